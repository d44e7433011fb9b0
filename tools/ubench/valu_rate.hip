// Micro-benchmark: VALU issue rate of the instructions the hot kernels are made of, with f32 control rows.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
//
// Every variant runs 8 independent dependency chains per lane, 8 waves per SIMD (2048 workgroups of 256 threads),
// and retires a known number of VALU instructions per chain step (column "instr/step": the integer / packed ops are
// paired with one v_add_u32 that keeps the optimiser from collapsing them; the dot products carry their own
// accumulate; control rows and multiplies are inline asm, one instruction per step).  Reported: lane-instructions per second chip-wide = 64 x wave-instructions / time.
// Reference points (MI355X_MICROARCH.md): a wave64 VALU instruction issues over 2 cycles on a SIMD (32 lanes per
// cycle) -> 4 SIMDs x 256 CUs x 32 lanes x 2.4 GHz = 78.6 T lane-instr/s if every instruction is full rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef _Float16 pkh __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { ADD_U32, ADD_F32, FMA_F32, PK_FMA_F32, MIN_I32, PK_MIN_I16, PK_SUB_I16, PERM, XOR_BCNT, DOT4, DOT2, PK_MAX3_F16,
          ALIGNBIT, ALIGNBYTE, MUL_LO, MUL_U24, MAX3_I32,
          S_MIN_I32, S_MAX3_I32, S_PERM, S_PK_MIN_I16, S_PK_MAX3_F16, S_PK_MIN3_F16, S_BCNT, S_XOR, S_AND, S_ALIGNBIT, S_DOT4,
          S_MAD_U32_U24, S_PK_SUB_I16 };

template <int OP> __global__ void k(unsigned *out, int iters)
{
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                unsigned x = a[i], y = a[(i + 3) & 7];
                // control rows: inline asm, so that the optimiser neither packs (SLP) nor fuses them
                if (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == ADD_F32) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == PK_FMA_F32 && (i & 1) == 0) {  // 4 register pairs per lane: one packed FMA per pair and step
                    unsigned long long p = ((unsigned long long)a[i + 1] << 32) | a[i];
                    const unsigned long long q = ((unsigned long long)a[(i + 3) & 7] << 32) | a[(i + 2) & 7];
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p) : "v"(q));
                    a[i] = (unsigned)p;
                    a[i + 1] = (unsigned)(p >> 32);
                }
                if (OP == MIN_I32) a[i] = (unsigned)min((int)x, (int)y) + 1;
                if (OP == PK_MIN_I16) { pk16 v = __builtin_elementwise_min(__builtin_bit_cast(pk16, x), __builtin_bit_cast(pk16, y)); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == PK_SUB_I16) { pk16 v = __builtin_bit_cast(pk16, x) - __builtin_bit_cast(pk16, y); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == PERM) a[i] = __builtin_amdgcn_perm(x, y, 0x0c020c00u) + 1;
                if (OP == XOR_BCNT) a[i] = __popc(x ^ y) + 1;
                if (OP == DOT4) a[i] = __builtin_amdgcn_udot4(x, y, 1u, false);
                if (OP == DOT2) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y), 1u, false);
                if (OP == PK_MAX3_F16) { pkh v = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(pkh, x & 0x00FF00FFu), __builtin_bit_cast(pkh, y & 0x00FF00FFu)), __builtin_bit_cast(pkh, a[(i + 5) & 7] & 0x00FF00FFu)); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == ALIGNBIT) a[i] = __builtin_amdgcn_alignbit(x, y, 16) + 1;
                if (OP == ALIGNBYTE) a[i] = __builtin_amdgcn_alignbyte(x, y, 3) + 1;
                // single-instruction rows (inline asm, one instruction per step)
                if (OP == S_MIN_I32) asm volatile("v_min_i32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_MAX3_I32) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_PERM) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(x), "v"(y), "v"(0x0c020c00u));
                if (OP == S_PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_PK_SUB_I16) asm volatile("v_pk_sub_i16 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_PK_MAX3_F16) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_PK_MIN3_F16) asm volatile("v_pk_minimum3_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_XOR) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_AND) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_DOT4) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == S_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(x), "v"(y));
                if (OP == MAX3_I32) a[i] = (unsigned)max(max((int)x, (int)y), (int)a[(i + 5) & 7]) + 1;
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, double instr_per_step)
{
    unsigned *d; hipMalloc(&d, 256 * 2048 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = 256.0 * 8 * 256 * (double)iters * 16 * 8;  // lane-steps
    printf("%-34s %7.3f ms  instr/step %.1f  %6.2f T lane-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms,
           instr_per_step, steps * instr_per_step / ms / 1e9,
           (ms * 1e-3 * 2.4e9) / (steps * instr_per_step / 64.0 / (256.0 * 4)));
    hipFree(d);
}
int main()
{
    printf("-- control rows\n");
    run<ADD_U32>("v_add_u32", 1);
    run<ADD_F32>("v_add_f32", 1);
    run<FMA_F32>("v_fma_f32", 1);
    run<PK_FMA_F32>("v_pk_fma_f32 (half as many instr)", 0.5);
    printf("-- integer / packed ops of the extractor and matcher kernels (each + 1 v_add_u32 unless noted)\n");
    run<MIN_I32>("v_min_i32 + add", 2);
    run<MAX3_I32>("v_max3_i32 + add", 2);
    run<PK_MIN_I16>("v_pk_min_i16 + add", 2);
    run<PK_SUB_I16>("v_pk_sub_i16 + add", 2);
    run<PK_MAX3_F16>("v_pk_maximum3_f16 + 2 and + add", 4);
    run<PERM>("v_perm_b32 + add", 2);
    run<ALIGNBIT>("v_alignbit_b32 + add", 2);
    run<ALIGNBYTE>("v_alignbyte_b32 + add", 2);
    run<XOR_BCNT>("v_xor + v_bcnt (carries the add)", 2);
    run<DOT4>("v_dot4_u32_u8 (carries the add)", 1);
    run<DOT2>("v_dot2_u32_u16 (carries the add)", 1);
    printf("-- the same instructions alone (inline asm, one per step)\n");
    run<S_XOR>("v_xor_b32", 1);
    run<S_AND>("v_and_b32", 1);
    run<S_MIN_I32>("v_min_i32", 1);
    run<S_MAX3_I32>("v_max3_i32", 1);
    run<S_PK_MIN_I16>("v_pk_min_i16", 1);
    run<S_PK_SUB_I16>("v_pk_sub_i16", 1);
    run<S_PK_MAX3_F16>("v_pk_maximum3_f16", 1);
    run<S_PK_MIN3_F16>("v_pk_minimum3_f16", 1);
    run<S_PERM>("v_perm_b32", 1);
    run<S_ALIGNBIT>("v_alignbit_b32", 1);
    run<S_BCNT>("v_bcnt_u32_b32", 1);
    run<S_DOT4>("v_dot4_u32_u8", 1);
    run<S_MAD_U32_U24>("v_mad_u32_u24", 1);
    run<MUL_LO>("v_mul_lo_u32", 1);
    run<MUL_U24>("v_mul_u32_u24", 1);
    return 0;
}
