// Micro-benchmark: issue rate of the VALU instructions the hot kernels are made of (each paired with one add, except the
// dot products and the 24-bit mad, which carry their add; the 3-input maximum carries two masking ands).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef _Float16 pkh __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
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
                if (OP == 0) { pk16 v = __builtin_elementwise_min(__builtin_bit_cast(pk16, x), __builtin_bit_cast(pk16, y)); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == 1) a[i] = (unsigned)min((int)x, (int)y) + 1;
                if (OP == 2) a[i] = __builtin_amdgcn_perm(x, y, 0x0c020c00u) + 1;
                if (OP == 3) { pk16 v = __builtin_bit_cast(pk16, x) - __builtin_bit_cast(pk16, y); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == 4) a[i] = __popc(x ^ y) + 1;
                if (OP == 5) a[i] = __builtin_amdgcn_udot4(x, y, 1u, false);
                if (OP == 6) { pkh v = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(pkh, x & 0x00FF00FFu), __builtin_bit_cast(pkh, y & 0x00FF00FFu)), __builtin_bit_cast(pkh, a[(i + 5) & 7] & 0x00FF00FFu)); a[i] = __builtin_bit_cast(unsigned, v) + 1; }
                if (OP == 9) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y), 1u, false);
                if (OP == 10) a[i] = __builtin_amdgcn_alignbit(x, y, 16) + 1;
                if (OP == 11) a[i] = __builtin_amdgcn_alignbyte(x, y, 3) + 1;
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name)
{
    unsigned *d; hipMalloc(&d, 256 * 2048 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per op: the op itself + 1 add  => 2 instructions per element-step
    double inst = 256.0 * 8 * 256 * (double)iters * 16 * 8 * 2;
    printf("%-14s %.3f ms  %.2f T lane-instr/s (op + add)\n", name, ms, inst / ms / 1e9);
    hipFree(d);
}
int main() { run<1>("v_min_i32"); run<0>("v_pk_min_i16"); run<3>("v_pk_sub_i16"); run<2>("v_perm_b32"); run<4>("xor+bcnt"); run<5>("v_dot4_u32_u8");
    run<9>("v_dot2_u32_u16"); run<6>("v_pk_maximum3_f16(+2 and)"); run<10>("v_alignbit"); run<11>("v_alignbyte"); return 0; }
