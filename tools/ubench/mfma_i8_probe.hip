// Probe of the gfx950 v_mfma_i32_32x32x32_i8 operand / accumulator lane maps with exact integer data
// (cdna_hip_programming.md: "Other dtypes: check the map with exact integer data before relying on it").
// Assumed (by analogy with the bf16 32x32x16 form): lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16h + j] and
// B[k = 16h + j][col r], j = 0..15 in its 16 operand bytes; C: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_i8_probe.hip -o /tmp/mfma_i8_probe && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const int8_t *A /*[32][32] row-major m,k*/, const int8_t *B /*[32][32] row-major k,n*/, int *C /*[32][32]*/)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    int8_t *pa = reinterpret_cast<int8_t *>(&a), *pb = reinterpret_cast<int8_t *>(&b);
    for (int j = 0; j < 16; j++) {
        pa[j] = A[r * 32 + 16 * h + j];
        pb[j] = B[(16 * h + j) * 32 + r];
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        C[row * 32 + r] = c[reg];
    }
}

int main()
{
    int8_t hA[1024], hB[1024];
    int hC[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 1024; i++) {
        hA[i] = (int8_t)(rand() % 255 - 127);
        hB[i] = (int8_t)(rand() % 255 - 127);
    }
    for (int m = 0; m < 32; m++)
        for (int n = 0; n < 32; n++) {
            int s = 0;
            for (int kk = 0; kk < 32; kk++)
                s += (int)hA[m * 32 + kk] * (int)hB[kk * 32 + n];
            ref[m * 32 + n] = s;
        }
    int8_t *dA, *dB;
    int *dC;
    hipMalloc(&dA, 1024), hipMalloc(&dB, 1024), hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++)
        bad += hC[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 assumed lane maps: %d of 1024 outputs differ\n", bad);
    return bad != 0;
}
