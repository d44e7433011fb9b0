// Cost of a cooperative-groups grid barrier on MI355X (8 XCDs): nwg workgroups of 1024 threads, nsync barriers, each
// workgroup also does a global atomic and reads a word another workgroup wrote (the pattern a multi-workgroup claim
// fixpoint would need).   hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip && ./grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(1024) void k_sync(int nsync, int *buf, int *out)
{
    cg::grid_group grid = cg::this_grid();
    int acc = 0;
    for (int s = 0; s < nsync; s++) {
        if (threadIdx.x == 0)
            atomicAdd(&buf[s & 7], 1);
        grid.sync();
        acc += __hip_atomic_load(&buf[s & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = acc;
}

int main()
{
    int *buf, *out;
    hipMalloc(&buf, 64);
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int nwg : {1, 2, 4, 10, 16, 32}) {
        for (int nsync : {0, 10, 50}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(buf, 0, 64);
                void *args[] = {&nsync, &buf, &out};
                hipEventRecord(e0, 0);
                hipError_t e = hipLaunchCooperativeKernel((const void *)k_sync, dim3(nwg), dim3(1024), args, 0, 0);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                if (e != hipSuccess) {
                    printf("launch failed: %s\n", hipGetErrorString(e));
                    return 1;
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("nwg %2d nsync %2d: %.1f us\n", nwg, nsync, best * 1e3f);
        }
    }
    return 0;
}
