// Does hipStreamSynchronize on thread B's stream return before B's kernels have finished while thread A does
// something else (hipMalloc/hipFree, hipDeviceSynchronize, a synchronous hipMemcpy, a stream capture)?
//   hipcc --offload-arch=gfx950 -O2 -o sync_race sync_race.hip -lpthread && ./sync_race [seconds per mode]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_slow(int *out, int v, int spin)
{
    // ~spin * 10 ns of dependent work, then the value
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin)
        ;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *out = v;
}
__global__ void k_nop(int *p) { if (p && threadIdx.x == 12345) *p = 0; }

static std::atomic<bool> stop{false};

static void mode_thread(int mode)
{
    hipSetDevice(0);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<char> host(1 << 16, 1);
    void *d = nullptr;
    hipMalloc(&d, 1 << 16);
    while (!stop.load()) {
        switch (mode) {
        case 1: {  // hipMalloc / hipFree
            void *p = nullptr;
            hipMalloc(&p, 1 << 20);
            hipFree(p);
            break;
        }
        case 2:
            hipDeviceSynchronize();
            break;
        case 3:
            hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
            break;
        case 4: {  // capture + instantiate + launch
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                for (int i = 0; i < 8; i++)
                    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (int *)nullptr);
                if (hipStreamEndCapture(s, &g) == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                    hipGraphLaunch(ge, s);
                    hipStreamSynchronize(s);
                    hipGraphExecDestroy(ge);
                }
                if (g)
                    hipGraphDestroy(g);
            }
            (void)hipGetLastError();
            break;
        }
        case 5: {  // stream create / destroy
            hipStream_t t;
            hipStreamCreateWithFlags(&t, hipStreamNonBlocking);
            hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, t, (int *)nullptr);
            hipStreamSynchronize(t);
            hipStreamDestroy(t);
            break;
        }
        case 6: {  // pageable async copies + sync on an own stream
            hipMemcpyAsync(d, host.data(), host.size(), hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            break;
        }
        default:
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    hipFree(d);
    hipStreamDestroy(s);
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    hipSetDevice(0);
    const char *names[] = {"idle", "hipMalloc+hipFree", "hipDeviceSynchronize", "hipMemcpy (sync)", "capture+graph", "stream create/destroy",
                           "pageable async copy + sync"};
    for (int mode = 0; mode <= 6; mode++) {
        stop = false;
        std::vector<std::thread> others;
        for (int t = 0; t < 3; t++)
            others.emplace_back(mode_thread, mode);
        long long iters = 0, early = 0, early_pinned = 0;
        {
            hipStream_t s;
            hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            int *d = nullptr, *pin = nullptr;
            hipMalloc(&d, 64);
            hipHostMalloc(&pin, 64, hipHostMallocDefault);
            int pageable = 0;
            const auto t0 = std::chrono::steady_clock::now();
            int v = 0;
            std::vector<char> img(307200, 3);
            void *dimg = nullptr;
            hipMalloc(&dimg, img.size());
            const bool fresh = getenv("FRESH") != nullptr;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
                v++;
                if (fresh) {  // a stream nobody has used yet, like the first call on a new handle
                    hipStreamDestroy(s);
                    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                }
                hipMemcpy2DAsync(dimg, 640, img.data(), 640, 640, 480, hipMemcpyHostToDevice, s);  // pageable input first
                for (int k = 0; k < 10; k++)  // a chain of kernels like one extraction; the last writes v
                    hipLaunchKernelGGL(k_slow, dim3(1), dim3(64), 0, s, d, k == 9 ? v : -v, 1000);
                if (v & 1) {
                    hipMemcpyAsync(&pageable, d, 4, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    if (pageable != v)
                        early++;
                } else {
                    hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    if (*pin != v)
                        early_pinned++;
                }
                hipStreamSynchronize(s);
                iters++;
            }
            hipFree(d);
            hipHostFree(pin);
            hipStreamDestroy(s);
        }
        stop = true;
        for (auto &t : others)
            t.join();
        printf("others: %-28s  %lld rounds, result not there after sync: %lld (pageable copy), %lld (pinned copy)\n", names[mode], iters, early,
               early_pinned);
        fflush(stdout);
    }
    return 0;
}
