// CPU-side unit test of the (thread, device) workspace lookup of the stateless matcher entry points
// (orb_slam2_map_amd/csrc/workspace.h; VERDICT r2 "weak" 6: a workspace created on device A was reused on device B,
// or dropped without release).  Plain g++, fake device ids, no HIP.
#include "workspace.h"

#include <cstdio>
#include <thread>

struct Fake {
    static int live, released, dropped;
    int device = -1;
    int stream_of_device = -1;  // stands for the hipStream_t / DevBufs created on `device`
    Fake() { live++; }
    ~Fake()  // the shape of ProjWorkspace / the brute-force Ws: release on thread exit, never while the process exits
    {
        live--;
        if (device < 0)
            return;
        if (orbgpu::process_exiting().load())
            dropped++;
        else
            released++;  // "hipSetDevice(device); hipStreamDestroy(stream); hipFree(...)"
    }
};
int Fake::live = 0, Fake::released = 0, Fake::dropped = 0;

static Fake &use(int device)
{
    Fake &w = orbgpu::per_device_workspace<Fake>(device);
    if (w.device != device) {  // first use: "create the stream and the buffers on the selected device"
        w.device = device;
        w.stream_of_device = device;
    }
    return w;
}

#define CHECK(c)                                                  \
    do {                                                          \
        if (!(c)) {                                               \
            std::printf("FAILED %s (line %d)\n", #c, __LINE__);   \
            return 1;                                             \
        }                                                         \
    } while (0)

int main()
{
    Fake *a0 = &use(0);
    Fake *a3 = &use(3);
    CHECK(a0 != a3);
    CHECK(a0->stream_of_device == 0 && a3->stream_of_device == 3);
    CHECK(&use(0) == a0);               // switching back finds device 0's own workspace again ...
    CHECK(a0->stream_of_device == 0);   // ... with the resources created there
    CHECK(&use(3) == a3);
    CHECK(&use(7) != a3 && use(7).stream_of_device == 7);
    CHECK(&use(3) == a3 && &use(0) == a0);  // growing the table keeps the existing workspaces where they are
    CHECK(Fake::live == 3);             // nothing was dropped or duplicated on the switches
    int other_thread_ok = 0;
    std::thread t([&] {
        Fake &b0 = use(0);
        other_thread_ok = (&b0 != a0) && b0.stream_of_device == 0;
    });
    t.join();
    CHECK(other_thread_ok);             // workspaces are per thread: no sharing of a stream between host threads
    CHECK(Fake::live == 3);             // the other thread's workspace went away with the thread
    CHECK(Fake::released == 1 && Fake::dropped == 0);  // ... and gave its device resources back (the runtime is alive)
    // a pool of short-lived workers, two devices each: everything they created is released at their exit
    for (int k = 0; k < 8; k++) {
        std::thread w([&] { use(1), use(2); });
        w.join();
    }
    CHECK(Fake::live == 3 && Fake::released == 1 + 16 && Fake::dropped == 0);
    // a workspace that was never bound to a device has nothing to release
    {
        std::thread w([&] { (void)orbgpu::per_device_workspace<Fake>(5); });
        w.join();
    }
    CHECK(Fake::released == 17);
    // process exit: the library's atexit handler raises the flag before the HIP runtime is torn down; a thread that
    // ends after that must not touch the device any more
    orbgpu::process_exiting().store(true);
    {
        std::thread w([&] { use(4); });
        w.join();
    }
    CHECK(Fake::released == 17 && Fake::dropped == 1);
    orbgpu::process_exiting().store(false);
    std::printf("workspace_test ok\n");
    return 0;
}
