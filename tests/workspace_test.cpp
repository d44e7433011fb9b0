// CPU-side unit test of the (thread, device) workspace lookup of the stateless matcher entry points
// (orb_slam2_map_amd/csrc/workspace.h; VERDICT r2 "weak" 6: a workspace created on device A was reused on device B,
// or dropped without release).  Plain g++, fake device ids, no HIP.
#include "workspace.h"

#include <cstdio>
#include <thread>

struct Fake {
    static int live;
    int device = -1;
    int stream_of_device = -1;  // stands for the hipStream_t / DevBufs created on `device`
    Fake() { live++; }
    ~Fake() { live--; }
};
int Fake::live = 0;

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
    std::printf("workspace_test ok\n");
    return 0;
}
