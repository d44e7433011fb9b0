// Per-thread, per-device workspaces of the stateless entry points (no HIP types here: tests/workspace_test.cpp
// compiles this header with plain g++).
//
// The reference builds an ORBmatcher on the stack per call site (Tracking.cc:1153, 1488, 1048), so the C ABI offers
// free functions with a `device_id` argument; allocating their staging buffers per call would dominate the kernel time,
// hence a workspace that lives as long as the calling thread.  A stream or a buffer belongs to the device it was
// created on: the workspace is therefore looked up by (thread, device) -- a thread that alternates between devices gets
// one workspace per device, nothing is reused across devices and nothing is dropped on a switch.
#pragma once

#include <atomic>
#include <memory>
#include <vector>

namespace orbgpu {

// A workspace dies with its thread (the thread_local table below) and must give its stream and device buffers back
// then: a caller that matches from short-lived worker threads would otherwise leak ~20 buffers and a stream per thread.
// The one moment it must NOT touch HIP is process exit, when the runtime may already be torn down: the library
// registers an atexit handler at load time (runtime.hip) that raises this flag; libamdhip64 was loaded before
// liborbgpu (it is a dependency), so its own handlers ran their registration earlier and run their teardown LATER than
// ours.  Workspace destructors release their device resources iff the flag is still down.
inline std::atomic<bool> &process_exiting()
{
    static std::atomic<bool> flag{false};
    return flag;
}

template <typename W> W &per_device_workspace(int device_id)
{
    static thread_local std::vector<std::unique_ptr<W>> table;
    const size_t slot = device_id < 0 ? 0 : (size_t)device_id;
    if (slot >= table.size())
        table.resize(slot + 1);
    if (!table[slot])
        table[slot].reset(new W());
    return *table[slot];
}

} // namespace orbgpu
