// fake_hip.hpp -- control surface of the malloc-backed stand-in for the HIP runtime (TEST INFRASTRUCTURE ONLY).
//
// The product's host logic (stochqn_amd/csrc/runtime.cpp, machines.cpp, group.cpp: context registry, mirrors, reclaim /
// spill / resume, host-range pinning, shard worker threads) calls 30 entry points of the HIP runtime.  In the product
// they resolve to libamdhip64; in tests/hostsim they resolve -- at link time, the sources are compiled unchanged --
// to fake_hip.cpp: "device" memory is malloc, copies are memcpy, streams run their work in enqueue order.  No kernels,
// no oracle, nothing of this is ever linked into libstochqn.so.  What it buys: the host logic runs on the CPU under
// -fsanitize=address,undefined and -fsanitize=thread, with every HIP call site failable on demand.
#pragma once
#include <cstddef>
#include <functional>

namespace fakehip {

// the entry points that can be made to fail / are counted
enum Fn {
	F_Malloc = 0, F_Free, F_HostMalloc, F_HostFree, F_HostRegister, F_HostUnregister, F_Memcpy, F_MemcpyAsync, F_Memcpy2D,
	F_Memset, F_MemsetAsync, F_PointerGetAttributes, F_StreamCreate, F_StreamCreateWithFlags, F_StreamDestroy,
	F_StreamSynchronize, F_StreamWaitEvent, F_EventCreate, F_EventCreateWithFlags, F_EventDestroy, F_EventRecord,
	F_EventElapsedTime, F_GetDevice, F_SetDevice, F_GetDeviceCount, F_DeviceGetAttribute, F_DeviceSynchronize, F_StreamQuery, F_COUNT
};
const char* fn_name(int fn);

void reset();                                   // forget injections, counters, violations; live objects stay
void set_devices(int count);                    // visible devices (default 1)
void set_capacity(size_t bytes);                // device memory per device; 0 = unlimited.  Beyond it hipMalloc reports hipErrorOutOfMemory
// streams: false (default) = every operation runs when it is enqueued; true = operations are queued and run, in enqueue
// order, only when something synchronises (stream / device synchronise, a blocking copy, hipFree) -- the latest moment a
// correct caller may count on.  Two legal schedules at opposite ends.
void set_lazy(bool on);
// model 2 (round 5): a queue PER STREAM.  An operation runs only when its own stream is synchronised (or a stream that waits for
// it through an event, or the whole device); hipStreamSynchronize(a) leaves what is queued on stream b where it is, and
// hipHostUnregister waits for nothing -- the schedule under which "the call returned while a copy through the caller's array
// was still queued on a side stream" shows.  0 = immediate, 1 = set_lazy(true).
void set_stream_model(int model);
long pending_ops();                             // operations queued anywhere (models 1 and 2)
long pending_host_ops();                        // model 2: queued copies with HOST memory on one side (what must be 0 when a host caller gets its call back)
// the caller's allocator: [p, p + bytes) is freed by its owner / handed out again.  Freed while a queued operation still goes
// through it, or while it is still registered: a violation; any copy through a dead range: a violation
void host_range_dies(const void* p, size_t bytes);
void host_range_lives(const void* p, size_t bytes);
void fail_nth(int fn, long nth);                // the nth call of `fn` from now (1 = the next) fails once; <= 0: off
long calls(int fn);                             // calls of `fn` since reset()
long injected();                                // injections that fired since reset()

// work that a "kernel launch" stands for: runs now, or at the next synchronisation in lazy mode (fake_launch.cpp)
void enqueue(void* stream, std::function<void()> work);
// an operation that may not be able to complete when its turn comes (a collective whose peers have not arrived yet): try_run is
// asked again -- at every synchronisation and every hipStreamQuery of its stream -- until it returns true
void enqueue_waitable(void* stream, std::function<bool()> try_run);
void flush();                                   // run everything that is queued (what a synchronisation does)

// what is alive (leak checks) and what went wrong (use of dead objects, copies outside allocations, double frees ...)
struct Live { long device_allocs, device_bytes, pinned_allocs, registered_ranges, streams, events; };
Live live();
long violations();
const char* last_violation();
bool is_registered(const void* p);              // p lies inside a range pinned with hipHostRegister

}  // namespace fakehip
