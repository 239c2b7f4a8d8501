// fake_hip.cpp -- malloc-backed stand-in for the 29 HIP runtime entry points the product's host logic calls
// (TEST INFRASTRUCTURE ONLY; see fake_hip.hpp).  Signatures come from the real <hip/hip_runtime_api.h>, the
// definitions here take libamdhip64's place at link time in tests/hostsim's sanitizer builds.
#include "fake_hip.hpp"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

struct ihipStream_t { int device; unsigned flags; };
struct ihipEvent_t { bool recorded; };

namespace fakehip {
namespace {

enum Kind { DEV = 0, PINNED, REGISTERED };
struct Range { size_t bytes; Kind kind; int device; };

std::mutex g_mu;                                 // registry of ranges / streams / events, counters
std::map<uintptr_t, Range> g_ranges;
std::set<const void*> g_streams, g_events;
int g_devices = 1;
size_t g_capacity = 0;
bool g_lazy = false;
long g_calls[F_COUNT];
long g_fail_at[F_COUNT];                         // fails when g_calls reaches this (0 = off)
long g_injected = 0, g_violations = 0;
std::string g_last_violation;
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;

std::mutex g_q_mu;                               // the queue of deferred work (lazy mode)
std::recursive_mutex g_flush_mu;                 // one flush at a time: work runs in enqueue order
std::deque<std::function<void()>> g_queue;

const char* const kNames[F_COUNT] = {
	"hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipHostRegister", "hipHostUnregister", "hipMemcpy", "hipMemcpyAsync",
	"hipMemcpy2D", "hipMemset", "hipMemsetAsync", "hipPointerGetAttributes", "hipStreamCreate", "hipStreamCreateWithFlags",
	"hipStreamDestroy", "hipStreamSynchronize", "hipStreamWaitEvent", "hipEventCreate", "hipEventCreateWithFlags", "hipEventDestroy",
	"hipEventRecord", "hipEventElapsedTime", "hipGetDevice", "hipSetDevice", "hipGetDeviceCount", "hipDeviceGetAttribute",
	"hipDeviceSynchronize"};

void violation(const std::string& what)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_violations++;
	g_last_violation = what;
	std::fprintf(stderr, "fake_hip: VIOLATION: %s\n", what.c_str());
}

hipError_t err(hipError_t e) { t_last = e; return e; }

// counts the call; true when this one has to fail
bool enter(int fn)
{
	std::lock_guard<std::mutex> lk(g_mu);
	const long k = ++g_calls[fn];
	if (g_fail_at[fn] > 0 && k == g_fail_at[fn]) { g_fail_at[fn] = 0; g_injected++; return true; }
	return false;
}

// the range that contains p, if any (g_mu held)
const Range* find_locked(const void* p, uintptr_t* base = nullptr)
{
	const uintptr_t a = (uintptr_t) p;
	auto it = g_ranges.upper_bound(a);
	if (it == g_ranges.begin()) return nullptr;
	--it;
	if (a >= it->first + it->second.bytes && !(it->second.bytes == 0 && a == it->first)) return nullptr;
	if (base) *base = it->first;
	return &it->second;
}

// [p, p + bytes) must lie inside ONE known range when p lies in any (plain host memory is the sanitizer's business)
bool span_ok(const void* p, size_t bytes, const char* who)
{
	std::unique_lock<std::mutex> lk(g_mu);
	uintptr_t base = 0;
	const Range* r = find_locked(p, &base);
	if (!r) return true;
	if ((uintptr_t) p + bytes <= base + r->bytes) return true;
	const size_t have = base + r->bytes - (uintptr_t) p;
	lk.unlock();
	violation(std::string(who) + ": " + std::to_string(bytes) + " bytes at an address with only " + std::to_string(have) + " left in its allocation");
	return false;
}

bool known(const void* p)
{
	std::lock_guard<std::mutex> lk(g_mu);
	return find_locked(p) != nullptr;
}

bool stream_ok(hipStream_t s, const char* who)
{
	if (!s) return true;                             // the NULL stream
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (g_streams.count(s)) return true;
	}
	violation(std::string(who) + ": stream that does not exist (destroyed?)");
	return false;
}

bool event_ok(hipEvent_t e, const char* who)
{
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (e && g_events.count(e)) return true;
	}
	violation(std::string(who) + ": event that does not exist (destroyed?)");
	return false;
}

}  // namespace

const char* fn_name(int fn) { return (fn >= 0 && fn < F_COUNT) ? kNames[fn] : "?"; }

void reset()
{
	flush();
	std::lock_guard<std::mutex> lk(g_mu);
	for (int i = 0; i < F_COUNT; i++) { g_calls[i] = 0; g_fail_at[i] = 0; }
	g_injected = 0;
	g_violations = 0;
	g_last_violation.clear();
}

void set_devices(int count) { std::lock_guard<std::mutex> lk(g_mu); g_devices = count < 1 ? 1 : count; }
void set_capacity(size_t bytes) { std::lock_guard<std::mutex> lk(g_mu); g_capacity = bytes; }
void set_lazy(bool on) { flush(); std::lock_guard<std::mutex> lk(g_mu); g_lazy = on; }
void fail_nth(int fn, long nth)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (fn >= 0 && fn < F_COUNT) g_fail_at[fn] = nth > 0 ? g_calls[fn] + nth : 0;
}
long calls(int fn) { std::lock_guard<std::mutex> lk(g_mu); return (fn >= 0 && fn < F_COUNT) ? g_calls[fn] : 0; }
long injected() { std::lock_guard<std::mutex> lk(g_mu); return g_injected; }
long violations() { std::lock_guard<std::mutex> lk(g_mu); return g_violations; }
const char* last_violation() { std::lock_guard<std::mutex> lk(g_mu); return g_last_violation.c_str(); }

void flush()
{
	std::lock_guard<std::recursive_mutex> fl(g_flush_mu);
	for (;;) {
		std::function<void()> work;
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			if (g_queue.empty()) return;
			work = std::move(g_queue.front());
			g_queue.pop_front();
		}
		work();
	}
}

void enqueue(void* stream, std::function<void()> work)
{
	(void) stream;                                   // one FIFO for all streams: enqueue order is a legal schedule
	bool lazy;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		lazy = g_lazy;
	}
	if (!lazy) {
		std::lock_guard<std::recursive_mutex> fl(g_flush_mu);      // not while another thread is half-way through the queue
		flush();
		work();
		return;
	}
	std::lock_guard<std::mutex> lk(g_q_mu);
	g_queue.push_back(std::move(work));
}

Live live()
{
	std::lock_guard<std::mutex> lk(g_mu);
	Live l{};
	for (auto& kv : g_ranges) {
		if (kv.second.kind == DEV) { l.device_allocs++; l.device_bytes += (long) kv.second.bytes; }
		else if (kv.second.kind == PINNED) l.pinned_allocs++;
		else l.registered_ranges++;
	}
	l.streams = (long) g_streams.size();
	l.events = (long) g_events.size();
	return l;
}

bool is_registered(const void* p)
{
	std::lock_guard<std::mutex> lk(g_mu);
	const Range* r = find_locked(p);
	return r && r->kind == REGISTERED;
}

}  // namespace fakehip

using namespace fakehip;

// ------------------------------------------------------------------------------------------------------------------
// the runtime entry points
// ------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int* count)
{
	if (enter(F_GetDeviceCount)) return err(hipErrorNoDevice);
	std::lock_guard<std::mutex> lk(g_mu);
	*count = g_devices;
	return hipSuccess;
}

hipError_t hipGetDevice(int* id)
{
	if (enter(F_GetDevice)) return err(hipErrorInvalidDevice);
	*id = t_device;
	return hipSuccess;
}

hipError_t hipSetDevice(int id)
{
	if (enter(F_SetDevice)) return err(hipErrorInvalidDevice);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (id < 0 || id >= g_devices) return err(hipErrorInvalidDevice);
	}
	t_device = id;
	return hipSuccess;
}

hipError_t hipDeviceGetAttribute(int* pi, hipDeviceAttribute_t attr, int)
{
	if (enter(F_DeviceGetAttribute)) return err(hipErrorInvalidValue);
	*pi = attr == hipDeviceAttributeMultiprocessorCount ? 8 : 0;
	return hipSuccess;
}

hipError_t hipDeviceSynchronize(void)
{
	const bool fail = enter(F_DeviceSynchronize);
	flush();
	return fail ? err(hipErrorUnknown) : hipSuccess;
}

hipError_t hipGetLastError(void)
{
	const hipError_t e = t_last;
	t_last = hipSuccess;
	return e;
}

const char* hipGetErrorString(hipError_t e)
{
	switch (e) {
	case hipSuccess: return "no error";
	case hipErrorOutOfMemory: return "out of memory";
	case hipErrorInvalidValue: return "invalid argument";
	case hipErrorInvalidDevice: return "invalid device ordinal";
	case hipErrorNoDevice: return "no ROCm-capable device is detected";
	case hipErrorLaunchFailure: return "unspecified launch failure";
	case hipErrorHostMemoryAlreadyRegistered: return "part or all of the requested memory range is already mapped";
	case hipErrorHostMemoryNotRegistered: return "pointer does not correspond to a registered memory region";
	default: return "unknown error (fake_hip)";
	}
}

hipError_t hipMalloc(void** ptr, size_t size)
{
	*ptr = nullptr;
	if (enter(F_Malloc)) return err(hipErrorOutOfMemory);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (g_capacity) {
			size_t used = 0;
			for (auto& kv : g_ranges) if (kv.second.kind == DEV && kv.second.device == t_device) used += kv.second.bytes;
			if (used + size > g_capacity) return err(hipErrorOutOfMemory);
		}
	}
	void* p = std::malloc(size ? size : 1);
	if (!p) return err(hipErrorOutOfMemory);
	std::lock_guard<std::mutex> lk(g_mu);
	g_ranges[(uintptr_t) p] = Range{size ? size : 1, DEV, t_device};
	*ptr = p;
	return hipSuccess;
}

hipError_t hipFree(void* p)
{
	const bool fail = enter(F_Free);
	if (!p) return hipSuccess;
	flush();                                         // hipFree synchronises the device
	{
		std::lock_guard<std::mutex> lk(g_mu);
		auto it = g_ranges.find((uintptr_t) p);
		if (it != g_ranges.end() && it->second.kind == DEV) {
			g_ranges.erase(it);
			std::free(p);
			return fail ? err(hipErrorUnknown) : hipSuccess;     // an injected failure still frees: the leak checks stay meaningful
		}
	}
	violation("hipFree of something that is not a live device allocation");
	return err(hipErrorInvalidValue);
}

hipError_t hipHostMalloc(void** ptr, size_t size, unsigned int)
{
	*ptr = nullptr;
	if (enter(F_HostMalloc)) return err(hipErrorOutOfMemory);
	void* p = std::malloc(size ? size : 1);
	if (!p) return err(hipErrorOutOfMemory);
	std::lock_guard<std::mutex> lk(g_mu);
	g_ranges[(uintptr_t) p] = Range{size ? size : 1, PINNED, -1};
	*ptr = p;
	return hipSuccess;
}

hipError_t hipHostFree(void* p)
{
	const bool fail = enter(F_HostFree);
	if (!p) return hipSuccess;
	flush();
	{
		std::lock_guard<std::mutex> lk(g_mu);
		auto it = g_ranges.find((uintptr_t) p);
		if (it != g_ranges.end() && it->second.kind == PINNED) {
			g_ranges.erase(it);
			std::free(p);
			return fail ? err(hipErrorUnknown) : hipSuccess;
		}
	}
	violation("hipHostFree of something that is not a live pinned allocation");
	return err(hipErrorInvalidValue);
}

hipError_t hipHostRegister(void* p, size_t bytes, unsigned int)
{
	if (enter(F_HostRegister)) return err(hipErrorOutOfMemory);
	if (!p || !bytes) return err(hipErrorInvalidValue);
	std::lock_guard<std::mutex> lk(g_mu);
	// overlap with anything known: already registered (or not host memory at all)
	auto it = g_ranges.lower_bound((uintptr_t) p);
	if (it != g_ranges.end() && it->first < (uintptr_t) p + bytes) return err(hipErrorHostMemoryAlreadyRegistered);
	if (find_locked(p)) return err(hipErrorHostMemoryAlreadyRegistered);
	g_ranges[(uintptr_t) p] = Range{bytes, REGISTERED, -1};
	return hipSuccess;
}

hipError_t hipHostUnregister(void* p)
{
	const bool fail = enter(F_HostUnregister);
	flush();
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_ranges.find((uintptr_t) p);
	if (it == g_ranges.end() || it->second.kind != REGISTERED) return err(hipErrorHostMemoryNotRegistered);
	g_ranges.erase(it);
	return fail ? err(hipErrorUnknown) : hipSuccess;
}

hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
	if (enter(F_PointerGetAttributes)) return err(hipErrorInvalidValue);
	std::lock_guard<std::mutex> lk(g_mu);
	const Range* r = find_locked(p);
	std::memset(a, 0, sizeof *a);
	if (!r) { a->type = hipMemoryTypeUnregistered; a->device = -1; return hipSuccess; }      // ordinary host memory, as ROCm 6+ reports it
	a->type = r->kind == DEV ? hipMemoryTypeDevice : hipMemoryTypeHost;
	a->device = r->device;
	return hipSuccess;
}

namespace {
void copy_now(void* dst, const void* src, size_t bytes)
{
	if (bytes) std::memmove(dst, src, bytes);
}
}  // namespace

hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind)
{
	const bool fail = enter(F_Memcpy);
	flush();                                          // a blocking copy: everything before it has run
	if (fail) return err(hipErrorUnknown);
	if (!span_ok(dst, bytes, "hipMemcpy dst") || !span_ok(src, bytes, "hipMemcpy src")) return err(hipErrorInvalidValue);
	copy_now(dst, src, bytes);
	return hipSuccess;
}

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t stream)
{
	if (enter(F_MemcpyAsync)) return err(hipErrorUnknown);
	if (!stream_ok(stream, "hipMemcpyAsync")) return err(hipErrorInvalidValue);
	if (!span_ok(dst, bytes, "hipMemcpyAsync dst") || !span_ok(src, bytes, "hipMemcpyAsync src")) return err(hipErrorInvalidValue);
	// pageable memory on either side makes the real call synchronous for the host: the source is read / the destination is
	// written before it returns.  Only copies between device and PINNED memory may still be in flight afterwards.
	if (!known(dst)) {
		flush();
		copy_now(dst, src, bytes);
		return hipSuccess;
	}
	if (!known(src)) {
		std::shared_ptr<std::vector<char>> snap(new std::vector<char>((const char*) src, (const char*) src + bytes));
		enqueue(stream, [dst, snap, bytes] { copy_now(dst, snap->data(), bytes); });
		return hipSuccess;
	}
	enqueue(stream, [dst, src, bytes] { copy_now(dst, src, bytes); });
	return hipSuccess;
}

hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind)
{
	const bool fail = enter(F_Memcpy2D);
	flush();
	if (fail) return err(hipErrorUnknown);
	if (height == 0 || width == 0) return hipSuccess;
	if (!span_ok(dst, (height - 1) * dpitch + width, "hipMemcpy2D dst") || !span_ok(src, (height - 1) * spitch + width, "hipMemcpy2D src"))
		return err(hipErrorInvalidValue);
	for (size_t r = 0; r < height; r++) copy_now((char*) dst + r * dpitch, (const char*) src + r * spitch, width);
	return hipSuccess;
}

hipError_t hipMemset(void* dst, int value, size_t bytes)
{
	const bool fail = enter(F_Memset);
	flush();
	if (fail) return err(hipErrorUnknown);
	if (!span_ok(dst, bytes, "hipMemset")) return err(hipErrorInvalidValue);
	if (bytes) std::memset(dst, value, bytes);
	return hipSuccess;
}

hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t stream)
{
	if (enter(F_MemsetAsync)) return err(hipErrorUnknown);
	if (!stream_ok(stream, "hipMemsetAsync") || !span_ok(dst, bytes, "hipMemsetAsync")) return err(hipErrorInvalidValue);
	enqueue(stream, [dst, value, bytes] { if (bytes) std::memset(dst, value, bytes); });
	return hipSuccess;
}

static hipError_t make_stream(int fn, hipStream_t* s, unsigned flags)
{
	*s = nullptr;
	if (enter(fn)) return err(hipErrorOutOfMemory);
	ihipStream_t* st = new ihipStream_t{t_device, flags};
	std::lock_guard<std::mutex> lk(g_mu);
	g_streams.insert(st);
	*s = st;
	return hipSuccess;
}

hipError_t hipStreamCreate(hipStream_t* s) { return make_stream(F_StreamCreate, s, 0); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int flags) { return make_stream(F_StreamCreateWithFlags, s, flags); }

hipError_t hipStreamDestroy(hipStream_t s)
{
	const bool fail = enter(F_StreamDestroy);
	flush();
	bool ok;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		ok = s && g_streams.erase(s) == 1;
	}
	if (!ok) { violation("hipStreamDestroy of a stream that does not exist"); return err(hipErrorInvalidValue); }
	delete s;
	return fail ? err(hipErrorUnknown) : hipSuccess;
}

hipError_t hipStreamSynchronize(hipStream_t s)
{
	const bool fail = enter(F_StreamSynchronize);
	if (!stream_ok(s, "hipStreamSynchronize")) return err(hipErrorInvalidValue);
	flush();
	return fail ? err(hipErrorLaunchFailure) : hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
	if (enter(F_StreamWaitEvent)) return err(hipErrorUnknown);
	if (!stream_ok(s, "hipStreamWaitEvent") || !event_ok(e, "hipStreamWaitEvent")) return err(hipErrorInvalidValue);
	return hipSuccess;                                // one FIFO: whatever was recorded earlier runs earlier
}

static hipError_t make_event(int fn, hipEvent_t* e)
{
	*e = nullptr;
	if (enter(fn)) return err(hipErrorOutOfMemory);
	ihipEvent_t* ev = new ihipEvent_t{false};
	std::lock_guard<std::mutex> lk(g_mu);
	g_events.insert(ev);
	*e = ev;
	return hipSuccess;
}

hipError_t hipEventCreate(hipEvent_t* e) { return make_event(F_EventCreate, e); }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return make_event(F_EventCreateWithFlags, e); }

hipError_t hipEventDestroy(hipEvent_t e)
{
	const bool fail = enter(F_EventDestroy);
	{
		std::lock_guard<std::mutex> lk(g_mu);
		if (!e || !g_events.erase(e)) e = nullptr;
	}
	if (!e) { violation("hipEventDestroy of an event that does not exist"); return err(hipErrorInvalidValue); }
	delete e;
	return fail ? err(hipErrorUnknown) : hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
	if (enter(F_EventRecord)) return err(hipErrorUnknown);
	if (!event_ok(e, "hipEventRecord") || !stream_ok(s, "hipEventRecord")) return err(hipErrorInvalidValue);
	std::lock_guard<std::mutex> lk(g_mu);
	e->recorded = true;
	return hipSuccess;
}

hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
	if (enter(F_EventElapsedTime)) return err(hipErrorUnknown);
	if (!event_ok(a, "hipEventElapsedTime") || !event_ok(b, "hipEventElapsedTime")) return err(hipErrorInvalidValue);
	std::lock_guard<std::mutex> lk(g_mu);
	if (!a->recorded || !b->recorded) return err(hipErrorInvalidValue);
	*ms = 0.001f;
	return hipSuccess;
}
