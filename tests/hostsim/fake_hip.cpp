// fake_hip.cpp -- malloc-backed stand-in for the 30 HIP runtime entry points the product's host logic calls
// (TEST INFRASTRUCTURE ONLY; see fake_hip.hpp).  Signatures come from the real <hip/hip_runtime_api.h>, the
// definitions here take libamdhip64's place at link time in tests/hostsim's sanitizer builds.
#include "fake_hip.hpp"
#include <execinfo.h>

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
#include <chrono>
#include <set>
#include <thread>
#include <string>
#include <vector>

struct ihipStream_t { int device; unsigned flags; };
struct ihipEvent_t { bool recorded; const void* stream; long ticket; };        // (per-stream model) recorded after `ticket` operations of `stream`

namespace fakehip {
namespace {

enum Kind { DEV = 0, PINNED, REGISTERED };
struct Range { size_t bytes; Kind kind; int device; };

std::mutex g_mu;                                 // registry of ranges / streams / events, counters
std::map<uintptr_t, Range> g_ranges;
std::set<const void*> g_streams, g_events;
int g_devices = 1;
size_t g_capacity = 0;
int g_mode = 0;                                  // 0: work runs when enqueued; 1: one FIFO, run at any synchronisation; 2: a queue per stream
long g_calls[F_COUNT];
long g_fail_at[F_COUNT];                         // fails when g_calls reaches this (0 = off)
long g_injected = 0, g_violations = 0;
std::string g_last_violation;
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;

std::mutex g_q_mu;                               // the queues of deferred work
std::recursive_mutex g_flush_mu;                 // one flush at a time: work runs in enqueue order
std::deque<std::function<void()>> g_queue;       // model 1: one FIFO over all streams

// model 2: a queue per stream.  An operation runs only when ITS stream (or one that waits for it through an event, or the
// device) is synchronised -- what a stream other than the one the caller waited for may legally do.  The NULL stream runs at
// once, behind everything on the blocking streams (a legal schedule of its own).
struct Op {
	std::function<void()> work;                  // empty (and no try_run): a wait marker
	std::function<bool()> try_run;               // an operation that may not be able to complete yet (a collective whose peers have not arrived): false = ask again
	const void* wait_stream = nullptr;
	long wait_ticket = 0;
	uintptr_t lo[2] = {0, 0}, hi[2] = {0, 0};    // host memory the operation touches (copies): what must not die under it
	const char* what = "work";
};
struct Queue { std::deque<Op> ops; long enqueued = 0, done = 0; };
std::map<const void*, Queue> g_sq;               // g_q_mu
std::map<uintptr_t, size_t> g_dead;              // host ranges their owner has freed (host_range_dies), g_mu

bool touches(const Op& op, uintptr_t lo, uintptr_t hi)
{
	for (int k = 0; k < 2; k++) if (op.hi[k] > op.lo[k] && op.lo[k] < hi && lo < op.hi[k]) return true;
	return false;
}

const char* const kNames[F_COUNT] = {
	"hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipHostRegister", "hipHostUnregister", "hipMemcpy", "hipMemcpyAsync",
	"hipMemcpy2D", "hipMemset", "hipMemsetAsync", "hipPointerGetAttributes", "hipStreamCreate", "hipStreamCreateWithFlags",
	"hipStreamDestroy", "hipStreamSynchronize", "hipStreamWaitEvent", "hipEventCreate", "hipEventCreateWithFlags", "hipEventDestroy",
	"hipEventRecord", "hipEventElapsedTime", "hipGetDevice", "hipSetDevice", "hipGetDeviceCount", "hipDeviceGetAttribute",
	"hipDeviceSynchronize", "hipStreamQuery"};

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
void set_lazy(bool on) { set_stream_model(on ? 1 : 0); }
void set_stream_model(int model) { flush(); std::lock_guard<std::mutex> lk(g_mu); g_mode = model < 0 || model > 2 ? 0 : model; }
void fail_nth(int fn, long nth)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (fn >= 0 && fn < F_COUNT) g_fail_at[fn] = nth > 0 ? g_calls[fn] + nth : 0;
}
long calls(int fn) { std::lock_guard<std::mutex> lk(g_mu); return (fn >= 0 && fn < F_COUNT) ? g_calls[fn] : 0; }
long injected() { std::lock_guard<std::mutex> lk(g_mu); return g_injected; }
long violations() { std::lock_guard<std::mutex> lk(g_mu); return g_violations; }
const char* last_violation() { std::lock_guard<std::mutex> lk(g_mu); return g_last_violation.c_str(); }

namespace {
int mode() { std::lock_guard<std::mutex> lk(g_mu); return g_mode; }

// model 2: run the operations of stream s until `upto` of them are done (< 0: all that are queued now).  g_flush_mu held.
// false: the operation at the head of the queue cannot complete yet (Op::try_run) -- it stays where it is.
bool drain(const void* s, long upto)
{
	for (;;) {
		Op op;
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			auto it = g_sq.find(s);
			if (it == g_sq.end() || it->second.ops.empty() || (upto >= 0 && it->second.done >= upto)) return true;
			op = std::move(it->second.ops.front());
			it->second.ops.pop_front();
		}
		bool done = true;
		if (op.try_run) done = op.try_run();
		else if (op.work) op.work();
		else done = drain(op.wait_stream, op.wait_ticket);       // an event of another stream: that stream runs up to its record first
		std::lock_guard<std::mutex> lk(g_q_mu);
		auto it = g_sq.find(s);
		if (!done) {
			if (it != g_sq.end()) it->second.ops.push_front(std::move(op));
			return false;
		}
		if (it != g_sq.end()) it->second.done++;
	}
}

// a stream is synchronised: its operations run; one that cannot complete yet is asked again until it can (a real stream waits
// the same way -- for ever, if the peer never comes: after two minutes the stand-in calls that a violation and moves on)
void drain_all_of(const void* s)
{
	for (long spins = 0;; spins++) {
		{
			std::lock_guard<std::recursive_mutex> fl(g_flush_mu);
			if (drain(s, -1)) return;
		}
		if (spins == 10000 && std::getenv("FAKE_HIP_TRACE_STUCK")) {         // 2 s without progress: who is waiting? (debugging aid)
			void* frames[48];
			const int nf = backtrace(frames, 48);
			std::fprintf(stderr, "fake_hip: a wait for stream %p has made no progress for 2 s; the waiter:\n", s);
			backtrace_symbols_fd(frames, nf, 2);
		}
		if (spins > 600000) { violation("a stream never finished: an operation on it waited for something that did not come"); return; }
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

bool is_blocking(const void* s)
{
	if (!s) return true;
	std::lock_guard<std::mutex> lk(g_mu);
	return g_streams.count(s) && !(static_cast<const ihipStream_t*>(s)->flags & hipStreamNonBlocking);
}

// what the NULL stream (and every blocking call: hipMemcpy, hipMemset) orders itself behind: the blocking streams
void drain_blocking()
{
	std::lock_guard<std::recursive_mutex> fl(g_flush_mu);
	std::vector<const void*> ss;
	{
		std::lock_guard<std::mutex> lk(g_q_mu);
		for (auto& kv : g_sq) ss.push_back(kv.first);
	}
	for (const void* st : ss) if (is_blocking(st)) (void) drain(st, -1);
}

void sync_stream(const void* s)
{
	const int m = mode();
	if (m != 2) { flush(); return; }
	if (!s) { drain_blocking(); return; }
	drain_all_of(s);
}

void push(const void* stream, Op op)
{
	const int m = mode();
	if (m == 2 && stream) {
		std::lock_guard<std::mutex> lk(g_q_mu);
		Queue& q = g_sq[stream];
		q.ops.push_back(std::move(op));
		q.enqueued++;
		return;
	}
	auto now = [](Op& o) {                              // run at once; an operation that cannot complete yet is asked until it can
		if (o.try_run) { while (!o.try_run()) std::this_thread::sleep_for(std::chrono::microseconds(200)); }
		else if (o.work) o.work();
	};
	if (m == 2) { drain_blocking(); now(op); return; }                       // the NULL stream
	if (op.try_run && !op.work) { std::function<bool()> t = op.try_run; op.work = [t] { while (!t()) std::this_thread::sleep_for(std::chrono::microseconds(200)); }; op.try_run = nullptr; }
	if (!op.work) return;                                                    // models 0 / 1: one FIFO, waits are implied
	if (m == 0) {
		std::lock_guard<std::recursive_mutex> fl(g_flush_mu);      // not while another thread is half-way through the queue
		flush();
		op.work();
		return;
	}
	std::lock_guard<std::mutex> lk(g_q_mu);
	g_queue.push_back(std::move(op.work));
}
}  // namespace

void flush()
{
	std::lock_guard<std::recursive_mutex> fl(g_flush_mu);
	for (;;) {
		std::function<void()> work;
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			if (g_queue.empty()) break;
			work = std::move(g_queue.front());
			g_queue.pop_front();
		}
		work();
	}
	for (;;) {                                       // model 2: every stream, until nothing is left anywhere
		std::vector<const void*> ss;
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			for (auto& kv : g_sq) if (!kv.second.ops.empty()) ss.push_back(kv.first);
		}
		if (ss.empty()) return;
		for (const void* st : ss) drain_all_of(st);
	}
}

void enqueue(void* stream, std::function<void()> work)
{
	Op op;
	op.work = std::move(work);
	op.what = "kernel";
	push(stream, std::move(op));
}

void enqueue_waitable(void* stream, std::function<bool()> try_run)
{
	Op op;
	op.try_run = std::move(try_run);
	op.what = "collective";
	push(stream, std::move(op));
}

long pending_host_ops()
{
	std::lock_guard<std::mutex> lk(g_q_mu);
	long k = 0;
	for (auto& kv : g_sq) for (const Op& op : kv.second.ops) k += (op.hi[0] > op.lo[0]) || (op.hi[1] > op.lo[1]);
	return k;
}

long pending_ops()
{
	std::lock_guard<std::mutex> lk(g_q_mu);
	long k = (long) g_queue.size();
	for (auto& kv : g_sq) k += (long) kv.second.ops.size();
	return k;
}

void host_range_dies(const void* p, size_t bytes)
{
	const uintptr_t lo = (uintptr_t) p, hi = lo + bytes;
	std::string hit;
	{
		std::lock_guard<std::mutex> lk(g_q_mu);
		for (auto& kv : g_sq) for (const Op& op : kv.second.ops) if (touches(op, lo, hi)) hit = op.what;
	}
	if (!hit.empty()) violation("a host range was freed by its owner while a queued " + hit + " still goes through it");
	{
		std::lock_guard<std::mutex> lk(g_mu);
		uintptr_t base = 0;
		const Range* r = find_locked(p, &base);
		if (r && r->kind == REGISTERED) hit = "registered";
		g_dead[lo] = bytes;
	}
	if (hit == "registered") violation("a host range was freed by its owner while it was still page-locked (hipHostRegister)");
}

void host_range_lives(const void* p, size_t bytes)
{
	const uintptr_t lo = (uintptr_t) p, hi = lo + bytes;
	std::lock_guard<std::mutex> lk(g_mu);
	for (auto it = g_dead.begin(); it != g_dead.end();)
		if (it->first < hi && lo < it->first + it->second) it = g_dead.erase(it);
		else ++it;
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
	if (mode() == 2) {                               // the real call waits for nothing: a copy still queued through the range loses its pages
		size_t bytes = 0;
		{
			std::lock_guard<std::mutex> lk(g_mu);
			auto it = g_ranges.find((uintptr_t) p);
			if (it != g_ranges.end() && it->second.kind == REGISTERED) bytes = it->second.bytes;
		}
		bool hit = false;
		{
			std::lock_guard<std::mutex> lk(g_q_mu);
			for (auto& kv : g_sq) for (const Op& op : kv.second.ops) hit = hit || (bytes && touches(op, (uintptr_t) p, (uintptr_t) p + bytes));
		}
		if (hit) violation("hipHostUnregister of a range that a queued copy still goes through");
	}
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
bool dead(const void* p, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	const uintptr_t lo = (uintptr_t) p, hi = lo + bytes;
	for (auto& kv : g_dead) if (kv.first < hi && lo < kv.first + kv.second) return true;
	return false;
}

void copy_now(void* dst, const void* src, size_t bytes)
{
	if (!bytes) return;
	if (dead(dst, bytes) || dead(src, bytes)) { violation("a copy goes through a host range that its owner has freed"); return; }
	std::memmove(dst, src, bytes);
}

// the host side(s) of a copy, for Op::lo / hi (device allocations of the fake are not the caller's to free)
void host_sides(Op& op, const void* dst, const void* src, size_t bytes)
{
	std::lock_guard<std::mutex> lk(g_mu);
	int k = 0;
	for (const void* p : {dst, src}) {
		const Range* r = find_locked(p);
		if (r && r->kind == DEV) continue;
		op.lo[k] = (uintptr_t) p; op.hi[k] = (uintptr_t) p + bytes; k++;
	}
}
}  // namespace

hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind)
{
	const bool fail = enter(F_Memcpy);
	sync_stream(nullptr);                             // a blocking copy: everything before it (on the blocking streams) has run
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
		sync_stream(stream);
		copy_now(dst, src, bytes);
		return hipSuccess;
	}
	Op op;
	op.what = "copy";
	if (!known(src)) {
		if (dead(src, bytes)) { violation("a copy reads a host range that its owner has freed"); return err(hipErrorInvalidValue); }
		std::shared_ptr<std::vector<char>> snap(new std::vector<char>((const char*) src, (const char*) src + bytes));
		op.work = [dst, snap, bytes] { copy_now(dst, snap->data(), bytes); };
		host_sides(op, dst, nullptr, bytes);
		push(stream, std::move(op));
		return hipSuccess;
	}
	op.work = [dst, src, bytes] { copy_now(dst, src, bytes); };
	host_sides(op, dst, src, bytes);
	push(stream, std::move(op));
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
	sync_stream(nullptr);
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
	sync_stream(s);                                   // (the real call lets the stream's work finish)
	if (mode() == 2) { std::lock_guard<std::mutex> lk(g_q_mu); g_sq.erase(s); }
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
	sync_stream(s);
	return fail ? err(hipErrorLaunchFailure) : hipSuccess;
}

hipError_t hipStreamQuery(hipStream_t s)
{
	if (enter(F_StreamQuery)) return err(hipErrorUnknown);
	if (!stream_ok(s, "hipStreamQuery")) return err(hipErrorInvalidValue);
	const int m = mode();
	if (m == 2 && s) {                                 // the device works while the host asks: what can run, runs
		std::lock_guard<std::recursive_mutex> fl(g_flush_mu);
		(void) drain(s, -1);
	}
	if (m == 1) flush();                               // one FIFO: a device that is asked has had time to work its queue off
	std::lock_guard<std::mutex> lk(g_q_mu);
	bool busy = m == 1 && !g_queue.empty();
	if (m == 2) { auto it = g_sq.find(s); busy = it != g_sq.end() && !it->second.ops.empty(); }
	return busy ? err(hipErrorNotReady) : hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int)
{
	if (enter(F_StreamWaitEvent)) return err(hipErrorUnknown);
	if (!stream_ok(s, "hipStreamWaitEvent") || !event_ok(e, "hipStreamWaitEvent")) return err(hipErrorInvalidValue);
	if (mode() == 2) {
		Op op;
		op.what = "wait";
		{
			std::lock_guard<std::mutex> lk(g_mu);
			if (!e->recorded) return hipSuccess;      // an event never recorded holds nobody up
			op.wait_stream = e->stream;
			op.wait_ticket = e->ticket;
		}
		if (op.wait_stream == s) return hipSuccess;   // same stream: already in order
		if (!s) { std::lock_guard<std::recursive_mutex> fl(g_flush_mu); (void) drain(op.wait_stream, op.wait_ticket); return hipSuccess; }
		push(s, std::move(op));
	}
	return hipSuccess;                                // models 0 / 1: one FIFO, whatever was recorded earlier runs earlier
}

static hipError_t make_event(int fn, hipEvent_t* e)
{
	*e = nullptr;
	if (enter(fn)) return err(hipErrorOutOfMemory);
	ihipEvent_t* ev = new ihipEvent_t{false, nullptr, 0};
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
	long ticket = 0;
	{
		std::lock_guard<std::mutex> lk(g_q_mu);
		auto it = g_sq.find(s);
		if (it != g_sq.end()) ticket = it->second.enqueued;
	}
	std::lock_guard<std::mutex> lk(g_mu);
	e->recorded = true;
	e->stream = s;
	e->ticket = ticket;
	return hipSuccess;
}

hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
	if (enter(F_EventElapsedTime)) return err(hipErrorUnknown);
	if (!event_ok(a, "hipEventElapsedTime") || !event_ok(b, "hipEventElapsedTime")) return err(hipErrorInvalidValue);
	if (mode() == 2) {                                // both must have happened: the product only asks after a synchronisation
		std::lock_guard<std::mutex> lk(g_q_mu);
		for (hipEvent_t e : {a, b}) { auto it = g_sq.find(e->stream); if (it != g_sq.end() && it->second.done < e->ticket) return err(hipErrorNotReady); }
	}
	std::lock_guard<std::mutex> lk(g_mu);
	if (!a->recorded || !b->recorded) return err(hipErrorInvalidValue);
	*ms = 0.001f;
	return hipSuccess;
}
