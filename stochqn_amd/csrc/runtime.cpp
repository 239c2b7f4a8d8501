// runtime.cpp -- context registry, pointer residency, mirrors, RCCL hook, options, profiler API.
#include "machines.hpp"
#include "stochqn_hip.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <unistd.h>
#include <cerrno>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <vector>

namespace sqn {

namespace {

std::recursive_mutex g_mu;          // recursive: an allocation made under the lock may reclaim another context (reclaim_one)
std::unordered_map<const void*, DevCtx*> g_ctx;
Options g_opt;
unsigned long long g_clock = 0;                 // registry clock: DevCtx::last_use

// ---- contexts reclaimed under memory pressure ---------------------------------------------------
// R and Python never call dealloc_*: an optimiser object that was dropped leaves its device context (the mirrors of its
// host arrays: S, Y, F ... tens of GB at the BASELINE shapes) behind until the process ends.  When device memory runs out
// -- or the mirrors exceed option "max_mirror_bytes" -- the least recently used context that is not inside a call is
// *spilled*: its mirrors are copied to host memory owned by the library and the context is destroyed.  Not into the
// caller's arrays: the object may be long gone and its arrays freed.  If the object does come back (same address, counters
// that continue where the spilled context stopped) the first call re-creates the context from the spill; if something
// else turns up at that address the spill is dropped.
struct SpillPiece { const void* caller = nullptr; size_t count = 0; real* data = nullptr; };      // data stays until the first call after the resume has completed
struct Spill {
	int kind = 0, n = 0;
	size_t m = 0, fsize = 0, niter = 0, bytes = 0;
	int section = 0;
	std::vector<SpillPiece> pieces;
	~Spill() { for (auto& p : pieces) std::free(p.data); }
};
std::unordered_map<const void*, Spill*> g_spill;
std::atomic<long> g_fail_alloc_after{-1};       // fault injection: < 0 off, else allocations left before one fails
std::atomic<int> g_inject_device_fault{0};      // fault injection: the next stream synchronisation reports a failure

std::atomic<long long> g_stats[ST_COUNT];
const char* const kStatNames[ST_COUNT] = {
	"steps_three_pass", "steps_sweeps", "steps_plain", "steps_kappa_fallback",
	"allreduces", "allreduce_doubles", "contexts_created", "contexts_reclaimed", "x_uploads", "x_uploads_skipped",
	"host_ranges_registered", "x_sent_ahead", "x_sent_again", "x_prefetched", "host_unpin_failed", "host_copies_in_flight", "host_pins_declined", "host_pins_foreign", "host_pin_errors"};

// Copies between device memory and ORDINARY host memory on the reclaim path, which by definition runs when the device is
// full: through a small pinned buffer made while memory was still plentiful (with the first mirror), so that the runtime
// does not have to pin hundreds of MB of the destination on the fly -- and find page-table space for that -- at the worst
// possible moment.  Falls back to a plain hipMemcpy when the buffer could not be made.
constexpr size_t kBounceBytes = (size_t) 32 << 20;
void* g_bounce = nullptr;
bool g_bounce_tried = false;

void ensure_bounce()
{
	if (g_bounce_tried) return;
	g_bounce_tried = true;
	if (hipHostMalloc(&g_bounce, kBounceBytes, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); g_bounce = nullptr; }
}

hipError_t bounced_copy(void* dst, const void* src, size_t bytes, bool to_host)
{
	if (!g_bounce) return hipMemcpy(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
	for (size_t off = 0; off < bytes; off += kBounceBytes) {
		const size_t k = bytes - off < kBounceBytes ? bytes - off : kBounceBytes;
		if (to_host) {
			const hipError_t e = hipMemcpy(g_bounce, (const char*) src + off, k, hipMemcpyDeviceToHost);
			if (e != hipSuccess) return e;
			std::memcpy((char*) dst + off, g_bounce, k);
		} else {
			std::memcpy(g_bounce, (const char*) src + off, k);
			const hipError_t e = hipMemcpy((char*) dst + off, g_bounce, k, hipMemcpyHostToDevice);
			if (e != hipSuccess) return e;
		}
	}
	return hipSuccess;
}

bool alloc_should_fail()
{
	long left = g_fail_alloc_after.load();
	while (left >= 0) {
		if (g_fail_alloc_after.compare_exchange_weak(left, left - 1)) return left == 0;
	}
	return false;
}
bool g_profile = false;
double g_retired_ms[K_COUNT] = {0};
long long g_retired_launches[K_COUNT] = {0};
bool g_atexit = false;

// ---- RCCL, loaded on demand -------------------------------------------------------------------
struct Comm {
	void* dl = nullptr;
	ncclComm_t comm = nullptr;
	int rank = 0, nranks = 1;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                       // optional: present in every RCCL of ROCm 5+
	ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr; // optional
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_comm;
// communicators that were aborted after a reduction failed or never completed: every later reduction over them fails at once
std::mutex g_dead_mu;
std::vector<void*> g_dead_comms;
bool comm_is_dead(void* comm)
{
	std::lock_guard<std::mutex> lk(g_dead_mu);
	return std::find(g_dead_comms.begin(), g_dead_comms.end(), comm) != g_dead_comms.end();
}

bool load_rccl()
{
	if (g_comm.dl) return true;
	// STOCHQN_HIP_RCCL_LIB names the library to load (a particular RCCL build; tests/hostsim: a host-side stand-in)
	const char* names[] = {std::getenv("STOCHQN_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* nm : names) {
		if (!nm || !*nm) continue;
		g_comm.dl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
		if (g_comm.dl) break;
	}
	if (!g_comm.dl) { std::fprintf(stderr, "stochqn: cannot dlopen RCCL: %s\n", dlerror()); return false; }
	g_comm.GetUniqueId = (decltype(g_comm.GetUniqueId)) dlsym(g_comm.dl, "ncclGetUniqueId");
	g_comm.CommInitRank = (decltype(g_comm.CommInitRank)) dlsym(g_comm.dl, "ncclCommInitRank");
	g_comm.CommInitAll = (decltype(g_comm.CommInitAll)) dlsym(g_comm.dl, "ncclCommInitAll");
	g_comm.AllReduce = (decltype(g_comm.AllReduce)) dlsym(g_comm.dl, "ncclAllReduce");
	g_comm.CommDestroy = (decltype(g_comm.CommDestroy)) dlsym(g_comm.dl, "ncclCommDestroy");
	g_comm.GetErrorString = (decltype(g_comm.GetErrorString)) dlsym(g_comm.dl, "ncclGetErrorString");
	g_comm.CommAbort = (decltype(g_comm.CommAbort)) dlsym(g_comm.dl, "ncclCommAbort");
	g_comm.CommGetAsyncError = (decltype(g_comm.CommGetAsyncError)) dlsym(g_comm.dl, "ncclCommGetAsyncError");
	if (!g_comm.GetUniqueId || !g_comm.CommInitRank || !g_comm.CommInitAll || !g_comm.AllReduce || !g_comm.CommDestroy) {
		std::fprintf(stderr, "stochqn: RCCL library lacks the expected symbols\n");
		dlclose(g_comm.dl);                      // a later attempt starts over instead of calling NULL pointers
		g_comm = Comm{};
		return false;
	}
	return true;
}

// A reduction that failed leaves un-reduced local partial sums where the next kernel expects global
// ones: the step would finish with wrong alpha / beta and the ranks would take different decisions.
// Mark the context instead: the call that saw it returns -1000 / invalid_input (machines.cpp: after_call).
void reducer_failed(void* user, const char* what)
{
	std::fprintf(stderr, "stochqn: %s -- the call fails (-1000) instead of continuing on un-reduced sums\n", what);
	if (user) static_cast<DevCtx*>(user)->fault = true;
}

void allreduce_hook(void* user, double* buf, int count, hipStream_t stream)
{
	DevCtx* c = static_cast<DevCtx*>(user);
	stat_add(ST_ALLREDUCE); stat_add(ST_ALLREDUCE_DOUBLES, count);
	if (comm_is_dead(c->red.comm)) { reducer_failed(user, "the communicator was aborted after an earlier failure"); return; }
	ncclResult_t r = g_comm.AllReduce(buf, buf, (size_t) count, ncclDouble, ncclSum, (ncclComm_t) c->red.comm, stream);
	if (r != ncclSuccess) {
		std::fprintf(stderr, "stochqn: ncclAllReduce: %s\n", g_comm.GetErrorString ? g_comm.GetErrorString(r) : "?");
		reducer_failed(user, "ncclAllReduce failed");
		// a communicator one of whose collectives failed is finished: the peers are waiting for a contribution that will not come,
		// this rank must not post the next one as if nothing had happened.  Aborted here, dead for every later reduction.
		std::lock_guard<std::mutex> lk(g_dead_mu);
		if (std::find(g_dead_comms.begin(), g_dead_comms.end(), c->red.comm) == g_dead_comms.end()) {
			g_dead_comms.push_back(c->red.comm);
			if (g_comm.CommAbort) (void) g_comm.CommAbort((ncclComm_t) c->red.comm);
		}
	}
}

// ---- caller-supplied reducer -------------------------------------------------------------------
struct Custom {
	stochqn_hip_allreduce_fn fn = nullptr;
	void* user = nullptr;
	int rank = 0, nranks = 1;
} g_custom;

void custom_hook(void* user, double* buf, int count, hipStream_t stream)
{
	stat_add(ST_ALLREDUCE); stat_add(ST_ALLREDUCE_DOUBLES, count);
	if (g_custom.fn(g_custom.user, buf, count, (void*) stream) != 0)
		reducer_failed(user, "the caller-supplied all-reduce reported a failure");
}

// ---- loop-back reducer: P shards of one problem driven by P host threads on ONE GPU ------------
// Rehearses the sharded path where only one device is available (tests): the all-reduce is a
// host-side rendezvous of the calling threads, summed in rank order.
Loopback g_loop;                     // the instance behind stochqn_hip_loopback_* (tests)
thread_local Reducer t_reducer;      // what contexts created by this thread reduce through (default: the process-wide one)

// false: a shard never arrived (it failed on its own and left the call early).  The rendezvous is then
// broken for good -- every later reduction fails at once instead of waiting -- and the calls return -1000.
bool loop_barrier(Loopback& lp)
{
	std::unique_lock<std::mutex> lk(lp.mu);
	if (lp.broken) return false;
	const long gen = lp.generation;
	if (++lp.arrived == lp.nranks) {
		lp.arrived = 0;
		lp.generation++;
		lp.cv.notify_all();
		return true;
	}
#if defined(__SANITIZE_THREAD__)
	// tests/hostsim's TSan build only: gcc 11's libtsan does not know pthread_cond_clockwait, which a steady-clock wait compiles to
	const auto deadline = std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(std::chrono::duration<double>(lp.patience_s));
	const bool arrived = lp.cv.wait_until(lk, deadline, [&] { return lp.generation != gen || lp.broken; });
#else
	// the steady clock: a step of the wall clock (NTP, an operator) neither fires the patience early nor stretches it
	const bool arrived = lp.cv.wait_for(lk, std::chrono::duration<double>(lp.patience_s), [&] { return lp.generation != gen || lp.broken; });
#endif
	if (!arrived || lp.broken) {
		lp.broken = true;
		lp.cv.notify_all();
		return false;
	}
	return true;
}

void loopback_hook(void* user, double* buf, int count, hipStream_t stream)
{
	DevCtx* c = static_cast<DevCtx*>(user);
	Loopback& lp = *c->red.loop;
	const int me = c->red.rank;
	double tmp[kRedMax];
	stat_add(ST_ALLREDUCE); stat_add(ST_ALLREDUCE_DOUBLES, count);       // per shard of this process, like the RCCL hook
	SQN_HIP_OK(hipStreamSynchronize(stream));
	for (int done = 0; done < count; done += kRedMax) {            // Fisher products reduce fisher_size scalars
		const int k = count - done < kRedMax ? count - done : kRedMax;
		SQN_HIP_OK(hipMemcpy(tmp, buf + done, (size_t) k * sizeof(double), hipMemcpyDeviceToHost));
		std::memcpy(&lp.slots[(size_t) me * kRedMax], tmp, (size_t) k * sizeof(double));
		if (!loop_barrier(lp)) { reducer_failed(user, "a shard did not reach the host-side reduction"); return; }
		for (int j = 0; j < k; j++) {
			double s = 0;
			for (int r = 0; r < lp.nranks; r++) s += lp.slots[(size_t) r * kRedMax + j];
			tmp[j] = s;
		}
		// nobody overwrites a slot before everybody has read it
		if (!loop_barrier(lp)) { reducer_failed(user, "a shard did not reach the host-side reduction"); return; }
		SQN_HIP_OK(hipMemcpy(buf + done, tmp, (size_t) k * sizeof(double), hipMemcpyHostToDevice));
	}
}

void free_view(DevCtx* c, View& v)
{
	if (v.mirror && v.dev) {
		SQN_HIP_OK(hipFree(v.dev));
		c->mirrored.fetch_sub(v.count * sizeof(real));
	}
	v = View{};
}

void destroy(DevCtx* c, bool keep_spill = false)
{
	// every wait of a context whose stream carries RCCL collectives is bounded (wait_stream): a context released or reclaimed
	// while a peer is gone -- without an earlier call that timed out -- must not hang here either
	if (!c->wedged) (void) wait_stream(c, c->sc.stream);
	if (!c->wedged && c->own_stream && c->own_stream != c->sc.stream) (void) wait_stream(c, c->own_stream);
	if (!c->wedged && c->copy_stream) (void) wait_stream(c, c->copy_stream);      // an upload of x started when the last call returned may still be writing the staging vector
	if (!c->wedged && c->down_stream) (void) wait_stream(c, c->down_stream);
	if (c->wedged) {
		// kernels that can never end still point into this context's memory and sit on its streams: nothing of it is freed or
		// destroyed (a leak, said out loud once, instead of a hang or a use-after-free on the device)
		std::fprintf(stderr, "stochqn: a context whose collectives cannot be ended is abandoned with its device memory (%zu bytes of mirrors)\n", c->mirrored.load());
		if (Spill* sp = static_cast<Spill*>(c->spill)) { delete sp; c->spill = nullptr; }
		for (auto& r : c->regs)
			if (r.p) { if (hipHostUnregister(const_cast<void*>(r.p)) != hipSuccess) (void) hipGetLastError(); note_unpinned(r.p); }
		return;                                          // the DevCtx object itself stays too: the reducer's `user` pointer is in flight
	}
	if (Spill* sp = static_cast<Spill*>(c->spill)) {
		// keep_spill: a resume that could not be completed (a mirror could not be had) -- the state goes back where the
		// object's next call looks for it, instead of being lost with the half-made context
		if (keep_spill && !g_spill.count(c->key)) g_spill[c->key] = sp;
		else delete sp;
		c->spill = nullptr;
	}
	c->prof.collect();
	for (int i = 0; i < K_COUNT; i++) { g_retired_ms[i] += c->prof.total_ms[i]; g_retired_launches[i] += c->prof.launches[i]; }
	for (auto& p : c->prof.pending) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
	for (auto e : c->prof.pool) (void) hipEventDestroy(e);
	View* vs[] = {&c->S, &c->Y, &c->sbak, &c->ybak, &c->gprev, &c->xsum, &c->xprev, &c->H0, &c->G, &c->F};
	for (View* v : vs) free_view(c, *v);
	if (c->pool) SQN_HIP_OK(hipFree(c->pool));
	if (c->sc.fisher_part) SQN_HIP_OK(hipFree(c->sc.fisher_part));
	if (c->fisher_t) SQN_HIP_OK(hipFree(c->fisher_t));
	for (real* p : c->stage) if (p) SQN_HIP_OK(hipFree(p));
	for (int i = 0; i < 2; i++) {
		if (c->host_stage[i] && c->host_stage_pinned[i]) SQN_HIP_OK(hipHostFree(c->host_stage[i]));
		else std::free(c->host_stage[i]);
	}
	if (c->pin) SQN_HIP_OK(hipHostFree(c->pin));
	if (c->copy_stream) (void) hipStreamDestroy(c->copy_stream);      // idle: waited for above
	if (c->down_stream) (void) hipStreamDestroy(c->down_stream);
	for (hipEvent_t e : c->xup_ev) (void) hipEventDestroy(e);
	for (hipEvent_t e : c->chunk_ev) (void) hipEventDestroy(e);
	for (hipEvent_t e : c->up_ev) (void) hipEventDestroy(e);
	if (c->carry) SQN_HIP_OK(hipFree(c->carry));
	if (c->spec) SQN_HIP_OK(hipFree(c->spec));
	if (c->x_pre_ev) (void) hipEventDestroy(c->x_pre_ev);
	for (auto& r : c->regs)
		if (r.p) { if (hipHostUnregister(const_cast<void*>(r.p)) != hipSuccess) (void) hipGetLastError(); note_unpinned(r.p); }   // the caller may have freed it already
	if (c->own_stream) SQN_HIP_OK(hipStreamDestroy(c->own_stream));
	delete c;
}

// STOCHQN_HIP_PROFILE=1 in the environment (R / Python users cannot easily call the profile API):
// the event profiler is on from the first context and a per-kernel table goes to stderr at exit.
void at_exit()
{
	if (std::getenv("STOCHQN_HIP_PROFILE")) {
		std::fprintf(stderr, "stochqn: per-kernel device time (HIP events)\n");
		for (int id = 0; id < K_COUNT; id++) {
			long long l = g_retired_launches[id];
			double ms = g_retired_ms[id];
			for (auto& kv : g_ctx) { l += kv.second->prof.launches[id]; ms += kv.second->prof.total_ms[id]; }
			if (l > 0) std::fprintf(stderr, "  %-12s %8lld launches  %12.3f ms total  %10.4f ms avg\n", kernel_name(id), l, ms, ms / (double) l);
		}
	}
	/* device memory dies with the process; destroying streams here can race with the HIP runtime's own
	   teardown, so contexts are simply abandoned */
}

}  // namespace

void stat_add(int id, long long v) { g_stats[id].fetch_add(v, std::memory_order_relaxed); }

namespace { thread_local bool t_hip_failed = false; }

void note_hip_failure(const char* expr, hipError_t e, const char* file, int line)
{
	std::fprintf(stderr, "stochqn: %s failed: %s (%s:%d)\n", expr, hipGetErrorString(e), file, line);
	(void) hipGetLastError();
	t_hip_failed = true;
}

bool take_hip_failure()
{
	const bool f = t_hip_failed;
	t_hip_failed = false;
	return f;
}

namespace {

void views_of(DevCtx* c, View** out)
{
	View* vs[] = {&c->S, &c->Y, &c->sbak, &c->ybak, &c->gprev, &c->xsum, &c->xprev, &c->H0, &c->G, &c->F};
	for (int i = 0; i < 10; i++) out[i] = vs[i];
}

size_t mirror_bytes(DevCtx* c) { return c->mirrored.load(); }

void drop_spill(const void* key)
{
	auto it = g_spill.find(key);
	if (it == g_spill.end()) return;
	delete it->second;
	g_spill.erase(it);
}

void destroy(DevCtx* c, bool keep_spill);

// Spill and destroy the least recently used idle context that holds mirrors ON DEVICE `device` (< 0: on any device: the cap on
// mirrors counts them all).  g_mu held.  false: nothing to reclaim.
bool reclaim_one(int device)
{
	for (;;) {
		DevCtx* victim = nullptr;
		for (auto& kv : g_ctx) {
			DevCtx* c = kv.second;
			if (c->in_call || c->no_spill || c->wedged || mirror_bytes(c) == 0 || (device >= 0 && c->device != device)) continue;
			if (!victim || c->last_use < victim->last_use) victim = c;
		}
		if (!victim) return false;
		const void* key = victim->key;
		// a context that never completed a call holds nothing the caller's own arrays do not; raw contexts re-upload every call
		if (victim->has_last && victim->kind != KIND_RAW) {
			std::unique_ptr<Spill> sp(new Spill());
			sp->kind = victim->kind; sp->n = victim->n; sp->m = victim->m; sp->fsize = victim->fsize;
			sp->niter = victim->last_niter; sp->section = victim->last_section;
			// a victim whose stream reports a failure holds nothing worth keeping a copy of -- and is not touched any further
			bool ok = !victim->wedged && wait_stream(victim, victim->sc.stream) == hipSuccess && !victim->fault;      // bounded like every wait
			if (!ok) (void) hipGetLastError();
			View* vs[10];
			views_of(victim, vs);
			for (View* v : vs) {
				if (!ok) break;
				if (!v->mirror || !v->dev || v->count == 0) continue;
				SpillPiece pc;
				pc.caller = v->caller; pc.count = v->count;
				pc.data = (real*) std::malloc(v->count * sizeof(real));
				if (!pc.data || bounced_copy(pc.data, v->dev, v->count * sizeof(real), true) != hipSuccess) {
					(void) hipGetLastError();
					std::free(pc.data);
					ok = false;
					break;
				}
				sp->bytes += v->count * sizeof(real);
				sp->pieces.push_back(pc);
			}
			if (!ok) { victim->no_spill = true; continue; }     // no host memory for it (or its stream failed): leave it alone, look for another
			drop_spill(key);
			g_spill[key] = sp.release();
		}
		if (std::getenv("STOCHQN_HIP_VERBOSE"))
			std::fprintf(stderr, "stochqn: device memory is short: the idle context of the workspace at %p (%zu MB of mirrors) was moved to host memory\n",
			             key, mirror_bytes(victim) >> 20);
		g_ctx.erase(key);
		destroy(victim, false);
		stat_add(ST_CTX_RECLAIMED);
		return true;
	}
}

}  // namespace

// Option "max_mirror_bytes": keep the mirrors of idle contexts under the cap (the context inside a call is never touched).
void enforce_mirror_cap()
{
	const long cap = options().max_mirror_bytes;
	if (cap <= 0) return;
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	for (;;) {
		size_t total = 0;
		for (auto& kv : g_ctx) total += mirror_bytes(kv.second);
		if (total <= (size_t) cap || !reclaim_one(-1)) return;
	}
}

Options& options()
{
	static const bool env_read = [] {            // environment defaults, for callers that cannot reach set_option (R, Python)
		if (const char* e = std::getenv("STOCHQN_HIP_DEVICES")) g_opt.devices = std::atoi(e);
		if (const char* e = std::getenv("STOCHQN_HIP_VIRTUAL_DEVICES")) g_opt.virtual_devices = std::atoi(e) != 0;
		if (const char* e = std::getenv("STOCHQN_HIP_DEVICES_MIN_N")) g_opt.devices_min_n = std::atol(e);
		return true;
	}();
	(void) env_read;
	return g_opt;
}

bool device_alloc(void** p, size_t bytes)
{
	for (;;) {
		*p = nullptr;
		const hipError_t e = alloc_should_fail() ? hipErrorOutOfMemory : hipMalloc(p, bytes ? bytes : 1);
		if (e == hipSuccess) return true;
		(void) hipGetLastError();
		*p = nullptr;
		// out of device memory (and only then: any other failure would not be cured by it): give up the mirrors of an optimiser
		// ON THIS DEVICE that nobody has called for the longest time, try again
		bool freed = false;
		if (e == hipErrorOutOfMemory) {
			int dev = 0;
			if (hipGetDevice(&dev) != hipSuccess) { (void) hipGetLastError(); dev = -1; }
			std::lock_guard<std::recursive_mutex> lk(g_mu);
			freed = dev >= 0 && reclaim_one(dev);
		}
		if (freed) continue;
		std::fprintf(stderr, "stochqn: could not allocate %zu bytes of device memory (%s)\n", bytes, hipGetErrorString(e));
		return false;
	}
}

bool pinned_alloc(void** p, size_t bytes)
{
	*p = nullptr;
	if (alloc_should_fail() || hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		(void) hipGetLastError();
		*p = nullptr;
		return false;
	}
	return true;
}

int default_grid_cap()
{
	static const int cus = [] {
		int dev = 0, n = 0;
		if (hipGetDevice(&dev) == hipSuccess &&
		    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
			return n > kMaxGrid ? kMaxGrid : n;
		return 256;
	}();
	return cus;
}

void begin_call(DevCtx* c)
{
	// the stream of this call: the context's own, or the NULL stream (whatever was enqueued on the other one has been
	// synchronised at the end of the call that enqueued it, stream-ordered calls excepted: those stay on the NULL stream)
	c->sc.stream = (g_opt.null_stream == 1 || g_opt.async_device) ? nullptr : c->own_stream;      // the automatic rule: open_call
	c->sc.nontemporal = g_opt.nontemporal;
	c->sc.grid_cap = g_opt.grid_cap > 0 ? g_opt.grid_cap : default_grid_cap();
	c->sc.reverse = g_opt.reverse;
	c->sc.rows_split = g_opt.rows_split;
	c->sc.fisher_rows = g_opt.fisher_rows;
	c->sc.fisher_split = g_opt.fisher_split;
	c->sc.fisher_split_per_cu = g_opt.fisher_split_per_cu;
	c->sc.fisher_lag = g_opt.fisher_lag;
	c->sc.fisher_tile = g_opt.fisher_tile;
	c->sc.phase_inv = g_opt.phase_ticks > 0 ? (uint32_t) (4294967296.0 / (double) g_opt.phase_ticks) : 0u;
	c->sc.keep_tail = g_opt.keep_tail;
	c->sc.qdot_per_cu = g_opt.qdot_per_cu; c->sc.sadd_per_cu = g_opt.sadd_per_cu; c->sc.sdot2_per_cu = g_opt.sdot2_per_cu; c->sc.sdot_per_cu = g_opt.sdot_per_cu;
	c->sc.pair_per_cu = g_opt.pair_per_cu;
	c->sc.sdot_tile = g_opt.sdot_tile;
	c->sc.prof = g_profile ? &c->prof : nullptr;
	c->sc.phase = &c->phase;
	c->phase = 1;
}

namespace {
struct Roctx {
	int (*push)(const char*) = nullptr;
	int (*pop)() = nullptr;
} g_roctx;

bool roctx_ready()
{
	static const bool ok = [] {
		if (!std::getenv("STOCHQN_HIP_ROCTX")) return false;
		for (const char* nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
			if (void* h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) {
				g_roctx.push = (int (*)(const char*)) dlsym(h, "roctxRangePushA");
				g_roctx.pop = (int (*)()) dlsym(h, "roctxRangePop");
				if (g_roctx.push && g_roctx.pop) return true;
			}
		}
		std::fprintf(stderr, "stochqn: STOCHQN_HIP_ROCTX is set but no roctx library could be loaded\n");
		return false;
	}();
	return ok;
}
}  // namespace

ApiRange::ApiRange(const char* name, int section)
{
	if (!roctx_ready()) return;
	char label[64];
	std::snprintf(label, sizeof label, "%s section %d", name, section);
	g_roctx.push(label);
	on = true;
}

ApiRange::~ApiRange()
{
	if (on) g_roctx.pop();
}

bool device_ready()
{
	static const bool ready = [] {               // evaluated once, thread-safe (C++11 static initialisation)
		int count = 0;
		const hipError_t e = hipGetDeviceCount(&count);
		if (e != hipSuccess || count <= 0) { (void) hipGetLastError(); return false; }
		return true;
	}();
	return ready;
}

bool is_device_pointer(const void* p)
{
	if (!p) return false;
	hipPointerAttribute_t a;
	hipError_t e = hipPointerGetAttributes(&a, p);
	if (e != hipSuccess) { (void) hipGetLastError(); return false; }   // ordinary malloc / R / numpy memory
	return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

DevCtx* lookup(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	return it == g_ctx.end() ? nullptr : it->second;
}

DevCtx* hold(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end() || it->second->in_call) return nullptr;      // inside a call of another thread: not ours to touch
	it->second->in_call = true;
	return it->second;
}

void forget_rows_of(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it != g_ctx.end()) it->second->forget_rows();
}

DevCtx* acquire(const void* key, int kind, int n, size_t m, size_t fsize, bool* fresh)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	*fresh = false;
	auto it = g_ctx.find(key);
	if (it != g_ctx.end()) {
		DevCtx* c = it->second;
		if (c->kind == kind && c->n == n && c->m == m && c->fsize == fsize) {
			begin_call(c);
			c->last_use = ++g_clock;
			c->in_call = true;
			return c;
		}
		destroy(c, false);       // same address, different problem: the old owner is gone
		g_ctx.erase(it);
	}
	DevCtx* c = new DevCtx();
	c->key = key; c->kind = kind; c->n = n; c->m = m; c->fsize = fsize;
	c->n_global = (double) n;
	c->last_use = ++g_clock;
	c->in_call = true;
	if (hipGetDevice(&c->device) != hipSuccess) { (void) hipGetLastError(); c->device = 0; }
	if (hipStreamCreate(&c->own_stream) != hipSuccess) {      // blocking flavour: ordered after the null stream
		(void) hipGetLastError();
		std::fprintf(stderr, "stochqn: could not create the stream of a device context\n");
		c->own_stream = nullptr;
		delete c;
		return nullptr;
	}
	c->sc.stream = c->own_stream;
	// pool layout: part0 | part1 | red0 | red1 | sy | yy | report | rho | alpha | rows_part x2 | gsy | kap | ticket
	const size_t part = (size_t) kMaxSums * kMaxGrid;
	const size_t rows_part = (size_t) kRedMax * kMaxGrid;
	const size_t total = 2 * part + 2 * kRedMax + 4 * m + 8 + 2 * rows_part + m * m + 3 * m + 1;
	c->pin_count = 16 + 2 * m + fsize + 3 * m + 8;
	c->pool_bytes = total * sizeof(double);
	if (!device_alloc((void**) &c->pool, c->pool_bytes) ||
	    !pinned_alloc((void**) &c->pin, c->pin_count * sizeof(double)) ||
	    (fsize > 0 && (!device_alloc((void**) &c->sc.fisher_part, fsize * kMaxGrid * sizeof(double)) ||
	                   !device_alloc((void**) &c->fisher_t, fsize * sizeof(double))))) {
		std::fprintf(stderr, "stochqn: could not allocate the scratch of a device context\n");
		destroy(c, false);
		return nullptr;
	}
	SQN_HIP_OK(hipMemset(c->pool, 0, total * sizeof(double)));
	double* p = c->pool;
	c->sc.part[0] = p; p += part;
	c->sc.part[1] = p; p += part;
	c->sc.red[0] = p; p += kRedMax;
	c->sc.red[1] = p; p += kRedMax;
	c->sc.sy = p; p += m;
	c->sc.yy = p; p += m;
	c->sc.report = p; p += 8;        // report | rho | alpha are contiguous: one D2H per step
	c->sc.rho = p; p += m;
	c->sc.alpha = p; p += m;
	c->sc.rows_part[0] = p; p += rows_part;
	c->sc.rows_part[1] = p; p += rows_part;
	c->sc.gsy = p; p += m * m;
	c->kap_dev = p; p += 3 * m;
	c->sc.ticket = reinterpret_cast<unsigned*>(p);             // zero (the pool is cleared above), and put back to zero by whoever draws the last ticket
	c->forget_rows();
	begin_call(c);
	c->sc.allreduce = nullptr;
	c->sc.user = c;
	g_ctx[key] = c;
	stat_add(ST_CTX_CREATED);
	if (!g_atexit) {
		g_atexit = true;
		std::atexit(at_exit);
		if (std::getenv("STOCHQN_HIP_PROFILE")) { g_profile = true; c->sc.prof = &c->prof; }
	}
	*fresh = true;
	return c;
}

bool prepare_context(const void* key, int kind, int n, size_t m, size_t fsize, int stages)
{
	bool fresh = false;
	DevCtx* c = acquire(key, kind, n, m, fsize, &fresh);
	if (!c) return false;
	bool ok = true;
	for (int i = 0; i < stages && i < 3 && ok; i++) ok = ensure_stage(c, i);
	end_use(c);
	if (!ok) { release(key); return false; }
	return true;
}

void end_use(DevCtx* c)
{
	if (!c) return;
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	c->in_call = false;
}

// The state a reclaimed context left in host memory, if it is this object's: same shape, and the counters continue where
// that context stopped.  Anything else at this address is a different object: the spill is dropped.
void attach_spill(DevCtx* c, size_t niter, int section)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_spill.find(c->key);
	if (it == g_spill.end()) return;
	Spill* sp = it->second;
	g_spill.erase(it);
	if (sp->kind == c->kind && sp->n == c->n && sp->m == c->m && sp->fsize == c->fsize && sp->niter == niter && sp->section == section)
		c->spill = sp;
	else delete sp;
}

void detach_spill(DevCtx* c)
{
	delete static_cast<Spill*>(c->spill);
	c->spill = nullptr;
}

bool has_spill(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	return g_spill.count(key) != 0;
}

// The call returns with *req == x -- the caller will now evaluate something AT x and must not modify it (reference
// include/stochqn.h:364-366) -- but the device copy of x is stale (the request before was at x_avg: the caller was free to edit
// x) and the next call will need it.  Its upload starts now, on the side stream, and runs while the caller computes; the next
// call finds x in place (stage_x orders itself behind the copy, and still compares the probe values).  Pinned arrays only:
// a copy from pageable memory would be done before this call returned.
static void prefetch_x(DevCtx* c, const real* x)
{
	const Options& o = options();
	if (!o.x_prefetch || o.x_upload != 0 || !x || c->fault || c->kind == KIND_RAW || c->async_call || (c->x_valid && c->x_host == x)) return;
	const size_t n = (size_t) c->n, bytes = n * sizeof(real);
	if ((long) bytes < o.register_min_bytes || is_device_pointer(x)) return;
	if (!ensure_stage(c, 0) || !ensure_copy_stream(c, 1) || !ensure_registered(c, x, bytes)) return;
	if (!c->x_pre_ev && hipEventCreateWithFlags(&c->x_pre_ev, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); c->x_pre_ev = nullptr; return; }
	if (c->x_pre_pending) SQN_HIP_OK(hipStreamWaitEvent(c->copy_stream, c->x_pre_ev, 0));
	SQN_HIP_OK(hipMemcpyAsync(c->stage[0], x, bytes, hipMemcpyHostToDevice, c->copy_stream));
	SQN_HIP_OK(hipEventRecord(c->x_pre_ev, c->copy_stream));
	c->x_pre_pending = true;
	x_handed_back(c, x, n);                          // the probe values of what is on its way
	stat_add(ST_X_PREFETCH);
}

// The invariant of the host path (callers whose arrays live in a garbage-collected heap: reference src/Rwrapper.c:106-123,
// stochqn/pywrapper.pxi:161-172): when a call returns, NOTHING the library enqueued on any of the context's streams is still
// running -- the caller may free x, grad or the requested vector the moment it has them back.  Asked of the runtime itself
// (hipStreamQuery), not of the library's own bookkeeping; every stream found busy counts one "host_copies_in_flight".  The one
// exception is the upload that option x_prefetch starts on purpose, after this check.
static void count_work_in_flight(DevCtx* c)
{
	if (c->async_call || !(c->stage[0] || c->stage[1] || c->copy_stream)) return;      // host callers only (stream-ordered device callers leave work behind by design)
	for (hipStream_t s : {c->own_stream, c->copy_stream, c->down_stream}) {
		if (!s || (s == c->copy_stream && c->x_pre_pending)) continue;
		const hipError_t e = hipStreamQuery(s);
		if (e == hipErrorNotReady) stat_add(ST_WORK_IN_FLIGHT);
		if (e != hipSuccess) (void) hipGetLastError();
	}
}

bool note_state(const void* key, size_t niter, int section, bool req_is_x, const real* x)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) return false;
	count_work_in_flight(it->second);
	it->second->in_call = false;
	detach_spill(it->second);                        // whatever the first call after a reclaim did not pick up is not this object's
	// x is only known to be untouched until the next call while it is what *req designates ("do NOT modify", reference
	// include/stochqn.h:364-366); a request at x_avg leaves the caller free to edit x
	if (!req_is_x) it->second->x_valid = false;
	else if (x && !it->second->fault) prefetch_x(it->second, x);
	it->second->call_index++;
	it->second->has_last = true;
	it->second->last_niter = niter;
	it->second->last_section = section;
	const bool hip_failed = take_hip_failure();       // always taken: the flag must not outlive the call
	const bool fault = it->second->fault || hip_failed;
	it->second->fault = false;
	return fault;
}

void release(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	drop_spill(key);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) return;
	destroy(it->second, false);
	g_ctx.erase(it);
}

void abandon_context(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) return;
	destroy(it->second, true);
	g_ctx.erase(it);
}

void release_all()
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	for (auto& kv : g_ctx) destroy(kv.second, false);
	g_ctx.clear();
	for (auto& kv : g_spill) delete kv.second;
	g_spill.clear();
}

bool bind(DevCtx* c, View& v, real* caller, size_t count, bool import)
{
	if (caller == v.caller && count == v.count && (v.dev || count == 0)) return true;
	free_view(c, v);
	v.caller = caller;
	v.count = count;
	if (!caller || count == 0) return true;
	if (is_device_pointer(caller)) { v.dev = caller; v.mirror = false; return true; }
	if (!device_alloc((void**) &v.dev, count * sizeof(real))) {
		std::fprintf(stderr, "stochqn: could not allocate a %zu-element device mirror\n", count);
		v = View{};
		return false;
	}
	v.mirror = true;
	c->mirrored.fetch_add(count * sizeof(real));
	{
		std::lock_guard<std::recursive_mutex> lk(g_mu);
		ensure_bounce();                         // reclaim_one is going to need it when memory is short
	}
	// a context that was reclaimed while idle comes back from the library's own host copy, not from the caller's (stale) arrays
	if (Spill* sp = static_cast<Spill*>(c->spill)) {
		for (auto& pc : sp->pieces)
			if (pc.caller == caller && pc.count == count && pc.data) {
				hipError_t e;
				{
					std::lock_guard<std::recursive_mutex> lk(g_mu);      // one bounce buffer for the process
					e = bounced_copy(v.dev, pc.data, count * sizeof(real), false);
				}
				if (e != hipSuccess) {                                   // the mirror holds garbage: the call fails, the host copy stays (abandon_context)
					(void) hipGetLastError();
					std::fprintf(stderr, "stochqn: could not bring a reclaimed array back to the device: %s\n", hipGetErrorString(e));
					free_view(c, v);
					return false;
				}
				enforce_mirror_cap();                                    // the host copy stays until the call has completed (detach_spill)
				return true;
			}
	}
	// without `import` the mirror starts with indeterminate contents, like the reference's malloc
	if (import) SQN_HIP_OK(hipMemcpy(v.dev, caller, count * sizeof(real), hipMemcpyHostToDevice));
	enforce_mirror_cap();
	return true;
}

// stochqn_hip_export of a workspace whose context was reclaimed: the caller asks for its arrays to be brought up to date,
// so they are alive -- copy the spilled state into them (the spill stays: the object may still be called again)
bool export_spill(const void* key)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	auto it = g_spill.find(key);
	if (it == g_spill.end()) return false;
	for (auto& pc : it->second->pieces)
		if (pc.data && pc.caller) std::memcpy(const_cast<void*>(pc.caller), pc.data, pc.count * sizeof(real));
	return true;
}

void export_view(DevCtx* c, View& v)
{
	if (v.mirror && v.dev && v.caller)
		SQN_HIP_OK(hipMemcpyAsync(const_cast<void*>(v.caller), v.dev, v.count * sizeof(real), hipMemcpyDeviceToHost, c->sc.stream));
}

bool ensure_stage(DevCtx* c, int which)
{
	return c->stage[which] || device_alloc((void**) &c->stage[which], (size_t) c->n * sizeof(real));
}

real* host_landing(DevCtx* c, int slot)
{
	if (c->host_stage[slot]) return c->host_stage[slot];
	const size_t bytes = (size_t) c->n * sizeof(real);
	c->host_stage_pinned[slot] = pinned_alloc((void**) &c->host_stage[slot], bytes);
	if (!c->host_stage_pinned[slot]) c->host_stage[slot] = (real*) std::malloc(bytes);    // slower copies, same result
	return c->host_stage[slot];
}

// ---- which host ranges may be page-locked in place -------------------------------------------------------------------------------
// hipHostRegister maps the PAGES that contain [p, p + bytes) into the device's address space at their own addresses (measured
// on the MI355X boxes, ROCm 7.0 and 7.2: device pointer == host pointer; tools/pin_probe.hip).  That is safe exactly as long as
// those pages belong to the array and to nothing else, and stay mapped until the unregister.  A block in the program-break heap
// gives neither: its first and last page hold its neighbours' bytes too, and glibc cuts the break back (an munmap of pages
// next to, or -- once the block is freed -- under a registration) whenever the top of the heap falls free.  Round 4 lost a test
// process twice to a GPU memory-access fault at a break-heap address.  Arrays with a mapping of their own -- what
// malloc / numpy / R hand out above the mmap threshold, what stochqn_amd/free.py makes for its own arrays -- start on a page of
// their own and are unmapped as a whole by their owner, after the owner has unpinned them.
bool g_pin_probe_by_maps = false;                    // runtime.hpp: a switch for tests/hostsim (the fallback of peek_words)
namespace {
std::mutex g_span_mu;
std::map<uintptr_t, uintptr_t> g_spans;            // first page -> one past the last page, of every range this library registered

uintptr_t break_heap_start()
{
	static std::atomic<uintptr_t> start{0};
	uintptr_t s = start.load(std::memory_order_relaxed);
	if (s) return s;
	if (FILE* f = std::fopen("/proc/self/maps", "r")) {     // the start of [heap] never moves; the break (its end) is sbrk(0)
		char line[512];
		while (std::fgets(line, sizeof line, f))
			if (std::strstr(line, "[heap]")) { s = (uintptr_t) std::strtoull(line, nullptr, 16); break; }
		std::fclose(f);
	}
	if (s) start.store(s, std::memory_order_relaxed);
	return s;
}

// four words at `at`, read without a fault if there is nothing to read: process_vm_readv on the process itself says EFAULT;
// where that call is not allowed (EPERM / ENOSYS under some seccomp profiles) /proc/self/maps says whether the page is readable
__attribute__((no_sanitize("address", "thread"))) bool peek_words(uintptr_t at, uintptr_t out[4])
{
	if (!g_pin_probe_by_maps) {
		struct iovec local{out, 4 * sizeof(uintptr_t)}, remote{(void*) at, 4 * sizeof(uintptr_t)};
		const ssize_t got = process_vm_readv(getpid(), &local, 1, &remote, 1, 0);
		if (got == (ssize_t) (4 * sizeof(uintptr_t))) return true;
		if (got >= 0 || errno == EFAULT) return false;
	}
	bool readable = false;
	if (FILE* f = std::fopen("/proc/self/maps", "r")) {
		char line[512];
		while (std::fgets(line, sizeof line, f)) {
			char* dash = nullptr;
			const uintptr_t a = (uintptr_t) std::strtoull(line, &dash, 16);
			if (!dash || *dash != '-') continue;
			char* sp = nullptr;
			const uintptr_t b = (uintptr_t) std::strtoull(dash + 1, &sp, 16);
			if (at + 4096 <= a) break;                              // the list is sorted by address
			if (a <= at && at + 4096 <= b) { readable = sp && sp[0] == ' ' && sp[1] == 'r'; break; }
		}
		std::fclose(f);
	}
	if (!readable) return false;
	const volatile uintptr_t* h = (const volatile uintptr_t*) at;
	for (int i = 0; i < 4; i++) out[i] = h[i];
	return true;
}

// glibc gives every thread but the first an arena of its own, and those grow in "heaps": anonymous mappings aligned to their
// maximum size (64 MiB on 64-bit), each beginning with {arena, previous heap, size in use, size made read-write}.  A block
// below the mmap threshold that a worker thread allocated lives in one: it shares its first and last page with its neighbours
// and sits under a top that glibc gives back (madvise, or munmap of the whole heap) as it does with the break.  Recognised by
// that header -- five conditions on four words; a wrong "yes" costs the staged copy path and nothing else, a wrong "no" is the
// behaviour before this check.
bool inside_a_thread_arena_heap(uintptr_t lo, uintptr_t hi)
{
	constexpr uintptr_t kHeap = (uintptr_t) 64 << 20;
	const uintptr_t base = lo & ~(kHeap - 1);
	if (hi > base + kHeap) return false;                            // a block never spans two heaps
	uintptr_t h[4];
	if (!peek_words(base, h)) return false;
	const uintptr_t arena = h[0], prev = h[1], size = h[2], prot = h[3];
	const uintptr_t arena_off = arena & (kHeap - 1);
	return arena && (arena & 7) == 0 && arena_off >= 32 && arena_off <= 256        // an arena begins right after the header of its first heap
	       && (prev & (kHeap - 1)) == 0 && (arena - arena_off == base ? prev == 0 : true)
	       && size && (size & 4095) == 0 && (prot & 4095) == 0 && size <= prot && prot <= kHeap && hi <= base + prot;
}
}  // namespace

// The verdict and the RESERVATION of the pages are one critical section: on "yes" the span [lo, hi) is on the books
// before anybody else can ask about a range that shares a page with it -- two threads asking at the same time about two ranges
// with a page in common cannot both be told yes.  Who is told yes and then fails to register gives the span back
// (note_unpinned).  `own`: a range of the asker's own that the new one REPLACES (the same array, longer now): its span does not
// count against the new range and is taken off the books in the same breath.
bool pinnable_in_place(const void* p, size_t bytes, const void* own)
{
	static const bool trace = std::getenv("STOCHQN_HIP_PIN_TRACE") != nullptr;
	const uintptr_t lo = (uintptr_t) p & ~(uintptr_t) 4095, hi = ((uintptr_t) p + bytes + 4095) & ~(uintptr_t) 4095;
	const uintptr_t heap = break_heap_start(), brk_now = (uintptr_t) sbrk(0);
	bool ok = !(heap && lo < brk_now && hi > heap);
	if (!ok && trace) std::fprintf(stderr, "stochqn: pin %p +%zu declined: inside the break heap [%#lx, %#lx)\n", p, bytes, (unsigned long) heap, (unsigned long) brk_now);
	if (ok && inside_a_thread_arena_heap(lo, hi)) {
		ok = false;
		if (trace) std::fprintf(stderr, "stochqn: pin %p +%zu declined: inside a heap of one of glibc's thread arenas (at %#lx)\n", p, bytes, (unsigned long) (lo & ~(((uintptr_t) 64 << 20) - 1)));
	}
	if (ok) {
		std::lock_guard<std::mutex> lk(g_span_mu);
		const uintptr_t own_lo = own ? ((uintptr_t) own & ~(uintptr_t) 4095) : 0;
		const std::pair<const uintptr_t, uintptr_t>* hit = nullptr;
		// every span that can reach into [lo, hi): the ones that start inside it, and the one before
		auto it = g_spans.lower_bound(lo);
		if (it != g_spans.begin()) { auto before = std::prev(it); if (before->second > lo && !(own && before->first == own_lo)) hit = &*before; }
		for (; !hit && it != g_spans.end() && it->first < hi; ++it)
			if (!(own && it->first == own_lo)) hit = &*it;
		if (hit) {
			ok = false;
			if (trace) std::fprintf(stderr, "stochqn: pin %p +%zu declined: its pages [%#lx, %#lx) overlap the pinned range [%#lx, %#lx)\n", p, bytes,
			                        (unsigned long) lo, (unsigned long) hi, (unsigned long) hit->first, (unsigned long) hit->second);
		} else {
			if (own) g_spans.erase(own_lo);
			g_spans[lo] = hi;                               // reserved: given back by note_unpinned if the registration fails
		}
	}
	if (!ok) stat_add(ST_HOST_PIN_DECLINED);
	return ok;
}

void note_unpinned(const void* p)
{
	std::lock_guard<std::mutex> lk(g_span_mu);
	g_spans.erase((uintptr_t) p & ~(uintptr_t) 4095);
}

bool ensure_registered(DevCtx* c, const void* p, size_t bytes)
{
	const Options& o = options();
	if (!p || (long) bytes < o.register_min_bytes) return false;
	for (auto& r : c->regs)
		if (r.p == p) {
			if (r.bytes >= bytes) return true;
			if (hipHostUnregister(const_cast<void*>(r.p)) != hipSuccess) (void) hipGetLastError();
			note_unpinned(r.p);
			r = DevCtx::HostRange{};
		}
	// pinned by its owner (hipHostMalloc, the caller's own hipHostRegister, stochqn_hip_pin_host)?  then it is as fast as it gets
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) == hipSuccess) { if (a.type == hipMemoryTypeHost) return true; }
	else (void) hipGetLastError();
	if (!o.register_host) return false;          // the library pins nothing by itself unless told to (runtime.hpp: register_host)
	// ... and then only an array that was already there, at this address, in an earlier call: a caller that hands over a fresh
	// array every time would pay a registration of n words per call for nothing
	DevCtx::Seen* mine = nullptr;
	DevCtx::Seen* oldest = &c->seen[0];
	for (auto& s : c->seen) {
		if (s.p == p) mine = &s;
		if (s.call < oldest->call) oldest = &s;
	}
	if (!mine) { *oldest = DevCtx::Seen{p, c->call_index}; return false; }
	if (mine->call == c->call_index) return false;             // first sighting was in this very call
	mine->call = c->call_index;
	if (!pinnable_in_place(p, bytes)) return false;          // a block in a malloc heap, or pages shared with another pin: left to the runtime's pageable path
	DevCtx::HostRange& slot = c->regs[c->reg_turn++ % (int) (sizeof(c->regs) / sizeof(c->regs[0]))];
	if (slot.p) { if (hipHostUnregister(const_cast<void*>(slot.p)) != hipSuccess) (void) hipGetLastError(); note_unpinned(slot.p); slot = DevCtx::HostRange{}; }
	if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
		(void) hipGetLastError();                        // not fatal: the runtime's staged copies still work
		note_unpinned(p);                                // the pages pinnable_in_place reserved
		return false;
	}
	slot.p = p;
	slot.bytes = bytes;
	stat_add(ST_HOST_REGISTERED);
	return true;
}

real* stage_in(DevCtx* c, int which, real* caller, size_t count, bool host)
{
	if (!host) return caller;
	if (!ensure_stage(c, which)) return nullptr;
	(void) ensure_registered(c, caller, count * sizeof(real));
	SQN_HIP_OK(hipMemcpyAsync(c->stage[which], caller, count * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	return c->stage[which];
}

static inline size_t probe_index(int j, size_t count)
{
	return count <= 1 ? 0 : (size_t) (((unsigned __int128) j * (count - 1)) / (DevCtx::kProbe - 1));
}

// Option "x_upload" = 0 (the caller vouches that it does not touch x while *req designates it): is the device copy of x still
// what the caller holds?  Orders the stream behind an upload of x that was started when the last call returned.
bool x_is_current(DevCtx* c, const real* caller, size_t count)
{
	if (c->x_pre_pending) {                  // whatever touches the staging vector comes after that upload
		SQN_HIP_OK(hipStreamWaitEvent(c->sc.stream, c->x_pre_ev, 0));
		c->x_pre_pending = false;
	}
	bool current = options().x_upload == 0 && c->kind != KIND_RAW && c->x_valid && c->x_host == caller;
	if (current) {
		// belt and braces: the values at kProbe spread-out positions must still be the ones handed back (a caller that
		// rescales, projects or resets x between calls is caught here and simply gets its x uploaded)
		for (int j = 0; j < DevCtx::kProbe && current; j++) {
			const double v = (double) caller[probe_index(j, count)];
			current = std::memcmp(&v, &c->x_probe[j], sizeof v) == 0;
		}
	}
	if (current) stat_add(ST_X_UPLOAD_SKIPPED);
	else { c->x_valid = false; stat_add(ST_X_UPLOAD); }
	return current;
}

real* stage_x(DevCtx* c, real* caller, size_t count)
{
	if (!ensure_stage(c, 0)) return nullptr;
	if (x_is_current(c, caller, count)) return c->stage[0];
	(void) ensure_registered(c, caller, count * sizeof(real));
	SQN_HIP_OK(hipMemcpyAsync(c->stage[0], caller, count * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	return c->stage[0];
}

// ---- checksum of a host buffer (sqn_device.hpp: XHash) -----------------------------------------------------------------------
void xhash_host(const void* buf, size_t bytes, size_t word_lo, size_t word_hi, XHash* out)
{
	const unsigned char* p = static_cast<const unsigned char*>(buf);
	const size_t whole = bytes / 8;                      // full words; a partial last one (float vectors of odd length) is zero-extended
	unsigned long long a0 = 0, a1 = 0, b0 = 0, b1 = 0;
	size_t i = word_lo;
	const size_t hi_whole = word_hi < whole ? word_hi : whole;
	for (; i + 1 < hi_whole; i += 2) {
		unsigned long long w0, w1;
		std::memcpy(&w0, p + 8 * i, 8);
		std::memcpy(&w1, p + 8 * i + 8, 8);
		xhash_word(w0, i, a0, b0);
		xhash_word(w1, i + 1, a1, b1);
	}
	for (; i < hi_whole; i++) {
		unsigned long long w;
		std::memcpy(&w, p + 8 * i, 8);
		xhash_word(w, i, a0, b0);
	}
	if (word_hi > whole && word_lo <= whole && bytes > 8 * whole) {
		unsigned long long w = 0;
		std::memcpy(&w, p + 8 * whole, bytes - 8 * whole);       // little-endian: the low bytes
		xhash_word(w, whole, a0, b0);
	}
	out->a = a0 + a1;
	out->b = b0 + b1;
}

struct XHashJob {
	std::vector<std::thread> team;
	std::vector<XHash> part;
};

XHashJob* xhash_start(const void* buf, size_t bytes, int threads)
{
	const size_t words = xhash_words(bytes);
	if (threads < 1) threads = 1;
	if ((size_t) threads > words / 4096 + 1) threads = (int) (words / 4096 + 1);
	XHashJob* job = new (std::nothrow) XHashJob;
	if (!job) return nullptr;
	try {
		job->part.resize((size_t) threads);
		const size_t per = ((words + (size_t) threads - 1) / (size_t) threads + 1) & ~(size_t) 1;     // even: the pairs of the inner loop stay pairs
		for (int t = 0; t < threads; t++) {
			const size_t lo = (size_t) t * per < words ? (size_t) t * per : words, hi = lo + per < words ? lo + per : words;
			XHash* out = &job->part[(size_t) t];
			job->team.emplace_back([=] { xhash_host(buf, bytes, lo, hi, out); });
		}
	} catch (...) {                                      // no thread to be had: whoever asked uploads x instead
		for (auto& th : job->team) th.join();
		delete job;
		return nullptr;
	}
	return job;
}

XHash xhash_finish(XHashJob* job)
{
	XHash h;
	for (auto& th : job->team) th.join();
	for (const XHash& p : job->part) { h.a += p.a; h.b += p.b; }
	delete job;
	return h;
}

int xhash_threads(const DevCtx* c)
{
	int t = options().hash_threads;
	if (t <= 0) {
		const unsigned hw = std::thread::hardware_concurrency();
		t = hw >= 16 ? 8 : (hw >= 2 ? (int) hw / 2 : 1);
		if (c->n > 0 && c->n_global > (double) c->n) {           // a shard of a group: the shards share the team
			const double shards = std::ceil(c->n_global / (double) c->n);
			t = (int) ((double) t / shards);
		}
	}
	return t < 1 ? 1 : (t > 64 ? 64 : t);
}

void x_handed_back(DevCtx* c, const real* caller, size_t count)
{
	c->x_host = caller;
	c->x_valid = !c->fault;
	for (int j = 0; j < DevCtx::kProbe; j++) c->x_probe[j] = (double) caller[probe_index(j, count)];
}

bool ensure_upload_slices(DevCtx* c, int slices, size_t carry_count)
{
	if (!ensure_copy_stream(c, 1)) return false;
	while ((int) c->up_ev.size() < slices) {
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); return false; }
		c->up_ev.push_back(e);
	}
	if (c->carry_count < carry_count) {
		double* bigger = nullptr;                        // the old one stays until the new one exists: a sliced pass of this very call may already count on it
		if (!device_alloc((void**) &bigger, carry_count * sizeof(double))) return false;
		if (c->carry) SQN_HIP_OK(hipFree(c->carry));
		c->carry = bigger;
		c->carry_count = carry_count;
	}
	return true;
}

bool ensure_copy_stream(DevCtx* c, int chunks)
{
	if (!c->copy_stream && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); c->copy_stream = nullptr; return false; }
	if (!c->down_stream && hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); c->down_stream = nullptr; return false; }
	for (std::vector<hipEvent_t>* evs : {&c->chunk_ev, &c->xup_ev})
		while ((int) evs->size() < chunks) {
			hipEvent_t e;
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); return false; }
			evs->push_back(e);
		}
	return true;
}

// Waiting for a stream that carries RCCL all-reduces (one process per GPU, stochqn_hip_comm_init; or the shards of one process).
// A collective whose peer never arrives -- a rank that failed on its own and left the call, a process that died, a fabric that
// never came up at first contact -- does not return an error: its kernel spins on the device and hipStreamSynchronize waits
// for it for ever.  So the wait is bounded: the stream is polled (spinning like hipStreamSynchronize itself for the first 50 ms, then every 200 us),
// the communicator's asynchronous error state is looked at, and after "reducer_patience_s" seconds the communicator is ABORTED
// (ncclCommAbort ends the kernels that wait), marked dead for every later reduction, and the call fails (-1000) -- on every
// rank that was waiting for the one that is gone.  Contexts without an RCCL reducer wait with hipStreamSynchronize as before.
hipError_t wait_stream(DevCtx* c, hipStream_t s)
{
	if (c->sc.allreduce != allreduce_hook || !c->red.comm) return hipStreamSynchronize(s);
	if (c->wedged) return hipErrorNotReady;            // its collectives can never end (below): nobody waits for this stream again
	// A communicator that was aborted -- earlier in this very call (allreduce_hook's failure path), by another context or thread
	// that shares it, by another shard of the group -- has been RELEASED by ncclCommAbort: it must not be handed to RCCL again
	// (ncclCommGetAsyncError on freed memory).  Its kernels were ended by the abort, so the stream drains.
	if (comm_is_dead(c->red.comm)) {
		c->fault = true;
		if (g_comm.CommAbort) return hipStreamSynchronize(s);
		c->wedged = true;                                  // dead and never aborted (this RCCL cannot): whether this stream can drain is unknowable
		return hipErrorNotReady;
	}
	const auto t0 = std::chrono::steady_clock::now();
	const double patience = options().reducer_patience_s;
	const char* why = nullptr;
	bool slow = false, dead_meanwhile = false;
	for (long spin = 0;; spin++) {
		const hipError_t q = hipStreamQuery(s);
		if (q != hipErrorNotReady) return q;
		(void) hipGetLastError();
		if (!slow && (spin & 255) != 255) continue;        // the clock is read every 256 polls (a poll is ~1 us)
		const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (waited > 50e-3) {                              // no step takes this long: from here on something is slow or gone -- stop burning the core
			{
				// the liveness test and the call into the communicator under ONE lock: whoever aborts does so under g_dead_mu too
				std::lock_guard<std::mutex> lk(g_dead_mu);
				if (std::find(g_dead_comms.begin(), g_dead_comms.end(), c->red.comm) != g_dead_comms.end()) { dead_meanwhile = true; break; }
				if (g_comm.CommGetAsyncError) {
					ncclResult_t st = ncclSuccess;
					if (g_comm.CommGetAsyncError((ncclComm_t) c->red.comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress) { why = "the communicator reports an asynchronous error"; break; }
				}
			}
			if (waited > patience) { why = "a reduction did not complete within reducer_patience_s"; break; }
			std::this_thread::sleep_for(std::chrono::microseconds(200));
			slow = true;                                   // ... one poll per sleep from here on, the clock read every time
		}
	}
	c->fault = true;
	if (dead_meanwhile) {                                  // somebody else gave up on the same communicator while this context waited
		std::fprintf(stderr, "stochqn: rank %d of %d: the communicator was given up by another context while this one waited -- the call fails (-1000)\n", c->red.rank, c->red.nranks);
		if (g_comm.CommAbort) return hipStreamSynchronize(s);
		c->wedged = true;
		return hipErrorNotReady;
	}
	if (!g_comm.CommAbort) {
		// This RCCL has no ncclCommAbort: nothing can end the kernels that wait for the missing peer, the stream will never drain.
		// Saying "the call fails" and then synchronising on it would block for ever.  The communicator is dead for
		// every later reduction, the context is WEDGED: its calls fail at once, its streams are never waited for again and its
		// device memory is abandoned with it (the kernels in flight still point into it).
		std::fprintf(stderr, "stochqn: rank %d of %d: %s, and this RCCL has no ncclCommAbort -- the context is abandoned; this call and every later one on it fail (-1000)\n", c->red.rank, c->red.nranks, why);
		std::lock_guard<std::mutex> lk(g_dead_mu);
		if (std::find(g_dead_comms.begin(), g_dead_comms.end(), c->red.comm) == g_dead_comms.end()) g_dead_comms.push_back(c->red.comm);
		c->wedged = true;
		return hipErrorNotReady;
	}
	std::fprintf(stderr, "stochqn: rank %d of %d: %s -- aborting the communicator; this call and every later reduction over it fail (-1000)\n", c->red.rank, c->red.nranks, why);
	{
		std::lock_guard<std::mutex> lk(g_dead_mu);
		if (std::find(g_dead_comms.begin(), g_dead_comms.end(), c->red.comm) == g_dead_comms.end()) {
			g_dead_comms.push_back(c->red.comm);
			(void) g_comm.CommAbort((ncclComm_t) c->red.comm);       // ends the kernels that wait for the missing peer
		}
	}
	return hipStreamSynchronize(s);                    // what was enqueued behind the collective drains now
}

// the arrival counter of the pair kernels' last-workgroup verdict must be zero when such a kernel starts, and only the workgroup
// that draws the last ticket puts it back: a launch that faulted or was cut short leaves it non-zero, no workgroup of the NEXT
// pair would draw grid-1, report[4..7] would keep the previous pair's sums and the host would decide on those.
// Whenever a call ends in a fault the counter is cleared on the (now idle) stream before anything else is enqueued.
static void reset_ticket(DevCtx* c)
{
	if (c->wedged || !c->sc.ticket) return;
	if (hipMemsetAsync(c->sc.ticket, 0, sizeof(unsigned), c->sc.stream) != hipSuccess) (void) hipGetLastError();
}

void sync(DevCtx* c)
{
	// a kernel that could not be launched (or failed) leaves its outputs stale: never hand that back
	// as a result -- the call reports invalid_input / -1000 instead (machines.cpp: after_call)
	if (c->async_call && !c->copy_busy && !c->sc.prof && mirror_bytes(c) == 0) {
		// stream-ordered call (option "async_device"): only what can be known without waiting
		const hipError_t l0 = hipGetLastError();
		if (l0 != hipSuccess) { std::fprintf(stderr, "stochqn: device work failed: %s\n", hipGetErrorString(l0)); c->fault = true; }
		return;
	}
	hipError_t e = wait_stream(c, c->sc.stream);
	if (c->copy_busy) {                                  // slices of x still on their way to the host
		// the copy streams wait for events of the compute stream: behind a collective whose peer is gone they are just as stuck,
		// so they are waited for with the same bound
		for (hipStream_t s : {c->copy_stream, c->down_stream}) {
			if (!s) continue;
			const hipError_t e2 = wait_stream(c, s);
			if (e == hipSuccess) e = e2;
		}
		c->copy_busy = false;
	}
	hipError_t l = hipGetLastError();
	if (g_inject_device_fault.exchange(0)) l = hipErrorLaunchFailure;
	if (e != hipSuccess || l != hipSuccess) {
		std::fprintf(stderr, "stochqn: device work failed: %s\n", hipGetErrorString(e != hipSuccess ? e : l));
		c->fault = true;
	}
	if (c->fault) reset_ticket(c);
	if (c->sc.prof && !c->wedged) c->prof.collect();
}

// The reducer a context created by the calling thread gets: the thread's own binding when one was
// installed (shard workers of the single-process multi-device mode, loop-back test threads), else
// the process-wide communicator (one process per GPU), else none.
Reducer current_reducer()
{
	if (t_reducer.kind != Reducer::NONE) return t_reducer;
	Reducer r;
	if (g_comm.comm) { r.kind = Reducer::RCCL; r.comm = g_comm.comm; r.rank = g_comm.rank; r.nranks = g_comm.nranks; }
	else if (g_custom.fn) { r.kind = Reducer::CUSTOM; r.rank = g_custom.rank; r.nranks = g_custom.nranks; }
	return r;
}

void set_thread_reducer(const Reducer& r) { t_reducer = r; }

int comm_nranks()
{
	const Reducer r = current_reducer();
	return r.kind == Reducer::NONE ? 1 : r.nranks;
}

void comm_attach(DevCtx* c)
{
	c->red = current_reducer();
	if (c->red.kind == Reducer::NONE || (c->red.kind == Reducer::LOOP && c->red.nranks <= 1)) {
		c->red = Reducer{};
		c->sc.allreduce = nullptr;
		c->n_global = (double) c->n;
		return;
	}
	c->sc.allreduce = c->red.kind == Reducer::LOOP ? loopback_hook : (c->red.kind == Reducer::CUSTOM ? custom_hook : allreduce_hook);
	// global problem size for the ||dir|| > 1e3*n guard (reference src/stochqn.c:829)
	double nn = (double) c->n;
	SQN_HIP_OK(hipMemcpyAsync(c->sc.red[0], &nn, sizeof(double), hipMemcpyHostToDevice, c->sc.stream));
	c->sc.allreduce(c, c->sc.red[0], 1, c->sc.stream);
	SQN_HIP_OK(hipMemcpyAsync(&nn, c->sc.red[0], sizeof(double), hipMemcpyDeviceToHost, c->sc.stream));
	SQN_HIP_OK(wait_stream(c, c->sc.stream));               // first contact with the peers: bounded like every later wait
	c->n_global = nn;
}

// ---- RCCL communicators for the shards of one process (single-process multi-device mode) --------
bool comm_init_all(int ndev, const int* devices, void** comms_out)
{
	if (!load_rccl()) return false;
	std::vector<ncclComm_t> comms((size_t) ndev);
	ncclResult_t r = g_comm.CommInitAll(comms.data(), ndev, devices);
	if (r != ncclSuccess) {
		std::fprintf(stderr, "stochqn: ncclCommInitAll over %d devices failed: %s\n", ndev, g_comm.GetErrorString ? g_comm.GetErrorString(r) : "?");
		return false;
	}
	for (int i = 0; i < ndev; i++) comms_out[i] = comms[(size_t) i];
	return true;
}

// true: the communicator is not to be destroyed -- it was aborted (ncclCommAbort released it), or it is dead and COULD not be
// aborted (an RCCL without ncclCommAbort): collectives that can never end still refer to it, so it is left where it is
static bool forget_dead(void* comm)
{
	std::lock_guard<std::mutex> lk(g_dead_mu);
	auto it = std::find(g_dead_comms.begin(), g_dead_comms.end(), comm);
	if (it == g_dead_comms.end()) return false;
	g_dead_comms.erase(it);
	return true;
}

void comm_destroy(void* comm)
{
	if (comm && !forget_dead(comm) && g_comm.CommDestroy) g_comm.CommDestroy((ncclComm_t) comm);
}

}  // namespace sqn

// ------------------------------------------------------------------------------------------------
// C entry points of stochqn_hip.h that do not involve the optimiser arithmetic
// ------------------------------------------------------------------------------------------------
using namespace sqn;

extern "C" {

int stochqn_hip_available(void) { return device_ready() ? 1 : 0; }

// ---- host arrays pinned by whoever owns them ------------------------------------------------------------
namespace {
struct Pin { size_t bytes = 0; int refs = 0; };
std::mutex g_pin_mu;
std::unordered_map<const void*, Pin> g_pins;
}

// STOCHQN_HIP_PIN_TRACE=1: every pin / unpin request and what became of it, on stderr (which arrays of a caller got page-locked,
// which were left pageable and why)
static int pin_says(void* p, size_t bytes, int rc, const char* why)
{
	static const bool on = std::getenv("STOCHQN_HIP_PIN_TRACE") != nullptr;
	if (on) std::fprintf(stderr, "stochqn: pin %p +%zu -> %d (%s)\n", p, bytes, rc, why);
	return rc;
}

int stochqn_hip_pin_host(void* p, size_t bytes)
{
	if (!p || bytes == 0 || !device_ready()) return -1;
	std::lock_guard<std::mutex> lk(g_pin_mu);
	auto it = g_pins.find(p);
	if (it != g_pins.end() && it->second.bytes >= bytes) { it->second.refs++; return pin_says(p, bytes, 0, "pinned here already: once more"); }
	if (it != g_pins.end()) {                                  // the same array, longer now: pinned anew
		// the verdict on the longer range FIRST (its own shorter span does not count against it): a decline leaves the shorter pin
		// and every holder's reference where they were -- their later unpin still finds them
		const size_t old_bytes = it->second.bytes;
		if (!pinnable_in_place(p, bytes, p)) return pin_says(p, bytes, 1, "declined: malloc heap or shared pages (the shorter pin stays)");
		if (hipHostUnregister(p) != hipSuccess) (void) hipGetLastError();
		if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
			(void) hipGetLastError();
			note_unpinned(p);
			// back to what the holders had: the shorter range, if the runtime still takes it
			if (pinnable_in_place(p, old_bytes) && hipHostRegister(p, old_bytes, hipHostRegisterPortable) == hipSuccess) return pin_says(p, bytes, -1, "hipHostRegister of the longer range failed: the shorter pin is back");
			(void) hipGetLastError();
			note_unpinned(p);
			g_pins.erase(it);
			return pin_says(p, bytes, -1, "hipHostRegister failed");
		}
		it->second.bytes = bytes;
		it->second.refs++;
		return pin_says(p, bytes, 0, "pinned anew, longer");
	}
	hipPointerAttribute_t a;                                   // ordinary memory: an error (older runtimes) or "unregistered" (ROCm 6+)
	if (hipPointerGetAttributes(&a, p) == hipSuccess) {
		if (a.type == hipMemoryTypeHost) { stat_add(ST_HOST_PIN_FOREIGN); return pin_says(p, bytes, 1, "page-locked by other means"); }
		if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) return pin_says(p, bytes, -1, "not host memory");
	} else (void) hipGetLastError();
	if (!pinnable_in_place(p, bytes)) return pin_says(p, bytes, 1, "declined: malloc heap or shared pages");                // stays pageable (works, a little slower)
	const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
	if (e != hipSuccess) { (void) hipGetLastError(); note_unpinned(p); stat_add(ST_HOST_PIN_ERRORS); return pin_says(p, bytes, -1, hipGetErrorString(e)); }
	g_pins[p] = Pin{bytes, 1};
	stat_add(ST_HOST_REGISTERED);
	return pin_says(p, bytes, 0, "pinned");
}

// Option "x_prefetch" leaves an upload of the caller's x in flight when a call returns.  A caller that drops x right then (its
// finaliser unpins, the allocator unmaps) would pull the pages from under that copy: the unpin waits for such uploads first.
static void drain_prefetches()
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	for (auto& kv : g_ctx)
		if (kv.second->x_pre_pending && kv.second->copy_stream && hipStreamSynchronize(kv.second->copy_stream) != hipSuccess) (void) hipGetLastError();
}

int stochqn_hip_unpin_host(void* p)
{
	if (!p) return -1;
	drain_prefetches();
	std::lock_guard<std::mutex> lk(g_pin_mu);
	auto it = g_pins.find(p);
	if (it == g_pins.end()) return -1;
	if (--it->second.refs > 0) return 0;
	g_pins.erase(it);
	note_unpinned(p);
	if (hipHostUnregister(p) != hipSuccess) {               // the range stays page-locked in the runtime's books: say so, the owner is about to free it
		std::fprintf(stderr, "stochqn: hipHostUnregister(%p) failed: %s\n", p, hipGetErrorString(hipGetLastError()));
		stat_add(ST_HOST_UNPIN_FAILED);
		return -1;
	}
	return 0;
}

// ---- host arrays the library hands out: a mapping of their own, pinned (include/stochqn_hip.h) ----------------------------
namespace {
std::mutex g_host_alloc_mu;
std::unordered_map<const void*, size_t> g_host_allocs;      // what stochqn_hip_alloc_host gave out -> bytes mapped
}

void* stochqn_hip_alloc_host(size_t bytes, int* pinned)
{
	if (pinned) *pinned = 0;
	if (bytes == 0) return nullptr;
	const size_t mapped = (bytes + 4095) & ~(size_t) 4095;
	void* p = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (p == MAP_FAILED) return nullptr;
	{
		std::lock_guard<std::mutex> lk(g_host_alloc_mu);
		g_host_allocs[p] = mapped;
	}
	const int rc = stochqn_hip_pin_host(p, mapped);          // 0 pinned; 1 declined / foreign; -1 no device: the memory works either way
	if (pinned) *pinned = rc == 0;
	return p;
}

int stochqn_hip_free_host(void* p, size_t bytes)
{
	(void) bytes;                                            // the mapped size is on the books
	size_t mapped = 0;
	{
		std::lock_guard<std::mutex> lk(g_host_alloc_mu);
		auto it = g_host_allocs.find(p);
		if (it == g_host_allocs.end()) return -1;
		mapped = it->second;
		g_host_allocs.erase(it);
	}
	(void) stochqn_hip_unpin_host(p);                        // -1 when the pin was declined: nothing to undo
	return munmap(p, mapped) == 0 ? 0 : -1;
}

void stochqn_hip_invalidate(const void* s_mem)
{
	if (group_invalidate(s_mem)) return;
	if (!s_mem) return;
	for (const void* key : {s_mem, raw_key(s_mem)}) forget_rows_of(key);
}

void stochqn_hip_release(const void* s_mem)
{
	if (!s_mem || group_release(s_mem)) return;
	release(s_mem);
	release(raw_key(s_mem));
}
void stochqn_hip_release_all(void) { group_release_all(); release_all(); }

int stochqn_hip_devices_active(const void* s_mem) { return group_shards(s_mem); }
int stochqn_hip_devices_reducer(const void* s_mem) { return group_reducer_kind(s_mem); }
int stochqn_hip_devices_layout(const void* s_mem, int shard, int* device, size_t* offset, size_t* count)
{
	return group_layout(s_mem, shard, device, offset, count);
}
int stochqn_hip_devices_bind(const void* s_mem, int shard, real_t* x, real_t* grad, real_t* hess_vec)
{
	return group_bind(s_mem, shard, x, grad, hess_vec);
}
int stochqn_hip_devices_request(const void* s_mem, int shard, real_t** req, real_t** req_vec)
{
	return group_request(s_mem, shard, req, req_vec);
}
int stochqn_hip_devices_foreach(const void* s_mem, stochqn_hip_shard_fn fn, void* user)
{
	return group_foreach(s_mem, fn, user);
}

int stochqn_hip_export(const void* s_mem)
{
	int grc = 0;
	if (group_export(s_mem, &grc)) return grc;
	DevCtx* c = hold(s_mem);                             // not reclaimed from under the copies (another thread's failed allocation)
	if (!c) return export_spill(s_mem) ? 0 : -1000;      // reclaimed while idle: its state sits in host memory
	(void) take_hip_failure();
	View* vs[] = {&c->S, &c->Y, &c->sbak, &c->ybak, &c->gprev, &c->xsum, &c->xprev, &c->H0, &c->G, &c->F};
	for (View* v : vs) export_view(c, *v);
	sync(c);
	const bool failed = c->fault || take_hip_failure();
	c->fault = false;
	end_use(c);
	return failed ? -1000 : 0;
}

int stochqn_hip_set_option(const char* name, double value)
{
	if (!name) return -1;
	(void) options();                            // environment defaults first, explicit settings win
	if (!std::strcmp(name, "nontemporal")) g_opt.nontemporal = value != 0;
	else if (!std::strcmp(name, "grid_cap")) {
		int g = (int) value;
		if (g < 0) g = 0;
		if (g > kMaxGrid) g = kMaxGrid;
		g_opt.grid_cap = g;
	}
	else if (!std::strcmp(name, "rows_split")) g_opt.rows_split = value != 0;
	else if (!std::strcmp(name, "fisher_rows")) g_opt.fisher_rows = (int) value;
	else if (!std::strcmp(name, "fisher_split")) g_opt.fisher_split = value != 0;
	else if (!std::strcmp(name, "fisher_tile")) g_opt.fisher_tile = value >= 2 ? 2 : 1;
	else if (!std::strcmp(name, "fisher_lag")) g_opt.fisher_lag = value < 0 ? 0 : (value > 1e6 ? 1000000 : (int) value);
	else if (!std::strcmp(name, "fisher_split_per_cu")) g_opt.fisher_split_per_cu = value < 0 ? 0 : (value > 8 ? 8 : (int) value);
	// 0 = off; otherwise at least 64 ticks (640 ns): the kernels multiply 2^32 / ticks by the number of output streams of the pass
	// (up to 4, adaQN's pass 2) in 32 bits, which a period of 2 or 3 ticks would wrap to a period that never ends
	else if (!std::strcmp(name, "phase_ticks")) g_opt.phase_ticks = value < 2 ? 0 : (value < 64 ? 64 : (value > 1e8 ? 100000000 : (int) value));
	else if (!std::strcmp(name, "spec_x")) g_opt.spec_x = value != 0;
	else if (!std::strcmp(name, "x_prefetch")) g_opt.x_prefetch = value != 0;
	else if (!std::strcmp(name, "keep_tail")) g_opt.keep_tail = value < 0 ? 0 : (value > 1 ? 1 : value);
	else if (!std::strcmp(name, "qdot_per_cu")) g_opt.qdot_per_cu = (int) value;
	else if (!std::strcmp(name, "sdot_tile")) g_opt.sdot_tile = value >= 2 ? 2 : 1;
	else if (!std::strcmp(name, "pair_per_cu")) g_opt.pair_per_cu = value < 0 ? 0 : (value > 8 ? 8 : (int) value);
	else if (!std::strcmp(name, "sadd_per_cu")) g_opt.sadd_per_cu = (int) value;
	else if (!std::strcmp(name, "sdot2_per_cu")) g_opt.sdot2_per_cu = (int) value;
	else if (!std::strcmp(name, "sdot_per_cu")) g_opt.sdot_per_cu = (int) value;
	else if (!std::strcmp(name, "reverse")) g_opt.reverse = value != 0;
	else if (!std::strcmp(name, "threepass")) g_opt.threepass = value != 0;
	else if (!std::strcmp(name, "kappa_max")) g_opt.kappa_max = value;
	else if (!std::strcmp(name, "strict_grad")) g_opt.strict_grad = value != 0;
	else if (!std::strcmp(name, "register_host")) g_opt.register_host = value != 0;
	else if (!std::strcmp(name, "register_min_bytes")) g_opt.register_min_bytes = value < 0 ? 0 : (long) value;
	else if (!std::strcmp(name, "x_upload")) g_opt.x_upload = value < 0 ? 0 : (value > 2 ? 2 : (int) value);
	else if (!std::strcmp(name, "hash_threads")) g_opt.hash_threads = value < 0 ? 0 : (int) value;
	else if (!std::strcmp(name, "upload_slices")) g_opt.upload_slices = value < 0 ? 0 : (value > 64 ? 64 : (int) value);
	else if (!std::strcmp(name, "apply_chunks")) g_opt.apply_chunks = value < 1 ? 1 : (value > 64 ? 64 : (int) value);
	else if (!std::strcmp(name, "host_slice_min")) g_opt.host_slice_min = value < 2 ? 2 : (long) value;
	else if (!std::strcmp(name, "max_mirror_bytes")) g_opt.max_mirror_bytes = value < 0 ? 0 : (long) value;
	else if (!std::strcmp(name, "devices")) options().devices = value < 0 ? 0 : (int) value;
	else if (!std::strcmp(name, "virtual_devices")) options().virtual_devices = value != 0;
	else if (!std::strcmp(name, "devices_rccl_single")) options().devices_rccl_single = value != 0;
	else if (!std::strcmp(name, "devices_min_n")) options().devices_min_n = (long) value;
	else if (!std::strcmp(name, "reducer_patience_s")) options().reducer_patience_s = value > 0 ? value : 120;
	else if (!std::strcmp(name, "verify_cache")) options().verify_cache = value != 0;
	else if (!std::strcmp(name, "raw_reuse_cache")) options().raw_reuse_cache = value != 0;
	else if (!std::strcmp(name, "async_device")) options().async_device = value != 0;
	else if (!std::strcmp(name, "null_stream")) options().null_stream = (int) value;
	else if (!std::strcmp(name, "fail_alloc_after")) g_fail_alloc_after.store((long) value);
	else if (!std::strcmp(name, "inject_device_fault")) g_inject_device_fault.store(value != 0);
	else return -1;
	return 0;
}

void stochqn_hip_profile_enable(int on)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	g_profile = on != 0;
	for (auto& kv : g_ctx) kv.second->sc.prof = g_profile ? &kv.second->prof : nullptr;
}

void stochqn_hip_profile_reset(void)
{
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	for (auto& kv : g_ctx) kv.second->prof.reset();
	for (int i = 0; i < K_COUNT; i++) { g_retired_ms[i] = 0; g_retired_launches[i] = 0; }
}

int stochqn_hip_profile_kernels(void) { return K_COUNT; }
const char* stochqn_hip_profile_name(int id) { return kernel_name(id); }

int stochqn_hip_profile_get(int id, long long* launches, double* total_ms)
{
	if (id < 0 || id >= K_COUNT) return -1;
	std::lock_guard<std::recursive_mutex> lk(g_mu);
	long long l = g_retired_launches[id];
	double ms = g_retired_ms[id];
	for (auto& kv : g_ctx) { l += kv.second->prof.launches[id]; ms += kv.second->prof.total_ms[id]; }
	if (launches) *launches = l;
	if (total_ms) *total_ms = ms;
	return 0;
}

// ---- synthetic inputs (measurement helpers; device pointers, enqueued on the null stream) ----------
int stochqn_hip_synth_uniform(real_t* out, size_t count, unsigned long long first_index, unsigned long long seed,
                              unsigned long long stream, unsigned long long t, double a, double b)
{
	if (!device_ready() || !out || !is_device_pointer(out)) return -1000;
	if (count) launch_synth_uniform(nullptr, out, count, first_index, synth_key(seed, stream, t), a, b);
	return hipGetLastError() == hipSuccess ? 0 : -1000;
}

int stochqn_hip_synth_noisy_grad(real_t* grad, const real_t* d, const real_t* x, size_t count, unsigned long long first_index,
                                 unsigned long long seed, unsigned long long stream, unsigned long long t, double amp)
{
	if (!device_ready() || !grad || !d || !x || !is_device_pointer(grad) || !is_device_pointer(d) || !is_device_pointer(x)) return -1000;
	if (count) launch_synth_grad(nullptr, grad, d, x, count, first_index, synth_key(seed, stream, t), amp);
	return hipGetLastError() == hipSuccess ? 0 : -1000;
}

int stochqn_hip_synth_batch_row(real_t* row, const real_t* d, size_t count, unsigned long long first_index, unsigned k, unsigned bs)
{
	if (!device_ready() || !row || !d || bs == 0 || !is_device_pointer(row) || !is_device_pointer(d)) return -1000;
	if (count) launch_synth_batch_row(nullptr, row, d, count, first_index, k, bs);
	return hipGetLastError() == hipSuccess ? 0 : -1000;
}

int stochqn_hip_comm_unique_id(void* out128)
{
	if (!load_rccl()) return -1;
	ncclUniqueId id;
	if (g_comm.GetUniqueId(&id) != ncclSuccess) return -1;
	std::memcpy(out128, &id, sizeof(id));
	return 0;
}

int stochqn_hip_comm_init(int rank, int nranks, const void* unique_id128)
{
	if (!device_ready() || g_custom.fn || !load_rccl()) return -1;
	if (g_comm.comm) return 0;
	ncclUniqueId id;
	std::memcpy(&id, unique_id128, sizeof(id));
	ncclResult_t r = g_comm.CommInitRank(&g_comm.comm, nranks, id, rank);
	if (r != ncclSuccess) {
		std::fprintf(stderr, "stochqn: ncclCommInitRank failed: %s\n", g_comm.GetErrorString ? g_comm.GetErrorString(r) : "?");
		g_comm.comm = nullptr;
		return -1;
	}
	g_comm.rank = rank;
	g_comm.nranks = nranks;
	return 0;
}

int stochqn_hip_comm_init_custom(int rank, int nranks, stochqn_hip_allreduce_fn fn, void* user)
{
	if (!device_ready() || !fn || nranks < 1 || rank < 0 || rank >= nranks || g_comm.comm) return -1;
	g_custom.fn = fn;
	g_custom.user = user;
	g_custom.rank = rank;
	g_custom.nranks = nranks;
	return 0;
}

int stochqn_hip_comm_nranks(void) { return comm_nranks(); }

// Latency of one reduction as the kernel chain pays it: `reps` in-place sums of `count` doubles through the reducer the
// calling thread's contexts use (RCCL communicator, caller-supplied reducer, loop-back), each followed by a stream
// synchronisation; median and minimum in microseconds.  Collective: every rank / shard thread calls it alike.
int stochqn_hip_comm_allreduce_probe(int count, int reps, double* median_us, double* min_us)
{
	if (!device_ready() || count < 1 || count > kRedMax || reps < 1) return -1000;
	DevCtx c;
	c.red = current_reducer();
	if (c.red.kind == Reducer::NONE) return -1;
	void (*hook)(void*, double*, int, hipStream_t) =
		c.red.kind == Reducer::LOOP ? loopback_hook : (c.red.kind == Reducer::CUSTOM ? custom_hook : allreduce_hook);
	double* buf = nullptr;
	hipStream_t st = nullptr;
	if (!device_alloc((void**) &buf, (size_t) count * sizeof(double))) return -1000;
	if (hipStreamCreate(&st) != hipSuccess) { (void) hipFree(buf); return -1000; }
	SQN_HIP_OK(hipMemsetAsync(buf, 0, (size_t) count * sizeof(double), st));
	std::vector<double> us((size_t) reps);
	for (int w = 0; w < 5 && !c.fault; w++) { hook(&c, buf, count, st); SQN_HIP_OK(hipStreamSynchronize(st)); }
	for (int r = 0; r < reps && !c.fault; r++) {
		const auto t0 = std::chrono::steady_clock::now();
		hook(&c, buf, count, st);
		SQN_HIP_OK(hipStreamSynchronize(st));
		us[(size_t) r] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
	}
	(void) hipStreamDestroy(st);
	(void) hipFree(buf);
	if (c.fault) return -1000;
	std::sort(us.begin(), us.end());
	if (median_us) *median_us = us[(size_t) reps / 2];
	if (min_us) *min_us = us[0];
	return 0;
}

long long stochqn_hip_stat(const char* name)
{
	if (!name) return -1;
	for (int i = 0; i < ST_COUNT; i++)
		if (!std::strcmp(name, kStatNames[i])) return g_stats[i].load(std::memory_order_relaxed);
	if (!std::strcmp(name, "host_pins_live")) {                // a gauge, not a counter: ranges pinned through stochqn_hip_pin_host right now
		std::lock_guard<std::mutex> lk(g_pin_mu);
		return (long long) g_pins.size();
	}
	return -1;
}

void stochqn_hip_stats_reset(void)
{
	for (int i = 0; i < ST_COUNT; i++)
		if (i != ST_HOST_UNPIN_FAILED && i != ST_WORK_IN_FLIGHT && i != ST_HOST_PIN_ERRORS && i != ST_HOST_PIN_FOREIGN) g_stats[i].store(0, std::memory_order_relaxed);       // those are facts about the process, not rates
}

int stochqn_hip_loopback_init(int nranks)
{
	if (nranks < 1 || g_comm.comm) return -1;
	std::lock_guard<std::mutex> lk(g_loop.mu);
	g_loop.nranks = nranks;
	g_loop.arrived = 0;
	g_loop.broken = false;
	g_loop.patience_s = options().reducer_patience_s;
	g_loop.slots.assign((size_t) nranks * kRedMax, 0.0);
	return 0;
}

int stochqn_hip_loopback_join(int rank)
{
	if (rank < 0 || rank >= g_loop.nranks) return -1;
	Reducer r;
	r.kind = Reducer::LOOP; r.loop = &g_loop; r.rank = rank; r.nranks = g_loop.nranks;
	set_thread_reducer(r);
	return 0;
}

void stochqn_hip_loopback_finalize(void)
{
	release_all();
	g_loop.nranks = 0;
	set_thread_reducer(Reducer{});
}

void stochqn_hip_comm_finalize(void)
{
	release_all();
	if (g_comm.comm) { if (!forget_dead(g_comm.comm)) g_comm.CommDestroy(g_comm.comm); g_comm.comm = nullptr; g_comm.nranks = 1; g_comm.rank = 0; }
	g_custom = Custom{};
}

}  // extern "C"
