// runtime.cpp -- context registry, pointer residency, mirrors, RCCL hook, options, profiler API.
#include "machines.hpp"
#include "stochqn_hip.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace sqn {

namespace {

std::mutex g_mu;
std::unordered_map<const void*, DevCtx*> g_ctx;
Options g_opt;
std::atomic<long> g_fail_alloc_after{-1};       // fault injection: < 0 off, else allocations left before one fails
std::atomic<int> g_inject_device_fault{0};      // fault injection: the next stream synchronisation reports a failure

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
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_comm;

bool load_rccl()
{
	if (g_comm.dl) return true;
	const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* nm : names) {
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
	ncclResult_t r = g_comm.AllReduce(buf, buf, (size_t) count, ncclDouble, ncclSum, (ncclComm_t) c->red.comm, stream);
	if (r != ncclSuccess) {
		std::fprintf(stderr, "stochqn: ncclAllReduce: %s\n", g_comm.GetErrorString ? g_comm.GetErrorString(r) : "?");
		reducer_failed(user, "ncclAllReduce failed");
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
	if (!lp.cv.wait_for(lk, std::chrono::seconds(lp.patience_s), [&] { return lp.generation != gen || lp.broken; }) || lp.broken) {
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

void free_view(View& v)
{
	if (v.mirror && v.dev) SQN_HIP_OK(hipFree(v.dev));
	v = View{};
}

void destroy(DevCtx* c)
{
	if (c->sc.stream) SQN_HIP_OK(hipStreamSynchronize(c->sc.stream));
	c->prof.collect();
	for (int i = 0; i < K_COUNT; i++) { g_retired_ms[i] += c->prof.total_ms[i]; g_retired_launches[i] += c->prof.launches[i]; }
	for (auto& p : c->prof.pending) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
	for (auto e : c->prof.pool) (void) hipEventDestroy(e);
	View* vs[] = {&c->S, &c->Y, &c->sbak, &c->ybak, &c->gprev, &c->xsum, &c->xprev, &c->H0, &c->G, &c->F};
	for (View* v : vs) free_view(*v);
	if (c->pool) SQN_HIP_OK(hipFree(c->pool));
	if (c->sc.fisher_part) SQN_HIP_OK(hipFree(c->sc.fisher_part));
	if (c->fisher_t) SQN_HIP_OK(hipFree(c->fisher_t));
	for (real* p : c->stage) if (p) SQN_HIP_OK(hipFree(p));
	for (int i = 0; i < 2; i++) {
		if (c->host_stage[i] && c->host_stage_pinned[i]) SQN_HIP_OK(hipHostFree(c->host_stage[i]));
		else std::free(c->host_stage[i]);
	}
	if (c->pin) SQN_HIP_OK(hipHostFree(c->pin));
	if (c->sc.stream) SQN_HIP_OK(hipStreamDestroy(c->sc.stream));
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
	*p = nullptr;
	if (alloc_should_fail() || hipMalloc(p, bytes ? bytes : 1) != hipSuccess) {
		(void) hipGetLastError();
		*p = nullptr;
		std::fprintf(stderr, "stochqn: could not allocate %zu bytes of device memory\n", bytes);
		return false;
	}
	return true;
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
	c->sc.nontemporal = g_opt.nontemporal;
	c->sc.grid_cap = g_opt.grid_cap > 0 ? g_opt.grid_cap : default_grid_cap();
	c->sc.rows_grid = g_opt.rows_grid;
	c->sc.reverse = g_opt.reverse;
	c->sc.rows_split = g_opt.rows_split;
	c->sc.rows_waves = g_opt.rows_waves;
	c->sc.combine_batch = g_opt.combine_batch;
	c->sc.h0_per_cu = g_opt.h0_per_cu;
	c->sc.fisher_rows = g_opt.fisher_rows;
	c->sc.stream_stores = g_opt.stream_stores;
	c->sc.qdot_stream = g_opt.qdot_stream;
	c->sc.qdot_per_cu = g_opt.qdot_per_cu; c->sc.sadd_per_cu = g_opt.sadd_per_cu; c->sc.sdot2_per_cu = g_opt.sdot2_per_cu; c->sc.sdot_per_cu = g_opt.sdot_per_cu;
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
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	return it == g_ctx.end() ? nullptr : it->second;
}

DevCtx* acquire(const void* key, int kind, int n, size_t m, size_t fsize, bool* fresh)
{
	std::lock_guard<std::mutex> lk(g_mu);
	*fresh = false;
	auto it = g_ctx.find(key);
	if (it != g_ctx.end()) {
		DevCtx* c = it->second;
		if (c->kind == kind && c->n == n && c->m == m && c->fsize == fsize) {
			begin_call(c);
			return c;
		}
		destroy(c);              // same address, different problem: the old owner is gone
		g_ctx.erase(it);
	}
	DevCtx* c = new DevCtx();
	c->key = key; c->kind = kind; c->n = n; c->m = m; c->fsize = fsize;
	c->n_global = (double) n;
	SQN_HIP_OK(hipStreamCreate(&c->sc.stream));   // blocking flavour: ordered after the null stream
	// pool layout: part0 | part1 | red0 | red1 | sy | yy | report | rho | alpha | rows_part x2 | gsy | gyy | coef
	const size_t part = (size_t) kMaxSums * kMaxGrid;
	const size_t rows_part = (size_t) kRedMax * kMaxGrid;
	const size_t total = 2 * part + 2 * kRedMax + 4 * m + 8 + 2 * rows_part + 2 * m * m + (2 + 2 * kPairsMax3) + 3 * m;
	c->pin_count = 16 + 2 * m + fsize + 3 * m + 8;
	if (!device_alloc((void**) &c->pool, total * sizeof(double)) ||
	    !pinned_alloc((void**) &c->pin, c->pin_count * sizeof(double)) ||
	    (fsize > 0 && (!device_alloc((void**) &c->sc.fisher_part, fsize * kMaxGrid * sizeof(double)) ||
	                   !device_alloc((void**) &c->fisher_t, fsize * sizeof(double))))) {
		std::fprintf(stderr, "stochqn: could not allocate the scratch of a device context\n");
		destroy(c);
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
	c->sc.gyy = p; p += m * m;
	c->sc.coef = p; p += 2 + 2 * kPairsMax3;
	c->kap_dev = p;
	c->forget_rows();
	begin_call(c);
	c->sc.allreduce = nullptr;
	c->sc.user = c;
	g_ctx[key] = c;
	if (!g_atexit) {
		g_atexit = true;
		std::atexit(at_exit);
		if (std::getenv("STOCHQN_HIP_PROFILE")) { g_profile = true; c->sc.prof = &c->prof; }
	}
	*fresh = true;
	return c;
}

bool prepare_context(const void* key, int kind, int n, size_t m, size_t fsize)
{
	bool fresh = false;
	return acquire(key, kind, n, m, fsize, &fresh) != nullptr;
}

bool note_state(const void* key, size_t niter, int section)
{
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) return false;
	it->second->has_last = true;
	it->second->last_niter = niter;
	it->second->last_section = section;
	const bool fault = it->second->fault;
	it->second->fault = false;
	return fault;
}

void release(const void* key)
{
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_ctx.find(key);
	if (it == g_ctx.end()) return;
	destroy(it->second);
	g_ctx.erase(it);
}

void release_all()
{
	std::lock_guard<std::mutex> lk(g_mu);
	for (auto& kv : g_ctx) destroy(kv.second);
	g_ctx.clear();
}

bool bind(DevCtx* c, View& v, real* caller, size_t count, bool import)
{
	(void) c;
	if (caller == v.caller && count == v.count && (v.dev || count == 0)) return true;
	free_view(v);
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
	// without `import` the mirror starts with indeterminate contents, like the reference's malloc
	if (import) SQN_HIP_OK(hipMemcpy(v.dev, caller, count * sizeof(real), hipMemcpyHostToDevice));
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

real* stage_in(DevCtx* c, int which, real* caller, size_t count, bool host)
{
	if (!host) return caller;
	if (!ensure_stage(c, which)) return nullptr;
	SQN_HIP_OK(hipMemcpyAsync(c->stage[which], caller, count * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	return c->stage[which];
}

void sync(DevCtx* c)
{
	// a kernel that could not be launched (or failed) leaves its outputs stale: never hand that back
	// as a result -- the call reports invalid_input / -1000 instead (machines.cpp: after_call)
	const hipError_t e = hipStreamSynchronize(c->sc.stream);
	hipError_t l = hipGetLastError();
	if (g_inject_device_fault.exchange(0)) l = hipErrorLaunchFailure;
	if (e != hipSuccess || l != hipSuccess) {
		std::fprintf(stderr, "stochqn: device work failed: %s\n", hipGetErrorString(e != hipSuccess ? e : l));
		c->fault = true;
	}
	if (c->sc.prof) c->prof.collect();
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
	SQN_HIP_OK(hipStreamSynchronize(c->sc.stream));
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

void comm_destroy(void* comm)
{
	if (comm && g_comm.CommDestroy) g_comm.CommDestroy((ncclComm_t) comm);
}

}  // namespace sqn

// ------------------------------------------------------------------------------------------------
// C entry points of stochqn_hip.h that do not involve the optimiser arithmetic
// ------------------------------------------------------------------------------------------------
using namespace sqn;

extern "C" {

int stochqn_hip_available(void) { return device_ready() ? 1 : 0; }

void stochqn_hip_invalidate(const void* s_mem)
{
	if (group_invalidate(s_mem)) return;
	if (!s_mem) return;
	for (const void* key : {s_mem, raw_key(s_mem)})
		if (DevCtx* c = lookup(key)) c->forget_rows();
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
	DevCtx* c = lookup(s_mem);
	if (!c) return -1000;
	View* vs[] = {&c->S, &c->Y, &c->sbak, &c->ybak, &c->gprev, &c->xsum, &c->xprev, &c->H0, &c->G, &c->F};
	for (View* v : vs) export_view(c, *v);
	sync(c);
	return 0;
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
	else if (!std::strcmp(name, "rows_grid")) {
		int g = (int) value;
		if (g < 0) g = 0;
		if (g > kMaxGrid) g = kMaxGrid;
		g_opt.rows_grid = g;
	}
	else if (!std::strcmp(name, "rows_split")) g_opt.rows_split = value != 0;
	else if (!std::strcmp(name, "rows_waves")) g_opt.rows_waves = (int) value;
	else if (!std::strcmp(name, "combine_batch")) g_opt.combine_batch = (int) value;
	else if (!std::strcmp(name, "h0_per_cu")) g_opt.h0_per_cu = (int) value;
	else if (!std::strcmp(name, "fisher_rows")) g_opt.fisher_rows = (int) value;
	else if (!std::strcmp(name, "stream_stores")) g_opt.stream_stores = value != 0;
	else if (!std::strcmp(name, "qdot_stream")) g_opt.qdot_stream = value != 0;
	else if (!std::strcmp(name, "qdot_per_cu")) g_opt.qdot_per_cu = (int) value;
	else if (!std::strcmp(name, "sadd_per_cu")) g_opt.sadd_per_cu = (int) value;
	else if (!std::strcmp(name, "sdot2_per_cu")) g_opt.sdot2_per_cu = (int) value;
	else if (!std::strcmp(name, "sdot_per_cu")) g_opt.sdot_per_cu = (int) value;
	else if (!std::strcmp(name, "reverse")) g_opt.reverse = value != 0;
	else if (!std::strcmp(name, "twopass")) g_opt.twopass = value != 0;
	else if (!std::strcmp(name, "twopass_h0")) g_opt.twopass_h0 = value != 0;
	else if (!std::strcmp(name, "threepass")) g_opt.threepass = value != 0;
	else if (!std::strcmp(name, "twopass_kappa_max")) g_opt.twopass_kappa_max = value;
	else if (!std::strcmp(name, "strict_grad")) g_opt.strict_grad = value != 0;
	else if (!std::strcmp(name, "devices")) options().devices = value < 0 ? 0 : (int) value;
	else if (!std::strcmp(name, "virtual_devices")) options().virtual_devices = value != 0;
	else if (!std::strcmp(name, "devices_min_n")) options().devices_min_n = (long) value;
	else if (!std::strcmp(name, "verify_cache")) options().verify_cache = value != 0;
	else if (!std::strcmp(name, "raw_reuse_cache")) options().raw_reuse_cache = value != 0;
	else if (!std::strcmp(name, "fail_alloc_after")) g_fail_alloc_after.store((long) value);
	else if (!std::strcmp(name, "inject_device_fault")) g_inject_device_fault.store(value != 0);
	else return -1;
	return 0;
}

void stochqn_hip_profile_enable(int on)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_profile = on != 0;
	for (auto& kv : g_ctx) kv.second->sc.prof = g_profile ? &kv.second->prof : nullptr;
}

void stochqn_hip_profile_reset(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	for (auto& kv : g_ctx) kv.second->prof.reset();
	for (int i = 0; i < K_COUNT; i++) { g_retired_ms[i] = 0; g_retired_launches[i] = 0; }
}

int stochqn_hip_profile_kernels(void) { return K_COUNT; }
const char* stochqn_hip_profile_name(int id) { return kernel_name(id); }

int stochqn_hip_profile_get(int id, long long* launches, double* total_ms)
{
	if (id < 0 || id >= K_COUNT) return -1;
	std::lock_guard<std::mutex> lk(g_mu);
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

int stochqn_hip_loopback_init(int nranks)
{
	if (nranks < 1 || g_comm.comm) return -1;
	std::lock_guard<std::mutex> lk(g_loop.mu);
	g_loop.nranks = nranks;
	g_loop.arrived = 0;
	g_loop.broken = false;
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
	if (g_comm.comm) { g_comm.CommDestroy(g_comm.comm); g_comm.comm = nullptr; g_comm.nranks = 1; g_comm.rank = 0; }
	g_custom = Custom{};
}

}  // extern "C"
