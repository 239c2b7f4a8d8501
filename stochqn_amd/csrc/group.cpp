// group.cpp -- single-process multi-device mode: ONE host process (an R session through .Call, a plain C
// program, the Cython binding) calls run_* once per step and the library drives P devices.
//
// Why: the reference's callers are single-process (reference src/Rwrapper.c:98-125, example/c_rosen.c:100-125,
// stochqn/pywrapper.pxi:161-207); a problem whose S and Y do not fit one device (BASELINE config 5: n = 1e9,
// m = 20 -> 320 GB) is out of their reach with one-process-per-GPU sharding (stochqn_hip_comm_init).
//
// How: with option "devices" = P >= 2 (or STOCHQN_HIP_DEVICES=P in the environment) the n dimension of a
// workspace whose arrays are host memory (profile B: R / numpy arrays) or library-owned (profile A:
// initialize_*) is cut into P contiguous slices.  Shard p lives on device p: its own slice of every
// n-vector and of every row of S, Y, F in that device's HBM, its own host thread (bound to the device),
// its own device context and stream.  A run_* call is fanned out: every shard thread uploads its slice
// of x / grad / hess_vec (P parallel PCIe streams), runs the ordinary single-device state machine
// (machines.cpp) on its slice, and copies its slice of x, of the direction and of whatever *req /
// *req_vec designate back into the caller's host arrays.  Inside the kernels chain every reduction is
// local partial sums + one all-reduce over the P shards: RCCL communicators from ncclCommInitAll, one
// per shard thread (one thread per device, so no group calls are needed), the scalars stay on the
// devices.  All shards therefore take bit-identical decisions; the front-end checks that they did.
//
// "virtual_devices" = 1 lets P exceed the number of physical devices (shard p on device p mod #devices)
// with the host-side rendezvous reducer instead of RCCL (which refuses two ranks on one device): the
// whole mode is testable on one GPU (tests/test_gpu_devices.py).
#include "machines.hpp"
#include "stochqn_hip.h"

#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <unordered_map>

namespace sqn {

namespace {

// ---- one host thread per shard, bound to the shard's device ------------------------------------
struct Worker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::function<void()> job;
	bool pending = false, quit = false;

	void start(int device, const Reducer& red)
	{
		th = std::thread([this, device, red] {
			(void) hipSetDevice(device);
			set_thread_reducer(red);
			std::unique_lock<std::mutex> lk(mu);
			for (;;) {
				cv.wait(lk, [&] { return pending || quit; });
				if (quit) return;
				lk.unlock();
				job();
				lk.lock();
				pending = false;
				cv.notify_all();
			}
		});
	}
	void submit(std::function<void()> f)
	{
		std::lock_guard<std::mutex> lk(mu);
		job = std::move(f);
		pending = true;
		cv.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return !pending; });
	}
	void stop()
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			quit = true;
			cv.notify_all();
		}
		if (th.joinable()) th.join();
	}
};

struct Shard {
	int rank = 0, device = 0;
	size_t off = 0, cnt = 0;                 // this shard's slice [off, off + cnt) of every n-vector
	Worker wk;
	// the shard's own ABI structs: same scalars as the caller's, arrays = this device's slices
	bfgs_mem b{};
	fisher_mem f{};
	workspace_oLBFGS wo{};
	workspace_SQN ws{};
	workspace_adaQN wa{};
	real *S = nullptr, *Y = nullptr, *sbak = nullptr, *ybak = nullptr, *gprev = nullptr, *xsum = nullptr, *xprev = nullptr,
	     *H0 = nullptr, *G = nullptr, *F = nullptr;
	real_t *lx = nullptr;                                    // x as the last call handed it to the shard's machine (host slice or bx)
	real *bx = nullptr, *bg = nullptr, *bhv = nullptr;       // device-resident caller: its own per-shard vectors, used in place
	bool bound = false;
	std::vector<real_t> rho, alpha, fy;                      // buffer_rho / buffer_alpha / buffer_y of ranks > 0
	// outcome of the last call
	int ret = 0;
	task_enum task = invalid_input;
	info_enum info = no_problems_encountered;
	real_t* req = nullptr;
	real_t* req_vec = nullptr;
	bool ok = true;
};

// the caller's host arrays as of the last call (profile B; stable for the life of the R / Python object,
// while the structs that carry them are rebuilt on the caller's stack every call): what export writes to
struct HostArrays {
	real_t *S = nullptr, *Y = nullptr, *sbak = nullptr, *ybak = nullptr, *gprev = nullptr, *xsum = nullptr, *xprev = nullptr,
	       *H0 = nullptr, *G = nullptr, *F = nullptr;
};

struct Group {
	HostArrays host;
	const void* key = nullptr;
	int kind = 0, n = 0, P = 0;
	size_t m = 0, fsize = 0;
	bool owned = false, virt = false;
	std::vector<std::unique_ptr<Shard>> sh;
	Loopback loop;
	std::vector<void*> comms;
	real* landing[2] = {nullptr, nullptr};   // owned workspaces: host-readable *req / *req_vec
	bool landing_pinned[2] = {false, false};
	void* token = nullptr;
	bool has_last = false;
	size_t last_niter = 0;
	int last_section = 0;
	// host callers: their x / grad / hess_vec pinned in place for all devices (hipHostRegisterPortable); everything else about
	// them -- x not sent again while a shard's copy is current, passes in slices under the transfers -- is the single-device
	// host path, run per shard on its slice (machines.cpp)
	struct HostRange { const void* p = nullptr; size_t bytes = 0; };
	HostRange regs[4];
	int reg_turn = 0;
};

std::mutex g_gmu;
std::unordered_map<const void*, Group*> g_groups;

void for_all(Group* g, const std::function<void(Shard&)>& fn)
{
	for (auto& s : g->sh) { Shard* sp = s.get(); sp->wk.submit([sp, &fn] { fn(*sp); }); }
	for (auto& s : g->sh) s->wk.wait();
}

int physical_devices()
{
	int c = 0;
	if (hipGetDeviceCount(&c) != hipSuccess) { (void) hipGetLastError(); return 0; }
	return c;
}

// shards a workspace of n variables runs on: 0 = the mode is off for it.  One shard only with option "devices_rccl_single":
// the whole mode -- worker thread, ncclCommInitAll, per-thread all-reduce, teardown -- on ONE device over a real communicator.
int shards_for(int n)
{
	const Options& o = options();
	if (n <= 0) return 0;
	if (o.devices < 2) return (o.devices_rccl_single && !o.virtual_devices && (long) n >= o.devices_min_n && physical_devices() >= 1) ? 1 : 0;
	if ((long) n < o.devices_min_n) return 0;
	int P = o.devices;
	const int phys = physical_devices();
	if (phys < 1) return 0;
	if (P > phys && !o.virtual_devices) {
		std::fprintf(stderr, "stochqn: option devices=%d but only %d device(s) are visible (set virtual_devices=1 to rehearse); "
		                     "using %d\n", P, phys, phys);
		P = phys;
	}
	if (P > n) P = n;
	return P >= 2 ? P : 0;
}


bool dalloc(real** p, size_t count, bool zero_fill)
{
	if (!device_alloc((void**) p, (count ? count : 1) * sizeof(real))) return false;
	if (zero_fill) SQN_HIP_OK(hipMemset(*p, 0, (count ? count : 1) * sizeof(real)));
	return true;
}

void dfree(real*& p)
{
	if (p) (void) hipFree(p);
	p = nullptr;
}

// rows x cnt block of a [rows][n] host array  <->  the shard's dense [rows][cnt] device array
void rows_h2d(real* dev, const real_t* host, size_t rows, size_t n, size_t off, size_t cnt)
{
	SQN_HIP_OK(hipMemcpy2D(dev, cnt * sizeof(real), host + off, n * sizeof(real), cnt * sizeof(real), rows, hipMemcpyHostToDevice));
}
void rows_d2h(real_t* host, const real* dev, size_t rows, size_t n, size_t off, size_t cnt)
{
	SQN_HIP_OK(hipMemcpy2D(host + off, n * sizeof(real), dev, cnt * sizeof(real), cnt * sizeof(real), rows, hipMemcpyDeviceToHost));
}

// pin [p, p + bytes) of the caller's memory for every device of the process (once; at most 4 ranges per group)
void pin_for_all_devices(Group* g, const void* p, size_t bytes)
{
	const Options& o = options();
	if (!o.register_host || !p || (long) bytes < o.register_min_bytes) return;
	for (auto& r : g->regs) if (r.p == p && r.bytes >= bytes) return;
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) == hipSuccess) { if (a.type == hipMemoryTypeHost) return; }      // pinned by somebody else already
	else (void) hipGetLastError();
	if (!pinnable_in_place(p, bytes)) return;                  // runtime.cpp: not in the break heap, no page shared with another pin
	Group::HostRange& slot = g->regs[g->reg_turn++ % 4];
	if (slot.p) { if (hipHostUnregister(const_cast<void*>(slot.p)) != hipSuccess) (void) hipGetLastError(); note_unpinned(slot.p); slot = Group::HostRange{}; }
	if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterPortable) != hipSuccess) { (void) hipGetLastError(); note_unpinned(p); return; }
	slot.p = p;
	slot.bytes = bytes;
	stat_add(ST_HOST_REGISTERED);
}

void destroy_group(Group* g)
{
	for (auto& r : g->regs)
		if (r.p) { if (hipHostUnregister(const_cast<void*>(r.p)) != hipSuccess) (void) hipGetLastError(); note_unpinned(r.p); }
	for_all(g, [&](Shard& s) {
		if (s.S) release(s.S);                                  // the shard's device context
		dfree(s.S); dfree(s.Y); dfree(s.sbak); dfree(s.ybak); dfree(s.gprev); dfree(s.xsum); dfree(s.xprev);
		dfree(s.H0); dfree(s.G); dfree(s.F);
	});
	for (auto& s : g->sh) s->wk.stop();
	for (void* c : g->comms) comm_destroy(c);
	for (int i = 0; i < 2; i++) {
		if (g->landing[i] && g->landing_pinned[i]) (void) hipHostFree(g->landing[i]);
		else std::free(g->landing[i]);
	}
	delete g;
}

// What the caller's workspace looks like, independent of the optimiser kind.
struct Shape {
	int kind = 0, n = 0;
	bfgs_mem* b = nullptr;
	fisher_mem* f = nullptr;            // adaQN without use_grad_diff
	bool need_gprev = false, need_avg = false, need_diag = false;
	// caller's arrays (host pointers for profile B, tokens for owned workspaces)
	real_t *gprev = nullptr, *xsum = nullptr, *xprev = nullptr, *H0 = nullptr, *G = nullptr;
	size_t niter = 0;
	int section = 0;
};

Group* create_group(const Shape& sp, int P, bool owned, bool resumed)
{
	std::unique_ptr<Group> g(new Group());
	g->key = sp.b->s_mem;
	g->kind = sp.kind; g->n = sp.n; g->P = P; g->m = sp.b->mem_size; g->fsize = sp.f ? sp.f->mem_size : 0;
	g->owned = owned;
	const int phys = physical_devices();
	if (phys < 1) return nullptr;
	g->virt = P > phys || options().virtual_devices;
	int dev_single = 0;                                    // one shard ("devices_rccl_single"): on the device the caller is on
	(void) hipGetDevice(&dev_single);
	if (!g->virt) {
		std::vector<int> devs((size_t) P);
		int cur = 0;
		(void) hipGetDevice(&cur);
		for (int p = 0; p < P; p++) devs[(size_t) p] = P == 1 ? cur : p;      // one shard ("devices_rccl_single"): where the caller is
		g->comms.assign((size_t) P, nullptr);
		const bool up = comm_init_all(P, devs.data(), g->comms.data());
		(void) hipSetDevice(cur);
		if (!up) return nullptr;
	} else {
		g->loop.nranks = P;
		g->loop.patience_s = options().reducer_patience_s;
		g->loop.slots.assign((size_t) P * kRedMax, 0.0);
	}
	const size_t n = (size_t) sp.n, base = n / (size_t) P, extra = n % (size_t) P;
	size_t off = 0;
	for (int p = 0; p < P; p++) {
		std::unique_ptr<Shard> s(new Shard());
		s->rank = p;
		s->device = g->virt ? p % phys : (P == 1 ? dev_single : p);
		s->off = off;
		s->cnt = base + ((size_t) p < extra ? 1 : 0);
		off += s->cnt;
		Reducer red;
		red.rank = p; red.nranks = P;
		if (g->virt) { red.kind = Reducer::LOOP; red.loop = &g->loop; }
		else { red.kind = Reducer::RCCL; red.comm = g->comms[(size_t) p]; }
		s->wk.start(s->device, red);
		g->sh.push_back(std::move(s));
	}
	// every shard allocates (and, for caller-owned host arrays, imports) its slices on its own device
	const size_t m = g->m, fsz = g->fsize;
	const bool bak = sp.b->min_curvature > 0;
	for_all(g.get(), [&](Shard& s) {
		const size_t c = s.cnt;
		bool ok = dalloc(&s.S, m * c, false) && dalloc(&s.Y, m * c, false);
		if (ok && bak) ok = dalloc(&s.sbak, c, true) && dalloc(&s.ybak, c, true);
		if (ok && sp.need_gprev) ok = dalloc(&s.gprev, c, false);
		if (ok && sp.need_avg) ok = dalloc(&s.xsum, c, true) && dalloc(&s.xprev, c, false);
		if (ok && sp.need_diag) ok = dalloc(&s.H0, c, false) && dalloc(&s.G, c, true);
		if (ok && fsz) ok = dalloc(&s.F, fsz * c, false);
		// the shard's device context (scratch pool, pinned read-back block, Fisher partials) now, not inside the first
		// call: a shard that ran out of memory there would leave the others waiting in an all-reduce
		// -- and with it the staging vectors a host caller's slices of x / grad / hess_vec go through
		if (ok) ok = prepare_context(s.S, sp.kind, (int) c, m, fsz, sp.kind == KIND_SQN ? 3 : 2);
		s.ok = ok;
		if (!ok || owned) return;
		// profile B: the caller's host arrays are the initial state (fresh R / numpy objects hold zeros
		// where the reference expects zeros; a resumed object holds everything)
		if (bak && sp.b->s_bak && sp.b->y_bak) { rows_h2d(s.sbak, sp.b->s_bak, 1, n, s.off, c); rows_h2d(s.ybak, sp.b->y_bak, 1, n, s.off, c); }
		if (sp.need_gprev && sp.gprev) rows_h2d(s.gprev, sp.gprev, 1, n, s.off, c);
		if (sp.need_avg) { rows_h2d(s.xsum, sp.xsum, 1, n, s.off, c); rows_h2d(s.xprev, sp.xprev, 1, n, s.off, c); }
		if (sp.need_diag) rows_h2d(s.G, sp.G, 1, n, s.off, c);
		if (resumed) {
			rows_h2d(s.S, sp.b->s_mem, m, n, s.off, c);
			rows_h2d(s.Y, sp.b->y_mem, m, n, s.off, c);
			if (fsz && sp.f->mem_used > 0) rows_h2d(s.F, sp.f->F, fsz, n, s.off, c);
		}
	});
	bool ok = true;
	for (auto& s : g->sh) ok = ok && s->ok;
	if (!ok) {
		std::fprintf(stderr, "stochqn: could not allocate the shards of a %d-device workspace (n = %d)\n", P, sp.n);
		destroy_group(g.release());
		return nullptr;
	}
	if (owned) {                                               // host-readable homes of *req / *req_vec (callers read them on the host)
		const size_t bytes = n * sizeof(real);
		for (int i = 0; i < (sp.kind == KIND_SQN ? 2 : 1) && ok; i++) {
			g->landing_pinned[i] = pinned_alloc((void**) &g->landing[i], bytes);
			if (!g->landing_pinned[i]) g->landing[i] = (real*) std::malloc(bytes);
			ok = g->landing[i] != nullptr;
		}
		if (!ok) { destroy_group(g.release()); return nullptr; }
	}
	if (std::getenv("STOCHQN_HIP_VERBOSE"))
		std::fprintf(stderr, "stochqn: workspace with n = %d sharded over %d device shards (%s workspace, reducer: %s)\n", sp.n, P,
		             owned ? "library-owned" : "caller-owned host", g->virt ? "host-side rendezvous (virtual devices)" : "RCCL");
	for (auto& s : g->sh) {
		s->rho.assign(m, 0); s->alpha.assign(m, 0); s->fy.assign(fsz ? fsz : 1, 0);
		s->b.s_mem = s->S; s->b.y_mem = s->Y; s->b.s_bak = s->sbak; s->b.y_bak = s->ybak;
		s->b.mem_size = m;
		s->f.F = s->F; s->f.mem_size = fsz;
	}
	return g.release();
}

Group* find_group(const void* key)
{
	std::lock_guard<std::mutex> lk(g_gmu);
	auto it = g_groups.find(key);
	return it == g_groups.end() ? nullptr : it->second;
}

void drop_group(const void* key)
{
	Group* g = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_gmu);
		auto it = g_groups.find(key);
		if (it == g_groups.end()) return;
		g = it->second;
		g_groups.erase(it);
	}
	destroy_group(g);
}

// The group of this call: the existing one when the call continues where the last one ended, else a
// new one (caller-owned arrays re-imported) -- same rule as the single-device contexts (open_call).
Group* group_for(const Shape& sp)
{
	Group* g = find_group(sp.b->s_mem);
	if (g && !g->owned) {
		const bool same = g->kind == sp.kind && g->n == sp.n && g->m == sp.b->mem_size && g->fsize == (sp.f ? sp.f->mem_size : 0);
		const bool continues = !g->has_last || (g->last_niter == sp.niter && g->last_section == sp.section);
		if (!same || !continues) { drop_group(sp.b->s_mem); g = nullptr; }
	}
	if (!g) {
		const int P = shards_for(sp.n);
		if (P < 1) return nullptr;
		g = create_group(sp, P, false, sp.niter > 0 || sp.b->mem_used > 0);
		if (!g) return nullptr;
		std::lock_guard<std::mutex> lk(g_gmu);
		g_groups[sp.b->s_mem] = g;
	}
	if (!g->owned) {
		HostArrays& h = g->host;
		h.S = sp.b->s_mem; h.Y = sp.b->y_mem;
		h.sbak = sp.b->min_curvature > 0 ? sp.b->s_bak : nullptr; h.ybak = sp.b->min_curvature > 0 ? sp.b->y_bak : nullptr;
		h.gprev = sp.need_gprev ? sp.gprev : nullptr;
		h.xsum = sp.need_avg ? sp.xsum : nullptr; h.xprev = sp.need_avg ? sp.xprev : nullptr;
		h.H0 = sp.need_diag ? sp.H0 : nullptr; h.G = sp.need_diag ? sp.G : nullptr;
		h.F = sp.f ? sp.f->F : nullptr;
	}
	return g;
}

// Scalars are the caller's on every call (R / Python rebuild the structs each time; hyper-parameters may
// change at any moment, reference include/stochqn.h:163-167).
void sync_bfgs(Shard& s, const bfgs_mem* b, bool lead)
{
	s.b.mem_used = b->mem_used; s.b.mem_st_ix = b->mem_st_ix; s.b.upd_freq = b->upd_freq;
	s.b.y_reg = b->y_reg; s.b.min_curvature = b->min_curvature;
	s.b.buffer_rho = (lead && b->buffer_rho) ? b->buffer_rho : s.rho.data();
	s.b.buffer_alpha = (lead && b->buffer_alpha) ? b->buffer_alpha : s.alpha.data();
	if (b->min_curvature > 0 && !s.sbak) {                     // switched on after the workspace was made
		s.ok = dalloc(&s.sbak, s.cnt, true) && dalloc(&s.ybak, s.cnt, true);
		s.b.s_bak = s.sbak; s.b.y_bak = s.ybak;
	}
}

int fail(task_enum* task, const char* who, const char* why)
{
	*task = invalid_input;
	std::fprintf(stderr, "%s: %s\n", who, why);
	return -1000;
}

// After the fan-out: every shard must have come to the same verdict (they all saw the same all-reduced
// scalars); then the caller's struct gets the counters and *req / *req_vec their host-readable targets.
bool agree(Group* g, const char* who)
{
	const Shard& a = *g->sh[0];
	for (auto& sp : g->sh) {
		const Shard& s = *sp;
		if (!s.ok || s.ret != a.ret || s.task != a.task || s.info != a.info || s.b.mem_used != a.b.mem_used || s.b.mem_st_ix != a.b.mem_st_ix) {
			std::fprintf(stderr, "%s: shard %d of %d disagrees with shard 0 (ret %d/%d task %d/%d info %d/%d): the sharded state is unusable\n",
			             who, s.rank, g->P, s.ret, a.ret, (int) s.task, (int) a.task, (int) s.info, (int) a.info);
			return false;
		}
	}
	return true;
}

real* landing(Group* g, int slot) { return g->landing[slot]; }      // allocated with the group (create_group)

// Which caller array does a shard-local *req designate, and where is its host-readable home?
// Returns nullptr for "the caller's x" (already current) and sets *is_x.
real_t* req_home(Group* g, const Shape& sp, const Shard& s, const real_t* req, real_t* x_caller, bool* is_x)
{
	*is_x = false;
	if (req == s.lx) { *is_x = true; return x_caller; }
	if (g->owned) return landing(g, 0);
	if (req == s.xsum) return sp.xsum;
	if (req == s.xprev) return sp.xprev;
	return nullptr;
}

}  // namespace

bool group_mode_for(int n) { return shards_for(n) >= 1; }

bool group_owns(const void* s_mem)
{
	Group* g = find_group(s_mem);
	return g && g->owned;
}

bool group_applies(const bfgs_mem* b, int n)
{
	if (!b || !b->s_mem) return false;
	if (group_owns(b->s_mem)) return true;
	if (options().devices < 2 && !options().devices_rccl_single) return false;
	if (is_device_pointer(b->s_mem)) return false;            // the caller keeps its arrays on one device: single-device path
	return shards_for(n) >= 1;
}

int group_shards(const void* key)
{
	Group* g = find_group(key);
	return g ? g->P : 0;
}

int group_reducer_kind(const void* key)
{
	Group* g = find_group(key);
	return !g ? 0 : (g->virt ? (int) Reducer::LOOP : (int) Reducer::RCCL);
}

bool group_release(const void* key)
{
	Group* g = find_group(key);
	if (!g) return false;
	if (g->owned) { for_all(g, [](Shard& s) { release(s.S); }); return true; }   // arrays stay: dealloc_* frees them
	drop_group(key);
	return true;
}

void group_release_all()
{
	std::vector<const void*> keys;
	{
		std::lock_guard<std::mutex> lk(g_gmu);
		for (auto& kv : g_groups) keys.push_back(kv.first);
	}
	for (const void* k : keys) group_release(k);
}

bool group_invalidate(const void* key)
{
	Group* g = find_group(key);
	if (!g) return false;
	for_all(g, [](Shard& s) { stochqn_hip_invalidate(s.S); });
	return true;
}

// ------------------------------------------------------------------------------------------------
// the fan-out of one run_* call
// ------------------------------------------------------------------------------------------------
namespace {

struct Io {                       // the caller's per-call vectors (host memory)
	real_t* x = nullptr;
	real_t* grad = nullptr;
	real_t* hv = nullptr;
	bool has_hv = false;                                // this call reads hess_vec (SQN section 4)
};

// The shard's view of the caller's per-call vectors: its own device vectors (bound), or ITS SLICE of the caller's host arrays
// -- the shard's machine then treats it like any host caller's: copies, slices and all (machines.cpp).
struct Local { real_t *x, *g, *hv; };
Local local_vectors(Shard& s, const Io& io)
{
	Local v;
	v.x = s.bound ? s.bx : (io.x ? io.x + s.off : nullptr);
	v.g = s.bound ? s.bg : (io.grad ? io.grad + s.off : nullptr);
	v.hv = s.bound ? s.bhv : ((io.hv && io.has_hv) ? io.hv + s.off : nullptr);   // a 1-element hess_vec (gradient differences) is never offset
	s.lx = v.x;
	set_thread_dev_requests(!s.bound);
	return v;
}

// what the machine left on the device for the caller: the vector *req designates, SQN's s-slot
void download(Group* g, const Shape& sp, Shard& s, const Io& io)
{
	const size_t bytes = s.cnt * sizeof(real);
	set_thread_dev_requests(false);
	if (s.ret == -1000 || s.bound) return;                     // bound: *req / *req_vec are fetched per shard (devices_request)
	bool is_x = false;
	if (s.req) {
		real_t* home = req_home(g, sp, s, s.req, io.x, &is_x);
		if (!is_x && home) SQN_HIP_OK(hipMemcpy(home + s.off, s.req, bytes, hipMemcpyDefault));
		else if (!is_x) s.ok = false;
	}
	if (s.req_vec) {                                           // SQN: the s-slot (reference src/stochqn.c:1104)
		real_t* home = g->owned ? landing(g, 1) : sp.b->s_mem + s.b.mem_st_ix * (size_t) sp.n;
		if (home) SQN_HIP_OK(hipMemcpy(home + s.off, s.req_vec, bytes, hipMemcpyDefault));
		else s.ok = false;
	}
}

// calls the front-end answers by itself ("evaluate the gradient at x"): the shards' requests follow
void request_x(const void* key)
{
	if (Group* g = find_group(key))
		for (auto& s : g->sh) { s->req = s->bound ? s->bx : nullptr; s->req_vec = nullptr; }
}

// Before the fan-out of a host caller's call: pin its vectors, and do not send x again when the shards still hold it.
void prepare_io(Group* g, Io& io, size_t n)
{
	if (g->sh[0]->bound) return;                              // device-resident caller: its vectors live on the devices
	pin_for_all_devices(g, io.x, n * sizeof(real));
	pin_for_all_devices(g, io.grad, n * sizeof(real));
	if (io.has_hv) pin_for_all_devices(g, io.hv, n * sizeof(real));
}

// (what the shards know about their copies of x is kept per shard by the machines: runtime.cpp stage_x / note_state)

void note(Group* g, size_t niter, int section)
{
	g->has_last = true;
	g->last_niter = niter;
	g->last_section = section;
}

}  // namespace

int group_run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task, workspace_oLBFGS* w,
                     info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 2) return fail(task, "oLBFGS", "got an invalid workspace as input.");
	if (w->section != 0 && (!x || !grad)) return fail(task, "oLBFGS", "got an invalid workspace as input.");      // the first call only asks for a gradient
	bfgs_mem* b = w->bfgs_memory;
	*req = x;
	if (w->section == 0) {
		if (!group_owns(b->s_mem)) drop_group(b->s_mem);        // a brand-new optimiser object at this address
		request_x(b->s_mem);
		*task = calc_grad;
		w->section = 1;
		return 0;
	}
	Shape sp;
	sp.kind = KIND_OLBFGS; sp.n = w->n; sp.b = b; sp.need_gprev = true; sp.gprev = w->grad_prev;
	sp.niter = w->niter; sp.section = w->section;
	Group* g = group_for(sp);
	if (!g) return fail(task, "oLBFGS", "could not set up the device shards of this workspace.");
	Io io;
	io.x = x; io.grad = grad;
	prepare_io(g, io, (size_t) w->n);
	for_all(g, [&](Shard& s) {
		sync_bfgs(s, b, s.rank == 0);
		s.wo.bfgs_memory = &s.b; s.wo.grad_prev = s.gprev; s.wo.hess_init = w->hess_init; s.wo.niter = w->niter;
		s.wo.section = w->section; s.wo.nthreads = w->nthreads; s.wo.check_nan = w->check_nan; s.wo.n = (int) s.cnt;
		const Local v = local_vectors(s, io);
		s.req = nullptr; s.req_vec = nullptr;
		s.ret = local_run_oLBFGS(step_size, v.x, v.g, &s.req, &s.task, &s.wo, &s.info);
		download(g, sp, s, io);
	});
	if (!agree(g, "run_oLBFGS")) { *task = invalid_input; return -1000; }
	const Shard& a = *g->sh[0];
	b->mem_used = a.b.mem_used; b->mem_st_ix = a.b.mem_st_ix;
	w->niter = a.wo.niter; w->section = a.wo.section;
	*task = a.task; *iter_info = a.info;
	note(g, w->niter, w->section);
	return a.ret;
}

int group_run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec,
                  task_enum* task, workspace_SQN* w, info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 4) return fail(task, "SQN", "got an invalid workspace as input.");
	if (w->section != 0 && (!x || !grad)) return fail(task, "SQN", "got an invalid workspace as input.");
	bfgs_mem* b = w->bfgs_memory;
	if (w->section == 0) {
		if (!group_owns(b->s_mem)) drop_group(b->s_mem);
		request_x(b->s_mem);
		*task = calc_grad;
		*req = x;
		w->section = 1;
		return 0;
	}
	Shape sp;
	sp.kind = KIND_SQN; sp.n = w->n; sp.b = b; sp.need_gprev = w->use_grad_diff != 0; sp.gprev = w->grad_prev;
	sp.need_avg = true; sp.xsum = w->x_sum; sp.xprev = w->x_avg_prev;
	sp.niter = w->niter; sp.section = w->section;
	Group* g = group_for(sp);
	if (!g) return fail(task, "SQN", "could not set up the device shards of this workspace.");
	if (w->section == 4 && !hess_vec) return fail(task, "SQN", "got an invalid workspace as input.");
	Io io;
	io.x = x; io.grad = grad; io.hv = hess_vec;
	io.has_hv = w->section == 4;
	prepare_io(g, io, (size_t) w->n);
	for_all(g, [&](Shard& s) {
		sync_bfgs(s, b, s.rank == 0);
		s.ws.bfgs_memory = &s.b; s.ws.grad_prev = s.gprev; s.ws.x_sum = s.xsum; s.ws.x_avg_prev = s.xprev;
		s.ws.use_grad_diff = w->use_grad_diff; s.ws.niter = w->niter; s.ws.section = w->section;
		s.ws.nthreads = w->nthreads; s.ws.check_nan = w->check_nan; s.ws.n = (int) s.cnt;
		if (w->use_grad_diff && !s.gprev) { s.ok = dalloc(&s.gprev, s.cnt, false); s.ws.grad_prev = s.gprev; }
		const Local v = local_vectors(s, io);
		s.req = nullptr; s.req_vec = nullptr;
		s.ret = local_run_SQN(step_size, v.x, v.g, v.hv, &s.req, &s.req_vec, &s.task, &s.ws, &s.info);
		if (s.task != calc_hess_vec) s.req_vec = nullptr;
		download(g, sp, s, io);
	});
	if (!agree(g, "run_SQN")) { *task = invalid_input; return -1000; }
	const Shard& a = *g->sh[0];
	b->mem_used = a.b.mem_used; b->mem_st_ix = a.b.mem_st_ix;
	w->niter = a.ws.niter; w->section = a.ws.section;
	*task = a.task; *iter_info = a.info;
	bool is_x = false;
	*req = req_home(g, sp, a, a.req, x, &is_x);
	if (a.req_vec) *req_vec = g->owned ? landing(g, 1) : b->s_mem + a.b.mem_st_ix * (size_t) w->n;
	note(g, w->niter, w->section);
	return a.ret;
}

int group_run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task, workspace_adaQN* w,
                    info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 5) return fail(task, "adaQN", "got an invalid workspace as input.");
	if (w->section != 0 && w->section != 3 && (!x || !grad)) return fail(task, "adaQN", "got an invalid workspace as input.");
	bfgs_mem* b = w->bfgs_memory;
	fisher_mem* fm = w->use_grad_diff ? nullptr : w->fisher_memory;
	if (w->section == 0) {
		if (!group_owns(b->s_mem)) drop_group(b->s_mem);
		request_x(b->s_mem);
		*task = calc_grad;
		*req = x;
		w->section = 1;
		return 0;
	}
	if (w->section == 3) {                                     // scalar only (reference src/stochqn.c:1258-1262)
		w->f_prev = f;
		w->section = 1;
		*task = calc_grad;
		*req = x;
		request_x(b->s_mem);
		if (Group* g = find_group(b->s_mem)) note(g, w->niter, w->section);
		return 0;
	}
	Shape sp;
	sp.kind = KIND_ADAQN; sp.n = w->n; sp.b = b; sp.f = fm; sp.need_gprev = w->use_grad_diff != 0; sp.gprev = w->grad_prev;
	sp.need_avg = true; sp.xsum = w->x_sum; sp.xprev = w->x_avg_prev;
	sp.need_diag = true; sp.H0 = w->H0; sp.G = w->grad_sum_sq;
	sp.niter = w->niter; sp.section = w->section;
	Group* g = group_for(sp);
	if (!g) return fail(task, "adaQN", "could not set up the device shards of this workspace.");
	Io io;
	io.x = x; io.grad = grad;
	prepare_io(g, io, (size_t) w->n);
	for_all(g, [&](Shard& s) {
		sync_bfgs(s, b, s.rank == 0);
		if (fm) {
			s.f.mem_used = fm->mem_used; s.f.mem_st_ix = fm->mem_st_ix;
			s.f.buffer_y = (s.rank == 0 && fm->buffer_y) ? fm->buffer_y : s.fy.data();
		}
		s.wa.bfgs_memory = &s.b; s.wa.fisher_memory = fm ? &s.f : nullptr; s.wa.H0 = s.H0; s.wa.grad_prev = s.gprev;
		s.wa.x_sum = s.xsum; s.wa.x_avg_prev = s.xprev; s.wa.grad_sum_sq = s.G;
		s.wa.f_prev = w->f_prev; s.wa.max_incr = w->max_incr; s.wa.scal_reg = w->scal_reg; s.wa.rmsprop_weight = w->rmsprop_weight;
		s.wa.use_grad_diff = w->use_grad_diff; s.wa.niter = w->niter; s.wa.section = w->section;
		s.wa.nthreads = w->nthreads; s.wa.check_nan = w->check_nan; s.wa.n = (int) s.cnt;
		if (w->use_grad_diff && !s.gprev) { s.ok = dalloc(&s.gprev, s.cnt, false); s.wa.grad_prev = s.gprev; }
		const Local v = local_vectors(s, io);
		s.req = nullptr; s.req_vec = nullptr;
		s.ret = local_run_adaQN(step_size, v.x, f, v.g, &s.req, &s.task, &s.wa, &s.info);
		download(g, sp, s, io);
	});
	if (!agree(g, "run_adaQN")) { *task = invalid_input; return -1000; }
	const Shard& a = *g->sh[0];
	for (auto& s : g->sh)
		if (fm && (s->f.mem_used != a.f.mem_used || s->f.mem_st_ix != a.f.mem_st_ix || s->wa.f_prev != a.wa.f_prev)) { *task = invalid_input; return -1000; }
	b->mem_used = a.b.mem_used; b->mem_st_ix = a.b.mem_st_ix;
	if (fm) { fm->mem_used = a.f.mem_used; fm->mem_st_ix = a.f.mem_st_ix; }
	w->f_prev = a.wa.f_prev;
	w->niter = a.wa.niter; w->section = a.wa.section;
	*task = a.task; *iter_info = a.info;
	bool is_x = false;
	*req = req_home(g, sp, a, a.req, x, &is_x);
	note(g, w->niter, w->section);
	return a.ret;
}

// ------------------------------------------------------------------------------------------------
// device-resident callers of the mode: per-shard vectors used in place, per-shard requests, per-shard work
// ------------------------------------------------------------------------------------------------
int group_layout(const void* key, int shard, int* device, size_t* offset, size_t* count)
{
	Group* g = find_group(key);
	if (!g || shard < 0 || shard >= g->P) return -1000;
	const Shard& s = *g->sh[(size_t) shard];
	if (device) *device = s.device;
	if (offset) *offset = s.off;
	if (count) *count = s.cnt;
	return 0;
}

int group_bind(const void* key, int shard, real_t* x, real_t* grad, real_t* hess_vec)
{
	Group* g = find_group(key);
	if (!g || shard < 0 || shard >= g->P) return -1000;
	Shard& s = *g->sh[(size_t) shard];
	if (!x && !grad && !hess_vec) { s.bound = false; s.bx = s.bg = s.bhv = nullptr; return 0; }
	if (!x || !grad || !is_device_pointer(x) || !is_device_pointer(grad) || (hess_vec && !is_device_pointer(hess_vec))) return -1000;
	s.bx = x; s.bg = grad; s.bhv = hess_vec;
	s.bound = true;
	return 0;
}

int group_request(const void* key, int shard, real_t** req, real_t** req_vec)
{
	Group* g = find_group(key);
	if (!g || shard < 0 || shard >= g->P) return -1000;
	const Shard& s = *g->sh[(size_t) shard];
	if (req) *req = s.req;
	if (req_vec) *req_vec = s.req_vec;
	return 0;
}

int group_foreach(const void* key, void (*fn)(void*, int, int, size_t, size_t), void* user)
{
	Group* g = find_group(key);
	if (!g || !fn) return -1000;
	for_all(g, [&](Shard& s) { fn(user, s.rank, s.device, s.off, s.cnt); });
	return 0;
}

// ------------------------------------------------------------------------------------------------
// checkpoint: shards -> the caller's host arrays (profile B); owned workspaces have no host arrays
// ------------------------------------------------------------------------------------------------
bool group_export(const void* key, int* rc)
{
	Group* g = find_group(key);
	if (!g) return false;
	*rc = 0;
	if (g->owned) return true;
	const HostArrays h = g->host;
	const size_t n = (size_t) g->n, m = g->m, fsz = g->fsize;
	for_all(g, [&](Shard& s) {
		const size_t c = s.cnt;
		if (h.S) rows_d2h(h.S, s.S, m, n, s.off, c);
		if (h.Y) rows_d2h(h.Y, s.Y, m, n, s.off, c);
		if (h.sbak && s.sbak) rows_d2h(h.sbak, s.sbak, 1, n, s.off, c);
		if (h.ybak && s.ybak) rows_d2h(h.ybak, s.ybak, 1, n, s.off, c);
		if (h.gprev && s.gprev) rows_d2h(h.gprev, s.gprev, 1, n, s.off, c);
		if (h.xsum && s.xsum) rows_d2h(h.xsum, s.xsum, 1, n, s.off, c);
		if (h.xprev && s.xprev) rows_d2h(h.xprev, s.xprev, 1, n, s.off, c);
		if (h.H0 && s.H0) rows_d2h(h.H0, s.H0, 1, n, s.off, c);
		if (h.G && s.G) rows_d2h(h.G, s.G, 1, n, s.off, c);
		if (h.F && s.F && fsz) rows_d2h(h.F, s.F, fsz, n, s.off, c);
		if (hipDeviceSynchronize() != hipSuccess) s.ok = false;
	});
	for (auto& s : g->sh) if (!s->ok) *rc = -1000;
	return true;
}

// ------------------------------------------------------------------------------------------------
// library-owned sharded workspaces (reference src/stochqn.c:300-547 in group mode)
// ------------------------------------------------------------------------------------------------
namespace {

// The arrays of an owned sharded workspace exist only as device slices; the struct fields the caller
// can see hold distinct non-NULL host addresses inside one small block ("tokens"): never dereferenced
// by the library, not device pointers, and s_mem's token is the registry key.
struct Tokens { char slot[16][8]; };

real_t* tok(void* block, int i) { return reinterpret_cast<real_t*>(static_cast<Tokens*>(block)->slot[i]); }

bfgs_mem* owned_bfgs(void* block, size_t mem_size, real_t min_curvature, real_t y_reg, size_t upd_freq)
{
	bfgs_mem* b = (bfgs_mem*) std::calloc(1, sizeof(bfgs_mem));
	if (!b) return nullptr;
	b->s_mem = tok(block, 0); b->y_mem = tok(block, 1);
	b->buffer_rho = (real_t*) std::calloc(mem_size, sizeof(real_t));
	b->buffer_alpha = (real_t*) std::calloc(mem_size, sizeof(real_t));
	if (min_curvature > 0) { b->s_bak = tok(block, 2); b->y_bak = tok(block, 3); }
	b->mem_size = mem_size; b->upd_freq = upd_freq; b->y_reg = y_reg; b->min_curvature = min_curvature;
	if (!b->buffer_rho || !b->buffer_alpha) { std::free(b->buffer_rho); std::free(b->buffer_alpha); std::free(b); return nullptr; }
	return b;
}

bool register_owned(const Shape& sp, void* block)
{
	const int P = shards_for(sp.n);
	if (P < 1) return false;
	Group* g = create_group(sp, P, true, false);
	if (!g) return false;
	g->token = block;
	std::lock_guard<std::mutex> lk(g_gmu);
	g_groups[sp.b->s_mem] = g;
	return true;
}

void free_owned_bfgs(bfgs_mem* b)
{
	if (!b) return;
	std::free(b->buffer_rho);
	std::free(b->buffer_alpha);
	std::free(b);
}

}  // namespace

workspace_oLBFGS* group_initialize_oLBFGS(int n, size_t mem_size, real_t hess_init, real_t y_reg, real_t min_curvature,
                                          int check_nan, int nthreads)
{
	void* block = std::calloc(1, sizeof(Tokens));
	workspace_oLBFGS* w = (workspace_oLBFGS*) std::calloc(1, sizeof(*w));
	bfgs_mem* b = block ? owned_bfgs(block, mem_size, min_curvature, y_reg, 1) : nullptr;
	if (!block || !w || !b) { std::free(block); std::free(w); free_owned_bfgs(b); return nullptr; }
	w->bfgs_memory = b; w->grad_prev = tok(block, 4); w->hess_init = hess_init; w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	Shape sp;
	sp.kind = KIND_OLBFGS; sp.n = n; sp.b = b; sp.need_gprev = true;
	if (!register_owned(sp, block)) { std::free(block); std::free(w); free_owned_bfgs(b); return nullptr; }
	return w;
}

workspace_SQN* group_initialize_SQN(int n, size_t mem_size, size_t bfgs_upd_freq, real_t min_curvature, int use_grad_diff,
                                    real_t y_reg, int check_nan, int nthreads)
{
	void* block = std::calloc(1, sizeof(Tokens));
	workspace_SQN* w = (workspace_SQN*) std::calloc(1, sizeof(*w));
	bfgs_mem* b = block ? owned_bfgs(block, mem_size, min_curvature, y_reg, bfgs_upd_freq) : nullptr;
	if (!block || !w || !b) { std::free(block); std::free(w); free_owned_bfgs(b); return nullptr; }
	w->bfgs_memory = b; w->grad_prev = use_grad_diff ? tok(block, 4) : nullptr; w->x_sum = tok(block, 5); w->x_avg_prev = tok(block, 6);
	w->use_grad_diff = use_grad_diff; w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	Shape sp;
	sp.kind = KIND_SQN; sp.n = n; sp.b = b; sp.need_gprev = use_grad_diff != 0; sp.need_avg = true;
	if (!register_owned(sp, block)) { std::free(block); std::free(w); free_owned_bfgs(b); return nullptr; }
	return w;
}

workspace_adaQN* group_initialize_adaQN(int n, size_t mem_size, size_t fisher_size, size_t bfgs_upd_freq, real_t max_incr,
                                        real_t min_curvature, real_t scal_reg, real_t rmsprop_weight, int use_grad_diff,
                                        real_t y_reg, int check_nan, int nthreads)
{
	void* block = std::calloc(1, sizeof(Tokens));
	workspace_adaQN* w = (workspace_adaQN*) std::calloc(1, sizeof(*w));
	bfgs_mem* b = block ? owned_bfgs(block, mem_size, min_curvature, y_reg, bfgs_upd_freq) : nullptr;
	fisher_mem* fm = nullptr;
	if (block && w && b && !use_grad_diff) {
		fm = (fisher_mem*) std::calloc(1, sizeof(fisher_mem));
		if (fm) { fm->F = tok(block, 7); fm->buffer_y = (real_t*) std::calloc(fisher_size ? fisher_size : 1, sizeof(real_t)); fm->mem_size = fisher_size; }
	}
	if (!block || !w || !b || (!use_grad_diff && (!fm || !fm->buffer_y || fisher_size == 0))) {
		if (fm) { std::free(fm->buffer_y); std::free(fm); }
		std::free(block); std::free(w); free_owned_bfgs(b);
		return nullptr;
	}
	w->bfgs_memory = b; w->fisher_memory = fm; w->H0 = tok(block, 8); w->grad_prev = use_grad_diff ? tok(block, 4) : nullptr;
	w->x_sum = tok(block, 5); w->x_avg_prev = tok(block, 6); w->grad_sum_sq = tok(block, 9);
	w->max_incr = max_incr; w->scal_reg = scal_reg; w->rmsprop_weight = rmsprop_weight; w->use_grad_diff = use_grad_diff;
	w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	Shape sp;
	sp.kind = KIND_ADAQN; sp.n = n; sp.b = b; sp.f = fm; sp.need_gprev = use_grad_diff != 0; sp.need_avg = true; sp.need_diag = true;
	if (!register_owned(sp, block)) {
		if (fm) { std::free(fm->buffer_y); std::free(fm); }
		std::free(block); std::free(w); free_owned_bfgs(b);
		return nullptr;
	}
	return w;
}

void group_dealloc(const void* s_mem)
{
	Group* g = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_gmu);
		auto it = g_groups.find(s_mem);
		if (it == g_groups.end()) return;
		g = it->second;
		g_groups.erase(it);
	}
	void* block = g->token;
	destroy_group(g);
	std::free(block);
}

}  // namespace sqn
