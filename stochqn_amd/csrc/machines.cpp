// machines.cpp -- the free-mode C ABI (include/stochqn.h) on top of the HIP sweeps.
//
// Host side of the hot path: the three reverse-communication state machines (reference
// src/stochqn.c:978-1315), take_step (:802-840), correction-pair construction (:861-966) and the
// ring bookkeeping (:554-610), written against the transition tables of SURVEY.md section 8a.
// Every vector operation is a kernel from kernels.hip on the context's stream; the host only
// moves counters, and reads back at most a handful of scalars per call.
//
// Residency rules (DESIGN.md "boundary"):
//   * struct arrays that are device pointers are used in place; host arrays are mirrored in HBM
//     (context keyed by bfgs_mem.s_mem) and only x, grad, *req, *req_vec are kept in step;
//   * x / grad / hess_vec are classified per call; host ones are staged over PCIe.
#include "machines.hpp"
#include "stochqn_hip.h"

#include <cmath>
#include <exception>
#include <cstdlib>
#include <cstring>

using namespace sqn;

#ifndef SQN_USE_FLOAT
static_assert(sizeof(bfgs_mem) == 96 && sizeof(fisher_mem) == 40, "ABI layout (reference include/stochqn.h:86-107)");
static_assert(sizeof(workspace_oLBFGS) == 48 && sizeof(workspace_SQN) == 64 && sizeof(workspace_adaQN) == 120,
              "ABI layout (reference include/stochqn.h:109-151)");
#else   // the same structs with real_t = float (reference -DUSE_FLOAT build)
static_assert(sizeof(bfgs_mem) == 88 && sizeof(fisher_mem) == 40, "ABI layout, float (reference include/stochqn.h:86-107)");
static_assert(sizeof(workspace_oLBFGS) == 48 && sizeof(workspace_SQN) == 64 && sizeof(workspace_adaQN) == 104,
              "ABI layout, float (reference include/stochqn.h:109-151)");
#endif
static_assert(sizeof(real) == sizeof(real_t), "library element type and ABI real_t must agree");

namespace {

// set by a shard thread of the single-process multi-device mode around its local_run_* calls (group.cpp)
thread_local bool t_dev_requests = false;

// ---- one run_* invocation -----------------------------------------------------------------------
struct Call {
	DevCtx* c = nullptr;
	bool host_caller = false;       // x lives in host memory: *req / *req_vec must be host-readable
	real* x_caller = nullptr;
	real* g_caller = nullptr;
	real* x = nullptr;              // device views
	real* g = nullptr;
	bool g_host = false;
	bool fresh = false;             // the device context was created by this call
	bool x_down = false, g_down = false;   // the update pass already sent x / the direction to the host, slice by slice
	bool g_pending = false;                // host gradient not uploaded yet: pass 1 of the three-pass form takes it in slices
	bool x_pending = false;                // host x not uploaded yet: the update takes it in slices, each just ahead of the slice of the update that reads it
	bool dev_requests = false;             // shard of a multi-device group: *req / *req_vec stay device pointers, the group copies them out
	bool x_spec = false;                   // slices of x went to the host before the guard had spoken (step_was_bad puts a rejected step right)
	XHashJob* x_job = nullptr;             // option "x_upload" = 2: the checksum of the caller's x is being taken; x_pending until it says "unchanged" (resolve_x)
	Call() = default;
	Call(const Call&) = delete;
	Call& operator=(const Call&) = delete;
	~Call() { if (x_job) (void) xhash_finish(x_job); }
};

inline size_t N(const DevCtx* c) { return (size_t) c->n; }
inline real* row(View& v, size_t r, const DevCtx* c) { return v.dev + r * N(c); }

void d2d(DevCtx* c, real* dst, const real* src, size_t count)
{
	const size_t pr = c->sc.prof ? c->sc.prof->begin(K_COPY, c->sc.stream) : 0;
	SQN_HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(real), hipMemcpyDeviceToDevice, c->sc.stream));
	if (c->sc.prof) c->sc.prof->end(pr, c->sc.stream);
}

void zero(DevCtx* c, real* dst, size_t count)
{
	const size_t pr = c->sc.prof ? c->sc.prof->begin(K_COPY, c->sc.stream) : 0;
	SQN_HIP_OK(hipMemsetAsync(dst, 0, count * sizeof(real), c->sc.stream));
	if (c->sc.prof) c->sc.prof->end(pr, c->sc.stream);
}

void to_host(DevCtx* c, void* dst, const double* src, size_t count)      // scalars of the recursion
{
	if (c->async_call) return;               // stream-ordered call: nothing is read back (and nothing that would have been is used)
	if (c->wedged) return;                   // the stream can never run again (runtime.cpp: wait_stream): the call fails, nothing is read
	SQN_HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToHost, c->sc.stream));
}

void vec_to_host(DevCtx* c, real* dst, const real* src, size_t count)    // an n-vector
{
	// a copy into pageable caller memory blocks INSIDE the runtime until the stream has run: on a wedged context that is for ever
	if (c->wedged) return;
	SQN_HIP_OK(hipMemcpyAsync(dst, src, count * sizeof(real), hipMemcpyDeviceToHost, c->sc.stream));
}

// Copy a few scalars that were read back into the pinned block on to a caller buffer (buffer_rho,
// buffer_alpha, buffer_y), which may itself live in host or in device memory.  After sync() only.
void hand_back(real_t* dst, const double* pinned, size_t count)
{
	if (!dst || count == 0) return;
	const bool on_device = is_device_pointer(dst);
	real_t tmp[256];
	for (size_t done = 0; done < count; done += 256) {           // converts double -> real_t on the way
		const size_t k = count - done < 256 ? count - done : 256;
		for (size_t i = 0; i < k; i++) tmp[i] = (real_t) pinned[done + i];
		if (on_device) SQN_HIP_OK(hipMemcpy(dst + done, tmp, k * sizeof(real_t), hipMemcpyHostToDevice));
		else std::memcpy(dst + done, tmp, k * sizeof(real_t));
	}
}

bool bind_bfgs(DevCtx* c, bfgs_mem* b, bool import_rows)
{
	const size_t n = N(c), m = b->mem_size;
	const size_t bak = b->min_curvature > 0 ? n : 0;
	return bind(c, c->S, b->s_mem, m * n, import_rows) && bind(c, c->Y, b->y_mem, m * n, import_rows) &&
	       bind(c, c->sbak, b->s_bak, bak, true) && bind(c, c->ybak, b->y_bak, bak, true);
}

// Start a call: find the context, bind every struct array, classify and stage x / grad.
bool open_call(Call& io, int kind, int n, bfgs_mem* b, size_t fsize, bool resumed, real* x, real* grad,
               size_t niter, int section, int check_nan)
{
	if (!b || !b->s_mem || !b->y_mem || n <= 0 || b->mem_size == 0) return false;
	(void) hipGetLastError();        // errors other code left behind on this thread are not ours (see sync())
	(void) take_hip_failure();
	bool fresh = false;
	DevCtx* c = acquire(b->s_mem, kind, n, b->mem_size, fsize, &fresh);
	if (!c) return false;
	if (!fresh && c->has_last && (c->last_niter != niter || c->last_section != section)) {
		// Same address, same shape, but not the state this context last handed back: the arrays
		// belong to another optimiser object now (R / Python never call dealloc_*).  Start over
		// from what the caller passes (host arrays are re-imported, caches dropped).
		release(b->s_mem);
		c = acquire(b->s_mem, kind, n, b->mem_size, fsize, &fresh);
		if (!c) return false;
	}
	io.c = c;
	io.fresh = fresh;
	io.dev_requests = t_dev_requests;
	if (fresh) attach_spill(c, niter, section);      // reclaimed while idle?  then its state comes back from the library's host copy
	// a context that could not be completed is dropped again: the next call starts over (and re-imports host arrays, or
	// finds the state a reclaimed predecessor left in host memory once more) instead of continuing on half-bound views
	if (!bind_bfgs(c, b, fresh && resumed)) { abandon_context(b->s_mem); return false; }
	if (!c->attached) {                  // first call on this context (created now, or ahead of time by a shard group)
		comm_attach(c);
		c->forget_rows();
		c->attached = true;
	}
	io.x_caller = x;
	io.g_caller = grad;
	io.host_caller = !is_device_pointer(x);
	io.g_host = !is_device_pointer(grad);
	// small problems of device-resident callers: no cross-stream hand-off per call (option "null_stream" = 2; nothing is
	// pending on either stream between calls, every synchronous call ends with a synchronisation of the stream it used)
	if (options().null_stream == 2 && !options().async_device)
		c->sc.stream = (!io.host_caller && !io.g_host && N(c) <= ((size_t) 1 << 22)) ? nullptr : c->own_stream;
	// option "async_device": a device-resident caller of a configuration where nothing can be rejected gets its call back
	// as soon as the kernels are enqueued (sync() then only looks for launch errors)
	c->async_call = options().async_device && !io.host_caller && !io.g_host && !c->S.mirror && !c->Y.mirror && check_nan == 0 &&
	                !(b->min_curvature > 0) && !options().verify_cache && c->sc.allreduce == nullptr;
	// staging vectors for host x / grad exist before anything is enqueued: stage_xg cannot fail later
	if ((io.host_caller && !ensure_stage(c, 0)) || (io.g_host && !ensure_stage(c, 1))) {
		if (fresh) abandon_context(b->s_mem);
		else end_use(c);
		return false;
	}
	return true;
}

// may_defer (the call that takes the step): a large host gradient is not sent in one piece here -- the first pass of the
// three-pass form fetches it slice by slice and starts on each slice as it lands (enqueue_three_pass); every other path
// asks for it with flush_g first.
void stage_xg(Call& io, bool need_x, bool need_g, bool may_defer = false)
{
	DevCtx* c = io.c;
	if (need_x) {
		// A large host x that has to go up (always, by default: the reference's *req aliases x) does not go up front: nothing
		// before the update reads it, so its slices travel while the two-loop runs and land just ahead of the slices of the
		// update -- whose results start their way down at once, on a stream of their own (the link is full duplex).
		if (io.host_caller && may_defer && options().apply_chunks >= 2 && N(c) >= (size_t) options().host_slice_min && ensure_stage(c, 0) &&
		    ensure_copy_stream(c, options().apply_chunks + 1)) {
			io.x = c->stage[0];
			if (options().x_upload == 2 && c->x_hash_valid && c->x_hash_count == N(c) && c->kind != KIND_RAW && !c->x_pre_pending &&
			    (io.x_job = xhash_start(io.x_caller, N(c) * sizeof(real), xhash_threads(c))) != nullptr) {
				// the checksum of the caller's x is on its way (threads of its own); whoever needs x first asks resolve_x
				(void) ensure_registered(c, io.x_caller, N(c) * sizeof(real));
				io.x_pending = true;
				c->x_valid = false;
			}
			else if (!x_is_current(c, io.x_caller, N(c))) {
				(void) ensure_registered(c, io.x_caller, N(c) * sizeof(real));
				io.x_pending = true;
			}
}
		else io.x = io.host_caller ? stage_x(c, io.x_caller, N(c)) : io.x_caller;
	}
	if (!need_g) return;
	const int slices = options().upload_slices;
	if (may_defer && io.g_host && slices >= 2 && N(c) >= (size_t) options().host_slice_min && ensure_stage(c, 1) &&
	    ensure_upload_slices(c, slices, sdot_carry_count(c->sc, N(c), (int) c->m))) {
		(void) ensure_registered(c, io.g_caller, N(c) * sizeof(real));
		io.g = c->stage[1];
		io.g_pending = true;
		return;
	}
	io.g = stage_in(c, 1, io.g_caller, N(c), io.g_host);
}

void flush_g(Call& io)
{
	if (!io.g_pending) return;
	SQN_HIP_OK(hipMemcpyAsync(io.g, io.g_caller, N(io.c) * sizeof(real), hipMemcpyHostToDevice, io.c->sc.stream));
	io.g_pending = false;
}

// Option "x_upload" = 2: the verdict of the checksum that stage_xg started.  Called by whoever is about to enqueue the first
// thing that reads or overwrites x (the caller's thread has enqueued pass 1 and pass 2 meanwhile; the device is busy with the
// upload of the gradient for longer than the checksum takes).  Equal sums: the device holds the caller's x, nothing goes up.
void resolve_x(Call& io)
{
	if (!io.x_job) return;
	DevCtx* c = io.c;
	const XHash h = xhash_finish(io.x_job);
	io.x_job = nullptr;
	const bool same = c->x_hash_valid && h.a == c->x_hash.a && h.b == c->x_hash.b;
	c->x_hash_valid = false;                // x is about to change on the device: the sum is taken anew when the call ends (close_call)
	if (same) { io.x_pending = false; stat_add(ST_X_UPLOAD_SKIPPED); }
	else stat_add(ST_X_UPLOAD);
}

void flush_x(Call& io)                      // whoever reads x in one piece asks for it here first
{
	resolve_x(io);
	if (!io.x_pending) return;
	SQN_HIP_OK(hipMemcpyAsync(io.x, io.x_caller, N(io.c) * sizeof(real), hipMemcpyHostToDevice, io.c->sc.stream));
	io.x_pending = false;
}

// elements [lo, hi) of a pending host x go up on the side stream; the main stream waits for them (event `slot` of xup_ev)
void x_slice_up(Call& io, size_t lo, size_t hi, int slot)
{
	DevCtx* c = io.c;
	SQN_HIP_OK(hipMemcpyAsync(io.x + lo, io.x_caller + lo, (hi - lo) * sizeof(real), hipMemcpyHostToDevice, c->copy_stream));
	SQN_HIP_OK(hipEventRecord(c->xup_ev[(size_t) slot], c->copy_stream));
	SQN_HIP_OK(hipStreamWaitEvent(c->sc.stream, c->xup_ev[(size_t) slot], 0));
	c->copy_busy = true;
}

// SliceFeed::arrive of a pending host gradient: elements [lo, hi) go up on the side stream, the main stream waits for them
void gradient_slice_arrives(void* user, size_t lo, size_t hi, int slice)
{
	Call& io = *static_cast<Call*>(user);
	DevCtx* c = io.c;
	SQN_HIP_OK(hipMemcpyAsync(io.g + lo, io.g_caller + lo, (hi - lo) * sizeof(real), hipMemcpyHostToDevice, c->copy_stream));
	SQN_HIP_OK(hipEventRecord(c->up_ev[(size_t) slice], c->copy_stream));
	SQN_HIP_OK(hipStreamWaitEvent(c->sc.stream, c->up_ev[(size_t) slice], 0));
	c->copy_busy = true;
	io.g_pending = false;
}

// Make a workspace array readable where the caller expects to read `*req` from.
real* publish(Call& io, View& v, size_t offset, int slot)
{
	DevCtx* c = io.c;
	real* dev = v.dev + offset;
	if (!io.host_caller || io.dev_requests) return dev;
	if (v.mirror) {                       // caller's own host array: refresh it, hand it back
		real* host = (real*) const_cast<void*>(v.caller) + offset;
		if (v.count == N(c)) (void) ensure_registered(c, host, N(c) * sizeof(real));      // x_sum / x_avg_prev: one array, requested every L steps
		vec_to_host(c, host, dev, N(c));
		return host;
	}
	real* landing = host_landing(c, slot);
	if (!landing) {                       // no host memory left: nothing sensible to hand out
		std::fprintf(stderr, "stochqn: could not allocate a host buffer for the requested vector\n");
		return nullptr;
	}
	vec_to_host(c, landing, dev, N(c));
	return landing;
}

// End of a call that touched x / grad: bring host copies up to date, then synchronise.
void close_call(Call& io, bool x_changed, bool g_changed)
{
	DevCtx* c = io.c;
	resolve_x(io);
	// ranks of a communicator: a copy into pageable caller memory blocks INSIDE the runtime until the stream has run -- for ever if a
	// collective on it waits for a peer that is gone.  The bounded wait comes first (runtime.cpp: wait_stream); a no-op without RCCL.
	// Whatever the wait reports -- the patience ran out, or the stream itself failed -- fails the call (c->fault): nothing is
	// copied into the caller's arrays behind an error that was looked away from.
	if (c->red.kind == Reducer::RCCL && io.host_caller && (x_changed || g_changed) && wait_stream(c, c->sc.stream) != hipSuccess) {
		(void) hipGetLastError();
		c->fault = true;
	}
	// a reduction that failed while the call was being enqueued (c->fault: the call is going to return -1000) left the update
	// working on un-reduced sums: the caller's arrays are not touched with that
	if (x_changed && io.host_caller && io.x && !io.x_down && !io.x_pending && !c->fault) vec_to_host(c, io.x_caller, io.x, N(c));
	if (g_changed && io.g_host && io.g && options().strict_grad && !io.g_down && !c->fault) vec_to_host(c, io.g_caller, io.g, N(c));
	// option "x_upload" = 2: the checksum of x as this call leaves it on the device (large host x only: the rest goes up in one piece)
	const bool sum = options().x_upload == 2 && io.host_caller && io.x && !c->fault && !c->async_call && c->kind != KIND_RAW &&
	                 N(c) >= (size_t) options().host_slice_min;
	if (sum) {
		SQN_HIP_OK(hipMemsetAsync(c->sc.report + 4, 0, 2 * sizeof(double), c->sc.stream));
		launch_xhash(c->sc, io.x, N(c), c->sc.report + 4);
		to_host(c, c->pin + 4, c->sc.report + 4, 2);
	}
	if (io.host_caller && io.x) c->x_hash_valid = false;
	sync(c);
	if (io.host_caller && io.x) x_handed_back(c, io.x_caller, N(c));      // device and host copies of x agree from here on
	if (sum && !c->fault) {
		std::memcpy(&c->x_hash.a, c->pin + 4, sizeof(double));
		std::memcpy(&c->x_hash.b, c->pin + 5, sizeof(double));
		c->x_hash_count = N(c);
		c->x_hash_valid = true;
	}
}

// The guarded update for a HOST caller of a large problem: the pass is element-wise, so it runs slice by slice (bit-identical
// to one launch) and each finished slice of x -- and of the direction when the caller wants it back -- starts its way over
// PCIe on a side stream while the next slice is still being updated.
void apply_step(Call& io, Partials guard, const real* r_in, real* grad_out, const ApplyArgs& ap, bool guarded)
{
	DevCtx* c = io.c;
	const Scratch& sc = c->sc;
	const size_t n = N(c);
	resolve_x(io);
	const bool want_x = io.host_caller && io.x == ap.x, want_g = io.g_host && options().strict_grad && io.g == grad_out;
	const int chunks = options().apply_chunks;
	const size_t min_chunk = (size_t) options().host_slice_min / 2;     // elements: below this a slice is all launch overhead
	if ((!want_x && !want_g) || c->fault || chunks < 2 || n < 2 * min_chunk || !ensure_copy_stream(c, chunks + 1)) {
		flush_x(io);
		launch_apply(sc, n, c->n_global, guard, r_in, grad_out, ap, guarded);
		return;
	}
	const bool x_up = io.x_pending && io.x == ap.x;          // x itself still has to come up: slice by slice, just ahead of the update
	if (!x_up) flush_x(io);
	(void) ensure_registered(c, io.x_caller, n * sizeof(real));
	if (want_g) (void) ensure_registered(c, io.g_caller, n * sizeof(real));
	size_t per = (n + (size_t) chunks - 1) / (size_t) chunks;
	if (per < min_chunk) per = min_chunk;
	per = (per + 2 * kVec - 1) / (2 * kVec) * (2 * kVec);   // whole packs: every slice keeps the alignment of the vector
	int j = 0;
	for (size_t off = 0; off < n; off += per, j++) {
		const size_t cnt = n - off < per ? n - off : per;
		ApplyArgs a = ap;
		a.x = ap.x + off;
		if (ap.x_sum) a.x_sum = ap.x_sum + off;
		if (ap.s_slot) a.s_slot = ap.s_slot + off;
		if (x_up) x_slice_up(io, off, off + cnt, j);
		launch_apply(sc, cnt, c->n_global, guard, r_in + off, grad_out + off, a, guarded);
		SQN_HIP_OK(hipEventRecord(c->chunk_ev[(size_t) j], sc.stream));
		SQN_HIP_OK(hipStreamWaitEvent(c->down_stream, c->chunk_ev[(size_t) j], 0));
		if (want_x) SQN_HIP_OK(hipMemcpyAsync(io.x_caller + off, ap.x + off, cnt * sizeof(real), hipMemcpyDeviceToHost, c->down_stream));
		if (want_g) SQN_HIP_OK(hipMemcpyAsync(io.g_caller + off, grad_out + off, cnt * sizeof(real), hipMemcpyDeviceToHost, c->down_stream));
	}
	if (x_up) io.x_pending = false;
	c->copy_busy = true;
	io.x_down = want_x;
	io.g_down = want_g;
}

// ---- ring bookkeeping (reference src/stochqn.c:554-579) ----------------------------------------
void ring_reset(bfgs_mem* b) { b->mem_used = 0; b->mem_st_ix = 0; }
void ring_advance(bfgs_mem* b)
{
	b->mem_st_ix = (b->mem_st_ix + 1) % b->mem_size;
	b->mem_used = (b->mem_used + 1 >= b->mem_size) ? b->mem_size : b->mem_used + 1;
}
void fisher_reset(fisher_mem* f) { if (f) { f->mem_used = 0; f->mem_st_ix = 0; } }
void fisher_advance(fisher_mem* f)
{
	f->mem_st_ix = (f->mem_st_ix + 1) % f->mem_size;
	f->mem_used = (f->mem_used + 1 >= f->mem_size) ? f->mem_size : f->mem_used + 1;
}

// ---- two-loop recursion (reference src/stochqn.c:663-708) as a chain of fused sweeps -----------
// Logical pair i lives in physical row (st + i) % m, st = row of the oldest pair.
void ensure_rho(DevCtx* c, size_t st, size_t used)
{
	for (size_t i = 0; i < used; i++) {
		const size_t r = (st + i) % c->m;
		if (c->rho_ok[r]) continue;
		Partials p = launch_dots3(c->sc, c->next_buf(), N(c), row(c->S, r, c), row(c->Y, r, c));
		launch_commit(c->sc, p, c->sc.sy + r, c->sc.yy + r);
		c->rho_ok[r] = 1;
	}
}

// the isolated entry points hold their context for the duration of one call
struct InUse {
	DevCtx* c;
	explicit InUse(DevCtx* c_) : c(c_) {}
	~InUse() { end_use(c); }
};

struct StepIn {
	double step = 0;
	real* x = nullptr;
	real* g = nullptr;
	size_t used = 0, st_ix = 0;      // ring counters as the caller's struct has them
	double h0 = 0;                   // oLBFGS hess_init
	real* H0 = nullptr;              // adaQN diagonal target
	real* G = nullptr;               // adaQN grad_sum_sq (NULL = no rescaling)
	double w = 0, eps = 0;
	real* gprev_out = nullptr;       // oLBFGS
	real* frow_out = nullptr;        // adaQN Fisher row
	real* x_sum = nullptr;           // SQN / adaQN
	real* s_slot = nullptr;          // oLBFGS
	int check_nan = 0;
};

// Returns the last-forward / guard partials; after this the direction is in `g`.
Partials enqueue_two_loop(DevCtx* c, real* g, size_t used, size_t st, const FirstArgs& fa_in, double h0,
                          real* H0, const ApplyArgs* fuse)
{
	const Scratch& sc = c->sc;
	const size_t n = N(c), m = c->m, k = used;
	auto r_of = [&](size_t i) { return (st + i) % m; };
	ensure_rho(c, st, k);

	FirstArgs fa = fa_in;
	fa.q = g;
	fa.s_newest = row(c->S, r_of(k - 1), c);
	Partials p = launch_first(sc, c->next_buf(), n, fa);
	for (size_t i = k - 1; i >= 1; i--)
		p = launch_bwd(sc, c->next_buf(), n, p, sc.sy + r_of(i), (int) i, row(c->Y, r_of(i), c), g, row(c->S, r_of(i - 1), c));

	MidScale ms{};
	if (H0 != nullptr) ms.H0 = H0;                                   // :695
	else if (h0 > 0) ms.h0 = h0;                                     // :698
	else { ms.sy_newest = sc.sy + r_of(k - 1); ms.yy_newest = sc.yy + r_of(k - 1); }   // :683-689
	p = launch_mid(sc, c->next_buf(), n, p, sc.sy + r_of(0), row(c->Y, r_of(0), c), g, ms);

	for (size_t i = 0; i + 1 < k; i++)
		p = launch_fwd(sc, c->next_buf(), n, p, sc.sy + r_of(i), (int) i, row(c->S, r_of(i), c), g, row(c->Y, r_of(i + 1), c));
	return launch_fwd_last(sc, c->next_buf(), n, p, sc.sy + r_of(k - 1), (int) (k - 1), row(c->S, r_of(k - 1), c), g, fuse);
}

// ---- the cached (three-pass) form and its fallback rule ------------------------------------------------
// Expanding the recursion over cached inner products is the same arithmetic as the reference's sequential
// sweeps associated differently: on every input class probed (tests/test_gpu_adversarial.py: Hessian
// condition 1e8, nearly collinear pairs, inconsistent and negative curvature, g in the span of Y,
// |g| ~ 1e+-150) both forms sit within a few ulps of an extended-precision evaluation.  The exception is
// pairs with s almost orthogonal to y (rho_i = 1/s'y huge): there every form loses digits -- about
// kappa_i = |s||y|/|s'y| per pair.  The three-pass form tracks the sweeps within 1.1x up to kappa = 1e10
// (profiles/r03_kappa_sweep_k20.json); it is used only while every pair in use has kappa_i <= option
// "kappa_max" (default 1e6, conservative); beyond, the step runs as the reference's own chain of sweeps.
// kappa comes from the all-reduced (s'y, s's, y'y) of the pair: identical on every rank.
bool pairs_tame(DevCtx* c, size_t st, size_t used)
{
	const double kmax = options().kappa_max;
	if (!(kmax > 0) || std::isinf(kmax)) return true;            // rule switched off
	const size_t m = c->m;
	size_t unknown = 0;
	for (size_t i = 0; i < used; i++) {
		const size_t r = (st + i) % m;
		if (c->kappa[r] >= 0) continue;
		if (c->async_call) { c->kappa[r] = 0; continue; }
		// a pair that did not come through accept_or_reject (imported state, isolated entry points, the bak->slot
		// quirk): its three dots now, with a read-back (rare path, one synchronisation for all such rows)
		Partials p = launch_dots3(c->sc, c->next_buf(), N(c), row(c->S, r, c), row(c->Y, r, c));
		launch_commit(c->sc, p, c->sc.sy + r, c->sc.yy + r);
		c->rho_ok[r] = 1;
		launch_fin(c->sc, p, 3, c->kap_dev + 3 * r);
		unknown++;
	}
	if (unknown) {
		double* land = c->pin + 16 + 2 * m + c->fsize;
		to_host(c, land, c->kap_dev, 3 * m);
		sync(c);
		for (size_t i = 0; i < used; i++) {
			const size_t r = (st + i) % m;
			if (c->kappa[r] >= 0) continue;
			const double sy = land[3 * r], ss = land[3 * r + 1], yy = land[3 * r + 2];
			const double k = std::sqrt(ss) * std::sqrt(yy) / std::fabs(sy);
			c->kappa[r] = (k >= 0) ? k : INFINITY;                 // NaN (0/0, non-finite dots) counts as untame
		}
	}
	for (size_t i = 0; i < used; i++)
		if (!(c->kappa[(st + i) % m] <= kmax)) return false;
	return true;
}

// ---- three-pass form: S twice, Y once (kernels.hip "three-pass form") ------------------------------------------
// The cached block holds s_a'y_b for pairs a older than b.  Column b is computed when pair b is the newest: in
// the normal course exactly one column is missing on the step after a pair was accepted, and it comes out of
// that step's pass 1 (one extra probe).  Anything else -- imported state, invalidate, the bak->slot quirk --
// rebuilds the missing columns with one stand-alone pass over S each (rare path).  Returns the ring row whose
// column pass 1 has to produce, or -1.
int ensure_sy_columns(DevCtx* c, size_t st, size_t used, const RowSet& s_rows, const CoefArgs& a)
{
	const size_t m = c->m;
	size_t stale = 0;
	for (size_t i = 0; i < used; i++) stale += !c->sy_ok[(st + i) % m];
	const size_t newest = (st + used - 1) % m;
	if (stale == 0) return -1;
	if (stale == 1 && !c->sy_ok[newest]) return (int) newest;
	for (size_t i = 0; i < used; i++) {
		const size_t b = (st + i) % m;
		if (c->sy_ok[b]) continue;
		Partials col = launch_sdot(c->sc, N(c), s_rows, row(c->Y, b, c), nullptr, nullptr);      // s_j'y_b for every pair j in use
		launch_store_column(c->sc, col, a, (int) b);
		c->sy_ok[b] = 1;
	}
	return -1;
}

bool threepass_ok(DevCtx* c, size_t st, size_t used)
{
	return options().threepass && used >= 1 && c->m <= (size_t) kPairsMax3 && pairs_tame(c, st, used);
}

// Returns the guard partials (sum r^2, nonfinite); the direction replaces g.  `qs` says how q0 is scaled:
// all-NULL = scalar (gamma of the newest pair, or h0 > 0), H0_in = a given diagonal, G = adaQN's step.
Partials enqueue_three_pass(DevCtx* c, real* g, size_t used, size_t st, double h0, real* gprev_out, const QdotScale& qs,
                            Call* pending = nullptr, const SliceFeed* drain = nullptr)
{
	const size_t m = c->m, k = used;
	RowSet ss{}, ys{};
	CoefArgs a{};
	a.k = (int) k;
	a.m = (int) m;
	a.h0 = h0;
	for (size_t i = 0; i < k; i++) {
		const size_t r = (st + i) % m;
		a.rows[i] = (int) r;
		ss.row[i] = row(c->S, r, c);
		ys.row[i] = row(c->Y, r, c);
	}
	ss.count = ys.count = (int) k;
	ensure_rho(c, st, k);                                     // s'y, y'y of every pair in use (rho_i, gamma)
	const int fresh = ensure_sy_columns(c, st, k, ss, a);
	const real* probe = fresh >= 0 ? row(c->Y, (size_t) fresh, c) : nullptr;
	SliceFeed feed{options().upload_slices, c->carry, c->carry_count, gradient_slice_arrives, pending};
	const bool sliced = pending && pending->g_pending && sdot_can_slice(c->sc, ss, g, gprev_out, probe);
	if (pending && !sliced) flush_g(*pending);
	Partials b = launch_sdot(c->sc, N(c), ss, g, gprev_out, probe, sliced ? &feed : nullptr);
	if (fresh >= 0) c->sy_ok[(size_t) fresh] = 1;             // stored by the coefficient kernel, ahead of the recursion
	Partials v = launch_qdot(c->sc, N(c), ys, g, qs, b, a, fresh);          // the scalar recursions run in the prologues of the passes
	if (pending) resolve_x(*pending);                         // pass 3 may send slices of x up and down: is the caller's x what the device holds?
	return launch_sadd(c->sc, c->next_buf(), N(c), ss, g, v, a, drain);
}

// Option "verify_cache" (debugging aid for DEVICE callers).  The library caches s'y, y'y and the products
// s_a'y_b of the pairs in the ring; it learns about changes to S and Y only through its own writes or
// stochqn_hip_invalidate.  A caller that restores or edits rows in place without saying so gets a direction
// built from stale inner products and no diagnostic.  With the option on, every step re-derives the cached
// numbers of ONE pair in use (round robin) from the arrays as they are and compares; a mismatch fails the
// call (-1000, message on stderr).  Costs a synchronisation and up to (k+4) n words per step.
void verify_cache(DevCtx* c, size_t st, size_t used)
{
	const size_t m = c->m;
	const size_t r = (st + (c->verify_turn++ % used)) % m;
	if (!c->rho_ok[r]) return;                                 // nothing cached for this row yet
	Partials p = launch_dots3(c->sc, c->next_buf(), N(c), row(c->S, r, c), row(c->Y, r, c));
	launch_fin(c->sc, p, 3, c->kap_dev);
	double* land = c->pin + 16 + 2 * m + c->fsize;             // [3 m]: fresh (s'y, s's, y'y) | cached s'y, y'y
	to_host(c, land, c->kap_dev, 3);
	to_host(c, land + 3, c->sc.sy + r, 1);
	to_host(c, land + 4, c->sc.yy + r, 1);
	std::vector<double> gsy;
	// three-pass form: the cached column s_q'y_r of every pair q in use that is older than r
	std::vector<double> col;
	size_t r_logical = used;
	for (size_t i = 0; i < used; i++) if ((st + i) % m == r) r_logical = i;
	const bool column = c->sy_ok[r] && m <= (size_t) kPairsMax3 && r_logical < used;
	if (column) {
		RowSet ss{};
		for (size_t i = 0; i < used; i++) ss.row[i] = row(c->S, (st + i) % m, c);
		ss.count = (int) used;
		Partials a = launch_sdot(c->sc, N(c), ss, row(c->Y, r, c), nullptr, nullptr);                  // s_i'y_r, logical order
		if (a.stride != 1) { launch_fin(c->sc, a, (int) used, c->sc.red[1]); a = Partials{c->sc.red[1], 1, 1}; }
		col.assign(used, 0.0);
		if (gsy.empty()) { gsy.assign(m * m, 0.0); SQN_HIP_OK(hipMemcpyAsync(gsy.data(), c->sc.gsy, m * m * sizeof(double), hipMemcpyDeviceToHost, c->sc.stream)); }
		SQN_HIP_OK(hipMemcpyAsync(col.data(), a.parts, used * sizeof(double), hipMemcpyDeviceToHost, c->sc.stream));
	}
	sync(c);
	const double scale = std::sqrt(std::fabs(land[1])) * std::sqrt(std::fabs(land[2]));     // |s_r||y_r|
	bool stale = !(std::fabs(land[0] - land[3]) <= 1e-8 * scale) || !(std::fabs(land[2] - land[4]) <= 1e-8 * std::fabs(land[2]));
	if (stale) std::fprintf(stderr, "stochqn: verify_cache: row %zu: s'y now %.17g cached %.17g, y'y now %.17g cached %.17g\n", r, land[0], land[3], land[2], land[4]);
	if (column && !stale) {
		double big = 0;
		for (size_t i = 0; i < r_logical; i++) big = std::fmax(big, std::fabs(col[i]));
		big = std::fmax(big, scale);
		for (size_t i = 0; i < r_logical && !stale; i++) {
			const size_t q = (st + i) % m;
			if (!(std::fabs(col[i] - gsy[q * m + r]) <= 1e-7 * big)) {
				stale = true;
				std::fprintf(stderr, "stochqn: verify_cache: s'y of rows (%zu, %zu): now %.17g cached %.17g\n", q, r, col[i], gsy[q * m + r]);
			}
		}
	}
	if (stale) {
		std::fprintf(stderr, "stochqn: verify_cache: the cached inner products of ring row %zu do not match s_mem / y_mem as they are now "
		                     "(rows changed outside the library without stochqn_hip_invalidate?) -- failing the call\n", r);
		c->fault = true;
	}
}

// apply_step for the three-pass form, one pass earlier.  Pass 3 runs in slices; as soon as a slice of the direction is final, x as
// the update WILL write it -- x - step r, ApplyOp's expression -- is computed into a vector of the library's own and sent to
// the caller's array, while the later slices of pass 3 are still running.  The guard (a sum over ALL of r) has not spoken at
// that point: the transfer is speculative.  The guarded update itself then runs as one launch on x proper, under the
// transfer; if it rejects the step (NaN / Inf / the norm test, reference src/stochqn.c:825-836: rare, the ring is flushed)
// the device x is untouched and step_was_bad sends it down again over what went ahead.  What the caller finds in x on
// return is what the one-launch path leaves there, bit for bit.
struct SpecDrain {
	Call* io;
	const real* r;
	const real* x;
	real* xs;
	double step;
	bool want_g;          // strict_grad: the direction goes back too (SQN / adaQN: the update leaves it as pass 3 wrote it)
};

void direction_slice_done(void* user, size_t lo, size_t hi, int slice)
{
	SpecDrain& d = *static_cast<SpecDrain*>(user);
	Call& io = *d.io;
	DevCtx* c = io.c;
	if (io.x_pending) x_slice_up(io, lo, hi, slice);      // the caller's x[lo, hi) comes up now, right ahead of what reads it (the slices cover x once)
	if (c->fault) return;                    // a reduction of this call failed: nothing of it reaches the caller's arrays (close_call)
	launch_spec_x(c->sc, hi - lo, d.r + lo, d.x + lo, d.step, d.xs + lo);
	SQN_HIP_OK(hipEventRecord(c->chunk_ev[(size_t) slice], c->sc.stream));
	SQN_HIP_OK(hipStreamWaitEvent(c->down_stream, c->chunk_ev[(size_t) slice], 0));
	SQN_HIP_OK(hipMemcpyAsync(io.x_caller + lo, d.xs + lo, (hi - lo) * sizeof(real), hipMemcpyDeviceToHost, c->down_stream));
	if (d.want_g) {
		SQN_HIP_OK(hipMemcpyAsync(io.g_caller + lo, d.r + lo, (hi - lo) * sizeof(real), hipMemcpyDeviceToHost, c->down_stream));
		io.g_down = true;
	}
	c->copy_busy = true;
	io.x_spec = true;
}

// may pass 3 of this call send x ahead?  (a host caller's x of a large problem, nobody waiting for the direction itself)
bool spec_x_ready(Call& io, const StepIn& in)
{
	DevCtx* c = io.c;
	const size_t n = N(c);
	const int chunks = options().apply_chunks;
	// oLBFGS with strict_grad: what goes back in grad is -step r, written by the update itself -- the plain path does that
	if (!options().spec_x || !io.host_caller || io.x != in.x || (io.g_host && options().strict_grad && in.s_slot) || chunks < 2 || n < (size_t) options().host_slice_min)
		return false;
	if (!ensure_upload_slices(c, 1, (size_t) 2 * kMaxGrid * kBlock) || !ensure_copy_stream(c, chunks + 1)) return false;   // + 1: the odd tail
	if (!c->spec && !device_alloc((void**) &c->spec, n * sizeof(real))) { c->spec = nullptr; return false; }
	(void) ensure_registered(c, io.x_caller, n * sizeof(real));
	if (io.g_host && options().strict_grad) (void) ensure_registered(c, io.g_caller, n * sizeof(real));
	return true;
}

// take_step (reference src/stochqn.c:802-840) + the caller's follow-up that only depends on the
// guard (x_sum += x, oLBFGS s-slot).  Enqueues everything and the read-back of the report block.
void enqueue_step(Call& io, const StepIn& in)
{
	DevCtx* c = io.c;
	const Scratch& sc = c->sc;
	const size_t n = N(c);
	if (options().verify_cache && in.used > 0) verify_cache(c, (in.st_ix == in.used) ? 0 : in.st_ix, in.used);
	ApplyArgs ap{in.x, in.x_sum, in.s_slot, in.step};
	FirstArgs fa{};
	fa.q = in.g;
	fa.gprev_out = in.gprev_out;
	fa.frow_out = in.frow_out;
	fa.G = in.G;
	fa.rmsprop_weight = in.w;
	fa.scal_reg = in.eps;

	if (in.used == 0) {
		flush_g(io);
		fa.H0_out = nullptr;                                          // rescale in place (:811)
		const bool need_first = in.check_nan || fa.gprev_out || fa.frow_out || fa.G;
		Partials guard{nullptr, 0, 0};
		if (need_first) guard = launch_first(sc, c->next_buf(), n, fa);
		apply_step(io, guard, in.g, in.g, ap, in.check_nan != 0);
		stat_add(ST_STEP_PLAIN);
	} else {
		fa.H0_out = in.H0;                                            // :818
		const size_t st = (in.st_ix == in.used) ? 0 : in.st_ix;      // :820
		const bool raw_cold = c->kind == KIND_RAW && !options().raw_reuse_cache;   // nothing cached survives the call: sweeps are cheapest
		if (!raw_cold && (!in.G || in.H0) && threepass_ok(c, st, in.used)) {
			QdotScale qs{};
			if (in.G) { qs.G = in.G; qs.H0_out = in.H0; qs.frow_out = in.frow_out; qs.rmsprop_weight = in.w; qs.scal_reg = in.eps; }
			SpecDrain sd{&io, in.g, in.x, c->spec, in.step, io.g_host && options().strict_grad && io.g == in.g};
			const bool ahead = spec_x_ready(io, in);
			sd.xs = c->spec;
			const SliceFeed drain{options().apply_chunks, c->carry, c->carry_count, direction_slice_done, &sd};
			Partials guard = enqueue_three_pass(c, in.g, in.used, st, in.h0, in.gprev_out, qs, &io, ahead ? &drain : nullptr);
			if (io.x_spec) {                                          // every slice of x is on its way already: the update in one launch, no copies
				io.x_pending = false;                                 // (and every slice of the caller's x came up with the slice that read it)
				launch_apply(sc, n, c->n_global, guard, in.g, in.g, ap, in.check_nan != 0);
				io.x_down = true;
				stat_add(ST_X_AHEAD);
			}
			else apply_step(io, guard, in.g, in.g, ap, in.check_nan != 0);
			stat_add(ST_STEP_THREE_PASS);
		} else {
			flush_g(io);
			if (!in.check_nan) flush_x(io);                               // the update rides in the last sweep
			stat_add(ST_STEP_SWEEPS);
			// was a cached form configured for this ring, and only the kappa rule said no?  (kappa is cached per row: no extra work)
			if (!raw_cold && options().threepass && (!in.G || in.H0) && c->m <= (size_t) kPairsMax3 && !pairs_tame(c, st, in.used))
				stat_add(ST_KAPPA_FALLBACK);
			Partials guard = enqueue_two_loop(c, in.g, in.used, st, fa, in.h0, in.G ? in.H0 : nullptr,
			                                  in.check_nan ? nullptr : &ap);
			if (in.check_nan) apply_step(io, guard, in.g, in.g, ap, true);
		}
	}
	// one read-back per step: report (bad flag, sum r^2, #nonfinite) | rho | alpha, contiguous in the pool
	if (in.check_nan || in.used > 0) to_host(c, c->pin, sc.report, 8 + 2 * c->m);
}

// after sync(): was the step rejected?  Also hands buffer_rho / buffer_alpha back to the caller.
bool step_was_bad(Call& io, bfgs_mem* b, size_t used_before, int check_nan)
{
	DevCtx* c = io.c;
	if (used_before > 0 && !c->async_call) {          // pinned block: report[8] | rho[m] | alpha[m]
		hand_back(b->buffer_rho, c->pin + 8, used_before);
		hand_back(b->buffer_alpha, c->pin + 8 + c->m, used_before);
	}
	const bool bad = check_nan && c->pin[0] != 0.0;       // unguarded: the step is always taken
	if (bad && io.x_spec) {                     // the caller's x received the step that was then rejected: send the device's (untouched) x again
		vec_to_host(c, io.x_caller, io.x, N(c));
		sync(c);
		x_handed_back(c, io.x_caller, N(c));
		stat_add(ST_X_RESENT);
	}
	return bad;
}

// ---- correction pairs ----------------------------------------------------------------------------
// reference "backup" = bak -> slot (src/stochqn.c:589-595).  The s half is dead (the slot's s is
// rewritten right after in every caller), the y half is observable.
void backup_pair(DevCtx* c, bfgs_mem* b)
{
	if (!(b->min_curvature > 0)) return;
	d2d(c, row(c->Y, b->mem_st_ix, c), c->ybak.dev, N(c));
	c->touch_row(b->mem_st_ix);
}

// update_s_vector (reference src/stochqn.c:861-870)
void make_s(DevCtx* c, bfgs_mem* b, bool needs_div)
{
	backup_pair(c, b);
	const bool scale = needs_div && b->upd_freq > 1;                  // :286-291
	launch_pair_s(c->sc, N(c), c->xsum.dev, 1 / (double) b->upd_freq, scale, c->xprev.dev, row(c->S, b->mem_st_ix, c));
	c->touch_row(b->mem_st_ix);
}

// check_min_curvature (reference src/stochqn.c:883-900) given the (s'y, s's, y'y) partials of the
// pair in the slot.  Synchronises (the decision is taken on the host).
// what the pair kernel needs to take the verdict on its own pair (its last workgroup; single device)
VerdictArgs verdict_of(DevCtx* c, const bfgs_mem* b)
{
	return VerdictArgs{(double) b->min_curvature, c->sc.sy + b->mem_st_ix, c->sc.yy + b->mem_st_ix, c->sc.report + 4};
}

void accept_or_reject(DevCtx* c, bfgs_mem* b, Partials p, info_enum* info)
{
	const size_t st = b->mem_st_ix;
	// one kernel reduces the three dots, decides and commits s'y / y'y of an accepted pair; the host
	// only needs the verdict for its bookkeeping (and for the rare rollback).  p.parts == NULL: the pair kernel's last
	// workgroup has done all of that already (kernels.hip: k_sweep_verdict)
	double* verdict = c->sc.report + 4;                                // report[4..7] <-> pin[4..7]
	if (p.parts) launch_verdict(c->sc, p, (double) b->min_curvature, c->sc.sy + st, c->sc.yy + st, verdict);
	if (c->async_call) {                 // min_curvature == 0: the pair is accepted whatever its dots are (:893); no read-back
		c->rho_ok[st] = 1;
		c->kappa[st] = 0;                // the kappa rule needs the dots on the host: off in stream-ordered calls (documented)
		ring_advance(b);
		return;
	}
	to_host(c, c->pin + 4, verdict, 4);
	sync(c);
	if (c->pin[7] != 0.0) {
		d2d(c, row(c->S, st, c), c->sbak.dev, N(c));                    // "rollback" = bak -> slot (:597-604)
		d2d(c, row(c->Y, st, c), c->ybak.dev, N(c));
		c->touch_row(st);
		c->sy_ok.assign(c->m, 0);                                    // row st may still be in use (full ring): its products with every newer pair are gone
		*info = curvature_too_small;
		return;
	}
	c->rho_ok[st] = 1;
	{
		const double k = std::sqrt(c->pin[5]) * std::sqrt(c->pin[6]) / std::fabs(c->pin[4]);   // |s||y| / |s'y|
		c->kappa[st] = (k >= 0) ? k : INFINITY;
	}
	ring_advance(b);
}

void average_sum(DevCtx* c, size_t L)                                   // average_from_sum (:286-291)
{
	if (L > 1) launch_scale(c->sc, N(c), c->xsum.dev, 1 / (double) L);
}

void archive_average(DevCtx* c)                                          // archive_x_avg (:606-610)
{
	d2d(c, c->xprev.dev, c->xsum.dev, N(c));
	zero(c, c->xsum.dev, N(c));
}

int invalid(task_enum* task, const char* who);
// failure after open_call succeeded (a view could not be bound)
int abandon(Call& io, const bfgs_mem* b, task_enum* task, const char* who)
{
	if (io.fresh) abandon_context(b->s_mem);
	return invalid(task, who);
}

int invalid(task_enum* task, const char* who)
{
	*task = invalid_input;
	std::fprintf(stderr, "%s got an invalid workspace as input.\n", who);
	return -1000;
}

int no_device(task_enum* task, const char* who)
{
	*task = invalid_input;
	std::fprintf(stderr, "%s: no usable HIP device (this build of libstochqn has no CPU path).\n", who);
	return -1000;
}

void* dev_alloc(size_t count, bool zero_fill)
{
	void* p = nullptr;
	if (!device_alloc(&p, count * sizeof(real))) return nullptr;
	if (zero_fill) SQN_HIP_OK(hipMemset(p, 0, count * sizeof(real)));
	return p;
}

// Section 0 is only ever seen on a brand-new optimiser object: whatever context still hangs on this
// address belongs to a dead one.  On return the caller-visible state is noted so that the next call
// can be recognised as its continuation (open_call).
template <class W> void before_call(W* w)
{
	if (w && w->bfgs_memory && w->section == 0) release(w->bfgs_memory->s_mem);
}
template <class W> int after_call(W* w, int rc, task_enum* task, bool req_is_x, const real_t* x)
{
	if (w && w->bfgs_memory && note_state(w->bfgs_memory->s_mem, w->niter, w->section, req_is_x, rc == -1000 ? nullptr : x)) {
		*task = invalid_input;                    // a HIP error surfaced during this call
		return -1000;
	}
	return rc;
}

// No C++ exception may cross the C ABI (R, Cython and C callers sit on
// the other side): anything thrown inside (std::bad_alloc from the registry, say) becomes the
// reference's own error convention, -1000 with task = invalid_input.
template <class F> int no_throw(task_enum* task, const char* who, F&& body)
{
	try {
		return body();
	} catch (const std::exception& e) {
		std::fprintf(stderr, "%s: %s\n", who, e.what());
	} catch (...) {
		std::fprintf(stderr, "%s: unexpected failure\n", who);
	}
	if (task) *task = invalid_input;
	return -1000;
}


}  // namespace

extern "C" {

// =================================================================================================
// oLBFGS (reference src/stochqn.c:978-1036)
// =================================================================================================
static int run_oLBFGS_impl(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task,
                           workspace_oLBFGS* w, info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 2) return invalid(task, "oLBFGS");
	if (!device_ready()) return no_device(task, "oLBFGS");
	bfgs_mem* b = w->bfgs_memory;
	*req = x;

	if (w->section == 0) {                     // first call: ask for a gradient
		*task = calc_grad;
		w->section = 1;
		return 0;
	}

	Call io;
	if (!open_call(io, KIND_OLBFGS, w->n, b, 0, w->niter > 0 || b->mem_used > 0, x, grad, w->niter, w->section, w->check_nan)) return invalid(task, "oLBFGS");
	DevCtx* c = io.c;
	if (!bind(c, c->gprev, w->grad_prev, N(c), true)) return abandon(io, b, task, "oLBFGS");

	if (w->section == 1) {                     // step; s-slot; ask for the gradient on the same batch
		stage_xg(io, true, true, true);
		const size_t used = b->mem_used;
		StepIn in;
		in.step = step_size; in.x = io.x; in.g = io.g;
		in.used = used; in.st_ix = b->mem_st_ix;
		in.h0 = w->hess_init;
		in.gprev_out = c->gprev.dev;                                  // :996
		in.s_slot = row(c->S, b->mem_st_ix, c);                       // :1006-1007, predicated on the guard
		in.check_nan = w->check_nan;
		enqueue_step(io, in);
		w->niter++;
		close_call(io, true, true);
		if (step_was_bad(io, b, used, w->check_nan)) {
			ring_reset(b);                                            // :831, :1015
			*iter_info = search_direction_was_nan;
			*task = calc_grad;
			w->section = 1;
			return 0;
		}
		c->touch_row(b->mem_st_ix);
		if (b->min_curvature > 0) { backup_pair(c, b); sync(c); }     // :1005 (y half)
		*task = calc_grad_same_batch;
		w->section = 2;
		return 1;
	}

	// section 2: y-slot from the gradient difference, accept / reject, ask for a new gradient
	stage_xg(io, false, true);
	const size_t st = b->mem_st_ix;
	const VerdictArgs va = verdict_of(c, b);
	Partials p = launch_pair_y_diff(c->sc, c->next_buf(), N(c), io.g, c->gprev.dev, row(c->S, st, c), b->y_reg, row(c->Y, st, c), &va);
	accept_or_reject(c, b, p, iter_info);
	sync(c);
	*task = calc_grad;
	w->section = 1;
	return 0;
}

// =================================================================================================
// SQN (reference src/stochqn.c:1038-1153)
// =================================================================================================
static int run_SQN_impl(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec,
                        task_enum* task, workspace_SQN* w, info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 4) return invalid(task, "SQN");
	if (!device_ready()) return no_device(task, "SQN");
	bfgs_mem* b = w->bfgs_memory;
	int ret = 0;

	if (w->section != 0) {
		Call io;
		if (!open_call(io, KIND_SQN, w->n, b, 0, w->niter > 0 || b->mem_used > 0, x, grad, w->niter, w->section, w->check_nan)) return invalid(task, "SQN");
		DevCtx* c = io.c;
		const size_t n = N(c);
		if (!bind(c, c->gprev, w->grad_prev, w->use_grad_diff ? n : 0, true) ||
		    !bind(c, c->xsum, w->x_sum, n, true) || !bind(c, c->xprev, w->x_avg_prev, n, true))
			return abandon(io, b, task, "SQN");

		switch (w->section) {
		case 1: {
			stage_xg(io, true, true, true);
			const size_t used = b->mem_used;
			StepIn in;
			in.step = step_size; in.x = io.x; in.g = io.g;
			in.used = used; in.st_ix = b->mem_st_ix;
			in.x_sum = c->xsum.dev;                                   // :1067, fused into the update
			in.check_nan = w->check_nan;
			enqueue_step(io, in);
			w->niter++;
			close_call(io, true, true);
			if (step_was_bad(io, b, used, w->check_nan)) { ring_reset(b); *iter_info = search_direction_was_nan; ret = 0; }
			else ret = 1;

			if (w->niter % b->upd_freq != 0) break;
			if (w->niter == b->upd_freq) {                            // :1078-1094: first average only
				average_sum(c, b->upd_freq);
				archive_average(c);
				if (!w->use_grad_diff) { sync(c); break; }
				*task = calc_grad_big_batch;
				*req = publish(io, c->xprev, 0, 0);
				sync(c);
				w->section = 2;
				return ret;
			}
			make_s(c, b, true);                                       // :1097
			*req = publish(io, c->xsum, 0, 0);
			if (w->use_grad_diff) { *task = calc_grad_big_batch; w->section = 3; }
			else {
				*task = calc_hess_vec;
				w->section = 4;
				*req_vec = publish(io, c->S, b->mem_st_ix * n, 1);
			}
			sync(c);
			return ret;
		}
		case 2:                                                       // :1118-1122
			stage_xg(io, false, true);
			d2d(c, c->gprev.dev, io.g, n);
			sync(c);
			break;
		case 3: {                                                     // :1125-1134
			stage_xg(io, false, true);
			const size_t st = b->mem_st_ix;
			const VerdictArgs va = verdict_of(c, b);
			Partials p = launch_pair_y_diff(c->sc, c->next_buf(), n, io.g, c->gprev.dev, row(c->S, st, c), b->y_reg, row(c->Y, st, c), &va);
			accept_or_reject(c, b, p, iter_info);
			if (*iter_info == no_problems_encountered) {
				d2d(c, c->gprev.dev, io.g, n);
				d2d(c, c->xprev.dev, c->xsum.dev, n);
			}
			zero(c, c->xsum.dev, n);
			sync(c);
			break;
		}
		case 4: {                                                     // :1137-1142
			const bool hv_host = !is_device_pointer(hess_vec);
			real* hv = stage_in(c, 2, hess_vec, n, hv_host);
			if (!hv) return invalid(task, "SQN");
			const size_t st = b->mem_st_ix;
			const VerdictArgs va = verdict_of(c, b);
			Partials p = launch_pair_y_hv(c->sc, c->next_buf(), n, hv, row(c->S, st, c), row(c->Y, st, c), c->xsum.dev, c->xprev.dev, &va);
			accept_or_reject(c, b, p, iter_info);
			sync(c);
			break;
		}
		}
	}
	w->section = 1;                                                   // resume_main_loop (:1148-1152)
	*task = calc_grad;
	*req = x;
	return ret;
}

// =================================================================================================
// adaQN (reference src/stochqn.c:1155-1315)
// =================================================================================================
static int run_adaQN_impl(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task,
                          workspace_adaQN* w, info_enum* iter_info)
{
	*iter_info = no_problems_encountered;
	if (!w || !w->bfgs_memory || w->section < 0 || w->section > 5) return invalid(task, "adaQN");
	if (!device_ready()) return no_device(task, "adaQN");
	bfgs_mem* b = w->bfgs_memory;
	fisher_mem* fm = w->use_grad_diff ? nullptr : w->fisher_memory;
	int ret = 0;
	bool build_y = false;

	if (w->section == 3) {                                             // :1258-1262, scalar only
		w->f_prev = f;
	} else if (w->section != 0) {
		Call io;
		const size_t fsize = fm ? fm->mem_size : 0;
		const bool resumed = w->niter > 0 || b->mem_used > 0;
		if (!open_call(io, KIND_ADAQN, w->n, b, fsize, resumed, x, grad, w->niter, w->section, w->check_nan)) return invalid(task, "adaQN");
		DevCtx* c = io.c;
		const size_t n = N(c);
		if (!bind(c, c->gprev, w->grad_prev, w->use_grad_diff ? n : 0, true) ||
		    !bind(c, c->xsum, w->x_sum, n, true) || !bind(c, c->xprev, w->x_avg_prev, n, true) ||
		    !bind(c, c->H0, w->H0, n, false) || !bind(c, c->G, w->grad_sum_sq, n, true) ||
		    !bind(c, c->F, fm ? fm->F : nullptr, fsize * n, resumed && fm && fm->mem_used > 0))
			return abandon(io, b, task, "adaQN");

		switch (w->section) {
		case 1: {
			stage_xg(io, true, true, true);
			const size_t used = b->mem_used;
			StepIn in;
			in.step = step_size; in.x = io.x; in.g = io.g;
			in.used = used; in.st_ix = b->mem_st_ix;
			in.H0 = c->H0.dev; in.G = c->G.dev;
			in.w = w->rmsprop_weight; in.eps = w->scal_reg;
			if (fm) { in.frow_out = row(c->F, fm->mem_st_ix, c); fisher_advance(fm); }   // :1174
			in.x_sum = c->xsum.dev;                                   // :1191
			in.check_nan = w->check_nan;
			enqueue_step(io, in);
			close_call(io, true, true);
			if (step_was_bad(io, b, used, w->check_nan)) { ring_reset(b); *iter_info = search_direction_was_nan; ret = 0; }
			else ret = 1;
			w->niter++;

			if (w->niter % b->upd_freq != 0) break;
			if (w->niter == b->upd_freq) {                            // :1206-1224
				average_sum(c, b->upd_freq);
				archive_average(c);
				if (w->use_grad_diff || w->max_incr > 0) {
					*task = w->use_grad_diff ? calc_grad_big_batch : calc_fun_val_batch;
					*req = publish(io, c->xprev, 0, 0);
					sync(c);
					w->section = w->use_grad_diff ? 2 : 3;
					return ret;
				}
				sync(c);
				break;
			}
			if (w->max_incr > 0) {                                    // :1227-1234
				average_sum(c, b->upd_freq);
				*task = calc_fun_val_batch;
				*req = publish(io, c->xsum, 0, 0);
				sync(c);
				w->section = 5;
				return ret;
			}
			make_s(c, b, true);                                       // :1237
			build_y = true;
			break;
		}
		case 2:                                                       // :1242-1255
			stage_xg(io, false, true);
			d2d(c, c->gprev.dev, io.g, n);
			if (w->max_incr) {
				*task = calc_fun_val_batch;
				*req = publish(io, c->xprev, 0, 0);
				sync(c);
				w->section = 3;
				return 0;
			}
			sync(c);
			break;
		case 4: {                                                     // :1265-1270
			stage_xg(io, false, true);
			const size_t st = b->mem_st_ix;
			const VerdictArgs va = verdict_of(c, b);
			Partials p = launch_pair_y_diff(c->sc, c->next_buf(), n, io.g, c->gprev.dev, row(c->S, st, c), b->y_reg, row(c->Y, st, c), &va);
			accept_or_reject(c, b, p, iter_info);
			if (*iter_info == no_problems_encountered) d2d(c, c->gprev.dev, io.g, n);
			zero(c, c->xsum.dev, n);
			sync(c);
			break;
		}
		case 5:                                                       // :1273-1291
			if (f > w->max_incr * w->f_prev || std::isinf(f) || std::isnan(f)) {
				ring_reset(b);
				fisher_reset(fm);
				stage_xg(io, true, false);
				d2d(c, io.x, c->xprev.dev, n);                        // x <- x_avg_prev
				*iter_info = func_increased;
				ret = 1;
				close_call(io, true, false);
				break;
			}
			w->f_prev = f;
			make_s(c, b, false);
			build_y = true;
			break;
		}

		if (build_y) {                                                // update_y (:1297-1308)
			if (!w->use_grad_diff && !fm) return invalid(task, "adaQN");
			if (w->use_grad_diff) {
				*req = publish(io, c->xsum, 0, 0);
				*task = calc_grad_big_batch;
				sync(c);
				w->section = 4;
				return ret;
			}
			const size_t st = b->mem_st_ix;
			Partials p = launch_fisher(c->sc, c->next_buf(), n, c->F.dev, fm->mem_used, row(c->S, st, c), c->fisher_t, row(c->Y, st, c));
			if (fm->buffer_y) to_host(c, c->pin + 8 + 2 * c->m, c->fisher_t, fm->mem_used);
			accept_or_reject(c, b, p, iter_info);
			if (!c->async_call) hand_back(fm->buffer_y, c->pin + 8 + 2 * c->m, fm->mem_used);
			if (*iter_info == no_problems_encountered) d2d(c, c->xprev.dev, c->xsum.dev, n);
			zero(c, c->xsum.dev, n);
			sync(c);
		}
	}
	w->section = 1;                                                   // resume_main_loop (:1310-1314)
	*task = calc_grad;
	*req = x;
	return ret;
}

}  // extern "C"

// One shard's call (group.cpp) or the whole call (below): the state machine between the two hooks that
// tie a device context to the caller-visible state.
namespace sqn {
void set_thread_dev_requests(bool on) { t_dev_requests = on; }
int local_run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task, workspace_oLBFGS* w,
                     info_enum* iter_info)
{
	return no_throw(task, "run_oLBFGS", [&] {
		ApiRange range("run_oLBFGS", w ? w->section : -1);
		before_call(w);
		const int rc = run_oLBFGS_impl(step_size, x, grad, req, task, w, iter_info);
		return after_call(w, rc, task, req && *req == x, x);
	});
}

int local_run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec, task_enum* task,
                  workspace_SQN* w, info_enum* iter_info)
{
	return no_throw(task, "run_SQN", [&] {
		ApiRange range("run_SQN", w ? w->section : -1);
		before_call(w);
		const int rc = run_SQN_impl(step_size, x, grad, hess_vec, req, req_vec, task, w, iter_info);
		return after_call(w, rc, task, req && *req == x, x);
	});
}

int local_run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task, workspace_adaQN* w,
                    info_enum* iter_info)
{
	return no_throw(task, "run_adaQN", [&] {
		ApiRange range("run_adaQN", w ? w->section : -1);
		before_call(w);
		const int rc = run_adaQN_impl(step_size, x, f, grad, req, task, w, iter_info);
		return after_call(w, rc, task, req && *req == x, x);
	});
}
}  // namespace sqn

extern "C" {

// The exported entry points (exception barrier: no_throw, above).  With option "devices" >= 2 a workspace
// whose arrays are host memory or library-owned shards runs on all devices (group.cpp); everything
// else -- device pointers in the structs, one device, small n -- takes the single-device path.
int run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t** req, task_enum* task, workspace_oLBFGS* w,
               info_enum* iter_info)
{
	if (w && w->bfgs_memory && device_ready() && group_applies(w->bfgs_memory, w->n))
		return no_throw(task, "run_oLBFGS", [&] { return group_run_oLBFGS(step_size, x, grad, req, task, w, iter_info); });
	return local_run_oLBFGS(step_size, x, grad, req, task, w, iter_info);
}

int run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t** req, real_t** req_vec, task_enum* task,
            workspace_SQN* w, info_enum* iter_info)
{
	if (w && w->bfgs_memory && device_ready() && group_applies(w->bfgs_memory, w->n))
		return no_throw(task, "run_SQN", [&] { return group_run_SQN(step_size, x, grad, hess_vec, req, req_vec, task, w, iter_info); });
	return local_run_SQN(step_size, x, grad, hess_vec, req, req_vec, task, w, iter_info);
}

int run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t** req, task_enum* task, workspace_adaQN* w,
              info_enum* iter_info)
{
	if (w && w->bfgs_memory && device_ready() && group_applies(w->bfgs_memory, w->n))
		return no_throw(task, "run_adaQN", [&] { return group_run_adaQN(step_size, x, f, grad, req, task, w, iter_info); });
	return local_run_adaQN(step_size, x, f, grad, req, task, w, iter_info);
}

// =================================================================================================
// library-owned workspaces (reference src/stochqn.c:300-547): structs and the small host-readable
// buffers in host memory, every n-sized array in HBM.
// =================================================================================================
bfgs_mem* initialize_bfgs_mem(const size_t mem_size, const int n, const real_t min_curvature, const real_t y_reg,
                              const size_t upd_freq)
{
	if (!device_ready() || n <= 0 || mem_size == 0) return nullptr;
	bfgs_mem* b = (bfgs_mem*) std::calloc(1, sizeof(bfgs_mem));
	if (!b) return nullptr;
	b->s_mem = (real_t*) dev_alloc((size_t) n * mem_size, false);
	b->y_mem = (real_t*) dev_alloc((size_t) n * mem_size, false);
	b->buffer_rho = (real_t*) std::calloc(mem_size, sizeof(real_t));
	b->buffer_alpha = (real_t*) std::calloc(mem_size, sizeof(real_t));
	if (min_curvature > 0) {                     // zero-filled: deterministic superset of the reference's malloc
		b->s_bak = (real_t*) dev_alloc((size_t) n, true);
		b->y_bak = (real_t*) dev_alloc((size_t) n, true);
	}
	b->mem_size = mem_size;
	b->upd_freq = upd_freq;
	b->y_reg = y_reg;
	b->min_curvature = min_curvature;
	if (!b->s_mem || !b->y_mem || !b->buffer_rho || !b->buffer_alpha || (min_curvature > 0 && (!b->s_bak || !b->y_bak))) {
		std::fprintf(stderr, "Error: Could not allocate memory for BFGS storage.\n");
		dealloc_bfgs_mem(b);
		return nullptr;
	}
	return b;
}

void dealloc_bfgs_mem(bfgs_mem* b)
{
	if (!b) return;
	release(b->s_mem);
	if (b->s_mem) (void) hipFree(b->s_mem);
	if (b->y_mem) (void) hipFree(b->y_mem);
	if (b->s_bak) (void) hipFree(b->s_bak);
	if (b->y_bak) (void) hipFree(b->y_bak);
	std::free(b->buffer_rho);
	std::free(b->buffer_alpha);
	std::free(b);
}

fisher_mem* initialize_fisher_mem(const size_t mem_size, const int n)
{
	if (!device_ready() || n <= 0 || mem_size == 0) return nullptr;
	fisher_mem* f = (fisher_mem*) std::calloc(1, sizeof(fisher_mem));
	if (!f) return nullptr;
	f->F = (real_t*) dev_alloc((size_t) n * mem_size, false);
	f->buffer_y = (real_t*) std::calloc(mem_size, sizeof(real_t));
	f->mem_size = mem_size;
	if (!f->F || !f->buffer_y) {
		std::fprintf(stderr, "Error: Could not allocate memory for Fisher storage.\n");
		dealloc_fisher_mem(f);
		return nullptr;
	}
	return f;
}

void dealloc_fisher_mem(fisher_mem* f)
{
	if (!f) return;
	if (f->F) (void) hipFree(f->F);
	std::free(f->buffer_y);
	std::free(f);
}

static bool gpu_or_complain(const char* who)
{
	if (device_ready()) return true;
	std::fprintf(stderr, "Error: %s: no usable HIP device (this build of libstochqn has no CPU path).\n", who);
	return false;
}

workspace_oLBFGS* initialize_oLBFGS(const int n, const size_t mem_size, const real_t hess_init, const real_t y_reg,
                                    const real_t min_curvature, const int check_nan, const int nthreads)
{
	if (!gpu_or_complain("initialize_oLBFGS")) return nullptr;
	if (mem_size > 0 && group_mode_for(n)) {                  // option "devices": sharded over the devices of this process
		workspace_oLBFGS* gw = group_initialize_oLBFGS(n, mem_size, hess_init, y_reg, min_curvature, check_nan, nthreads);
		if (!gw) std::fprintf(stderr, "Error: Could not allocate memory for oLBFGS.\n");
		return gw;
	}
	workspace_oLBFGS* w = (workspace_oLBFGS*) std::calloc(1, sizeof(*w));
	if (!w) return nullptr;
	w->bfgs_memory = initialize_bfgs_mem(mem_size, n, min_curvature, y_reg, 1);
	w->grad_prev = (real_t*) dev_alloc((size_t) (n > 0 ? n : 1), false);
	w->hess_init = hess_init;
	w->check_nan = check_nan;
	w->nthreads = nthreads;
	w->n = n;
	if (!w->bfgs_memory || !w->grad_prev) {
		std::fprintf(stderr, "Error: Could not allocate memory for oLBFGS.\n");
		dealloc_oLBFGS(w);
		return nullptr;
	}
	return w;
}

// a workspace made in group mode: the struct's array fields are tokens, the arrays are the shards
static bool dealloc_group(bfgs_mem* b, fisher_mem* f)
{
	if (!b || !group_owns(b->s_mem)) return false;
	group_dealloc(b->s_mem);
	std::free(b->buffer_rho);
	std::free(b->buffer_alpha);
	std::free(b);
	if (f) { std::free(f->buffer_y); std::free(f); }
	return true;
}

void dealloc_oLBFGS(workspace_oLBFGS* w)
{
	if (!w) return;
	if (dealloc_group(w->bfgs_memory, nullptr)) { std::free(w); return; }
	dealloc_bfgs_mem(w->bfgs_memory);
	if (w->grad_prev) (void) hipFree(w->grad_prev);
	std::free(w);
}

workspace_SQN* initialize_SQN(const int n, const size_t mem_size, const size_t bfgs_upd_freq, const real_t min_curvature,
                              const int use_grad_diff, const real_t y_reg, const int check_nan, const int nthreads)
{
	if (!gpu_or_complain("initialize_SQN")) return nullptr;
	if (mem_size > 0 && group_mode_for(n)) {
		workspace_SQN* gw = group_initialize_SQN(n, mem_size, bfgs_upd_freq, min_curvature, use_grad_diff, y_reg, check_nan, nthreads);
		if (!gw) std::fprintf(stderr, "Error: Could not allocate memory for SQN.\n");
		return gw;
	}
	workspace_SQN* w = (workspace_SQN*) std::calloc(1, sizeof(*w));
	if (!w) return nullptr;
	const size_t nn = (size_t) (n > 0 ? n : 1);
	w->bfgs_memory = initialize_bfgs_mem(mem_size, n, min_curvature, y_reg, bfgs_upd_freq);
	w->grad_prev = use_grad_diff ? (real_t*) dev_alloc(nn, false) : nullptr;
	w->x_sum = (real_t*) dev_alloc(nn, true);
	w->x_avg_prev = (real_t*) dev_alloc(nn, false);
	w->use_grad_diff = use_grad_diff;
	w->check_nan = check_nan;
	w->nthreads = nthreads;
	w->n = n;
	if (!w->bfgs_memory || !w->x_sum || !w->x_avg_prev || (use_grad_diff && !w->grad_prev)) {
		std::fprintf(stderr, "Error: Could not allocate memory for SQN.\n");
		dealloc_SQN(w);
		return nullptr;
	}
	return w;
}

void dealloc_SQN(workspace_SQN* w)
{
	if (!w) return;
	if (dealloc_group(w->bfgs_memory, nullptr)) { std::free(w); return; }
	dealloc_bfgs_mem(w->bfgs_memory);
	if (w->grad_prev) (void) hipFree(w->grad_prev);
	if (w->x_sum) (void) hipFree(w->x_sum);
	if (w->x_avg_prev) (void) hipFree(w->x_avg_prev);
	std::free(w);
}

workspace_adaQN* initialize_adaQN(const int n, const size_t mem_size, const size_t fisher_size, const size_t bfgs_upd_freq,
                                  const real_t max_incr, const real_t min_curvature, const real_t scal_reg,
                                  const real_t rmsprop_weight, const int use_grad_diff, const real_t y_reg,
                                  const int check_nan, const int nthreads)
{
	if (!gpu_or_complain("initialize_adaQN")) return nullptr;
	if (mem_size > 0 && group_mode_for(n)) {
		workspace_adaQN* gw = group_initialize_adaQN(n, mem_size, fisher_size, bfgs_upd_freq, max_incr, min_curvature, scal_reg,
		                                             rmsprop_weight, use_grad_diff, y_reg, check_nan, nthreads);
		if (!gw) std::fprintf(stderr, "Error: Could not allocate memory for adaQN.\n");
		return gw;
	}
	workspace_adaQN* w = (workspace_adaQN*) std::calloc(1, sizeof(*w));
	if (!w) return nullptr;
	const size_t nn = (size_t) (n > 0 ? n : 1);
	w->bfgs_memory = initialize_bfgs_mem(mem_size, n, min_curvature, y_reg, bfgs_upd_freq);
	if (use_grad_diff) w->grad_prev = (real_t*) dev_alloc(nn, false);
	else               w->fisher_memory = initialize_fisher_mem(fisher_size, n);
	w->H0 = (real_t*) dev_alloc(nn, false);
	w->x_sum = (real_t*) dev_alloc(nn, true);
	w->x_avg_prev = (real_t*) dev_alloc(nn, false);
	w->grad_sum_sq = (real_t*) dev_alloc(nn, true);
	w->max_incr = max_incr;
	w->scal_reg = scal_reg;
	w->rmsprop_weight = rmsprop_weight;
	w->use_grad_diff = use_grad_diff;
	w->check_nan = check_nan;
	w->nthreads = nthreads;
	w->n = n;
	if (!w->bfgs_memory || !w->H0 || !w->x_sum || !w->x_avg_prev || !w->grad_sum_sq ||
	    (use_grad_diff ? !w->grad_prev : !w->fisher_memory)) {
		std::fprintf(stderr, "Error: Could not allocate memory for adaQN.\n");
		dealloc_adaQN(w);
		return nullptr;
	}
	return w;
}

void dealloc_adaQN(workspace_adaQN* w)
{
	if (!w) return;
	if (dealloc_group(w->bfgs_memory, w->fisher_memory)) { std::free(w); return; }
	dealloc_bfgs_mem(w->bfgs_memory);
	dealloc_fisher_mem(w->fisher_memory);
	if (w->H0) (void) hipFree(w->H0);
	if (w->grad_prev) (void) hipFree(w->grad_prev);
	if (w->x_sum) (void) hipFree(w->x_sum);
	if (w->x_avg_prev) (void) hipFree(w->x_avg_prev);
	if (w->grad_sum_sq) (void) hipFree(w->grad_sum_sq);
	std::free(w);
}

// =================================================================================================
// isolated kernels of stochqn_hip.h
// =================================================================================================
// Context of an isolated entry point: keyed apart from the optimiser context of the same arrays (an odd
// address is never an array of reals), so a raw call never tears down a live optimiser's state.
static DevCtx* raw_context(real_t s_mem[], real_t y_mem[], int n, size_t mem_size, bool* fresh)
{
	DevCtx* c = acquire(raw_key(s_mem), KIND_RAW, n, mem_size, 0, fresh);
	if (!c) return nullptr;
	const size_t nn = (size_t) n;
	if (!bind(c, c->S, s_mem, mem_size * nn, true) || !bind(c, c->Y, y_mem, mem_size * nn, true)) { end_use(c); return nullptr; }
	// host arrays may have changed since the last call: refresh the mirrors
	if (c->S.mirror && !*fresh) SQN_HIP_OK(hipMemcpyAsync(c->S.dev, s_mem, mem_size * nn * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	if (c->Y.mirror && !*fresh) SQN_HIP_OK(hipMemcpyAsync(c->Y.dev, y_mem, mem_size * nn * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	// The pure functions these entries stand for (approx_inv_hess_grad, take_step) recompute every inner
	// product on every call.  Here s'y, y'y and the products s_a'y_b are cached per row, and nothing tells the
	// library that a caller rewrote S / Y in place (or that an allocator handed the same address to new
	// arrays): so the caches are dropped on every call unless the caller vouches for the arrays with
	// option "raw_reuse_cache" = 1 (the two-loop micro-benchmark does).
	if (!options().raw_reuse_cache || c->S.mirror || c->Y.mirror) c->forget_rows();
	if (!c->attached) { comm_attach(c); c->attached = true; }
	return c;
}

static int two_loop_impl(real_t grad[], int n, real_t H0[], real_t h0, real_t y_mem[], real_t s_mem[],
                         size_t mem_size, size_t mem_used, size_t mem_st_ix, real_t buffer_rho[], real_t buffer_alpha[])
{
	if (!device_ready() || !grad || !y_mem || !s_mem || n <= 0 || mem_size == 0 || mem_used == 0 || mem_used > mem_size)
		return -1000;
	(void) hipGetLastError();
	(void) take_hip_failure();
	bool fresh = false;
	DevCtx* c = raw_context(s_mem, y_mem, n, mem_size, &fresh);
	if (!c) return -1000;
	InUse hold(c);
	const size_t nn = (size_t) n;
	if (!bind(c, c->H0, H0, H0 ? nn : 0, true)) return -1000;
	if (c->H0.mirror) SQN_HIP_OK(hipMemcpyAsync(c->H0.dev, H0, nn * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	const bool g_host = !is_device_pointer(grad);
	real* g = stage_in(c, 1, grad, nn, g_host);
	if (!g) return -1000;
	const bool cold = !options().raw_reuse_cache;             // every inner product would be rebuilt for this one call: sweeps are cheapest
	if (!cold && threepass_ok(c, mem_st_ix % mem_size, mem_used)) {
		QdotScale qs{};
		qs.H0_in = H0 ? c->H0.dev : nullptr;
		(void) enqueue_three_pass(c, g, mem_used, mem_st_ix % mem_size, h0, nullptr, qs);
	} else {
		FirstArgs fa{};
		(void) enqueue_two_loop(c, g, mem_used, mem_st_ix % mem_size, fa, h0, H0 ? c->H0.dev : nullptr, nullptr);
	}
	to_host(c, c->pin, c->sc.report, 8 + 2 * c->m);
	if (g_host) vec_to_host(c, grad, g, nn);
	sync(c);
	if (const bool hip_failed = take_hip_failure(); c->fault || hip_failed) { c->fault = false; return -1000; }
	hand_back(buffer_rho, c->pin + 8, mem_used);
	hand_back(buffer_alpha, c->pin + 8 + c->m, mem_used);
	return 0;
}

int stochqn_hip_two_loop(real_t grad[], int n, real_t H0[], real_t h0, real_t y_mem[], real_t s_mem[],
                         size_t mem_size, size_t mem_used, size_t mem_st_ix, real_t buffer_rho[], real_t buffer_alpha[])
{
	return no_throw(nullptr, "stochqn_hip_two_loop", [&] {
		return two_loop_impl(grad, n, H0, h0, y_mem, s_mem, mem_size, mem_used, mem_st_ix, buffer_rho, buffer_alpha);
	});
}

// take_step of reference src/stochqn.c:802-840 on its own: [diagonal rescale ->] two-loop -> guard -> x update.
// With grad_sum_sq != NULL this is adaQN's step (H0 receives g/sqrt(G+eps), G is updated): pass 2 of the
// three-pass form in its adaQN mode, or the sweeps with option "threepass" = 0.
static int take_step_impl(real_t step_size, int n, real_t x[], real_t grad[], bfgs_mem* b, real_t rmsprop_weight, real_t H0[],
                          real_t h0, real_t grad_sum_sq[], real_t scal_reg, int check_nan, info_enum* iter_info)
{
	if (iter_info) *iter_info = no_problems_encountered;
	if (!device_ready() || !x || !grad || !b || !b->s_mem || !b->y_mem || n <= 0 || b->mem_size == 0 || b->mem_used > b->mem_size ||
	    b->mem_st_ix >= b->mem_size || (grad_sum_sq && b->mem_used > 0 && !H0))
		return -1000;
	(void) hipGetLastError();
	(void) take_hip_failure();
	bool fresh = false;
	DevCtx* c = raw_context(b->s_mem, b->y_mem, n, b->mem_size, &fresh);
	if (!c) return -1000;
	InUse hold(c);
	const size_t nn = (size_t) n;
	real_t* H0_used = grad_sum_sq ? H0 : nullptr;          // a caller-supplied diagonal is approx_inv_hess_grad's business (two_loop)
	if (!bind(c, c->H0, H0_used, H0_used ? nn : 0, false) || !bind(c, c->G, grad_sum_sq, grad_sum_sq ? nn : 0, true)) return -1000;
	if (c->G.mirror && !fresh) SQN_HIP_OK(hipMemcpyAsync(c->G.dev, grad_sum_sq, nn * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	Call io;
	io.c = c;
	io.x_caller = x; io.g_caller = grad;
	io.host_caller = !is_device_pointer(x);
	io.g_host = !is_device_pointer(grad);
	if ((io.host_caller && !ensure_stage(c, 0)) || (io.g_host && !ensure_stage(c, 1))) return -1000;
	stage_xg(io, true, true);
	StepIn in;
	in.step = step_size; in.x = io.x; in.g = io.g;
	in.used = b->mem_used; in.st_ix = b->mem_st_ix;
	in.h0 = h0;
	in.H0 = c->H0.dev; in.G = c->G.dev;
	in.w = rmsprop_weight; in.eps = scal_reg;
	in.check_nan = check_nan;
	enqueue_step(io, in);
	if (c->G.mirror) export_view(c, c->G);
	if (c->H0.mirror && b->mem_used > 0) export_view(c, c->H0);
	close_call(io, true, true);
	if (const bool hip_failed = take_hip_failure(); c->fault || hip_failed) { c->fault = false; return -1000; }
	if (step_was_bad(io, b, b->mem_used, check_nan)) {
		ring_reset(b);                                         // :831
		if (iter_info) *iter_info = search_direction_was_nan;
	}
	return 0;
}

int stochqn_hip_take_step(real_t step_size, int n, real_t x[], real_t grad[], bfgs_mem* bfgs_memory, real_t rmsprop_weight,
                          real_t H0[], real_t h0, real_t grad_sum_sq[], real_t scal_reg, int check_nan, info_enum* iter_info)
{
	return no_throw(nullptr, "stochqn_hip_take_step", [&] {
		return take_step_impl(step_size, n, x, grad, bfgs_memory, rmsprop_weight, H0, h0, grad_sum_sq, scal_reg, check_nan, iter_info);
	});
}

static int fisher_product_impl(real_t F[], size_t fu, int n, real_t s[], real_t buffer_y[], real_t y[])
{
	if (!device_ready() || !F || !s || !y || n <= 0 || fu == 0) return -1000;
	(void) hipGetLastError();
	(void) take_hip_failure();
	bool fresh = false;
	DevCtx* c = acquire(raw_key(F), KIND_RAW, n, 1, fu, &fresh);
	if (!c) return -1000;
	InUse hold(c);
	const size_t nn = (size_t) n;
	if (!c->attached) { comm_attach(c); c->attached = true; }
	// F: used in place when on the device, else mirrored (re-uploaded every call: contents may have changed)
	if (!bind(c, c->F, F, fu * nn, false)) return -1000;
	if (c->F.mirror) SQN_HIP_OK(hipMemcpyAsync(c->F.dev, F, fu * nn * sizeof(real), hipMemcpyHostToDevice, c->sc.stream));
	const bool s_host = !is_device_pointer(s), y_host = !is_device_pointer(y);
	real* sd = stage_in(c, 0, s, nn, s_host);
	if (!sd || (y_host && !ensure_stage(c, 1))) return -1000;
	real* yd = y_host ? c->stage[1] : y;
	launch_fisher(c->sc, c->next_buf(), nn, c->F.dev, fu, sd, c->fisher_t, yd);
	to_host(c, c->pin + 8 + 2 * c->m, c->fisher_t, fu);
	if (y_host) vec_to_host(c, y, yd, nn);
	sync(c);
	if (const bool hip_failed = take_hip_failure(); c->fault || hip_failed) { c->fault = false; return -1000; }
	hand_back(buffer_y, c->pin + 8 + 2 * c->m, fu);
	return 0;
}

int stochqn_hip_fisher_product(real_t F[], size_t fu, int n, real_t s[], real_t buffer_y[], real_t y[])
{
	return no_throw(nullptr, "stochqn_hip_fisher_product", [&] { return fisher_product_impl(F, fu, n, s, buffer_y, y); });
}

}  // extern "C"
