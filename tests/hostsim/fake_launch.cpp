// fake_launch.cpp -- stand-ins for the kernel launchers of stochqn_amd/csrc/sqn_device.hpp (TEST INFRASTRUCTURE ONLY).
//
// tests/hostsim builds the product's host logic without kernels.hip: what machines.cpp "launches" lands here.  A stand-in
// does three things and no numerics of the recursion:
//   * it TOUCHES the memory footprint the real kernel has -- every input vector is read over its n elements, every output
//     is rewritten -- on the fake stream (fake_hip.hpp: now, or at the next synchronisation in lazy mode), so that the
//     sanitizers see every pointer and size the host logic hands to a kernel (a dead mirror, a carry scratch that is too
//     small for the grid, a staging vector freed too early, two shard threads on one array);
//   * it produces the few scalars the state machines branch on -- the guard's verdict, a pair's verdict, (s'y, s's, y'y) --
//     from switches the tests set (fakelaunch::script), so that the rejection / rollback / resend branches run;
//   * the update itself is done for real (x -= step r, x_sum += x, the oLBFGS s-slot, copies of the raw gradient): one
//     line each, and it lets a test follow x through host staging, sliced downloads and spill / resume.  The direction
//     is the gradient itself (the two-loop stand-ins leave g alone).
// Reductions go through launch_fin + the context's all-reduce hook exactly as in kernels.hip (finish()), so the
// reducers (fake RCCL, loop-back) are exercised by every sweep.
#include "fake_launch.hpp"
#include "fake_hip.hpp"

#include "sqn_device.hpp"

#include <cstring>
#include <cmath>

namespace fakelaunch {
Script& script()
{
	static Script s;
	return s;
}
}  // namespace fakelaunch

namespace sqn {

namespace {

volatile double g_sink;

void rd(const real* p, size_t n)
{
	if (!p) return;
	double s = 0;
	for (size_t i = 0; i < n; i++) s += (double) p[i];
	g_sink = s;
}

void rw(real* p, size_t n)                       // rewrite in place: a write access to every element, values kept
{
	if (!p) return;
	volatile real* v = p;
	for (size_t i = 0; i < n; i++) v[i] = v[i];
}

void rdd(const double* p, size_t n)
{
	if (!p) return;
	double s = 0;
	for (size_t i = 0; i < n; i++) s += p[i];
	g_sink = s;
}

void rows_rd(const RowSet& r, size_t n) { for (int j = 0; j < r.count; j++) rd(r.row[j], n); }

void read_partials(Partials in, int nq)
{
	if (!in.parts) return;
	for (int j = 0; j < nq; j++) rdd(in.parts + (size_t) j * in.stride, (size_t) in.count);
}

// per-workgroup partials of `nq` quantities whose totals are `total`
void write_partials(double* parts, int grid, int nq, double total)
{
	for (int j = 0; j < nq; j++)
		for (int b = 0; b < grid; b++) parts[(size_t) j * kMaxGrid + b] = total / grid;
}

void run(const Scratch& sc, int id, std::function<void()> work)
{
	const size_t pr = sc.prof ? sc.prof->begin(id, sc.stream) : 0;
	fakehip::enqueue(sc.stream, std::move(work));
	if (sc.prof) sc.prof->end(pr, sc.stream);
}

int next_rev(const Scratch& sc) { return (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0; }

Partials finish(const Scratch& sc, double* raw_parts, double* red, int nsums, int grid)
{
	Partials raw{raw_parts, grid, kMaxGrid};
	if (!sc.allreduce) return raw;
	launch_fin(sc, raw, nsums, red);
	sc.allreduce(sc.user, red, nsums, sc.stream);
	return Partials{red, 1, 1};
}

}  // namespace

const char* kernel_name(int id)
{
	static const char* names[K_COUNT] = {
		"first", "bwd", "mid", "fwd", "fwd_last", "apply", "pair_s", "pair_y_diff", "pair_y_hv",
		"dots3", "fisher_t", "fisher_y", "fin", "small", "copy", "sdot", "sdot2", "qdot", "sadd"};
	return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}

int sweep_grid(const Scratch& sc, size_t n, int per_cu)
{
	size_t g = (n + 1023) / 1024, cap = (size_t) sc.grid_cap * per_cu;
	if (cap > (size_t) kMaxGrid) cap = kMaxGrid;
	if (g < 1) g = 1;
	return (int) (g > cap ? cap : g);
}

Partials launch_first(const Scratch& sc, int buf, size_t n, const FirstArgs& a)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	run(sc, K_FIRST, [=] {
		rd(a.q, n); rd(a.s_newest, n);
		if (a.gprev_out) for (size_t i = 0; i < n; i++) a.gprev_out[i] = a.q[i];
		if (a.frow_out) for (size_t i = 0; i < n; i++) a.frow_out[i] = a.q[i];
		if (a.G) { rw(a.G, n); rw(a.H0_out ? a.H0_out : a.q, n); }
		write_partials(parts, grid, 2, 1.0);
		if (!a.s_newest) write_partials(parts + kMaxGrid, grid, 1, 0.0);     // guard: no non-finite entries
	});
	return finish(sc, parts, sc.red[buf], a.s_newest ? 1 : 2, grid);
}

Partials launch_bwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i, const real* y_i, real* q, const real* s_prev)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	double *alpha = sc.alpha + i, *rho = sc.rho + i;
	run(sc, K_BWD, [=] { read_partials(in, 1); rdd(sy_row, 1); rd(y_i, n); rw(q, n); rd(s_prev, n); *alpha = 1; *rho = 1; write_partials(parts, grid, 1, 1.0); });
	return finish(sc, parts, sc.red[buf], 1, grid);
}

Partials launch_mid(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, const real* y_0, real* q, const MidScale& ms)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	double *alpha = sc.alpha, *rho = sc.rho;
	run(sc, K_MID, [=] {
		read_partials(in, 1); rdd(sy_row, 1); rdd(ms.sy_newest, 1); rdd(ms.yy_newest, 1); rd(ms.H0, n);
		rd(y_0, n); rw(q, n); *alpha = 1; *rho = 1; write_partials(parts, grid, 1, 1.0);
	});
	return finish(sc, parts, sc.red[buf], 1, grid);
}

Partials launch_fwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i, const real* s_i, real* r, const real* y_next)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	const double* alpha = sc.alpha + i;
	run(sc, K_FWD, [=] { read_partials(in, 1); rdd(sy_row, 1); rdd(alpha, 1); rd(s_i, n); rw(r, n); rd(y_next, n); write_partials(parts, grid, 1, 1.0); });
	return finish(sc, parts, sc.red[buf], 1, grid);
}

namespace {
void do_apply(const real* r, real* grad_out, const ApplyArgs& ap, size_t n, bool bad)
{
	for (size_t i = 0; i < n; i++) {
		if (!bad) {
			const double sg = -ap.step * (double) r[i];
			ap.x[i] = (real) ((double) ap.x[i] + sg);
			if (ap.s_slot) { ap.s_slot[i] = (real) sg; grad_out[i] = (real) sg; }
		}
		if (ap.x_sum) ap.x_sum[i] = (real) ((double) ap.x_sum[i] + (double) ap.x[i]);
	}
}
}  // namespace

Partials launch_fwd_last(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i, const real* s_i, real* r, const ApplyArgs* fuse)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	const double* alpha = sc.alpha + i;
	if (fuse) {
		const ApplyArgs ap = *fuse;
		run(sc, K_FWD_LAST, [=] { read_partials(in, 1); rdd(sy_row, 1); rdd(alpha, 1); rd(s_i, n); rw(r, n); do_apply(r, r, ap, n, false); });
		return Partials{nullptr, 0, 0};
	}
	run(sc, K_FWD_LAST, [=] {
		read_partials(in, 1); rdd(sy_row, 1); rdd(alpha, 1); rd(s_i, n); rw(r, n);
		write_partials(parts, grid, 1, 1.0); write_partials(parts + kMaxGrid, grid, 1, 0.0);
	});
	return finish(sc, parts, sc.red[buf], 2, grid);
}

void launch_apply(const Scratch& sc, size_t n, double n_global, Partials guard, const real* r_in, real* grad_out, const ApplyArgs& a, bool guarded)
{
	(void) n_global;
	double* report = sc.report;
	const bool bad = guarded && fakelaunch::script().reject_step;
	run(sc, K_APPLY, [=] {
		if (guarded) read_partials(guard, 2);
		report[0] = bad ? 1.0 : 0.0; report[1] = guarded ? 1.0 : 0.0; report[2] = 0.0;
		do_apply(r_in, grad_out, a, n, bad);
	});
}

void launch_spec_x(const Scratch& sc, size_t n, const real* r, const real* x, double step, real* out)
{
	run(sc, K_APPLY, [=] { for (size_t i = 0; i < n; i++) out[i] = (real) ((double) x[i] - step * (double) r[i]); });
}

// the checksum of x on the "device": the host's own routine (runtime.cpp) over the device copy, added into out2[0..1]
void launch_xhash(const Scratch& sc, const real* x, size_t n, double* out2)
{
	run(sc, K_XHASH, [=] {
		XHash h;
		xhash_host(x, n * sizeof(real), 0, xhash_words(n * sizeof(real)), &h);
		unsigned long long w[2];
		std::memcpy(w, out2, sizeof w);
		w[0] += h.a; w[1] += h.b;
		std::memcpy(out2, w, sizeof w);
	});
}

void launch_pair_s(const Scratch& sc, size_t n, real* x_sum, double inv_L, bool scale, const real* x_avg_prev, real* s_out)
{
	(void) next_rev(sc);
	run(sc, K_PAIR_S, [=] {
		for (size_t i = 0; i < n; i++) {
			if (scale) x_sum[i] = (real) ((double) x_sum[i] * inv_L);
			s_out[i] = (real) ((double) x_sum[i] - (double) x_avg_prev[i]);
		}
	});
}

namespace {
Partials three_dots(const Scratch& sc, int id, int buf, size_t n, std::function<void()> touch)
{
	const int grid = sweep_grid(sc, n);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	const fakelaunch::Script s = fakelaunch::script();
	run(sc, id, [=] {
		touch();
		write_partials(parts, grid, 1, s.sy); write_partials(parts + kMaxGrid, grid, 1, s.ss); write_partials(parts + 2 * kMaxGrid, grid, 1, s.yy);
	});
	return finish(sc, parts, sc.red[buf], 3, grid);
}
}  // namespace

void launch_verdict(const Scratch& sc, Partials in, double min_curvature, double* sy_dst, double* yy_dst, double* out);

// the pair kernel's last workgroup takes the verdict itself (kernels.hip: k_sweep_verdict): same stream, right behind the sweep
static Partials with_verdict(const Scratch& sc, Partials p, const VerdictArgs* v)
{
	if (!v || sc.allreduce || !sc.ticket) return p;
	launch_verdict(sc, p, v->min_curvature, v->sy_dst, v->yy_dst, v->out);
	return Partials{nullptr, 0, 0};
}

Partials launch_pair_y_diff(const Scratch& sc, int buf, size_t n, const real* g, const real* g_prev, const real* s, double lambda, real* y_out, const VerdictArgs* v)
{
	(void) lambda;
	return with_verdict(sc, three_dots(sc, K_PAIR_Y_DIFF, buf, n, [=] { rd(g_prev, n); rd(s, n); for (size_t i = 0; i < n; i++) y_out[i] = g[i]; }), v);
}

Partials launch_pair_y_hv(const Scratch& sc, int buf, size_t n, const real* hv, const real* s, real* y_out, real* x_sum, real* x_avg_prev, const VerdictArgs* v)
{
	return with_verdict(sc, three_dots(sc, K_PAIR_Y_HV, buf, n, [=] {
		rd(s, n);
		for (size_t i = 0; i < n; i++) { y_out[i] = hv[i]; x_avg_prev[i] = x_sum[i]; x_sum[i] = 0; }
	}), v);
}

Partials launch_dots3(const Scratch& sc, int buf, size_t n, const real* s, const real* y)
{
	return three_dots(sc, K_DOTS3, buf, n, [=] { rd(s, n); rd(y, n); });
}

Partials launch_fisher(const Scratch& sc, int buf, size_t n, const real* F, size_t fu, const real* s, double* t_dev, real* y_out)
{
	const int grid = sweep_grid(sc, n);
	double* fpart = sc.fisher_part;
	run(sc, K_FISHER_T, [=] {
		rd(F, fu * n); rd(s, n);
		for (size_t j = 0; j < fu; j++) for (int b = 0; b < grid; b++) fpart[j * kMaxGrid + b] = 1.0 / grid;
	});
	launch_fin(sc, Partials{fpart, grid, kMaxGrid}, (int) fu, t_dev);
	if (sc.allreduce) sc.allreduce(sc.user, t_dev, (int) fu, sc.stream);
	return three_dots(sc, K_FISHER_Y, buf, n, [=] { rd(F, fu * n); rdd(t_dev, fu); rd(s, n); rw(y_out, n); });
}

// ---- three-pass form ------------------------------------------------------------------------------------------------
static int sdot_grid(const Scratch& sc, size_t n, bool two)
{
	return sweep_grid(sc, n, two ? (sc.sdot2_per_cu > 0 ? sc.sdot2_per_cu : 3) : (sc.sdot_per_cu > 0 ? sc.sdot_per_cu : 1));
}

size_t sdot_carry_count(const Scratch& sc, size_t n, int k)
{
	const size_t one = (size_t) k * (size_t) sdot_grid(sc, n, false), two = 2 * (size_t) k * (size_t) sdot_grid(sc, n, true);
	return (one > two ? one : two) * kBlock;
}

bool sdot_can_slice(const Scratch& sc, const RowSet&, const real*, real*, const real* probe_y) { return probe_y || !sc.rows_split; }

Partials launch_sdot(const Scratch& sc, size_t n, const RowSet& s_rows, const real* g, real* copy_out, const real* probe_y, const SliceFeed* feed)
{
	const int grid = sdot_grid(sc, n, probe_y != nullptr);
	(void) next_rev(sc);
	const int nq = (probe_y ? 2 : 1) * s_rows.count;
	double* parts = sc.rows_part[0];
	const size_t lanes = (size_t) grid * kBlock;
	const bool sliced = feed && feed->slices >= 2 && n >= (size_t) 2 * feed->slices && feed->carry && feed->carry_count >= (size_t) nq * lanes;
	const int slices = sliced ? feed->slices : 1;
	for (int s = 0; s < slices; s++) {
		const size_t lo = n * (size_t) s / (size_t) slices, hi = n * (size_t) (s + 1) / (size_t) slices;
		if (feed) feed->arrive(feed->user, lo, hi, s);
		double* carry = sliced ? feed->carry : nullptr;
		const RowSet rows = s_rows;
		run(sc, probe_y ? K_SDOT2 : K_SDOT, [=] {
			for (int j = 0; j < rows.count; j++) rd(rows.row[j] + lo, hi - lo);
			rd(g + lo, hi - lo);
			if (probe_y) rd(probe_y + lo, hi - lo);
			if (copy_out) for (size_t i = lo; i < hi; i++) copy_out[i] = g[i];
			if (carry) for (size_t e = 0; e < (size_t) nq * lanes; e++) carry[e] = 0.0;     // the accumulators of every lane, between the slices
			if (s == slices - 1) for (int j = 0; j < nq; j++) for (int b = 0; b < grid; b++) parts[(size_t) j * kMaxGrid + b] = 1.0 / grid;
		});
	}
	return finish(sc, parts, sc.red[0], nq, grid);
}

namespace {
// the recursion in the prologue of pass 2: the partials of pass 1, the cached block, s'y / y'y; writes the new column, alpha, rho
void fold_a(const Scratch& sc, const Partials& in, const CoefArgs& a, int fresh_row)
{
	read_partials(in, (fresh_row >= 0 ? 2 : 1) * a.k);
	if (fresh_row >= 0) for (int i = 0; i < a.k; i++) sc.gsy[(size_t) a.rows[i] * a.m + fresh_row] = 1.0;
	for (int i = 0; i < a.k; i++) { rdd(sc.sy + a.rows[i], 1); rdd(sc.yy + a.rows[i], 1); sc.alpha[i] = 1; sc.rho[i] = 1; }
}
}  // namespace

Partials launch_qdot(const Scratch& sc, size_t n, const RowSet& y_rows, real* g, const QdotScale& q, const Partials& pass1, const CoefArgs& a, int fresh_row)
{
	const int grid = sweep_grid(sc, n, sc.qdot_per_cu > 0 ? sc.qdot_per_cu : 1);
	(void) next_rev(sc);
	double* parts = sc.rows_part[1];
	const Scratch scc = sc;
	const RowSet rows = y_rows;
	const Partials fin = pass1;
	const CoefArgs ca = a;
	run(sc, K_QDOT, [=] {
		fold_a(scc, fin, ca, fresh_row);
		rows_rd(rows, n); rd(q.H0_in, n);
		if (q.G) { rw(q.G, n); rw(q.H0_out, n); }
		if (q.frow_out) for (size_t i = 0; i < n; i++) q.frow_out[i] = g[i];
		rw(g, n);
		for (int j = 0; j < rows.count; j++) for (int b = 0; b < grid; b++) parts[(size_t) j * kMaxGrid + b] = 1.0 / grid;
	});
	return finish(sc, parts, sc.red[1], y_rows.count, grid);
}

bool sadd_can_slice(const Scratch& sc, size_t n, const RowSet&, const real*, const SliceFeed* drain)
{
	if (!drain || drain->slices < 2 || n < (size_t) 2 * drain->slices) return false;
	const int grid = sweep_grid(sc, n, sc.sadd_per_cu > 0 ? sc.sadd_per_cu : 2);
	return drain->carry && drain->carry_count >= 2 * (size_t) grid * kBlock;
}

Partials launch_sadd(const Scratch& sc, int buf, size_t n, const RowSet& s_rows, real* r, const Partials& pass2, const CoefArgs& a, const SliceFeed* drain)
{
	const int grid = sweep_grid(sc, n, sc.sadd_per_cu > 0 ? sc.sadd_per_cu : 2);
	(void) next_rev(sc);
	double* parts = sc.part[buf];
	const Scratch scc = sc;
	const RowSet rows = s_rows;
	const Partials fin = pass2;
	const CoefArgs ca = a;
	const bool sliced = sadd_can_slice(sc, n, s_rows, r, drain);
	const int slices = sliced ? drain->slices : 1;
	for (int s = 0; s < slices; s++) {
		const size_t lo = n * (size_t) s / (size_t) slices, hi = n * (size_t) (s + 1) / (size_t) slices;
		double* carry = sliced ? drain->carry : nullptr;
		const size_t lanes = (size_t) grid * kBlock;
		run(sc, K_SADD, [=] {
			if (s == 0) { read_partials(fin, ca.k); for (int i = 0; i < ca.k; i++) { rdd(scc.gsy + (size_t) ca.rows[i] * ca.m, (size_t) ca.m); rdd(scc.alpha + i, 1); } }
			for (int j = 0; j < rows.count; j++) rd(rows.row[j] + lo, hi - lo);
			rw(r + lo, hi - lo);
			if (carry) for (size_t e = 0; e < 2 * lanes; e++) carry[e] = 0.0;
			if (s == slices - 1) { write_partials(parts, grid, 1, 1.0); write_partials(parts + kMaxGrid, grid, 1, 0.0); }
		});
		if (sliced) drain->arrive(drain->user, lo, hi, s);
	}
	return finish(sc, parts, sc.red[buf], 2, grid);
}

void launch_store_column(const Scratch& sc, Partials in, const CoefArgs& a, int col_row)
{
	double* gsy = sc.gsy;
	run(sc, K_SMALL, [=] { read_partials(in, a.k); for (int i = 0; i < a.k; i++) gsy[(size_t) a.rows[i] * a.m + col_row] = 1.0; });
}

void launch_fin(const Scratch& sc, Partials in, int nsums, double* out)
{
	run(sc, K_FIN, [=] {
		for (int j = 0; j < nsums; j++) {
			double t = 0;
			for (int b = 0; b < in.count; b++) t += in.parts[(size_t) j * in.stride + b];
			out[j] = t;
		}
	});
}

void launch_commit(const Scratch& sc, Partials in, double* sy_dst, double* yy_dst)
{
	run(sc, K_SMALL, [=] { read_partials(in, 3); *sy_dst = 1.0; *yy_dst = 1.0; });
}

void launch_verdict(const Scratch& sc, Partials in, double min_curvature, double* sy_dst, double* yy_dst, double* out)
{
	const fakelaunch::Script s = fakelaunch::script();
	run(sc, K_SMALL, [=] {
		read_partials(in, 3);
		const bool rejected = min_curvature > 0 && s.reject_pair;
		out[0] = s.sy; out[1] = s.ss; out[2] = s.yy; out[3] = rejected ? 1.0 : 0.0;
		if (!rejected) { *sy_dst = s.sy; *yy_dst = s.yy; }
	});
}

uint64_t synth_key(uint64_t seed, uint64_t stream, uint64_t t) { return seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ (t * 0xD1B54A32D192ED03ull); }
void launch_synth_uniform(hipStream_t st, real* out, size_t count, uint64_t, uint64_t, double a, double) { fakehip::enqueue(st, [=] { for (size_t i = 0; i < count; i++) out[i] = (real) a; }); }
void launch_synth_grad(hipStream_t st, real* g, const real* d, const real* x, size_t count, uint64_t, uint64_t, double) { fakehip::enqueue(st, [=] { for (size_t i = 0; i < count; i++) g[i] = d[i] * x[i]; }); }
void launch_synth_batch_row(hipStream_t st, real* row, const real* d, size_t count, uint64_t, uint32_t, uint32_t) { fakehip::enqueue(st, [=] { for (size_t i = 0; i < count; i++) row[i] = d[i]; }); }

void launch_scale(const Scratch& sc, size_t n, real* x, double a)
{
	run(sc, K_SMALL, [=] { for (size_t i = 0; i < n; i++) x[i] = (real) ((double) x[i] * a); });
}

}  // namespace sqn
