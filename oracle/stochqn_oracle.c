/*  stochqn_oracle.c -- CPU restatement of the reference's stochastic quasi-Newton step path.
 *
 *  THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  It is the parity oracle for the HIP library
 *  in stochqn_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 *  leg may load it; libstochqn.so never links, loads or calls anything in oracle/.
 *
 *  What it restates (all citations are into /root/reference):
 *      src/stochqn.c:663-708    approx_inv_hess_grad  (L-BFGS two-loop recursion)
 *      src/stochqn.c:720-783    update_sum_sq, diag_rescal (AdaGrad / RMSProp scaling)
 *      src/stochqn.c:802-840    take_step
 *      src/stochqn.c:861-966    update_s_vector, check_min_curvature, update_y_{grad_diff,fisher,hessvec}
 *      src/stochqn.c:554-610    ring counters, backup/rollback, Fisher append, archive_x_avg
 *      src/stochqn.c:978-1315   run_oLBFGS, run_SQN, run_adaQN state machines
 *      src/stochqn.c:300-547    workspace allocation
 *  on the struct / enum definitions of include/stochqn.h:86-151,268-291 (re-declared, ABI
 *  identical, in this repository's include/stochqn.h).
 *
 *  Third-party arithmetic: the reference calls CBLAS ddot/daxpy/dscal/dnrm2/dgemv from whatever
 *  BLAS it was linked with (no pinned version; call sites src/stochqn.c:676-706,829,838,892,
 *  923,946-949,1006).  Their semantics are the textbook ones; only the summation order is
 *  implementation defined.  The loops below restate them with a fixed, documented order.
 *
 *  Pinning status: the reference itself cannot be built under this project's rules (its
 *  non-R/non-Python build needs a CMake-generated blasfuns.h plus an external BLAS; see
 *  DESIGN.md "oracle").  The oracle is pinned by the known-answer trajectories recorded from
 *  the compiled reference in SURVEY.md section 4 (committed as tests/golden/known_answers.json)
 *  and by example/c_rosen.c's printed output; see tests/test_oracle_known_answers.py.
 *
 *  Precision: compiled as is for the double ABI (liboracle.so) and with -DUSE_FLOAT for the float ABI
 *  (liboracle_f32.so): vectors and element-wise arithmetic in real_t, as in the reference's float
 *  build, inner products accumulated in double (a BLAS is free to do either).
 *
 *  Symbols are prefixed `oracle_` so that the oracle and libstochqn.so can live in one process.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "stochqn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * BLAS-1 restatements.  Dots are accumulated in 8 interleaved partial sums per thread-chunk
 * (the shape every SIMD BLAS kernel uses) and the chunks are combined in index order, so a
 * result depends only on (n, number of threads).  `oracle_set_threads` fixes the latter.
 * ------------------------------------------------------------------------------------------ */
static int g_threads = 1;

void oracle_set_threads(int nthreads) { g_threads = (nthreads < 1) ? 1 : nthreads; }
int  oracle_get_threads(void) { return g_threads; }

#define PAR_MIN 262144  /* below this a vector op stays on the calling thread */

static double dot_chunk(const real_t *a, const real_t *b, size_t lo, size_t hi)
{
	double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	size_t i = lo;
	for (; i + 8 <= hi; i += 8)
		for (int k = 0; k < 8; k++) acc[k] += (double) a[i + k] * (double) b[i + k];
	double tail = 0;
	for (; i < hi; i++) tail += (double) a[i] * (double) b[i];
	return (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) + tail;
}

/* ddot (reference call sites: src/stochqn.c:676,677,686,687,705,892) */
static double v_dot(size_t n, const real_t *a, const real_t *b)
{
	int nt = (n < PAR_MIN) ? 1 : g_threads;
	if (nt == 1) return dot_chunk(a, b, 0, n);
	double part[256];
	if (nt > 256) nt = 256;
	#pragma omp parallel for num_threads(nt) schedule(static, 1)
	for (int t = 0; t < nt; t++) {
		size_t lo = n * (size_t) t / (size_t) nt, hi = n * (size_t) (t + 1) / (size_t) nt;
		part[t] = dot_chunk(a, b, lo, hi);
	}
	double s = 0;
	for (int t = 0; t < nt; t++) s += part[t];
	return s;
}

/* daxpy: y += a*x (src/stochqn.c:678,706,838,923) */
static void v_axpy(size_t n, real_t a, const real_t *x, real_t *y)
{
	#pragma omp parallel for num_threads(g_threads) schedule(static) if (n >= PAR_MIN)
	for (size_t i = 0; i < n; i++) y[i] += a * x[i];
}

/* dscal: x *= a (src/stochqn.c:289,688,698,1006) */
static void v_scal(size_t n, real_t a, real_t *x)
{
	#pragma omp parallel for num_threads(g_threads) schedule(static) if (n >= PAR_MIN)
	for (size_t i = 0; i < n; i++) x[i] *= a;
}

/* dnrm2 (src/stochqn.c:829): overflow-safe two-pass form */
static double v_nrm2(size_t n, const real_t *x)
{
	double big = 0;
	for (size_t i = 0; i < n; i++) { double a = fabs(x[i]); if (a > big) big = a; }
	if (big == 0 || isnan(big) || isinf(big)) return big;
	double s = 0;
	#pragma omp parallel for num_threads(g_threads) schedule(static) reduction(+ : s) if (n >= PAR_MIN)
	for (size_t i = 0; i < n; i++) { double t = x[i] / big; s += t * t; }
	return big * sqrt(s);
}

static void v_copy(size_t n, const real_t *src, real_t *dst) { memcpy(dst, src, n * sizeof(real_t)); } /* :145-169 */
static void v_zero(size_t n, real_t *x) { memset(x, 0, n * sizeof(real_t)); }                          /* :171-194 */

/* ------------------------------------------------------------------------------------------
 * Ring-buffer bookkeeping (src/stochqn.c:554-610)
 * ------------------------------------------------------------------------------------------ */
static void ring_reset(bfgs_mem *b) { b->mem_used = 0; b->mem_st_ix = 0; }                 /* :554-558 */
static void fisher_reset(fisher_mem *f) { if (f) { f->mem_used = 0; f->mem_st_ix = 0; } } /* :560-567 */

static void ring_advance(bfgs_mem *b)                                                      /* :569-573 */
{
	b->mem_st_ix = (b->mem_st_ix + 1) % b->mem_size;
	b->mem_used = (b->mem_used + 1 >= b->mem_size) ? b->mem_size : b->mem_used + 1;
}

static void fisher_append(const real_t *g, fisher_mem *f, int n)                           /* :575-587 */
{
	if (!f) return;
	v_copy(n, g, f->F + f->mem_st_ix * (size_t) n);
	f->mem_st_ix = (f->mem_st_ix + 1) % f->mem_size;
	f->mem_used = (f->mem_used + 1 >= f->mem_size) ? f->mem_size : f->mem_used + 1;
}

/* The reference's "backup" and "rollback" both copy bak -> slot (argument order of copy_arr
 * is (src, dest); src/stochqn.c:589-604).  Restated as written, not as intended. */
static void slot_from_bak(bfgs_mem *b, int n)
{
	v_copy(n, b->s_bak, b->s_mem + b->mem_st_ix * (size_t) n);
	v_copy(n, b->y_bak, b->y_mem + b->mem_st_ix * (size_t) n);
}
static void pair_backup(bfgs_mem *b, int n) { if (b->min_curvature > 0) slot_from_bak(b, n); }
static void pair_rollback(bfgs_mem *b, int n, info_enum *info)
{
	if (b->min_curvature > 0) { slot_from_bak(b, n); *info = curvature_too_small; }
}

/* x_avg_prev <- x_avg ; x_sum <- 0 (x_avg is the same array as x_sum; :134,606-610) */
static void archive_average(real_t *x_sum, real_t *x_avg_prev, int n)
{
	v_copy(n, x_sum, x_avg_prev);
	v_zero(n, x_sum);
}

/* x_sum *= 1/L, skipped for L <= 1 (:286-291) */
static void sum_to_average(real_t *x_sum, size_t L, int n)
{
	if (L > 1) v_scal(n, 1 / (double) L, x_sum);
}

/* ------------------------------------------------------------------------------------------
 * Two-loop recursion (src/stochqn.c:663-708).  `st` is the physical row of the OLDEST pair;
 * logical pair i lives in row (st + i) % m.  rho/alpha are stored by logical index.
 * ------------------------------------------------------------------------------------------ */
void oracle_two_loop(real_t *q, int n, const real_t *H0, real_t h0, const real_t *Y, const real_t *S,
                     size_t m, size_t used, size_t st, real_t *rho, real_t *alpha)
{
	size_t N = (size_t) n;
	for (size_t k = used; k-- > 0;) {                                  /* newest -> oldest, :671-679 */
		size_t row = (st + k) % m;
		rho[k] = 1 / v_dot(N, Y + row * N, S + row * N);
		alpha[k] = rho[k] * v_dot(N, q, S + row * N);
		v_axpy(N, -alpha[k], Y + row * N, q);
	}

	if (H0 == NULL && h0 <= 0) {                                       /* :683-689 */
		size_t newest = (st - 1 + used) % m;
		double gamma = v_dot(N, S + newest * N, Y + newest * N) / v_dot(N, Y + newest * N, Y + newest * N);
		v_scal(N, gamma, q);
	} else if (H0 != NULL) {                                           /* :695 */
		for (size_t i = 0; i < N; i++) q[i] *= H0[i];
	} else {                                                           /* :698 */
		v_scal(N, h0, q);
	}

	for (size_t k = 0; k < used; k++) {                                /* oldest -> newest, :702-707 */
		size_t row = (st + k) % m;
		double beta = rho[k] * v_dot(N, Y + row * N, q);
		v_axpy(N, alpha[k] - beta, S + row * N, q);
	}
}

/* G <- w*G + (1-w)*g^2 for 0<w<1, else G <- G + g^2 (:720-747); then out <- g / sqrt(G + eps)
 * where out is `direction` or, when that is NULL, g itself (:762-783). */
void oracle_diag_rescale(real_t *direction, real_t *g, real_t *G, int n, real_t eps, real_t w)
{
	size_t N = (size_t) n;
	if (w > 0 && w < 1) {
		double wn = 1 - w;
		for (size_t i = 0; i < N; i++) G[i] = w * G[i] + wn * (g[i] * g[i]);
	} else {
		for (size_t i = 0; i < N; i++) G[i] += g[i] * g[i];
	}
	if (direction == NULL) for (size_t i = 0; i < N; i++) g[i] /= sqrt(G[i] + eps);
	else                   for (size_t i = 0; i < N; i++) direction[i] = g[i] / sqrt(G[i] + eps);
}

static int has_nonfinite(const real_t *a, size_t n)                    /* :228-266 */
{
	for (size_t i = 0; i < n; i++) if (isinf(a[i]) || isnan(a[i])) return 1;
	return 0;
}

/* take_step (:802-840).  `G == NULL` means "no diagonal rescaling" (oLBFGS, SQN). */
void oracle_take_step(real_t step, int n, real_t *x, real_t *g, bfgs_mem *b, real_t w, real_t *H0,
                      real_t h0, real_t *G, real_t eps, int check_nan, info_enum *info)
{
	if (b->mem_used == 0) {
		if (G != NULL) oracle_diag_rescale(NULL, g, G, n, eps, w);             /* :808-812 */
	} else {
		if (G != NULL) oracle_diag_rescale(H0, g, G, n, eps, w);               /* :818 */
		size_t st = (b->mem_st_ix == b->mem_used) ? 0 : b->mem_st_ix;          /* :820 */
		oracle_two_loop(g, n, H0, h0, b->y_mem, b->s_mem, b->mem_size, b->mem_used, st,
		                b->buffer_rho, b->buffer_alpha);
	}
	if (check_nan) {                                                           /* :825-835 */
		if (has_nonfinite(g, (size_t) n) || v_nrm2((size_t) n, g) > 1e3 * n) {
			ring_reset(b);
			*info = search_direction_was_nan;
			return;
		}
	}
	v_axpy((size_t) n, -step, g, x);                                           /* :838 */
}

/* ------------------------------------------------------------------------------------------
 * Correction-pair construction (src/stochqn.c:861-966)
 * ------------------------------------------------------------------------------------------ */
static void make_s(real_t *x_sum, const real_t *x_avg_prev, int n, int needs_div, bfgs_mem *b) /* :861-870 */
{
	pair_backup(b, n);
	if (needs_div) sum_to_average(x_sum, b->upd_freq, n);
	real_t *s = b->s_mem + b->mem_st_ix * (size_t) n;
	for (size_t i = 0; i < (size_t) n; i++) s[i] = x_sum[i] - x_avg_prev[i];
}

static void accept_or_reject(bfgs_mem *b, int n, info_enum *info)                              /* :883-900 */
{
	const real_t *s = b->s_mem + b->mem_st_ix * (size_t) n;
	const real_t *y = b->y_mem + b->mem_st_ix * (size_t) n;
	if (b->min_curvature > 0) {
		double curv = v_dot((size_t) n, s, y) / v_dot((size_t) n, s, s);
		if (curv <= b->min_curvature) { pair_rollback(b, n, info); return; }   /* NaN curvature passes */
	}
	ring_advance(b);
}

static void make_y_graddiff(const real_t *g, const real_t *g_prev, bfgs_mem *b, int n, info_enum *info) /* :915-926 */
{
	const real_t *s = b->s_mem + b->mem_st_ix * (size_t) n;
	real_t *y = b->y_mem + b->mem_st_ix * (size_t) n;
	for (size_t i = 0; i < (size_t) n; i++) y[i] = g[i] - g_prev[i];
	if (b->y_reg > 0) v_axpy((size_t) n, b->y_reg, s, y);
	accept_or_reject(b, n, info);
}

/* y = F' (F s) / fu over the fu = mem_used stored gradients (row-major gemv N then T, :936-952) */
void oracle_fisher_product(const real_t *F, size_t fu, int n, const real_t *s, real_t *t, real_t *y)
{
	size_t N = (size_t) n;
	for (size_t k = 0; k < fu; k++) t[k] = v_dot(N, F + k * N, s);
	double inv = 1 / (double) fu;
	#pragma omp parallel for num_threads(g_threads) schedule(static) if (N >= PAR_MIN)
	for (size_t i = 0; i < N; i++) {
		double acc = 0;
		for (size_t k = 0; k < fu; k++) acc += (double) F[k * N + i] * (double) t[k];
		y[i] = (real_t) (inv * acc);
	}
}

static void make_y_fisher(fisher_mem *f, bfgs_mem *b, int n, info_enum *info)
{
	const real_t *s = b->s_mem + b->mem_st_ix * (size_t) n;
	real_t *y = b->y_mem + b->mem_st_ix * (size_t) n;
	oracle_fisher_product(f->F, f->mem_used, n, s, f->buffer_y, y);
	accept_or_reject(b, n, info);
}

static void make_y_hessvec(const real_t *hv, bfgs_mem *b, info_enum *info, int n)              /* :962-966 */
{
	v_copy(n, hv, b->y_mem + b->mem_st_ix * (size_t) n);
	accept_or_reject(b, n, info);
}

/* ------------------------------------------------------------------------------------------
 * Workspaces (src/stochqn.c:300-547).  Deviation, documented: s_bak/y_bak are zero-filled
 * (the reference leaves malloc garbage there and then reads it).
 * ------------------------------------------------------------------------------------------ */
bfgs_mem* oracle_initialize_bfgs_mem(size_t m, int n, real_t min_curvature, real_t y_reg, size_t L)
{
	bfgs_mem *b = (bfgs_mem*) calloc(1, sizeof(bfgs_mem));
	if (!b) return NULL;
	b->s_mem = (real_t*) malloc(sizeof(real_t) * (size_t) n * m);
	b->y_mem = (real_t*) malloc(sizeof(real_t) * (size_t) n * m);
	b->buffer_rho = (real_t*) malloc(sizeof(real_t) * m);
	b->buffer_alpha = (real_t*) malloc(sizeof(real_t) * m);
	if (min_curvature > 0) {
		b->s_bak = (real_t*) calloc((size_t) n, sizeof(real_t));
		b->y_bak = (real_t*) calloc((size_t) n, sizeof(real_t));
	}
	b->mem_size = m; b->upd_freq = L; b->y_reg = y_reg; b->min_curvature = min_curvature;
	return b;
}

void oracle_dealloc_bfgs_mem(bfgs_mem *b)
{
	if (!b) return;
	free(b->s_mem); free(b->y_mem); free(b->buffer_rho); free(b->buffer_alpha); free(b->s_bak); free(b->y_bak);
	free(b);
}

fisher_mem* oracle_initialize_fisher_mem(size_t f, int n)
{
	fisher_mem *o = (fisher_mem*) calloc(1, sizeof(fisher_mem));
	if (!o) return NULL;
	o->F = (real_t*) malloc(sizeof(real_t) * (size_t) n * f);
	o->buffer_y = (real_t*) malloc(sizeof(real_t) * f);
	o->mem_size = f;
	return o;
}

void oracle_dealloc_fisher_mem(fisher_mem *f) { if (f) { free(f->F); free(f->buffer_y); free(f); } }

workspace_oLBFGS* oracle_initialize_oLBFGS(int n, size_t m, real_t hess_init, real_t y_reg,
                                           real_t min_curvature, int check_nan, int nthreads)
{
	workspace_oLBFGS *w = (workspace_oLBFGS*) calloc(1, sizeof(*w));
	w->bfgs_memory = oracle_initialize_bfgs_mem(m, n, min_curvature, y_reg, 1);
	w->grad_prev = (real_t*) malloc(sizeof(real_t) * (size_t) n);
	w->hess_init = hess_init; w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	return w;
}
void oracle_dealloc_oLBFGS(workspace_oLBFGS *w) { oracle_dealloc_bfgs_mem(w->bfgs_memory); free(w->grad_prev); free(w); }

workspace_SQN* oracle_initialize_SQN(int n, size_t m, size_t L, real_t min_curvature, int use_grad_diff,
                                     real_t y_reg, int check_nan, int nthreads)
{
	workspace_SQN *w = (workspace_SQN*) calloc(1, sizeof(*w));
	w->bfgs_memory = oracle_initialize_bfgs_mem(m, n, min_curvature, y_reg, L);
	w->grad_prev = use_grad_diff ? (real_t*) malloc(sizeof(real_t) * (size_t) n) : NULL;
	w->x_sum = (real_t*) calloc((size_t) n, sizeof(real_t));
	w->x_avg_prev = (real_t*) malloc(sizeof(real_t) * (size_t) n);
	w->use_grad_diff = use_grad_diff; w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	return w;
}
void oracle_dealloc_SQN(workspace_SQN *w)
{
	oracle_dealloc_bfgs_mem(w->bfgs_memory); free(w->grad_prev); free(w->x_sum); free(w->x_avg_prev); free(w);
}

workspace_adaQN* oracle_initialize_adaQN(int n, size_t m, size_t fisher_size, size_t L, real_t max_incr,
                                         real_t min_curvature, real_t scal_reg, real_t rmsprop_weight,
                                         int use_grad_diff, real_t y_reg, int check_nan, int nthreads)
{
	workspace_adaQN *w = (workspace_adaQN*) calloc(1, sizeof(*w));
	w->bfgs_memory = oracle_initialize_bfgs_mem(m, n, min_curvature, y_reg, L);
	if (use_grad_diff) w->grad_prev = (real_t*) malloc(sizeof(real_t) * (size_t) n);
	else               w->fisher_memory = oracle_initialize_fisher_mem(fisher_size, n);
	w->H0 = (real_t*) malloc(sizeof(real_t) * (size_t) n);
	w->x_sum = (real_t*) calloc((size_t) n, sizeof(real_t));
	w->x_avg_prev = (real_t*) malloc(sizeof(real_t) * (size_t) n);
	w->grad_sum_sq = (real_t*) calloc((size_t) n, sizeof(real_t));
	w->max_incr = max_incr; w->scal_reg = scal_reg; w->rmsprop_weight = rmsprop_weight;
	w->use_grad_diff = use_grad_diff; w->check_nan = check_nan; w->nthreads = nthreads; w->n = n;
	return w;
}
void oracle_dealloc_adaQN(workspace_adaQN *w)
{
	oracle_dealloc_bfgs_mem(w->bfgs_memory); oracle_dealloc_fisher_mem(w->fisher_memory);
	free(w->H0); free(w->grad_prev); free(w->x_sum); free(w->x_avg_prev); free(w->grad_sum_sq); free(w);
}

/* ------------------------------------------------------------------------------------------
 * State machines
 * ------------------------------------------------------------------------------------------ */

/* oLBFGS (src/stochqn.c:978-1036): 0 -> ask grad; 1 -> step, s-slot, ask same-batch grad;
 * 2 -> y-slot, accept/reject, ask grad. */
int oracle_run_oLBFGS(real_t step, real_t *x, real_t *g, real_t **req, task_enum *task,
                      workspace_oLBFGS *w, info_enum *info)
{
	bfgs_mem *b = w->bfgs_memory;
	int n = w->n;
	*info = no_problems_encountered;
	*req = x;
	switch (w->section) {
	case 0:
		*task = calc_grad; w->section = 1;
		return 0;
	case 1:
		v_copy(n, g, w->grad_prev);                                                    /* :996 */
		oracle_take_step(step, n, x, g, b, 0, NULL, w->hess_init, NULL, 0, w->check_nan, info);
		w->niter++;
		if (*info == no_problems_encountered) {
			pair_backup(b, n);                                                         /* :1005 */
			v_scal(n, -step, g);                                                       /* :1006 */
			v_copy(n, g, b->s_mem + b->mem_st_ix * (size_t) n);                        /* :1007 */
			*task = calc_grad_same_batch; w->section = 2;
			return 1;
		}
		ring_reset(b);                                                                 /* :1015 */
		*task = calc_grad; w->section = 1;
		return 0;
	case 2:
		make_y_graddiff(g, w->grad_prev, b, n, info);                                  /* :1026 */
		*task = calc_grad; w->section = 1;
		return 0;
	default:
		*task = invalid_input;
		fprintf(stderr, "oLBFGS got an invalid workspace as input.\n");
		return -1000;
	}
}

/* SQN (src/stochqn.c:1038-1153) */
int oracle_run_SQN(real_t step, real_t *x, real_t *g, real_t *hv, real_t **req, real_t **req_vec,
                   task_enum *task, workspace_SQN *w, info_enum *info)
{
	bfgs_mem *b = w->bfgs_memory;
	int n = w->n;
	int ret = 0;
	*info = no_problems_encountered;

	switch (w->section) {
	case 0:
		break;
	case 1:
		oracle_take_step(step, n, x, g, b, 0, NULL, 0, NULL, 0, w->check_nan, info);   /* :1055 */
		w->niter++;
		ret = (*info == search_direction_was_nan) ? 0 : 1;
		for (size_t i = 0; i < (size_t) n; i++) w->x_sum[i] += x[i];                   /* :1067 */
		if (w->niter % b->upd_freq != 0) break;
		if (w->niter == b->upd_freq) {                                                 /* :1078-1094 */
			sum_to_average(w->x_sum, b->upd_freq, n);
			archive_average(w->x_sum, w->x_avg_prev, n);
			if (!w->use_grad_diff) break;
			*task = calc_grad_big_batch; *req = w->x_avg_prev; w->section = 2;
			return ret;
		}
		make_s(w->x_sum, w->x_avg_prev, n, 1, b);                                      /* :1097 */
		*req = w->x_sum;
		if (w->use_grad_diff) { *task = calc_grad_big_batch; w->section = 3; }
		else { *task = calc_hess_vec; w->section = 4; *req_vec = b->s_mem + (size_t) n * b->mem_st_ix; }
		return ret;
	case 2:
		v_copy(n, g, w->grad_prev);                                                    /* :1120 */
		break;
	case 3:
		make_y_graddiff(g, w->grad_prev, b, n, info);                                  /* :1127-1133 */
		if (*info == no_problems_encountered) {
			v_copy(n, g, w->grad_prev);
			v_copy(n, w->x_sum, w->x_avg_prev);
		}
		v_zero(n, w->x_sum);
		break;
	case 4:
		archive_average(w->x_sum, w->x_avg_prev, n);                                   /* :1139 */
		make_y_hessvec(hv, b, info, n);                                                /* :1140 */
		break;
	default:
		*task = invalid_input;
		fprintf(stderr, "SQN got an invalid workspace as input.\n");
		return -1000;
	}
	w->section = 1; *task = calc_grad; *req = x;                                       /* :1148-1152 */
	return ret;
}

/* adaQN (src/stochqn.c:1155-1315) */
int oracle_run_adaQN(real_t step, real_t *x, real_t f, real_t *g, real_t **req, task_enum *task,
                     workspace_adaQN *w, info_enum *info)
{
	bfgs_mem *b = w->bfgs_memory;
	int n = w->n;
	int ret = 0;
	int build_y = 0;
	*info = no_problems_encountered;

	switch (w->section) {
	case 0:
		break;
	case 1:
		/* :1174.  R/Python hand over a 1-element F together with use_grad_diff (the reference then
		 * writes n words into it); the append is skipped in that mode. */
		if (!w->use_grad_diff) fisher_append(g, w->fisher_memory, n);
		oracle_take_step(step, n, x, g, b, w->rmsprop_weight, w->H0, 0, w->grad_sum_sq,
		                 w->scal_reg, w->check_nan, info);                             /* :1177 */
		ret = (*info == search_direction_was_nan) ? 0 : 1;
		w->niter++;
		for (size_t i = 0; i < (size_t) n; i++) w->x_sum[i] += x[i];                   /* :1191 */
		if (w->niter % b->upd_freq != 0) break;
		if (w->niter == b->upd_freq) {                                                 /* :1206-1224 */
			sum_to_average(w->x_sum, b->upd_freq, n);
			archive_average(w->x_sum, w->x_avg_prev, n);
			if (w->use_grad_diff) { *task = calc_grad_big_batch; *req = w->x_avg_prev; w->section = 2; return ret; }
			if (w->max_incr > 0)  { *task = calc_fun_val_batch;  *req = w->x_avg_prev; w->section = 3; return ret; }
			break;
		}
		if (w->max_incr > 0) {                                                         /* :1227-1234 */
			sum_to_average(w->x_sum, b->upd_freq, n);
			*task = calc_fun_val_batch; *req = w->x_sum; w->section = 5;
			return ret;
		}
		make_s(w->x_sum, w->x_avg_prev, n, 1, b);                                      /* :1237 */
		build_y = 1;
		break;
	case 2:
		v_copy(n, g, w->grad_prev);                                                    /* :1244 */
		if (w->max_incr) { *task = calc_fun_val_batch; *req = w->x_avg_prev; w->section = 3; return 0; }
		break;
	case 3:
		w->f_prev = f;                                                                 /* :1260 */
		break;
	case 4:
		make_y_graddiff(g, w->grad_prev, b, n, info);                                  /* :1266-1268 */
		if (*info == no_problems_encountered) v_copy(n, g, w->grad_prev);
		v_zero(n, w->x_sum);
		break;
	case 5:
		if (f > w->max_incr * w->f_prev || isinf(f) || isnan(f)) {                     /* :1275-1283 */
			ring_reset(b);
			fisher_reset(w->fisher_memory);
			v_copy(n, w->x_avg_prev, x);
			*info = func_increased;
			ret = 1;
			break;
		}
		w->f_prev = f;                                                                 /* :1287-1289 */
		make_s(w->x_sum, w->x_avg_prev, n, 0, b);
		build_y = 1;
		break;
	default:
		*task = invalid_input;
		fprintf(stderr, "adaQN got an invalid workspace as input.\n");
		return -1000;
	}

	if (build_y) {                                                                     /* :1297-1308 */
		if (w->use_grad_diff) { *req = w->x_sum; *task = calc_grad_big_batch; w->section = 4; return ret; }
		make_y_fisher(w->fisher_memory, b, n, info);
		if (*info == no_problems_encountered) v_copy(n, w->x_sum, w->x_avg_prev);
		v_zero(n, w->x_sum);
	}
	w->section = 1; *task = calc_grad; *req = x;                                       /* :1310-1314 */
	return ret;
}

#ifdef __cplusplus
}
#endif
