/*  stochqn.h -- free-mode C ABI of the MI355X-native stochastic quasi-Newton step library.
 *
 *  This header is the drop-in boundary.  Every type, enumerator value and prototype below is
 *  ABI-identical (x86-64 LP64) to david-cortes/stochQN's public header, so that the reference's
 *  own callers -- the R `.Call` shim (reference src/Rwrapper.c:9-229), the Cython shim
 *  (reference stochqn/pywrapper.pxi:1-207), the C and C++ examples (reference
 *  example/c_rosen.c:100-125, example/cpp_rosen.cpp) -- compile and link against
 *  `libstochqn.so` built from this repository without modification.
 *
 *  Interface replaced                                   reference location
 *  ---------------------------------------------------  ------------------------------
 *  real_t / precision switch                            include/stochqn.h:62-76
 *  bfgs_mem, fisher_mem                                 include/stochqn.h:86-107
 *  workspace_oLBFGS / workspace_SQN / workspace_adaQN   include/stochqn.h:109-151
 *  initialize_* / dealloc_*                             include/stochqn.h:227-238
 *  task_enum / info_enum / iter_status                  include/stochqn.h:268-291
 *  run_oLBFGS / run_SQN / run_adaQN                     include/stochqn.h:381-383
 *  C++ RAII wrappers oLBFGS / SQN / adaQN               include/stochqn.h:397-511
 *
 *  What is different underneath: all vector arithmetic runs as hand-written HIP kernels on an
 *  AMD MI355X (gfx950).  Array pointers handed to the library -- both the ones inside the
 *  structs and the per-call `x`, `grad`, `hess_vec` -- may be EITHER ordinary host pointers
 *  (what R, numpy and the C examples pass) OR device pointers (hipMalloc / torch tensors).
 *  Host arrays are mirrored in device memory behind the ABI; device arrays are used in place.
 *  See DESIGN.md ("boundary") and INTEGRATION.md.  Device-only extras live in stochqn_hip.h.
 *
 *  libstochqn.so is the double-precision library (the R package is double-only, reference
 *  src/Makevars:1; every BASELINE configuration is fp64).  Compiling a caller with -DUSE_FLOAT selects
 *  the single-precision ABI, which libstochqn_f32.so implements: vectors are stored and streamed as
 *  float, every inner product and scalar of the recursion is accumulated in double.
 */
#ifndef STOCHQN_INCLUDE
#define STOCHQN_INCLUDE

#include <stddef.h>

/* Precision switch of reference include/stochqn.h:62-76.  Symbol names are the same in both
 * precisions; as with the reference, a float caller links its own copy: libstochqn_f32.so. */
#if defined(USE_FLOAT) && !defined(USE_DOUBLE)
#   define real_t float
#else
#   define real_t double
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * State containers.  Field order, types and names are part of the ABI: the Cython shim
 * re-declares every field, C callers read `->niter`, the R shim fills the structs on its
 * stack on every call.  sizeof: bfgs_mem 96, fisher_mem 40, workspace_oLBFGS 48,
 * workspace_SQN 64, workspace_adaQN 120 (checked by static assertions in the library).
 * ------------------------------------------------------------------------------------------ */

/* Ring buffer of `mem_size` correction pairs; pair k occupies [k*n, (k+1)*n) of each array. */
typedef struct {
	real_t *s_mem;          /* [mem_size][n] position differences                          */
	real_t *y_mem;          /* [mem_size][n] gradient differences / H*s / Fisher*s          */
	real_t *buffer_rho;     /* [mem_size] 1/(y's) of the last two-loop, logical order       */
	real_t *buffer_alpha;   /* [mem_size] alpha of the last two-loop, logical order         */
	real_t *s_bak;          /* [n], only when min_curvature > 0                             */
	real_t *y_bak;          /* [n], only when min_curvature > 0                             */
	size_t mem_size;
	size_t mem_used;
	size_t mem_st_ix;       /* slot the next pair is written to (= oldest once full)        */
	size_t upd_freq;        /* L: iterations between pair updates (1 for oLBFGS)            */
	real_t y_reg;           /* lambda: y += lambda * s (gradient-difference pairs)          */
	real_t min_curvature;   /* reject pair when s'y / s's <= this (0 = accept everything)   */
} bfgs_mem;

/* Ring buffer of the last `mem_size` raw gradients (adaQN empirical Fisher). */
typedef struct {
	real_t *F;              /* [mem_size][n]                                                */
	real_t *buffer_y;       /* [mem_size] F*s of the last Fisher product                    */
	size_t mem_size;
	size_t mem_used;
	size_t mem_st_ix;
} fisher_mem;

typedef struct {
	bfgs_mem *bfgs_memory;
	real_t *grad_prev;      /* [n] */
	real_t hess_init;       /* H0 = hess_init * I when > 0, else s'y/y'y of newest pair     */
	size_t niter;
	int section;            /* position in the reverse-communication cycle; do not touch    */
	int nthreads;           /* accepted for ABI compatibility; the GPU path ignores it       */
	int check_nan;
	int n;
} workspace_oLBFGS;

typedef struct {
	bfgs_mem *bfgs_memory;
	real_t *grad_prev;      /* [n], only with use_grad_diff */
	real_t *x_sum;          /* [n] running sum of iterates; holds the average after /L      */
	real_t *x_avg_prev;     /* [n] */
	int use_grad_diff;
	size_t niter;
	int section;
	int nthreads;
	int check_nan;
	int n;
} workspace_SQN;

typedef struct {
	bfgs_mem *bfgs_memory;
	fisher_mem *fisher_memory;  /* NULL with use_grad_diff */
	real_t *H0;                 /* [n] */
	real_t *grad_prev;          /* [n], only with use_grad_diff */
	real_t *x_sum;              /* [n] */
	real_t *x_avg_prev;         /* [n] */
	real_t *grad_sum_sq;        /* [n] AdaGrad sum / RMSProp moving average of g*g */
	real_t f_prev;
	real_t max_incr;
	real_t scal_reg;
	real_t rmsprop_weight;
	int use_grad_diff;
	size_t niter;
	int section;
	int nthreads;
	int check_nan;
	int n;
} workspace_adaQN;

/* ------------------------------------------------------------------------------------------
 * Library-owned workspaces.  `initialize_*` places every n-sized array in device memory
 * (HBM) and the struct itself in host memory, so `->niter`, `->bfgs_memory->mem_used` etc.
 * stay readable exactly as with the reference.  Returns NULL (after a message on stderr)
 * when no usable HIP device exists or an allocation fails.  Argument meaning and defaults:
 * reference include/stochqn.h:170-226.
 * ------------------------------------------------------------------------------------------ */
workspace_oLBFGS* initialize_oLBFGS(const int n, const size_t mem_size, const real_t hess_init,
	const real_t y_reg, const real_t min_curvature, const int check_nan, const int nthreads);
void dealloc_oLBFGS(workspace_oLBFGS *oLBFGS);

workspace_SQN* initialize_SQN(const int n, const size_t mem_size, const size_t bfgs_upd_freq,
	const real_t min_curvature, const int use_grad_diff, const real_t y_reg,
	const int check_nan, const int nthreads);
void dealloc_SQN(workspace_SQN *SQN);

workspace_adaQN* initialize_adaQN(const int n, const size_t mem_size, const size_t fisher_size,
	const size_t bfgs_upd_freq, const real_t max_incr, const real_t min_curvature,
	const real_t scal_reg, const real_t rmsprop_weight, const int use_grad_diff,
	const real_t y_reg, const int check_nan, const int nthreads);
void dealloc_adaQN(workspace_adaQN *adaQN);

/* Same symbols the reference exports without declaring (reference src/stochqn.c:300,331,342,355). */
bfgs_mem* initialize_bfgs_mem(const size_t mem_size, const int n, const real_t min_curvature,
	const real_t y_reg, const size_t upd_freq);
void dealloc_bfgs_mem(bfgs_mem *bfgs_memory);
fisher_mem* initialize_fisher_mem(const size_t mem_size, const int n);
void dealloc_fisher_mem(fisher_mem *fisher_memory);

/* What the caller has to compute before calling run_* again (evaluated at *req). */
typedef enum task_enum {
	calc_grad = 101,
	calc_grad_same_batch = 102,
	calc_grad_big_batch = 103,
	calc_hess_vec = 104,
	calc_fun_val_batch = 105,
	invalid_input = 100
} task_enum;

/* What, if anything, went wrong inside the call. */
typedef enum info_enum {
	func_increased = 201,
	curvature_too_small = 202,
	search_direction_was_nan = 203,
	no_problems_encountered = 200
} info_enum;

/* Meaning of the int returned by run_*. */
typedef enum iter_status {did_not_update_x = 0, updated_x = 1, received_invalid_input = -1000} iter_status;

/* ------------------------------------------------------------------------------------------
 * Reverse-communication step functions (reference include/stochqn.h:293-383).
 *
 *   x, grad, hess_vec : host or device pointers to n doubles.  `x` is updated in place;
 *                       `grad` is overwritten with the search direction (oLBFGS: with
 *                       s = -step*direction), as in the reference.
 *   *req, *req_vec    : where the next quantity must be evaluated.  They are host-readable
 *                       when `x` is a host pointer and device pointers when `x` is a device
 *                       pointer.  Never write through them.
 *   All results are visible when the call returns (the library synchronises its stream).
 * ------------------------------------------------------------------------------------------ */
int run_oLBFGS(real_t step_size, real_t x[], real_t grad[], real_t **req, task_enum *task,
	workspace_oLBFGS *oLBFGS, info_enum *iter_info);
int run_SQN(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[], real_t **req,
	real_t **req_vec, task_enum *task, workspace_SQN *SQN, info_enum *iter_info);
int run_adaQN(real_t step_size, real_t x[], real_t f, real_t grad[], real_t **req,
	task_enum *task, workspace_adaQN *adaQN, info_enum *iter_info);

#ifdef __cplusplus
}
#endif


/* ------------------------------------------------------------------------------------------
 * C++ RAII front-ends, source-compatible with reference include/stochqn.h:397-511
 * (class names, public members, constructor defaults, run()/get_*() methods).
 * ------------------------------------------------------------------------------------------ */
#ifdef __cplusplus
#include <new>

namespace stochqn_detail {
/* Shared bookkeeping of the three front-ends: owns one workspace, frees it on scope exit. */
template <class Workspace, void (*Release)(Workspace*)>
struct owner {
	Workspace *workspace;
	task_enum task;
	info_enum info;
	iter_status status;
	real_t *req;

	explicit owner(Workspace *w)
		: workspace(w), task(calc_grad), info(no_problems_encountered), status(did_not_update_x), req(NULL)
	{
		if (w == NULL) throw std::bad_alloc();
	}
	~owner() { if (workspace != NULL) Release(workspace); }

	task_enum get_task()      { return task; }
	info_enum get_iter_info() { return info; }
	size_t    get_n_iter()    { return workspace->niter; }
	real_t*   get_req()       { return req; }

private:
	owner(const owner&);
	owner& operator=(const owner&);
};
}

class oLBFGS : public stochqn_detail::owner<workspace_oLBFGS, dealloc_oLBFGS>
{
public:
	oLBFGS(const int n, const size_t mem_size = 10, const real_t hess_init = 0, const real_t y_reg = 0,
		   const real_t min_curvature = 0, const int check_nan = 1, const int nthreads = 1)
		: owner(initialize_oLBFGS(n, mem_size, hess_init, y_reg, min_curvature, check_nan, nthreads)) {}

	iter_status run(real_t step_size, real_t x[], real_t grad[])
	{
		return (iter_status) run_oLBFGS(step_size, x, grad, &req, &task, workspace, &info);
	}
};

class SQN : public stochqn_detail::owner<workspace_SQN, dealloc_SQN>
{
public:
	real_t *req_vec;

	SQN(const int n, const size_t mem_size = 10, const size_t bfgs_upd_freq = 10,
		const real_t min_curvature = 1e-4, const int use_grad_diff = 0, const real_t y_reg = 0,
		const int check_nan = 1, const int nthreads = 1)
		: owner(initialize_SQN(n, mem_size, bfgs_upd_freq, min_curvature, use_grad_diff, y_reg, check_nan, nthreads)),
		  req_vec(NULL) {}

	iter_status run(real_t step_size, real_t x[], real_t grad[], real_t hess_vec[])
	{
		return (iter_status) run_SQN(step_size, x, grad, hess_vec, &req, &req_vec, &task, workspace, &info);
	}
	real_t* get_req_vec() { return req_vec; }
};

class adaQN : public stochqn_detail::owner<workspace_adaQN, dealloc_adaQN>
{
public:
	adaQN(const int n, const size_t mem_size = 10, const size_t fisher_size = 100,
		  const size_t bfgs_upd_freq = 10, const real_t max_incr = 1.01, const real_t min_curvature = 1e-4,
		  const real_t scal_reg = 1e-4, const real_t rmsprop_weight = 0.9, const int use_grad_diff = 0,
		  const real_t y_reg = 0, const int check_nan = 1, const int nthreads = 1)
		: owner(initialize_adaQN(n, mem_size, fisher_size, bfgs_upd_freq, max_incr, min_curvature, scal_reg,
								 rmsprop_weight, use_grad_diff, y_reg, check_nan, nthreads)) {}

	iter_status run(real_t step_size, real_t x[], real_t f, real_t grad[])
	{
		return (iter_status) run_adaQN(step_size, x, f, grad, &req, &task, workspace, &info);
	}
};

#endif /* __cplusplus */

#endif /* STOCHQN_INCLUDE */
