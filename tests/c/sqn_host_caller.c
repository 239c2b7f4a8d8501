/* A plain C99 program against include/stochqn.h + libstochqn.so: library-owned SQN workspace
 * (initialize_SQN / run_SQN / dealloc_SQN), caller vectors in ordinary host memory, *req and
 * *req_vec dereferenced on the host -- the calling convention of the reference's C example, on a
 * different problem (chained quadratic + quartic, analytic gradient and Hessian-vector product).
 * Prints every 20th iterate and the final state with 17 significant digits; tests/test_c_callers.py
 * compares the numbers with the CPU oracle driven through the same protocol.
 *
 *   gcc -std=c99 -I include tests/c/sqn_host_caller.c -L stochqn_amd/lib -lstochqn -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "stochqn.h"

#define N 12

static void gradient(const double *x, double *g)
{
	for (int i = 0; i < N; i++) {
		double d = 1.0 + 0.5 * i;
		g[i] = d * x[i] + 0.1 * x[i] * x[i] * x[i];
		if (i > 0) g[i] += 0.25 * (x[i] - x[i - 1]);
		if (i < N - 1) g[i] += 0.25 * (x[i] - x[i + 1]);
	}
}

static void hess_vec(const double *x, const double *v, double *hv)
{
	for (int i = 0; i < N; i++) {
		double d = 1.0 + 0.5 * i;
		hv[i] = (d + 0.3 * x[i] * x[i]) * v[i];
		if (i > 0) hv[i] += 0.25 * (v[i] - v[i - 1]);
		if (i < N - 1) hv[i] += 0.25 * (v[i] - v[i + 1]);
	}
}

int main(void)
{
	double x[N], grad[N], hv[N];
	double *req = NULL, *req_vec = NULL;
	task_enum task;
	info_enum info;
	for (int i = 0; i < N; i++) x[i] = 1.0 + 0.1 * i;

	workspace_SQN *w = initialize_SQN(N, 4, 3, 1e-6, 0, 0.0, 1, 1);
	if (w == NULL) { fprintf(stderr, "initialize_SQN failed\n"); return 2; }
	run_SQN(0.05, x, grad, hv, &req, &req_vec, &task, w, &info);
	int n_hv = 0, n_info = 0;
	while (w->niter < 120) {
		if (task == calc_grad) gradient(req, grad);
		else if (task == calc_hess_vec) { hess_vec(req, req_vec, hv); n_hv++; }
		else { fprintf(stderr, "unexpected task %d\n", (int) task); return 3; }
		int changed = run_SQN(0.05, x, grad, hv, &req, &req_vec, &task, w, &info);
		if (info != no_problems_encountered) n_info++;
		if (changed && w->niter % 20 == 0) {
			printf("iter %zu", w->niter);
			for (int i = 0; i < N; i++) printf(" %.17g", x[i]);
			printf("\n");
		}
	}
	printf("final niter %zu mem_used %zu mem_st_ix %zu hv %d info %d\n", w->niter, w->bfgs_memory->mem_used,
	       w->bfgs_memory->mem_st_ix, n_hv, n_info);
	dealloc_SQN(w);
	return 0;
}
