/* tools/c5_host_caller.c -- BASELINE config 5 as the reference's own kind of caller sees it: ONE plain C
 * process, the unchanged ABI of include/stochqn.h (initialize_SQN / run_SQN / dealloc_SQN, reference
 * example/c_rosen.c:100-125), x / grad / hess_vec in ordinary host memory, *req and *req_vec read on the host.
 * With STOCHQN_HIP_DEVICES=P in the environment the library shards the workspace over P GPUs (group.cpp):
 *
 *   gcc -O2 -fopenmp -std=c99 -I include tools/c5_host_caller.c -L stochqn_amd/lib -lstochqn -lm -o c5_host_caller
 *   STOCHQN_HIP_DEVICES=8 ./c5_host_caller 1000000000 20 30        # n = 1e9, m = 20: 320 GB of S and Y, 40 GB per GPU
 *   ./c5_host_caller 200000000 20 64 2                              # optional 4th argument: L (default 10)
 *
 * Problem: f(x) = 1/2 sum d_i x_i^2, gradient d.x with a deterministic +-1 % ripple, Hessian-vector product
 * d.v, all computed on the host with OpenMP (the caller's business; timed separately).  Prints the time
 * spent inside run_SQN per step -- PCIe-inclusive by construction: every step moves x and grad up and x and
 * the direction down (4 n words), split over the P devices' links. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "stochqn.h"
#include "stochqn_hip.h"

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static double d_of(long i) { return 0.5 + (double) ((i * 2654435761u) % 1000003) / 1000003.0; }

int main(int argc, char **argv)
{
	const long n = argc > 1 ? atol(argv[1]) : 1000000000L;
	const size_t m = argc > 2 ? (size_t) atol(argv[2]) : 20;
	const int steps = argc > 3 ? atoi(argv[3]) : 30;
	const size_t L = argc > 4 ? (size_t) atol(argv[4]) : 10;      /* a small L fills the ring quickly: the steady state of a full ring within a short run */
	if (n <= 0 || n > 2147483647L) { fprintf(stderr, "n must fit an int (reference include/stochqn.h:172-174)\n"); return 2; }
	double *x = malloc((size_t) n * sizeof(double)), *grad = malloc((size_t) n * sizeof(double)), *hv = malloc((size_t) n * sizeof(double));
	if (!x || !grad || !hv) { fprintf(stderr, "host memory\n"); return 2; }
	/* first touch by this one thread: the pages of x / grad / hv then sit on one NUMA node, like the arrays of an R or numpy
	 * caller.  (Touched by the OpenMP team of an unpinned process on a two-socket host they end up spread over both sockets,
	 * and the PCIe copies of the half behind the inter-socket link ran at less than half the speed: 97 instead of 57 ms per
	 * step at n = 2e8, profiles/r03_c5_host_caller_one_device.log.) */
	for (long i = 0; i < n; i++) { x[i] = 1.0 + (double) (i % 97) / 97.0; grad[i] = 0; hv[i] = 0; }
	/* this program owns the three arrays until it exits: it pins them (the library pins nothing behind a caller's back) */
	const int pinned = (stochqn_hip_pin_host(x, (size_t) n * sizeof(double)) == 0) + (stochqn_hip_pin_host(grad, (size_t) n * sizeof(double)) == 0) +
	                   (stochqn_hip_pin_host(hv, (size_t) n * sizeof(double)) == 0);
	printf("arrays pinned by their owner: %d of 3\n", pinned);

	double t0 = now();
	workspace_SQN *w = initialize_SQN((int) n, m, L, 0.0, 0, 0.0, 1, 1);
	if (!w) { fprintf(stderr, "initialize_SQN(n = %ld, m = %zu) failed\n", n, m); return 3; }
	printf("initialize_SQN(n = %ld, m = %zu): %.2f s\n", n, m, now() - t0);

	double *req = NULL, *req_vec = NULL;
	task_enum task;
	info_enum info;
	double t_lib = 0, t_caller = 0, f0 = 0, f1 = 0, t_first = 0, t_min = 1e30, t_full_min = 1e30, t_full_sum = 0;
	int full_steps = 0;
	#pragma omp parallel for reduction(+ : f0)
	for (long i = 0; i < n; i++) f0 += 0.5 * d_of(i) * x[i] * x[i];
	run_SQN(0.05, x, grad, hv, &req, &req_vec, &task, w, &info);
	int calls = 0, hvs = 0, bad = 0;
	while (w->niter < (size_t) steps) {
		double tc = now();
		if (task == calc_grad) {
			const long ripple = (long) w->niter;
			#pragma omp parallel for
			for (long i = 0; i < n; i++) grad[i] = d_of(i) * req[i] * (1.0 + 0.01 * (double) (((i + ripple) % 21) - 10) / 10.0);
		} else if (task == calc_hess_vec) {
			hvs++;
			#pragma omp parallel for
			for (long i = 0; i < n; i++) hv[i] = d_of(i) * req_vec[i];
		} else { fprintf(stderr, "unexpected task %d\n", (int) task); return 4; }
		double tl = now();
		t_caller += tl - tc;
		int rc = run_SQN(0.05, x, grad, hv, &req, &req_vec, &task, w, &info);
		const double dt = now() - tl;
		t_lib += dt;
		if (getenv("C5_VERBOSE")) printf("  call %3d niter %3zu next task %d: %.1f ms\n", calls, w->niter, (int) task, 1e3 * dt);
		if (calls < 2) t_first += dt;                 /* one-time work: the caller's vectors are pinned, staging is allocated */
		else if (task == calc_grad && dt < t_min) t_min = dt;
		if (calls >= 2 && task == calc_grad && w->bfgs_memory->mem_used == m && rc == 1) {   /* ordinary steps on a full ring */
			full_steps++; t_full_sum += dt;
			if (dt < t_full_min) t_full_min = dt;
		}
		if (rc != 0 && rc != 1) { fprintf(stderr, "run_SQN returned %d\n", rc); return 5; }
		if (info != no_problems_encountered) bad++;
		calls++;
	}
	#pragma omp parallel for reduction(+ : f1)
	for (long i = 0; i < n; i++) f1 += 0.5 * d_of(i) * x[i] * x[i];
	printf("n %ld m %zu steps %zu calls %d hess_vec %d flagged %d mem_used %zu  f %.6e -> %.6e\n", n, m, w->niter, calls, hvs, bad,
	       w->bfgs_memory->mem_used, f0, f1);
	printf("inside run_SQN: %.3f s = %.1f ms per step (PCIe-inclusive); caller's own gradient / Hv loops: %.3f s\n", t_lib,
	       1e3 * t_lib / (double) w->niter, t_caller);
	printf("  of which the first two calls (pinning the caller's vectors, staging): %.3f s; after them %.1f ms per step, fastest ordinary step %.1f ms\n",
	       t_first, 1e3 * (t_lib - t_first) / (double) (w->niter > 2 ? w->niter - 2 : 1), 1e3 * t_min);
	if (full_steps) printf("  ordinary steps with the ring full (%zu pairs): %d, fastest %.1f ms, mean %.1f ms\n", m, full_steps, 1e3 * t_full_min, 1e3 * t_full_sum / full_steps);
	printf("  x uploads %lld, skipped %lld; host ranges pinned %lld; steps by form: three-pass %lld, sweeps %lld, no pairs yet %lld\n",
	       stochqn_hip_stat("x_uploads"), stochqn_hip_stat("x_uploads_skipped"), stochqn_hip_stat("host_ranges_registered"),
	       stochqn_hip_stat("steps_three_pass"), stochqn_hip_stat("steps_sweeps"), stochqn_hip_stat("steps_plain"));
	dealloc_SQN(w);
	stochqn_hip_unpin_host(x); stochqn_hip_unpin_host(grad); stochqn_hip_unpin_host(hv);      /* before the arrays go */
	free(x); free(grad); free(hv);
	return (f1 < f0 && bad == 0) ? 0 : 6;
}
