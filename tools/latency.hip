// tools/latency.hip -- per-step wall time of the C ABI with a device-resident caller, as a function of n
// (the fixed per-call cost dominates below n ~ 1e6).  Driven by tools/latency_table.py.
// build: hipcc --offload-arch=gfx950 -O2 -I include tools/latency.hip -L stochqn_amd/lib -lstochqn -Wl,-rpath,$PWD/stochqn_amd/lib -o tools/latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "stochqn.h"
#include "stochqn_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_grad(const double* d, const double* x, double* g, int n, double noise)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) g[i] = d[i] * x[i] * (1.0 + noise * ((i * 2654435761u >> 8 & 1023) / 1024.0 - 0.5));
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
	const char* kind = argc > 1 ? argv[1] : "olbfgs";
	const int n = argc > 2 ? (int) atof(argv[2]) : 1000;
	const int m = argc > 3 ? atoi(argv[3]) : 10;
	const int steps = argc > 4 ? atoi(argv[4]) : 500;
	std::vector<double> hd(n), hx(n);
	for (int i = 0; i < n; i++) { hd[i] = 0.5 + (i % 97) / 97.0; hx[i] = 1.0 + (i % 89) / 89.0; }
	double *d, *x, *g, *hv;
	CK(hipMalloc(&d, n * 8.0)); CK(hipMalloc(&x, n * 8.0)); CK(hipMalloc(&g, n * 8.0)); CK(hipMalloc(&hv, n * 8.0));
	CK(hipMemcpy(d, hd.data(), n * 8.0, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), n * 8.0, hipMemcpyHostToDevice));
	double* req = nullptr; double* req_vec = nullptr; task_enum task; info_enum info;
	const bool sqn = kind[0] == 's';
	if (argc > 6) stochqn_hip_set_option("threepass", atof(argv[6]));      // 0: the two-pass form (A/B of the per-call cost)
	// argv[7]: 0 = the synchronous ABI with the guard (default); 1 = check_nan = 0, still synchronous; 2 = check_nan = 0 and
	// option "async_device" (stream-ordered calls: nothing can be rejected, so nothing is waited for)
	const int mode = argc > 7 ? atoi(argv[7]) : 0;
	// 3 = the synchronous ABI with the guard, on the NULL stream (option "null_stream")
	if (mode == 2) stochqn_hip_set_option("async_device", 1);
	if (mode == 3) stochqn_hip_set_option("null_stream", 1);
	if (mode == 4) stochqn_hip_set_option("null_stream", 0);     // the context's own stream, as in rounds 1-2
	const int check_nan = (mode == 0 || mode == 3 || mode == 4) ? 1 : 0;
	workspace_oLBFGS* wo = sqn ? nullptr : initialize_oLBFGS(n, m, 0, 0, 0, check_nan, 1);
	workspace_SQN* ws = sqn ? initialize_SQN(n, m, 10, 0, 0, 0, check_nan, 1) : nullptr;
	auto iter = [&](long& calls) {
		const size_t target = (sqn ? ws->niter : wo->niter) + 1;
		while ((sqn ? ws->niter : wo->niter) < target) {
			if (sqn) run_SQN(0.01, x, g, hv, &req, &req_vec, &task, ws, &info); else run_oLBFGS(0.01, x, g, &req, &task, wo, &info);
			calls++;
			if (task == calc_grad || task == calc_grad_same_batch) hipLaunchKernelGGL(k_grad, dim3((n + 255) / 256), dim3(256), 0, 0, d, req, g, n, task == calc_grad ? 0.01 : 0.0);
			else if (task == calc_hess_vec) hipLaunchKernelGGL(k_grad, dim3((n + 255) / 256), dim3(256), 0, 0, d, req_vec, hv, n, 0.0);
		}
	};
	long calls = 0;
	for (int i = 0; i < 3 * m + 25; i++) iter(calls);
	CK(hipDeviceSynchronize());
	const bool prof = !(argc > 5 && atoi(argv[5]) == 0);
	stochqn_hip_profile_enable(prof ? 1 : 0); stochqn_hip_profile_reset();
	calls = 0;
	const double t0 = now();
	for (int i = 0; i < steps; i++) iter(calls);
	CK(hipDeviceSynchronize());
	const double dt = now() - t0;
	stochqn_hip_profile_enable(0);
	double ksum = 0;
	printf("%s n=%d m=%d mode=%d: %.1f us/step, %.2f calls/step;", kind, n, m, mode, 1e6 * dt / steps, (double) calls / steps);
	for (int i = 0; i < stochqn_hip_profile_kernels(); i++) {
		long long cnt; double ms;
		stochqn_hip_profile_get(i, &cnt, &ms);
		if (cnt) { printf(" %s %.1fx%.1fus", stochqn_hip_profile_name(i), (double) cnt / steps, 1e3 * ms / cnt); ksum += ms; }
	}
	printf("; kernels %.1f us/step\n", 1e3 * ksum / steps);
	return 0;
}
