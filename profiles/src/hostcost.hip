// profiles/src/hostcost.hip -- cost of the host-side pieces of one API call (not product)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_write(double* out, double v) { if (threadIdx.x == 0) { out[0] = v; out[1] = v + 1; out[2] = v + 2; } }
int main()
{
	hipStream_t st; CK(hipStreamCreate(&st));
	double *dev, *pin; CK(hipMalloc(&dev, 4096)); CK(hipHostMalloc(&pin, 4096, hipHostMallocDefault));
	void* host = malloc(4096);
	const int R = 2000;
	hipPointerAttribute_t a;
	double t0 = now();
	for (int i = 0; i < R; i++) { (void) hipPointerGetAttributes(&a, dev); }
	printf("hipPointerGetAttributes(device ptr): %.2f us\n", 1e6 * (now() - t0) / R);
	t0 = now();
	for (int i = 0; i < R; i++) { if (hipPointerGetAttributes(&a, host) != hipSuccess) (void) hipGetLastError(); }
	printf("hipPointerGetAttributes(malloc ptr): %.2f us\n", 1e6 * (now() - t0) / R);
	for (int w = 0; w < 2; w++) {
		t0 = now();
		for (int i = 0; i < R; i++) { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, (double) i); CK(hipStreamSynchronize(st)); }
		printf("kernel + sync: %.2f us\n", 1e6 * (now() - t0) / R);
		t0 = now();
		for (int i = 0; i < R; i++) { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, (double) i); CK(hipMemcpyAsync(pin, dev, 64, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
		printf("kernel + D2H(64 B, pinned) + sync: %.2f us (last %.0f)\n", 1e6 * (now() - t0) / R, pin[0]);
		t0 = now();
		for (int i = 0; i < R; i++) { hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, pin, (double) i); CK(hipStreamSynchronize(st)); }
		printf("kernel writing host-mapped + sync: %.2f us (last %.0f)\n", 1e6 * (now() - t0) / R, pin[0]);
		t0 = now();
		for (int i = 0; i < R; i++) { for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, (double) i); CK(hipStreamSynchronize(st)); }
		printf("4 kernels + sync: %.2f us\n", 1e6 * (now() - t0) / R);
		hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		t0 = now();
		for (int i = 0; i < R; i++) { for (int k = 0; k < 4; k++) { CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, (double) i); CK(hipEventRecord(e1, st)); } CK(hipStreamSynchronize(st)); }
		printf("4 x (event, kernel, event) + sync: %.2f us\n", 1e6 * (now() - t0) / R);
		t0 = now();
		float acc_ms = 0;
		for (int i = 0; i < R; i++) { for (int k = 0; k < 4; k++) hipExtLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, e0, e1, 0, dev, (double) i); CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); acc_ms += ms; }
		printf("4 x hipExtLaunchKernelGGL(start, stop events) + sync: %.2f us (kernel %.2f us by its events)\n", 1e6 * (now() - t0) / R, 1e3 * acc_ms / R);
	}
	// The same four dependent kernels (+ the 64-byte read-back) as ONE captured graph, replayed: what a graph of
	// the default step's chain would cost per call.
	for (int with_copy = 0; with_copy < 2; with_copy++) {
		hipGraph_t g; hipGraphExec_t ge;
		CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
		for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, 1.0 + k);
		if (with_copy) CK(hipMemcpyAsync(pin, dev, 64, hipMemcpyDeviceToHost, st));
		CK(hipStreamEndCapture(st, &g));
		CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		for (int w = 0; w < 2; w++) {
			t0 = now();
			for (int i = 0; i < R; i++) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
			printf("graph of 4 kernels%s, launch + sync: %.2f us\n", with_copy ? " + D2H(64 B)" : "", 1e6 * (now() - t0) / R);
		}
		// stream-ordered (no wait per call): what async_device would see
		t0 = now();
		for (int i = 0; i < R; i++) CK(hipGraphLaunch(ge, st));
		CK(hipStreamSynchronize(st));
		printf("graph of 4 kernels%s, %d launches then one sync: %.2f us each\n", with_copy ? " + D2H(64 B)" : "", R, 1e6 * (now() - t0) / R);
		CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
	}
	t0 = now();
	for (int i = 0; i < R; i++) for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, st, dev, (double) i);
	CK(hipStreamSynchronize(st));
	printf("4 kernels, %d rounds then one sync: %.2f us per round\n", R, 1e6 * (now() - t0) / R);
	return 0;
}
