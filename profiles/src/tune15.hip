// profiles/src/tune15.hip -- round 5: what does the 64-byte read-back at the end of every synchronous call cost?  A call of the
// library ends with [last kernel] -> hipMemcpyAsync(pinned <- device, 64 B) -> hipStreamSynchronize.  The alternative: the last
// kernel writes its report straight into host-mapped pinned memory (hipHostMalloc memory is device-accessible at its own
// address) and the call only waits for the kernel.  Measured like tune13's empty calls -- and behind REAL kernels (a 1.2 GB
// read + 80 MB write, the shape of a C2 pass), where the host reaches the wait long before the device is done.
// hipcc --offload-arch=gfx950 -O3 tune15.hip -o tune15
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k_empty(double* p) { if (p == nullptr) p[0] = 0; }
__global__ void k_report(double* dst, double v) { if (threadIdx.x < 8 && blockIdx.x == 0) dst[threadIdx.x] = v + threadIdx.x; }
__global__ void __launch_bounds__(256) k_pass(const double* rows, const double* v, double* w, uint32_t n)
{
	const uint32_t packs = n / 2, stride = gridDim.x * 256;
	for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < packs; p += stride) {
		d2 q = *reinterpret_cast<const d2*>(v + (size_t) p * 2);
		#pragma unroll
		for (int j = 0; j < 10; j++) { d2 f = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rows + (size_t) j * n + (size_t) p * 2)); q.x += f.x; q.y += f.y; }
		*reinterpret_cast<d2*>(w + (size_t) p * 2) = q;
	}
}
int main()
{
	hipStream_t s; CK(hipStreamCreate(&s));
	const uint32_t n = 10000000;
	double *dev, *pin, *rows, *v, *w;
	CK(hipMalloc(&dev, 64)); CK(hipHostMalloc(&pin, 64));
	CK(hipMalloc(&rows, (size_t) 10 * n * 8)); CK(hipMalloc(&v, (size_t) n * 8)); CK(hipMalloc(&w, (size_t) n * 8));
	CK(hipMemset(rows, 0, (size_t) 10 * n * 8)); CK(hipMemset(v, 0, (size_t) n * 8));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int grid = 3 * prop.multiProcessorCount;
	for (int real = 0; real < 2; real++)
		for (int zero_copy = 0; zero_copy < 2; zero_copy++) {
			auto call = [&](int i) {
				for (int k = 0; k < 3; k++) {
					if (real) hipLaunchKernelGGL(k_pass, dim3(grid), dim3(256), 0, s, rows, v, w, n);
					else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, dev);
				}
				hipLaunchKernelGGL(k_report, dim3(1), dim3(64), 0, s, zero_copy ? pin : dev, (double) i);
				if (!zero_copy) CK(hipMemcpyAsync(pin, dev, 64, hipMemcpyDeviceToHost, s));
				CK(hipStreamSynchronize(s));
				if (pin[3] != (double) i + 3) { printf("report not there: %g\n", pin[3]); exit(1); }
			};
			const int reps = real ? 1000 : 4000;
			for (int i = 0; i < 200; i++) call(i);
			auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < reps; i++) call(i);
			const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e6;
			printf("%s kernels, report %s: %.2f us per call\n", real ? "three C2-pass-shaped" : "three empty", zero_copy ? "written into host-mapped memory by the kernel" : "copied back with hipMemcpyAsync", us);
		}
	return 0;
}
