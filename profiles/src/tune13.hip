// profiles/src/tune13.hip -- round 5, VERDICT r04 task 4: where is the floor of a C2 step (oLBFGS, n = 1e7, m = 10, fp64)?
// A step is two synchronous calls: [pass 1, pass 2, pass 3, update] + read-back + wait, then [y = g - g_prev with its verdict] +
// read-back + wait.  Measured here, with nothing of the product in the way:
//   * the bare traffic shapes of the five kernels at n = 1e7 (reads only / reads + one store / element-wise), best grid each:
//     what the memory system gives a 0.1 - 1 GB kernel including its ramp and tail;
//   * the same shapes at n = 1e8, where ramp and tail are 1 % instead of 10 %;
//   * what a call costs when its kernels do nothing: 4 dependent empty launches + 64-byte read-back + hipStreamSynchronize, and
//     1 launch + read-back + synchronise, wall clock, in a loop like the caller's.
// The sum is the floor of the step in this ABI; DESIGN.md 4.1 quotes it next to the product's number.  Not part of the product.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tune13.hip -o tune13
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ldnt(const double* p, size_t i) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)); }
__device__ __forceinline__ d2 ldd(const double* p, size_t i) { return *reinterpret_cast<const d2*>(p + i); }
constexpr int BLOCK = 256;

// NR rows streamed (nt) + one vector read (cached) [+ one vector written]: passes 1 / 2 / 3 without their arithmetic
template <int NR, bool WRITE>
__global__ void __launch_bounds__(BLOCK) k_rows(const double* rows, size_t ld_, const double* v, double* w, uint32_t n, double* out)
{
	double acc = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < packs; p += stride) {
		d2 q = ldd(v, (size_t) p * 2), f[NR];
		#pragma unroll
		for (int j = 0; j < NR; j++) f[j] = ldnt(rows + (size_t) j * ld_, (size_t) p * 2);
		#pragma unroll
		for (int j = 0; j < NR; j++) { q.x += f[j].x; q.y += f[j].y; }
		acc += q.x + q.y;
		if (WRITE) *reinterpret_cast<d2*>(w + (size_t) p * 2) = q;
	}
	if (acc == 12345.678) out[0] = acc;
}

// element-wise: R inputs read, W outputs written
template <int R, int W, int U>
__global__ void __launch_bounds__(BLOCK) k_ew(const double* a, const double* b, const double* c, double* o0, double* o1, double* o2, uint32_t n)
{
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	for (uint32_t p0 = blockIdx.x * BLOCK + threadIdx.x; p0 < packs; p0 += U * stride) {
		d2 x[U], y[U], z[U];
		#pragma unroll
		for (int u = 0; u < U; u++) { const uint32_t p = p0 + u * stride; if (p < packs) { x[u] = ldd(a, (size_t) p * 2); if (R > 1) y[u] = ldd(b, (size_t) p * 2); if (R > 2) z[u] = ldd(c, (size_t) p * 2); } }
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = p0 + u * stride;
			if (p < packs) {
				d2 r = x[u]; if (R > 1) { r.x -= y[u].x; r.y -= y[u].y; } if (R > 2) { r.x += 0.5 * z[u].x; r.y += 0.5 * z[u].y; }
				*reinterpret_cast<d2*>(o0 + (size_t) p * 2) = r;
				if (W > 1) *reinterpret_cast<d2*>(o1 + (size_t) p * 2) = r;
				if (W > 2) __builtin_nontemporal_store(r, reinterpret_cast<d2*>(o2 + (size_t) p * 2));
			}
		}
	}
}

__global__ void k_empty(double* p) { if (p == nullptr) p[0] = 0; }

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <class F> double time_ms(F&& launch, int reps = 15)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 3; i++) launch(i);
	CK(hipDeviceSynchronize());
	std::vector<float> t;
	for (int i = 0; i < reps; i++) {
		CK(hipEventRecord(a)); launch(i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
	}
	return median(t);
}

int main()
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	constexpr int M = 10;
	double floor_ms[2] = {0, 0};
	for (int big = 0; big < 2; big++) {
		const uint32_t n = big ? 100000000u : 10000000u;
		double *rows, *v, *w, *x, *y, *out;
		CK(hipMalloc(&rows, (size_t) (M + 1) * n * 8)); CK(hipMalloc(&v, (size_t) n * 8)); CK(hipMalloc(&w, (size_t) n * 8));
		CK(hipMalloc(&x, (size_t) n * 8)); CK(hipMalloc(&y, (size_t) n * 8)); CK(hipMalloc(&out, 64));
		CK(hipMemset(rows, 0, (size_t) (M + 1) * n * 8)); CK(hipMemset(v, 0, (size_t) n * 8)); CK(hipMemset(w, 0, (size_t) n * 8)); CK(hipMemset(x, 0, (size_t) n * 8)); CK(hipMemset(y, 0, (size_t) n * 8));
		printf("---- n = %u, m = %d, %d CUs\n", n, M, cus);
		double best[5] = {1e9, 1e9, 1e9, 1e9, 1e9};
		const char* what[5] = {"pass 1 shape: g + 11 rows read (12 n words; + g_prev written: 13 n in the product)", "pass 2 shape: g + 10 rows read, r0 written (12 n words)",
		                       "pass 3 shape: r0 + 10 rows read, r written (12 n words)", "update shape: 2 read, 3 written (5 n words)", "pair shape: 3 read, 1 written (4 n words)"};
		const double words[5] = {12, 12, 12, 5, 4};
		for (int g : {1, 2, 3, 4}) {
			double ms;
			ms = time_ms([&](int) { hipLaunchKernelGGL((k_rows<M + 1, false>), dim3(g * cus), dim3(BLOCK), 0, 0, rows, (size_t) n, v, w, n, out); }); best[0] = std::min(best[0], ms);
			ms = time_ms([&](int) { hipLaunchKernelGGL((k_rows<M, true>), dim3(g * cus), dim3(BLOCK), 0, 0, rows, (size_t) n, v, w, n, out); }); best[1] = std::min(best[1], ms); best[2] = std::min(best[2], ms);
			for (int u = 0; u < 2; u++) {
				ms = u ? time_ms([&](int) { hipLaunchKernelGGL((k_ew<2, 3, 4>), dim3(g * cus), dim3(BLOCK), 0, 0, v, x, y, x, w, rows, n); })
				       : time_ms([&](int) { hipLaunchKernelGGL((k_ew<2, 3, 2>), dim3(g * cus), dim3(BLOCK), 0, 0, v, x, y, x, w, rows, n); });
				best[3] = std::min(best[3], ms);
				ms = u ? time_ms([&](int) { hipLaunchKernelGGL((k_ew<3, 1, 4>), dim3(g * cus), dim3(BLOCK), 0, 0, v, x, y, w, w, w, n); })
				       : time_ms([&](int) { hipLaunchKernelGGL((k_ew<3, 1, 2>), dim3(g * cus), dim3(BLOCK), 0, 0, v, x, y, w, w, w, n); });
				best[4] = std::min(best[4], ms);
			}
		}
		double sum = 0;
		for (int k = 0; k < 5; k++) { printf("  %-90s best %.4f ms  %.0f GB/s\n", what[k], best[k], words[k] * 8.0 * n / best[k] / 1e6); sum += best[k]; }
		printf("  sum of the five: %.4f ms for %.3f GB = %.0f GB/s\n", sum, 45 * 8.0 * n / 1e9, 45 * 8.0 * n / sum / 1e6);
		floor_ms[big] = sum;
		CK(hipFree(rows)); CK(hipFree(v)); CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(out));
	}
	// ---- what the two calls of a step cost when their kernels do nothing ---------------------------------------------
	hipStream_t s; CK(hipStreamCreate(&s));
	double *dev, *pin;
	CK(hipMalloc(&dev, 64)); CK(hipHostMalloc(&pin, 64));
	auto call = [&](int launches) {
		for (int k = 0; k < launches; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, dev);
		CK(hipMemcpyAsync(pin, dev, 64, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s));
	};
	for (int i = 0; i < 200; i++) { call(4); call(1); }
	const int reps = 2000;
	auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < reps; i++) { call(4); call(1); }
	const double both = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e3;
	t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < reps; i++) call(4);
	const double four = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e3;
	printf("---- empty calls: [4 launches + read-back + synchronise] %.4f ms; the pair of calls of one oLBFGS step %.4f ms\n", four, both);
	printf("---- floor of a C2 step in this ABI: %.4f ms of bare traffic + %.4f ms of empty calls = %.4f ms  (%.3f GB -> %.0f GB/s, %.2f of 8 TB/s)\n",
	       floor_ms[0], both, floor_ms[0] + both, 45 * 8.0 * 1e7 / 1e9, 45 * 8.0 * 1e7 / (floor_ms[0] + both) / 1e6, 45 * 8.0 * 1e7 / (floor_ms[0] + both) / 1e6 / 8000);
	return 0;
}
