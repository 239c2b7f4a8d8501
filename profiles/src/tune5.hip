// profiles/src/tune5.hip -- 16-byte loads/stores at 8-byte alignment (odd n: every other ring row is misaligned); not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef d2 d2u __attribute__((aligned(8)));
template <bool NT> __device__ __forceinline__ d2 ldu(const double* p, size_t i)
{ return NT ? __builtin_nontemporal_load(reinterpret_cast<const d2u*>(p + i)) : *reinterpret_cast<const d2u*>(p + i); }
__device__ __forceinline__ double wave_sum(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
constexpr int K = 20;

// rows-dot over 2K rows with leading dimension ld_ (odd ld_ => odd rows are 8-byte aligned only)
template <int G, bool VEC>
__global__ void __launch_bounds__(256) k_rows(const double* S, const double* Y, size_t ld_, const double* g, uint32_t n, double* parts)
{
	__shared__ double sh[4];
	double acc[2 * K];
	#pragma unroll
	for (int j = 0; j < 2 * K; j++) acc[j] = 0;
	if (VEC) {
		const uint32_t packs = n / 2, stride = gridDim.x * 256;
		for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < packs; p += stride) {
			const size_t i = (size_t) p * 2;
			const d2 pv = ldu<false>(g, i);
			#pragma unroll
			for (int j0 = 0; j0 < 2 * K; j0 += G) {
				d2 f[G];
				#pragma unroll
				for (int u = 0; u < G; u++) { const int j = j0 + u; f[u] = ldu<true>((j < K ? S + (size_t) j * ld_ : Y + (size_t) (j - K) * ld_), i); }
				#pragma unroll
				for (int u = 0; u < G; u++) { acc[j0 + u] = fma(f[u].x, pv.x, acc[j0 + u]); acc[j0 + u] = fma(f[u].y, pv.y, acc[j0 + u]); }
			}
		}
	} else {
		const uint32_t stride = gridDim.x * 256;
		for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
			const double pv = g[i];
			#pragma unroll
			for (int j0 = 0; j0 < 2 * K; j0 += G) {
				double f[G];
				#pragma unroll
				for (int u = 0; u < G; u++) { const int j = j0 + u; f[u] = __builtin_nontemporal_load((j < K ? S + (size_t) j * ld_ : Y + (size_t) (j - K) * ld_) + i); }
				#pragma unroll
				for (int u = 0; u < G; u++) acc[j0 + u] = fma(f[u], pv, acc[j0 + u]);
			}
		}
	}
	for (int j = 0; j < 2 * K; j++) {
		double a = wave_sum(acc[j]);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
		__syncthreads();
		if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < 4; w++) t += sh[w]; parts[(size_t) j * 4096 + blockIdx.x] = t; }
	}
}

// copy with 16-byte stores at 8-byte alignment
__global__ void __launch_bounds__(256) k_copy(const double* a, double* b, uint32_t n)
{
	const uint32_t packs = n / 2, stride = gridDim.x * 256;
	for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < packs; p += stride) {
		const d2 v = ldu<false>(a, (size_t) p * 2);
		*reinterpret_cast<d2u*>(b + (size_t) p * 2) = v;
	}
}

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <class F> double time_ms(F&& launch, int reps = 7)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 2; i++) launch(i);
	CK(hipDeviceSynchronize());
	std::vector<float> t;
	for (int i = 0; i < reps; i++) {
		CK(hipEventRecord(a)); launch(i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
	}
	return median(t);
}

int main(int argc, char** argv)
{
	const uint32_t n = argc > 1 ? (uint32_t) atof(argv[1]) : 100000000u;
	double *S, *Y, *g, *parts, *c;
	CK(hipMalloc(&S, (size_t) K * (n + 2) * 8)); CK(hipMalloc(&Y, (size_t) K * (n + 2) * 8)); CK(hipMalloc(&g, (size_t) (n + 2) * 8));
	CK(hipMalloc(&c, (size_t) (n + 4) * 8));
	CK(hipMalloc(&parts, 4096 * 64 * 8));
	CK(hipMemset(S, 0, (size_t) K * (n + 2) * 8)); CK(hipMemset(Y, 0, (size_t) K * (n + 2) * 8)); CK(hipMemset(g, 0, (size_t) (n + 2) * 8));
	printf("n = %u, K = %d\n", n, K);
	for (int grid : {256, 512}) {
		for (size_t ld_ : {(size_t) n, (size_t) n + 1}) {
			double ms = time_ms([&](int) { hipLaunchKernelGGL((k_rows<8, true>), dim3(grid), dim3(256), 0, 0, S, Y, ld_, g, n, parts); });
			printf("rows vec16  grid %-4d ld %s: %.3f ms  %.0f GB/s\n", grid, ld_ == n ? "even (all rows 16B aligned)" : "odd (every other row 8B aligned)", ms, (2.0 * K + 1) * 8.0 * n / ms / 1e6);
		}
		double ms = time_ms([&](int) { hipLaunchKernelGGL((k_rows<8, false>), dim3(grid), dim3(256), 0, 0, S, Y, (size_t) n + 1, g, n, parts); });
		printf("rows scalar grid %-4d ld odd: %.3f ms  %.0f GB/s\n", grid, ms, (2.0 * K + 1) * 8.0 * n / ms / 1e6);
	}
	for (int off : {0, 1}) {
		double ms = time_ms([&](int) { hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, 0, g + off, c + off, n); });
		printf("copy 16B at offset %d: %.3f ms %.0f GB/s\n", off, ms, 16.0 * n / ms / 1e6);
		ms = time_ms([&](int) { hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, 0, g, c + off, n); });
		printf("copy 16B aligned src, dst offset %d: %.3f ms %.0f GB/s\n", off, ms, 16.0 * n / ms / 1e6);
	}
	return 0;
}
