// profiles/src/tune3.hip -- sweep of the two-pass kernels' shape (rows-dot / combine); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ d2 ld(const double* p, size_t i)
{ return NT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)) : *reinterpret_cast<const d2*>(p + i); }
__device__ __forceinline__ void st(double* p, size_t i, d2 v) { *reinterpret_cast<d2*>(p + i) = v; }
__device__ __forceinline__ double wave_sum(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
constexpr int K = 20;

// combine: r = c0*g + sum_j cy_j y_j + cs_j s_j ; G pairs (2G loads) in flight per group
template <int BLOCK, int G, bool NT, int MODE>
__global__ void __launch_bounds__(BLOCK) k_combine(const double* S, const double* Y, size_t ld_, const double* coef, double* g, double* out, uint32_t n, int rev, double* parts)
{
	__shared__ double cf[1 + 2 * K];
	__shared__ double sh[BLOCK / 64];
	for (int e = threadIdx.x; e < 1 + 2 * K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, last = packs - 1;
	double acc = 0;
	for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < packs; p += stride) {
		const size_t i = (size_t) (rev ? last - p : p) * 2;
		d2 r = ld<false>(g, i);
		r.x *= cf[0]; r.y *= cf[0];
		#pragma unroll
		for (int j0 = 0; j0 < K; j0 += G) {
			d2 fy[G], fs[G];
			#pragma unroll
			for (int u = 0; u < G; u++) { fy[u] = ld<NT>(Y + (size_t) (j0 + u) * ld_, i); fs[u] = ld<NT>(S + (size_t) (j0 + u) * ld_, i); }
			#pragma unroll
			for (int u = 0; u < G; u++) {
				r.x = fma(cf[1 + j0 + u], fy[u].x, r.x); r.y = fma(cf[1 + j0 + u], fy[u].y, r.y);
				r.x = fma(cf[1 + K + j0 + u], fs[u].x, r.x); r.y = fma(cf[1 + K + j0 + u], fs[u].y, r.y);
			}
		}
		acc = fma(r.x, r.x, acc); acc = fma(r.y, r.y, acc);
		if (MODE == 0) st(g, i, r);
		else if (MODE == 2) st(out, i, r);
		else if (MODE == 3) __builtin_nontemporal_store(r, reinterpret_cast<d2*>(g + i));
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BLOCK / 64; w++) t += sh[w]; parts[blockIdx.x] = t; }
}

// rows-dot over 2K rows
template <int BLOCK, int G, bool NT>
__global__ void __launch_bounds__(BLOCK) k_rows(const double* S, const double* Y, size_t ld_, const double* g, uint32_t n, int rev, double* parts)
{
	__shared__ double sh[BLOCK / 64];
	double acc[2 * K];
	#pragma unroll
	for (int j = 0; j < 2 * K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, last = packs - 1;
	for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < packs; p += stride) {
		const size_t i = (size_t) (rev ? last - p : p) * 2;
		const d2 pv = ld<false>(g, i);
		#pragma unroll
		for (int j0 = 0; j0 < 2 * K; j0 += G) {
			d2 f[G];
			#pragma unroll
			for (int u = 0; u < G; u++) { const int j = j0 + u; f[u] = ld<NT>((j < K ? S + (size_t) j * ld_ : Y + (size_t) (j - K) * ld_), i); }
			#pragma unroll
			for (int u = 0; u < G; u++) { acc[j0 + u] = fma(f[u].x, pv.x, acc[j0 + u]); acc[j0 + u] = fma(f[u].y, pv.y, acc[j0 + u]); }
		}
	}
	for (int j = 0; j < 2 * K; j++) {
		double a = wave_sum(acc[j]);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
		__syncthreads();
		if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BLOCK / 64; w++) t += sh[w]; parts[(size_t) j * 4096 + blockIdx.x] = t; }
	}
}

// combine with deferred stores: each lane computes T packs (strided by the grid) and stores them together
template <int BLOCK, int G, int T>
__global__ void __launch_bounds__(BLOCK) k_combine_batched(const double* S, const double* Y, size_t ld_, const double* coef, double* g, uint32_t n, double* parts)
{
	__shared__ double cf[1 + 2 * K];
	__shared__ double sh[BLOCK / 64];
	for (int e = threadIdx.x; e < 1 + 2 * K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	double acc = 0;
	for (uint32_t p0 = blockIdx.x * BLOCK + threadIdx.x; p0 < packs; p0 += T * stride) {
		d2 out[T];
		#pragma unroll
		for (int t = 0; t < T; t++) {
			const uint32_t p = p0 + t * stride;
			if (p < packs) {
				const size_t i = (size_t) p * 2;
				d2 r = ld<false>(g, i);
				r.x *= cf[0]; r.y *= cf[0];
				#pragma unroll
				for (int j0 = 0; j0 < K; j0 += G) {
					d2 fy[G], fs[G];
					#pragma unroll
					for (int u = 0; u < G; u++) { fy[u] = ld<true>(Y + (size_t) (j0 + u) * ld_, i); fs[u] = ld<true>(S + (size_t) (j0 + u) * ld_, i); }
					#pragma unroll
					for (int u = 0; u < G; u++) {
						r.x = fma(cf[1 + j0 + u], fy[u].x, r.x); r.y = fma(cf[1 + j0 + u], fy[u].y, r.y);
						r.x = fma(cf[1 + K + j0 + u], fs[u].x, r.x); r.y = fma(cf[1 + K + j0 + u], fs[u].y, r.y);
					}
				}
				acc = fma(r.x, r.x, acc); acc = fma(r.y, r.y, acc);
				out[t] = r;
			}
		}
		#pragma unroll
		for (int t = 0; t < T; t++) {
			const uint32_t p = p0 + t * stride;
			if (p < packs) st(g, (size_t) p * 2, out[t]);
		}
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BLOCK / 64; w++) t += sh[w]; parts[blockIdx.x] = t; }
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
	double *S, *Y, *g, *parts, *coef;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&Y, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8));
	CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(Y, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(coef, 0, 64 * 8));
	printf("n = %u, K = %d\n", n, K);
#define RUNC(BLOCK, G, NT, GRID, REV, MODE) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((k_combine<BLOCK, G, NT, MODE>), dim3(GRID), dim3(BLOCK), 0, 0, S, Y, (size_t) n, coef, g, out, n, (REV) ? (i & 1) : 0, parts); }); \
	printf("combine B%-4d G%-2d nt%d grid %-5d rev %d mode %d : %.3f ms  %.0f GB/s\n", BLOCK, G, NT, GRID, REV, MODE, ms, (2.0 * K + 2) * 8.0 * n / ms / 1e6); }
#define RUNR(BLOCK, G, NT, GRID, REV) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((k_rows<BLOCK, G, NT>), dim3(GRID), dim3(BLOCK), 0, 0, S, Y, (size_t) n, g, n, (REV) ? (i & 1) : 0, parts); }); \
	printf("rows    B%-4d G%-2d nt%d grid %-5d rev %d : %.3f ms  %.0f GB/s\n", BLOCK, G, NT, GRID, REV, ms, (2.0 * K + 1) * 8.0 * n / ms / 1e6); }
	double* out; CK(hipMalloc(&out, (size_t) n * 8));
#define RUNB(BLOCK, G, T, GRID) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((k_combine_batched<BLOCK, G, T>), dim3(GRID), dim3(BLOCK), 0, 0, S, Y, (size_t) n, coef, g, n, parts); }); \
	printf("combine-batched B%-4d G%-2d T%-2d grid %-5d : %.3f ms  %.0f GB/s\n", BLOCK, G, T, GRID, ms, (2.0 * K + 2) * 8.0 * n / ms / 1e6); }
	for (int grid : {512, 2048}) {
		RUNC(256, 4, true, grid, 0, 0); RUNC(256, 4, true, grid, 0, 1);
		RUNB(256, 4, 2, grid); RUNB(256, 4, 4, grid); RUNB(256, 4, 8, grid); RUNB(256, 4, 16, grid); RUNB(256, 4, 32, grid);
	}
	return 0;
}
