// profiles/src/tune8.hip -- shapes of the three-pass kernels: pass 1 (k rows of S . g: read-only, k accumulators per lane)
// and pass 2 (q0 = g - sum alpha_j y_j; r0 = gamma q0; v_j = y_j'r0; write r0).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ldnt(const double* p, size_t i) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)); }
__device__ __forceinline__ d2 ldd(const double* p, size_t i) { return *reinterpret_cast<const d2*>(p + i); }
__device__ __forceinline__ double wave_sum(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
constexpr int K = 20;

// pass 1, T packs per lane and iteration, OCC waves per SIMD asked from the compiler
template <int BLOCK, int T, int OCC, bool NT>
__global__ void __launch_bounds__(BLOCK, OCC) k_sdot(const double* S, size_t ld_, const double* g, uint32_t n, double* parts)
{
	__shared__ double sh[K * (BLOCK / 64)];
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	for (uint32_t p0 = blockIdx.x * BLOCK + threadIdx.x; p0 < packs; p0 += T * stride) {
		d2 gv[T], f[T][K];
		#pragma unroll
		for (int t = 0; t < T; t++) {
			const uint32_t p = p0 + t * stride;
			if (p < packs) {
				gv[t] = ldd(g, (size_t) p * 2);
				#pragma unroll
				for (int j = 0; j < K; j++) f[t][j] = NT ? ldnt(S + (size_t) j * ld_, (size_t) p * 2) : ldd(S + (size_t) j * ld_, (size_t) p * 2);
			}
		}
		#pragma unroll
		for (int t = 0; t < T; t++) {
			const uint32_t p = p0 + t * stride;
			if (p < packs) {
				#pragma unroll
				for (int j = 0; j < K; j++) { acc[j] = fma(f[t][j].x, gv[t].x, acc[j]); acc[j] = fma(f[t][j].y, gv[t].y, acc[j]); }
			}
		}
	}
	#pragma unroll
	for (int j = 0; j < K; j++) {
		const double t = wave_sum(acc[j]);
		if ((threadIdx.x & 63) == 0) sh[j * (BLOCK / 64) + (threadIdx.x >> 6)] = t;
	}
	__syncthreads();
	for (int j = threadIdx.x; j < K; j += BLOCK) {
		double t = 0;
		for (int w = 0; w < BLOCK / 64; w++) t += sh[j * (BLOCK / 64) + w];
		parts[(size_t) j * 4096 + blockIdx.x] = t;
	}
}

// pass 1 with the rows split over the NW waves of a workgroup (all waves walk the same columns)
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_sdot_split(const double* S, size_t ld_, const double* g, uint32_t n, double* parts)
{
	constexpr int RPW = (K + NW - 1) / NW;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row0 = wave * RPW;
	double acc[RPW];
	#pragma unroll
	for (int j = 0; j < RPW; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * 64;
	for (uint32_t p = blockIdx.x * 64 + lane; p < packs; p += stride) {
		const d2 gv = ldd(g, (size_t) p * 2);
		d2 f[RPW];
		#pragma unroll
		for (int j = 0; j < RPW; j++) if (row0 + j < K) f[j] = ldnt(S + (size_t) (row0 + j) * ld_, (size_t) p * 2);
		#pragma unroll
		for (int j = 0; j < RPW; j++) if (row0 + j < K) { acc[j] = fma(f[j].x, gv.x, acc[j]); acc[j] = fma(f[j].y, gv.y, acc[j]); }
	}
	#pragma unroll
	for (int j = 0; j < RPW; j++) {
		const double t = wave_sum(acc[j]);
		if (lane == 0 && row0 + j < K) parts[(size_t) (row0 + j) * 4096 + blockIdx.x] = t;
	}
}

// pass 2
template <int BLOCK, int OCC, int POL>
__global__ void __launch_bounds__(BLOCK, OCC) k_qdot(const double* Y, size_t ld_, const double* coef, double* g, uint32_t n, double* parts)
{
	__shared__ double sh[K * (BLOCK / 64)];
	__shared__ double cf[1 + K];
	for (int e = threadIdx.x; e < 1 + K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < packs; p += stride) {
		d2 q = ldd(g, (size_t) p * 2), f[K];
		#pragma unroll
		for (int j = 0; j < K; j++) f[j] = ldnt(Y + (size_t) j * ld_, (size_t) p * 2);
		#pragma unroll
		for (int j = K - 1; j >= 0; j--) { q.x = fma(-cf[1 + j], f[j].x, q.x); q.y = fma(-cf[1 + j], f[j].y, q.y); }
		q.x *= cf[0]; q.y *= cf[0];
		#pragma unroll
		for (int j = 0; j < K; j++) { acc[j] = fma(f[j].x, q.x, acc[j]); acc[j] = fma(f[j].y, q.y, acc[j]); }
		d2* dst = reinterpret_cast<d2*>(g + (size_t) p * 2);
		if (POL == 0) *dst = q;
		else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dst), "v"(q) : "memory");
	}
	#pragma unroll
	for (int j = 0; j < K; j++) {
		const double t = wave_sum(acc[j]);
		if ((threadIdx.x & 63) == 0) sh[j * (BLOCK / 64) + (threadIdx.x >> 6)] = t;
	}
	__syncthreads();
	for (int j = threadIdx.x; j < K; j += BLOCK) {
		double t = 0;
		for (int w = 0; w < BLOCK / 64; w++) t += sh[j * (BLOCK / 64) + w];
		parts[(size_t) j * 4096 + blockIdx.x] = t;
	}
}

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <class F> double time_ms(F&& launch, int reps = 9)
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
	double *S, *g, *parts, *coef;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8)); CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(coef, 0, 64 * 8));
	printf("n = %u, K = %d\n", n, K);
#define RS(BLOCK, T, OCC, NT, GRID) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sdot<BLOCK, T, OCC, NT>), dim3(GRID), dim3(BLOCK), 0, 0, S, (size_t) n, g, n, parts); }); \
	printf("sdot  B%-4d T%d occ%d nt%d grid %-5d : %.3f ms  %.0f GB/s\n", BLOCK, T, OCC, NT, GRID, ms, (K + 1.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
#define RP(NW, GRID) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sdot_split<NW>), dim3(GRID), dim3(64 * NW), 0, 0, S, (size_t) n, g, n, parts); }); \
	printf("sdot-split NW%d grid %-5d : %.3f ms  %.0f GB/s\n", NW, GRID, ms, (K + 1.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
#define RQ(BLOCK, OCC, POL, GRID) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_qdot<BLOCK, OCC, POL>), dim3(GRID), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts); }); \
	printf("qdot  B%-4d occ%d pol%d grid %-5d : %.3f ms  %.0f GB/s\n", BLOCK, OCC, POL, GRID, ms, (K + 2.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
	RS(256, 1, 1, true, 256); RS(256, 1, 1, true, 512); RS(256, 1, 2, true, 512); RS(256, 1, 3, true, 768); RS(256, 1, 4, true, 1024);
	RS(256, 2, 1, true, 256); RS(256, 2, 2, true, 512); RS(256, 4, 1, true, 256);
	RS(512, 1, 1, true, 256); RS(512, 1, 2, true, 256); RS(1024, 1, 1, true, 256); RS(128, 1, 2, true, 512); RS(128, 1, 4, true, 1024);
	RS(256, 1, 1, false, 256); RS(256, 1, 2, false, 512);
	RP(4, 1024); RP(4, 2048); RP(8, 512); RP(8, 1024); RP(5, 1024); RP(10, 512); RP(2, 2048);
	RQ(256, 1, 1, 256); RQ(256, 1, 0, 256); RQ(256, 2, 1, 512); RQ(256, 3, 1, 768); RQ(512, 1, 1, 256); RQ(128, 2, 1, 512); RQ(128, 4, 1, 1024);
	return 0;
}
