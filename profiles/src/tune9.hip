// profiles/src/tune9.hip -- the store stream of pass 2 once more: does it help when the WHOLE CHIP writes at the same time?
// Pass 2 of the three-pass form (q0 = g - sum alpha_j y_j; r0 = gamma q0; v_j = y_j'r0; r0 over g) with the results of NB
// iterations parked in LDS and written in one burst per workgroup -- alone (BURST) or behind a grid-wide barrier so that all 256
// workgroups leave the read phase and enter the write phase together (PHASED).  The barrier is a counter in device memory with
// a bounded spin: a workgroup that does not see the others within 2^22 polls goes on alone (every wave has an exit).
// Not part of the product.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tune9.hip -o tune9
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
__device__ __forceinline__ void st_stream(double* dstp, d2 q)
{
	d2* dst = reinterpret_cast<d2*>(dstp);
	asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dst), "v"(q) : "memory");
}
constexpr int K = 20;
constexpr int BLOCK = 256;

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		for (unsigned spin = 0; spin < (1u << 22); spin++) {
			if (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
			__builtin_amdgcn_s_sleep(2);
		}
	}
	__syncthreads();
}

// MODE 0: store at once (the product's shape); 1: no store; 2: burst of NB iterations per workgroup; 3: burst behind a grid
// barrier; 4: grid barrier only, no store (what the barrier itself costs)
template <int NB, int MODE>
__global__ void __launch_bounds__(BLOCK, 1) k_qdot(const double* Y, size_t ld_, const double* coef, double* g, uint32_t n, double* parts, unsigned* bar)
{
	__shared__ double sh[K * (BLOCK / 64)];
	__shared__ double cf[1 + K];
	__shared__ d2 park[(MODE == 2 || MODE == 3) ? NB * BLOCK : 1];
	for (int e = threadIdx.x; e < 1 + K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	const uint32_t iters = (packs + stride - 1) / stride;          // the same for every workgroup: barriers match
	unsigned phase = 0;
	for (uint32_t it0 = 0; it0 < iters; it0 += NB) {
		#pragma unroll 1
		for (int b = 0; b < NB; b++) {
			const uint32_t p = (it0 + b) * stride + blockIdx.x * BLOCK + threadIdx.x;
			if (it0 + b < iters && p < packs) {
				d2 q = ldd(g, (size_t) p * 2), f[K];
				#pragma unroll
				for (int j = 0; j < K; j++) f[j] = ldnt(Y + (size_t) j * ld_, (size_t) p * 2);
				#pragma unroll
				for (int j = K - 1; j >= 0; j--) { q.x = fma(-cf[1 + j], f[j].x, q.x); q.y = fma(-cf[1 + j], f[j].y, q.y); }
				q.x *= cf[0]; q.y *= cf[0];
				#pragma unroll
				for (int j = 0; j < K; j++) { acc[j] = fma(f[j].x, q.x, acc[j]); acc[j] = fma(f[j].y, q.y, acc[j]); }
				if (MODE == 0) st_stream(g + (size_t) p * 2, q);
				if (MODE == 2 || MODE == 3) park[b * BLOCK + threadIdx.x] = q;
			}
		}
		if (MODE == 3 || MODE == 4) { phase++; grid_barrier(bar, phase * gridDim.x); }
		if (MODE == 2 || MODE == 3) {
			#pragma unroll 1
			for (int b = 0; b < NB; b++) {
				const uint32_t p = (it0 + b) * stride + blockIdx.x * BLOCK + threadIdx.x;
				if (it0 + b < iters && p < packs) st_stream(g + (size_t) p * 2, park[b * BLOCK + threadIdx.x]);
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
	double *S, *g, *parts, *coef; unsigned* bar;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8)); CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8));
	CK(hipMalloc(&bar, 64));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(coef, 0, 64 * 8));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("n = %u, K = %d, %d CUs\n", n, K, cus);
#define RQ(NB, MODE, GRID, WHAT) { double ms = time_ms([&](int) { CK(hipMemsetAsync(bar, 0, 64, 0)); \
		hipLaunchKernelGGL((k_qdot<NB, MODE>), dim3(GRID), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, bar); }); \
	printf("qdot %-28s NB %-3d grid %-4d : %.3f ms  %.0f GB/s\n", WHAT, NB, GRID, ms, (K + 2.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
	for (int rep = 0; rep < 2; rep++) {
		RQ(1, 0, cus, "store at once"); RQ(1, 1, cus, "no store");
		RQ(4, 2, cus, "burst"); RQ(8, 2, cus, "burst"); RQ(16, 2, cus, "burst"); RQ(32, 2, cus, "burst");
		RQ(8, 4, cus, "barrier only, no store"); RQ(32, 4, cus, "barrier only, no store");
		RQ(4, 3, cus, "phased"); RQ(8, 3, cus, "phased"); RQ(16, 3, cus, "phased"); RQ(32, 3, cus, "phased");
	}
	return 0;
}
