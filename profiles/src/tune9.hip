// profiles/src/tune9.hip -- the store stream of pass 2 once more: does it help when the WHOLE CHIP writes at the same time?
// Pass 2 of the three-pass form (q0 = g - sum alpha_j y_j; r0 = gamma q0; v_j = y_j'r0; r0 over g) with the results of NB
// iterations parked in LDS and written in one burst per workgroup -- alone (BURST) or behind a grid-wide barrier so that all 256
// workgroups leave the read phase and enter the write phase together (PHASED).  The barrier is a counter in device memory with
// a bounded spin: a workgroup that does not see the others within 2^14 polls goes on alone (every wave has an exit).
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

// the same barrier with relaxed atomics: nothing but time is communicated, so no cache write-back / invalidate is wanted
__device__ __forceinline__ void grid_barrier_relaxed(unsigned* bar, unsigned target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (unsigned spin = 0; spin < (1u << 14); spin++) {
			if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	__syncthreads();
}

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		for (unsigned spin = 0; spin < (1u << 14); spin++) {
			if (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
			__builtin_amdgcn_s_sleep(2);
		}
	}
	__syncthreads();
}

// MODE 0: store at once (the product's shape); 1: no store; 2: burst of NB iterations per workgroup; 3: burst behind a grid
// barrier (acquire / release); 4: that barrier only, no store; 5 / 6: the same two with the relaxed barrier; 7: no barrier at all --
// every wave flushes what it has parked when the chip-wide 100 MHz clock (s_memrealtime) enters a new period of `period` ticks,
// or when its NB slots are full; 8: the clock checks only, no store
template <int NB, int MODE>
__global__ void __launch_bounds__(BLOCK, 1) k_qdot(const double* Y, size_t ld_, const double* coef, double* g, uint32_t n, double* parts, unsigned* bar,
                                                  uint32_t period_inv)
{
	constexpr bool PARK = MODE == 2 || MODE == 3 || MODE == 5 || MODE == 7;
	__shared__ double sh[K * (BLOCK / 64)];
	__shared__ double cf[1 + K];
	__shared__ d2 park[PARK ? NB * BLOCK : 1];
	for (int e = threadIdx.x; e < 1 + K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	const uint32_t iters = (packs + stride - 1) / stride;          // the same for every workgroup: barriers match
	unsigned phase = 0;
	int b = 0;
	uint32_t it_first = 0;
	uint32_t epoch = (MODE == 7 || MODE == 8) ? __umulhi((uint32_t) __builtin_amdgcn_s_memrealtime(), period_inv) : 0;
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++) {
		const uint32_t p = it * stride + blockIdx.x * BLOCK + threadIdx.x;
		if (p < packs) {
			d2 q = ldd(g, (size_t) p * 2), f[K];
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = ldnt(Y + (size_t) j * ld_, (size_t) p * 2);
			#pragma unroll
			for (int j = K - 1; j >= 0; j--) { q.x = fma(-cf[1 + j], f[j].x, q.x); q.y = fma(-cf[1 + j], f[j].y, q.y); }
			q.x *= cf[0]; q.y *= cf[0];
			#pragma unroll
			for (int j = 0; j < K; j++) { acc[j] = fma(f[j].x, q.x, acc[j]); acc[j] = fma(f[j].y, q.y, acc[j]); }
			if (MODE == 0) st_stream(g + (size_t) p * 2, q);
			if (PARK) park[b * BLOCK + threadIdx.x] = q;
		}
		b++;
		bool flush = b == NB || it == iters - 1;
		if (MODE == 7 || MODE == 8) {
			const uint32_t e = __umulhi((uint32_t) __builtin_amdgcn_s_memrealtime(), period_inv);
			if (e != epoch) { flush = true; epoch = e; }
		}
		if (flush) {
			if (MODE == 3 || MODE == 4) { phase++; grid_barrier(bar, phase * gridDim.x); }
			if (MODE == 5 || MODE == 6) { phase++; grid_barrier_relaxed(bar, phase * gridDim.x); }
			if (PARK) {
				#pragma unroll 1
				for (int bb = 0; bb < b; bb++) {
					const uint32_t pp = (it_first + bb) * stride + blockIdx.x * BLOCK + threadIdx.x;
					if (pp < packs) st_stream(g + (size_t) pp * 2, park[bb * BLOCK + threadIdx.x]);
				}
			}
			b = 0; it_first = it + 1;
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
#define RQP(NB, MODE, GRID, WHAT, TICKS) { const uint32_t inv = (TICKS) ? (uint32_t) (4294967296.0 / (TICKS)) : 0; double ms = time_ms([&](int) { CK(hipMemsetAsync(bar, 0, 64, 0)); \
		hipLaunchKernelGGL((k_qdot<NB, MODE>), dim3(GRID), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, bar, inv); }); \
	printf("qdot %-30s NB %-3d period %-6d grid %-4d : %.3f ms  %.0f GB/s\n", WHAT, NB, (int) (TICKS), GRID, ms, (K + 2.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
#define RQ(NB, MODE, GRID, WHAT) RQP(NB, MODE, GRID, WHAT, 0)
	for (int rep = 0; rep < 2; rep++) {
		RQ(1, 0, cus, "store at once"); RQ(1, 1, cus, "no store");
		RQ(8, 2, cus, "burst"); RQ(32, 2, cus, "burst");
		RQ(32, 4, cus, "acq/rel barrier only, no store"); RQ(32, 3, cus, "phased, acq/rel barrier");
		RQ(8, 6, cus, "relaxed barrier only, no store"); RQ(16, 6, cus, "relaxed barrier only, no store"); RQ(32, 6, cus, "relaxed barrier only, no store");
		RQ(8, 5, cus, "phased, relaxed barrier"); RQ(16, 5, cus, "phased, relaxed barrier"); RQ(32, 5, cus, "phased, relaxed barrier");
		RQP(32, 8, cus, "clock checks only, no store", 8000);
		RQP(32, 7, cus, "clock-phased", 2000); RQP(32, 7, cus, "clock-phased", 4000); RQP(32, 7, cus, "clock-phased", 6000);
		RQP(32, 7, cus, "clock-phased", 8000); RQP(32, 7, cus, "clock-phased", 10000); RQP(32, 7, cus, "clock-phased", 16000);
	}
	return 0;
}
