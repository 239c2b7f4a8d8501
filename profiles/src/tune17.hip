// profiles/src/tune17.hip -- round 6: do the three passes of the recursion gain from longer bursts per row?
// Fisher pass 1 gained 3.3 % when a workgroup took TWO adjacent column tiles per trip (2 KB instead of 1 KB of every row back to back,
// option fisher_tile).  The passes of the three-pass form read 4 KB of every row per workgroup and iteration (256 lanes x 16 B, one
// pack of all K rows per lane).  Here: the shapes of pass 1 (g + K rows read, K sums), pass 2 (g + K rows read, r0 written through
// the clock-phased parking, K sums) and pass 3 (r0 + K rows read, r written, 2 sums) with U = 1 (as the product has them) and with
// U = 2 / 4 adjacent tiles per iteration: a lane then holds U packs of every row -- U x 4 KB of a row per workgroup and iteration,
// U x (K + 1) loads in flight per lane (one wave per SIMD either way: the parked results take the CU's LDS).
// Not part of the product.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tune17.hip -o tune17
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
__device__ __forceinline__ uint32_t epoch_now(uint32_t inv) { return __umulhi((uint32_t) __builtin_amdgcn_s_memrealtime(), inv); }

// pass 1's shape: g and K rows read, K sums
template <int U>
__global__ void __launch_bounds__(BLOCK, 1) k_sdot(const double* S, size_t ld_, const double* g, uint32_t n, double* parts)
{
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK * U;
	for (uint32_t p0 = blockIdx.x * BLOCK * U + threadIdx.x; p0 < packs; p0 += stride) {
		d2 v[U], f[U][K];
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = p0 + u * BLOCK;
			if (p < packs) {
				v[u] = ldd(g, (size_t) p * 2);
				#pragma unroll
				for (int j = 0; j < K; j++) f[u][j] = ldnt(S + (size_t) j * ld_, (size_t) p * 2);
			}
		}
		#pragma unroll
		for (int u = 0; u < U; u++)
			if (p0 + u * BLOCK < packs) {
				#pragma unroll
				for (int j = 0; j < K; j++) { acc[j] = fma(f[u][j].x, v[u].x, acc[j]); acc[j] = fma(f[u][j].y, v[u].y, acc[j]); }
			}
	}
	#pragma unroll
	for (int j = 0; j < K; j++) { const double t = wave_sum(acc[j]); if ((threadIdx.x & 63) == 0) parts[(size_t) j * 4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t; }
}

// passes 2 and 3: MODE 2: q = g - sum cf_j y_j, scaled, K sums, q written; MODE 3: r = r0 + sum cf_j s_j, 2 sums, r written.
// Results are parked in LDS (NB packs per lane) and flushed when the chip-wide clock enters a new period, as in the product.
template <int MODE, int U>
__global__ void __launch_bounds__(BLOCK, 1) k_pass(const double* R, size_t ld_, const double* coef, double* g, uint32_t n, double* parts, uint32_t inv)
{
	constexpr int NB = 32;
	__shared__ double cf[K];
	__shared__ d2 park[NB * BLOCK];
	for (int e = threadIdx.x; e < K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc[MODE == 2 ? K : 2];
	#pragma unroll
	for (int j = 0; j < (MODE == 2 ? K : 2); j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK * U;
	int b = 0; uint32_t epoch = epoch_now(inv);
	uint32_t where[NB];                                  // pack number of every parked result (registers: NB is a compile-time constant)
	#pragma unroll 1
	for (uint32_t p0 = blockIdx.x * BLOCK * U + threadIdx.x; p0 < packs; p0 += stride) {
		d2 v[U], f[U][K];
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = p0 + u * BLOCK;
			if (p < packs) {
				v[u] = ldd(g, (size_t) p * 2);
				#pragma unroll
				for (int j = 0; j < K; j++) f[u][j] = ldnt(R + (size_t) j * ld_, (size_t) p * 2);
			}
		}
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = p0 + u * BLOCK;
			if (p < packs) {
				d2 q = v[u];
				if (MODE == 2) {
					#pragma unroll
					for (int j = K - 1; j >= 0; j--) { q.x = fma(-cf[j], f[u][j].x, q.x); q.y = fma(-cf[j], f[u][j].y, q.y); }
					q.x *= 0.75; q.y *= 0.75;
					#pragma unroll
					for (int j = 0; j < K; j++) { acc[j] = fma(f[u][j].x, q.x, acc[j]); acc[j] = fma(f[u][j].y, q.y, acc[j]); }
				} else {
					#pragma unroll
					for (int j = 0; j < K; j++) { q.x = fma(cf[j], f[u][j].x, q.x); q.y = fma(cf[j], f[u][j].y, q.y); }
					acc[0] = fma(q.x, q.x, acc[0]); acc[0] = fma(q.y, q.y, acc[0]);
					acc[1] += (isfinite(q.x) ? 0.0 : 1.0) + (isfinite(q.y) ? 0.0 : 1.0);
				}
				park[b * BLOCK + threadIdx.x] = q;
				#pragma unroll
				for (int s = 0; s < NB; s++) if (s == b) where[s] = p;
				b++;
			}
		}
		const uint32_t e = epoch_now(inv);
		if (b + U > NB || e != epoch) {
			epoch = e;
			#pragma unroll
			for (int s = 0; s < NB; s++) if (s < b) st_stream(g + (size_t) where[s] * 2, park[s * BLOCK + threadIdx.x]);
			b = 0;
		}
	}
	#pragma unroll
	for (int s = 0; s < NB; s++) if (s < b) st_stream(g + (size_t) where[s] * 2, park[s * BLOCK + threadIdx.x]);
	#pragma unroll
	for (int j = 0; j < (MODE == 2 ? K : 2); j++) { const double t = wave_sum(acc[j]); if ((threadIdx.x & 63) == 0) parts[(size_t) j * 4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t; }
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
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8));
	CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(coef, 0, 64 * 8));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("n = %u, K = %d, %d CUs, one workgroup of %d per CU\n", n, K, cus, BLOCK);
#define INV(TICKS) ((uint32_t) (4294967296.0 / (TICKS)))
#define R1(U) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sdot<U>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, g, n, parts); }); \
	printf("pass 1 shape, %d tile(s) per iteration : %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", U, ms, (K + 1.0) * 8.0 * n / ms / 1e6, (K + 1.0) * 8.0 * n / ms / 1e6 / 8000); fflush(stdout); }
#define RP(MODE, U) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_pass<MODE, U>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, INV(8000)); }); \
	printf("pass %d shape, %d tile(s) per iteration : %.3f ms  %.0f GB/s  %.3f of 8 TB/s\n", MODE, U, ms, (K + 2.0) * 8.0 * n / ms / 1e6, (K + 2.0) * 8.0 * n / ms / 1e6 / 8000); fflush(stdout); }
	for (int rep = 0; rep < 3; rep++) {
		R1(1); R1(2); R1(4);
		RP(2, 1); RP(2, 2); RP(2, 4);
		RP(3, 1); RP(3, 2); RP(3, 4);
	}
	return 0;
}
