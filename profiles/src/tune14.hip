// profiles/src/tune14.hip -- round 5: passes 2 and 3 run ONE wave per SIMD (a workgroup's parked results take most of a CU's LDS), so
// nothing hides a wave's own latencies but the wave itself.  Per pack a lane of pass 3 waits for its 21 loads, then runs a chain
// of 20 DEPENDENT fma's per element (r = r0 + c_0 s_0 + c_1 s_1 + ...), then parks the result -- and only then asks for the next
// pack's rows: the memory pipe idles for the length of the chain.  Variants of pass 3's shape (n = 1e8, K = 20, clock-phased stores):
//   V0  as the product has it;
//   V1  the chain split into four independent partial sums (different rounding: allowed, parity is 1e-10, not bits);
//   V2  software-pipelined: the NEXT pack's 21 loads are issued before this pack's arithmetic (two register buffers);
//   V3  V2 + V1.
// Not part of the product.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tune14.hip -o tune14
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
__device__ __forceinline__ void st_plain(double* dstp, d2 q) { *reinterpret_cast<d2*>(dstp) = q; }
constexpr int K = 20;
constexpr int BLOCK = 256;
__device__ __forceinline__ uint32_t epoch_now(uint32_t inv) { return __umulhi((uint32_t) __builtin_amdgcn_s_memrealtime(), inv); }


template <int V>
__global__ void __launch_bounds__(BLOCK, 1) k_sadd(const double* S, size_t ld_, const double* coef, double* r, uint32_t n, double* parts, uint32_t inv)
{
	constexpr int NB = 32;
	__shared__ double sh[2 * (BLOCK / 64)];
	__shared__ double cf[K];
	__shared__ d2 park[NB * BLOCK];
	for (int e = threadIdx.x; e < K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc0 = 0, acc1 = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, iters = (packs + stride - 1) / stride;
	int b = 0; uint32_t it_first = 0, epoch = epoch_now(inv);
	d2 v, f[K], vn, fn[K];
	uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
	if (V >= 2 && p < packs) {
		vn = ldd(r, (size_t) p * 2);
		#pragma unroll
		for (int j = 0; j < K; j++) fn[j] = ldnt(S + (size_t) j * ld_, (size_t) p * 2);
	}
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++, p += stride) {
		const bool live = p < packs;
		if (V >= 2) {
			v = vn;
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = fn[j];
			const uint32_t pn = p + stride;
			if (pn < packs) {                                  // the next pack's loads go out before this pack's arithmetic
				vn = ldd(r, (size_t) pn * 2);
				#pragma unroll
				for (int j = 0; j < K; j++) fn[j] = ldnt(S + (size_t) j * ld_, (size_t) pn * 2);
			}
		} else if (live) {
			v = ldd(r, (size_t) p * 2);
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = ldnt(S + (size_t) j * ld_, (size_t) p * 2);
		}
		if (live) {
			if (V == 1 || V == 3) {
				d2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
				#pragma unroll
				for (int j = 0; j < K; j += 4) {
					s0.x = fma(cf[j], f[j].x, s0.x); s0.y = fma(cf[j], f[j].y, s0.y);
					s1.x = fma(cf[j + 1], f[j + 1].x, s1.x); s1.y = fma(cf[j + 1], f[j + 1].y, s1.y);
					s2.x = fma(cf[j + 2], f[j + 2].x, s2.x); s2.y = fma(cf[j + 2], f[j + 2].y, s2.y);
					s3.x = fma(cf[j + 3], f[j + 3].x, s3.x); s3.y = fma(cf[j + 3], f[j + 3].y, s3.y);
				}
				v.x += (s0.x + s1.x) + (s2.x + s3.x); v.y += (s0.y + s1.y) + (s2.y + s3.y);
			} else {
				#pragma unroll
				for (int j = 0; j < K; j++) { v.x = fma(cf[j], f[j].x, v.x); v.y = fma(cf[j], f[j].y, v.y); }
			}
			acc0 = fma(v.x, v.x, acc0); acc0 = fma(v.y, v.y, acc0);
			acc1 += (isfinite(v.x) ? 0.0 : 1.0) + (isfinite(v.y) ? 0.0 : 1.0);
			park[b * BLOCK + threadIdx.x] = v;
		}
		b++;
		bool flush = b == NB || it == iters - 1;
		const uint32_t e = epoch_now(inv);
		if (e != epoch) { flush = true; epoch = e; }
		if (flush) {
			#pragma unroll 1
			for (int bb = 0; bb < b; bb++) {
				const uint32_t pp = (it_first + bb) * stride + blockIdx.x * BLOCK + threadIdx.x;
				if (pp < packs) st_stream(r + (size_t) pp * 2, park[bb * BLOCK + threadIdx.x]);
			}
			b = 0; it_first = it + 1;
		}
	}
	double t0 = wave_sum(acc0), t1 = wave_sum(acc1);
	if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = t0; sh[BLOCK / 64 + (threadIdx.x >> 6)] = t1; }
	__syncthreads();
	if (threadIdx.x == 0) { double a = 0, c = 0; for (int w = 0; w < BLOCK / 64; w++) { a += sh[w]; c += sh[BLOCK / 64 + w]; } parts[blockIdx.x] = a; parts[4096 + blockIdx.x] = c; }
}

// pass 2's shape: g and K rows read, K dots accumulated (independent), r0 = a combination written.  V0 as is, V2 software-pipelined.
template <int V>
__global__ void __launch_bounds__(BLOCK, 1) k_qdot(const double* Y, size_t ld_, const double* coef, double* g, uint32_t n, double* parts, uint32_t inv)
{
	constexpr int NB = 32;
	__shared__ double cf[K];
	__shared__ d2 park[NB * BLOCK];
	for (int e = threadIdx.x; e < K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc[K];
	#pragma unroll
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, iters = (packs + stride - 1) / stride;
	int b = 0; uint32_t it_first = 0, epoch = epoch_now(inv);
	d2 v, f[K], vn, fn[K];
	uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
	if (V >= 2 && p < packs) {
		vn = ldd(g, (size_t) p * 2);
		#pragma unroll
		for (int j = 0; j < K; j++) fn[j] = ldnt(Y + (size_t) j * ld_, (size_t) p * 2);
	}
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++, p += stride) {
		const bool live = p < packs;
		if (V >= 2) {
			v = vn;
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = fn[j];
			const uint32_t pn = p + stride;
			if (pn < packs) {
				vn = ldd(g, (size_t) pn * 2);
				#pragma unroll
				for (int j = 0; j < K; j++) fn[j] = ldnt(Y + (size_t) j * ld_, (size_t) pn * 2);
			}
		} else if (live) {
			v = ldd(g, (size_t) p * 2);
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = ldnt(Y + (size_t) j * ld_, (size_t) p * 2);
		}
		if (live) {
			d2 q = v;
			#pragma unroll
			for (int j = 0; j < K; j++) { q.x = fma(-cf[j], f[j].x, q.x); q.y = fma(-cf[j], f[j].y, q.y); }
			q.x *= 0.75; q.y *= 0.75;
			#pragma unroll
			for (int j = 0; j < K; j++) { acc[j] = fma(f[j].x, q.x, acc[j]); acc[j] = fma(f[j].y, q.y, acc[j]); }
			park[b * BLOCK + threadIdx.x] = q;
		}
		b++;
		bool flush = b == NB || it == iters - 1;
		const uint32_t e = epoch_now(inv);
		if (e != epoch) { flush = true; epoch = e; }
		if (flush) {
			#pragma unroll 1
			for (int bb = 0; bb < b; bb++) {
				const uint32_t pp = (it_first + bb) * stride + blockIdx.x * BLOCK + threadIdx.x;
				if (pp < packs) st_stream(g + (size_t) pp * 2, park[bb * BLOCK + threadIdx.x]);
			}
			b = 0; it_first = it + 1;
		}
	}
	#pragma unroll
	for (int j = 0; j < K; j++) { const double t = wave_sum(acc[j]); if ((threadIdx.x & 63) == 0) parts[(size_t) j * 4096 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t; }
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
	printf("n = %u, K = %d, %d CUs\n", n, K, cus);
#define INV(TICKS) ((uint32_t) (4294967296.0 / (TICKS)))
#define RS(V, WHAT) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sadd<V>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, INV(8000)); }); \
	printf("pass 3 shape  %-58s : %.3f ms  %.0f GB/s\n", WHAT, ms, (K + 2.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
#define RQ(V, WHAT) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_qdot<V>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, INV(8000)); }); \
	printf("pass 2 shape  %-58s : %.3f ms  %.0f GB/s\n", WHAT, ms, (K + 2.0) * 8.0 * n / ms / 1e6); fflush(stdout); }
	for (int rep = 0; rep < 3; rep++) {
		RS(0, "V0 as the product has it"); RS(1, "V1 four independent partial sums"); RS(2, "V2 next pack's loads before this pack's arithmetic"); RS(3, "V3 = V2 + V1");
		RQ(0, "V0 as the product has it"); RQ(2, "V2 next pack's loads before this pack's arithmetic");
	}
	return 0;
}
