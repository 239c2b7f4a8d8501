// profiles/src/tune12.hip -- round 5, VERDICT r04 task 5: is the guarded update cheaper INSIDE pass 3 now that stores are
// clock-phased?  pass 3 (r = r0 + sum c_j s_j: 21 reads, 1 write) followed by the update (x -= step r; x_sum += x: 3 reads,
// 2 writes) against ONE kernel that reads r0, the K rows, x and x_sum and parks three (or, without the direction, two) output
// streams in LDS.  The fused form saves the update's read of r (n words of 70 n) and one launch; it costs two more read
// streams and two more store streams in the pass that is already the widest.  Not part of the product.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tune12.hip -o tune12
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

// pass 3, all K rows of an iteration in registers (the shape of pass 2)
template <int NB, int PH, int OCC>
__global__ void __launch_bounds__(BLOCK, OCC) k_sadd(const double* S, size_t ld_, const double* coef, double* r, uint32_t n, double* parts, uint32_t inv)
{
	__shared__ double sh[2 * (BLOCK / 64)];
	__shared__ double cf[K];
	__shared__ d2 park[PH ? NB * BLOCK : 1];
	for (int e = threadIdx.x; e < K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	double acc0 = 0, acc1 = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, iters = (packs + stride - 1) / stride;
	int b = 0; uint32_t it_first = 0, epoch = PH ? epoch_now(inv) : 0;
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++) {
		const uint32_t p = it * stride + blockIdx.x * BLOCK + threadIdx.x;
		if (p < packs) {
			d2 v = ldd(r, (size_t) p * 2), f[K];
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = ldnt(S + (size_t) j * ld_, (size_t) p * 2);
			#pragma unroll
			for (int j = 0; j < K; j++) { v.x = fma(cf[j], f[j].x, v.x); v.y = fma(cf[j], f[j].y, v.y); }
			acc0 = fma(v.x, v.x, acc0); acc0 = fma(v.y, v.y, acc0);
			acc1 += (isfinite(v.x) ? 0.0 : 1.0) + (isfinite(v.y) ? 0.0 : 1.0);
			if (PH) park[b * BLOCK + threadIdx.x] = v; else st_stream(r + (size_t) p * 2, v);
		}
		if (PH) {
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
	}
	double t0 = wave_sum(acc0), t1 = wave_sum(acc1);
	if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = t0; sh[BLOCK / 64 + (threadIdx.x >> 6)] = t1; }
	__syncthreads();
	if (threadIdx.x == 0) { double a = 0, c = 0; for (int w = 0; w < BLOCK / 64; w++) { a += sh[w]; c += sh[BLOCK / 64 + w]; } parts[blockIdx.x] = a; parts[4096 + blockIdx.x] = c; }
}

// element-wise shapes: NR inputs (the first NWR of them are also outputs), U packs per lane in flight
// out_0 = in_0 - 0.5 * in_last ; out_1 = in_1 + out_0   (the update: x, x_sum, r)       [NR = 3, NWR = 2]
// out_0 = in_0 - 0.5 * in_1, dot with in_2                 (a backward sweep: q, y, s)      [NR = 3, NWR = 1]
template <int NR, int NWR, int NB, int PH, int U, int POL>
__global__ void __launch_bounds__(BLOCK) k_ew(double* a0, double* a1, const double* a2, uint32_t n, double* parts, uint32_t inv)
{
	__shared__ double sh[BLOCK / 64];
	__shared__ d2 park[PH ? NWR * NB * BLOCK : 1];
	double acc = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, iters = (packs + U * stride - 1) / (U * stride);
	int b = 0; uint32_t it_first = 0, epoch = PH ? epoch_now(inv) : 0;
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++) {
		d2 x0[U], x1[U], x2[U];
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = (it * U + u) * stride + blockIdx.x * BLOCK + threadIdx.x;
			if (p < packs) { x0[u] = ldd(a0, (size_t) p * 2); x1[u] = (NWR == 2) ? ldd(a1, (size_t) p * 2) : ldnt(a1, (size_t) p * 2); x2[u] = (NWR == 2) ? ldd(a2, (size_t) p * 2) : ldnt(a2, (size_t) p * 2); }
		}
		#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t p = (it * U + u) * stride + blockIdx.x * BLOCK + threadIdx.x;
			if (p < packs) {
				d2 o0, o1;
				if (NWR == 2) { o0.x = fma(-0.5, x2[u].x, x0[u].x); o0.y = fma(-0.5, x2[u].y, x0[u].y); o1.x = x1[u].x + o0.x; o1.y = x1[u].y + o0.y; }
				else { o0.x = fma(-0.5, x1[u].x, x0[u].x); o0.y = fma(-0.5, x1[u].y, x0[u].y); acc = fma(o0.x, x2[u].x, acc); acc = fma(o0.y, x2[u].y, acc); }
				if (PH) { park[((b * U + u) * NWR) * BLOCK + threadIdx.x] = o0; if (NWR == 2) park[((b * U + u) * NWR + 1) * BLOCK + threadIdx.x] = o1; }
				else if (POL) { st_stream(a0 + (size_t) p * 2, o0); if (NWR == 2) st_stream(a1 + (size_t) p * 2, o1); }
				else { st_plain(a0 + (size_t) p * 2, o0); if (NWR == 2) st_plain(a1 + (size_t) p * 2, o1); }
			}
		}
		if (PH) {
			b++;
			bool flush = (b + 1) * U > NB || it == iters - 1;
			const uint32_t e = epoch_now(inv);
			if (e != epoch) { flush = true; epoch = e; }
			if (flush) {
				#pragma unroll 1
				for (int bb = 0; bb < b * U; bb++) {
					const uint32_t pp = (it_first * U + bb) * stride + blockIdx.x * BLOCK + threadIdx.x;
					if (pp < packs) {
						if (POL) { st_stream(a0 + (size_t) pp * 2, park[(bb * NWR) * BLOCK + threadIdx.x]); if (NWR == 2) st_stream(a1 + (size_t) pp * 2, park[(bb * NWR + 1) * BLOCK + threadIdx.x]); }
						else { st_plain(a0 + (size_t) pp * 2, park[(bb * NWR) * BLOCK + threadIdx.x]); if (NWR == 2) st_plain(a1 + (size_t) pp * 2, park[(bb * NWR + 1) * BLOCK + threadIdx.x]); }
					}
				}
				b = 0; it_first = it + 1;
			}
		}
	}
	double t0 = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t0;
	__syncthreads();
	if (threadIdx.x == 0) { double a = 0; for (int w = 0; w < BLOCK / 64; w++) a += sh[w]; parts[blockIdx.x] = a; }
}


// pass 3 + the update in one kernel: NS parked output streams (r, x, x_sum | x, x_sum), NB slots each
template <int NB, int NS>
__global__ void __launch_bounds__(BLOCK, 1) k_sadd_fused(const double* S, size_t ld_, const double* coef, double* r, double* x, double* xs, uint32_t n, double* parts, uint32_t inv0)
{
	__shared__ double sh[2 * (BLOCK / 64)];
	__shared__ double cf[K];
	__shared__ d2 park[NS * NB * BLOCK];
	for (int e = threadIdx.x; e < K; e += BLOCK) cf[e] = coef[e];
	__syncthreads();
	const uint32_t inv = inv0 * NS;                      // fewer slots per stream: a shorter period, like the product's Parked<W, NS>
	double acc0 = 0, acc1 = 0;
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK, iters = (packs + stride - 1) / stride;
	int b = 0; uint32_t it_first = 0, epoch = epoch_now(inv);
	#pragma unroll 1
	for (uint32_t it = 0; it < iters; it++) {
		const uint32_t p = it * stride + blockIdx.x * BLOCK + threadIdx.x;
		if (p < packs) {
			d2 v = ldd(r, (size_t) p * 2), f[K];
			const d2 xv = ldd(x, (size_t) p * 2), sv = ldd(xs, (size_t) p * 2);
			#pragma unroll
			for (int j = 0; j < K; j++) f[j] = ldnt(S + (size_t) j * ld_, (size_t) p * 2);
			#pragma unroll
			for (int j = 0; j < K; j++) { v.x = fma(cf[j], f[j].x, v.x); v.y = fma(cf[j], f[j].y, v.y); }
			acc0 = fma(v.x, v.x, acc0); acc0 = fma(v.y, v.y, acc0);
			acc1 += (isfinite(v.x) ? 0.0 : 1.0) + (isfinite(v.y) ? 0.0 : 1.0);
			d2 xn, sn;
			xn.x = fma(-0.5, v.x, xv.x); xn.y = fma(-0.5, v.y, xv.y);
			sn.x = sv.x + xn.x; sn.y = sv.y + xn.y;
			int s = 0;
			if (NS == 3) park[(b * NS + s++) * BLOCK + threadIdx.x] = v;
			park[(b * NS + s++) * BLOCK + threadIdx.x] = xn;
			park[(b * NS + s) * BLOCK + threadIdx.x] = sn;
		}
		b++;
		bool flush = b == NB || it == iters - 1;
		const uint32_t e = epoch_now(inv);
		if (e != epoch) { flush = true; epoch = e; }
		if (flush) {
			#pragma unroll 1
			for (int bb = 0; bb < b; bb++) {
				const uint32_t pp = (it_first + bb) * stride + blockIdx.x * BLOCK + threadIdx.x;
				if (pp < packs) {
					int s = 0;
					if (NS == 3) st_stream(r + (size_t) pp * 2, park[(bb * NS + s++) * BLOCK + threadIdx.x]);
					st_plain(x + (size_t) pp * 2, park[(bb * NS + s++) * BLOCK + threadIdx.x]);
					st_plain(xs + (size_t) pp * 2, park[(bb * NS + s) * BLOCK + threadIdx.x]);
				}
			}
			b = 0; it_first = it + 1;
		}
	}
	double t0 = wave_sum(acc0), t1 = wave_sum(acc1);
	if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = t0; sh[BLOCK / 64 + (threadIdx.x >> 6)] = t1; }
	__syncthreads();
	if (threadIdx.x == 0) { double a = 0, c = 0; for (int w = 0; w < BLOCK / 64; w++) { a += sh[w]; c += sh[BLOCK / 64 + w]; } parts[blockIdx.x] = a; parts[4096 + blockIdx.x] = c; }
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
	double *S, *g, *x, *xs, *parts, *coef;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8)); CK(hipMalloc(&x, (size_t) n * 8)); CK(hipMalloc(&xs, (size_t) n * 8));
	CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(x, 0, (size_t) n * 8)); CK(hipMemset(xs, 0, (size_t) n * 8)); CK(hipMemset(coef, 0, 64 * 8));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("n = %u, K = %d, %d CUs\n", n, K, cus);
#define INV(TICKS) ((uint32_t) (4294967296.0 / (TICKS)))
	for (int rep = 0; rep < 3; rep++) {
		// the two launches of the product, back to back (what one step pays for pass 3 + the update)
		for (int ugrid : {2, 3}) {
			double ms = time_ms([&](int) {
				hipLaunchKernelGGL((k_sadd<32, 1, 1>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, INV(8000));
				hipLaunchKernelGGL((k_ew<3, 2, 16, 0, 2, 0>), dim3(ugrid * cus), dim3(BLOCK), 0, 0, x, xs, g, n, parts, INV(8000)); });
			printf("separate: pass 3 (phased, 32 slots) + update (stores at once, default policy, grid %d/CU) : %.3f ms\n", ugrid, ms); fflush(stdout);
		}
		{
			double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sadd<32, 1, 1>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, n, parts, INV(8000)); });
			printf("          pass 3 alone : %.3f ms\n", ms);
			ms = time_ms([&](int) { hipLaunchKernelGGL((k_ew<3, 2, 16, 0, 2, 0>), dim3(2 * cus), dim3(BLOCK), 0, 0, x, xs, g, n, parts, INV(8000)); });
			printf("          update alone : %.3f ms\n", ms); fflush(stdout);
		}
#define RF(NB, NS, TICKS) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_sadd_fused<NB, NS>), dim3(cus), dim3(BLOCK), 0, 0, S, (size_t) n, coef, g, x, xs, n, parts, INV(TICKS)); }); \
	printf("fused   : %d output streams, %d slots each, period %d ticks (x NS shorter in the kernel) : %.3f ms\n", NS, NB, (int) (TICKS), ms); fflush(stdout); }
		RF(10, 3, 8000); RF(10, 3, 4000); RF(10, 3, 16000); RF(8, 3, 8000);
		RF(16, 2, 8000); RF(16, 2, 4000); RF(16, 2, 16000);
	}
	return 0;
}
