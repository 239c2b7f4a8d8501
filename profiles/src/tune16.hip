// profiles/src/tune16.hip -- round 6: the C2 step (oLBFGS, n = 1e7, m = 10) as ONE launch.
// The product runs a step's two-loop recursion as three dependent kernels (pass 1: g + 10 rows of S + the new y -> 20 sums;
// pass 2: g + 10 rows of Y -> r0, 10 sums; pass 3: r0 + 10 rows of S -> r, 2 sums) and the guarded update as a fourth; every
// kernel begins with every workgroup re-adding the previous kernel's per-workgroup partial sums (`total_of`) and ends with a
// workgroup reduction per quantity.  Round 5 priced the kernel boundaries only with EMPTY kernels.  Here the same four shapes,
// with the same prologues and epilogues, run (A) as four launches -- at the product's grids (3 / 1 / 1 / 2 workgroups per CU)
// and at one common grid -- and (B) as ONE cooperative launch of that common grid whose phases are separated by a grid barrier
// (ticket counter, agent-scope fences: every workgroup's partials and its part of r0 / r written back before anybody reads
// them).  Same loads, same stores, same sums in both.
// hipcc --offload-arch=gfx950 -O3 tune16.hip -o tune16
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int K = 10, kBlock = 256, kWaves = 4, kMaxGrid = 2048;

__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ __forceinline__ double block_sum(double v, double* sh)
{
	v = wave_sum(v);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// totals of `nq` quantities over `count` partials each (quantity q at parts[q * kMaxGrid ..]), every workgroup for itself
__device__ __forceinline__ void totals(const double* parts, int nq, int count, double* out, double* sh)
{
	for (int q = 0; q < nq; q++) {
		double a = 0;
		for (int i = threadIdx.x; i < count; i += kBlock) a += parts[(size_t) q * kMaxGrid + i];
		const double t = block_sum(a, sh);
		if (threadIdx.x == 0) out[q] = t;
	}
	__syncthreads();
}

struct Args {
	const double* S; const double* Y; const double* ynew; double* g; double* gprev; double* x; double* sslot;
	double* part1; double* part2; double* part3; unsigned* ticket; uint32_t n; double step;
};

// pass 1: 20 sums (s_j'g, s_j'y_new), g copied to g_prev
__device__ __forceinline__ void phase1(const Args& a, uint32_t nb, uint32_t b, double* sh)
{
	double acc[2 * K];
	for (int j = 0; j < 2 * K; j++) acc[j] = 0;
	const uint32_t packs = a.n / 2, stride = nb * kBlock;
	for (uint32_t p = b * kBlock + threadIdx.x; p < packs; p += stride) {
		const d2 gv = *reinterpret_cast<const d2*>(a.g + (size_t) p * 2), yv = *reinterpret_cast<const d2*>(a.ynew + (size_t) p * 2);
		d2 f[K];
		#pragma unroll
		for (int j = 0; j < K; j++) f[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.S + (size_t) j * a.n + (size_t) p * 2));
		*reinterpret_cast<d2*>(a.gprev + (size_t) p * 2) = gv;
		#pragma unroll
		for (int j = 0; j < K; j++) { acc[j] = fma(f[j].x, gv.x, fma(f[j].y, gv.y, acc[j])); acc[K + j] = fma(f[j].x, yv.x, fma(f[j].y, yv.y, acc[K + j])); }
	}
	for (int j = 0; j < 2 * K; j++) { const double t = block_sum(acc[j], sh); if (threadIdx.x == 0) a.part1[(size_t) j * kMaxGrid + b] = t; }
}
// pass 2: coefficients from pass 1's totals; q = g - sum cf_j y_j; r0 = cf q; 10 sums y_j'r0; r0 replaces g
__device__ __forceinline__ void phase2(const Args& a, uint32_t nb1, uint32_t nb, uint32_t b, double* sh, double* cf)
{
	totals(a.part1, 2 * K, (int) nb1, cf, sh);
	double acc[K];
	for (int j = 0; j < K; j++) acc[j] = 0;
	const uint32_t packs = a.n / 2, stride = nb * kBlock, last = packs - 1;
	for (uint32_t pp = b * kBlock + threadIdx.x; pp < packs; pp += stride) {
		const uint32_t p = last - pp;                                       // the product alternates the direction of the traversal
		d2 q = *reinterpret_cast<const d2*>(a.g + (size_t) p * 2);
		d2 f[K];
		#pragma unroll
		for (int j = 0; j < K; j++) f[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.Y + (size_t) j * a.n + (size_t) p * 2));
		#pragma unroll
		for (int j = K - 1; j >= 0; j--) { q.x = fma(-1e-9 * cf[j], f[j].x, q.x); q.y = fma(-1e-9 * cf[j], f[j].y, q.y); }
		q.x *= 0.5; q.y *= 0.5;
		#pragma unroll
		for (int j = 0; j < K; j++) acc[j] = fma(f[j].x, q.x, fma(f[j].y, q.y, acc[j]));
		*reinterpret_cast<d2*>(a.g + (size_t) p * 2) = q;
	}
	for (int j = 0; j < K; j++) { const double t = block_sum(acc[j], sh); if (threadIdx.x == 0) a.part2[(size_t) j * kMaxGrid + b] = t; }
}
// pass 3: r = r0 + sum c_j s_j; guard sums
__device__ __forceinline__ void phase3(const Args& a, uint32_t nb2, uint32_t nb, uint32_t b, double* sh, double* cf)
{
	totals(a.part2, K, (int) nb2, cf, sh);
	double s0 = 0, s1 = 0;
	const uint32_t packs = a.n / 2, stride = nb * kBlock;
	for (uint32_t p = b * kBlock + threadIdx.x; p < packs; p += stride) {
		d2 v = *reinterpret_cast<const d2*>(a.g + (size_t) p * 2);
		d2 f[K];
		#pragma unroll
		for (int j = 0; j < K; j++) f[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.S + (size_t) j * a.n + (size_t) p * 2));
		#pragma unroll
		for (int j = 0; j < K; j++) { v.x = fma(1e-9 * cf[j], f[j].x, v.x); v.y = fma(1e-9 * cf[j], f[j].y, v.y); }
		s0 = fma(v.x, v.x, fma(v.y, v.y, s0)); s1 += (isfinite(v.x) ? 0.0 : 1.0) + (isfinite(v.y) ? 0.0 : 1.0);
		*reinterpret_cast<d2*>(a.g + (size_t) p * 2) = v;
	}
	const double t0 = block_sum(s0, sh), t1 = block_sum(s1, sh);
	if (threadIdx.x == 0) { a.part3[b] = t0; a.part3[kMaxGrid + b] = t1; }
}
// the guarded update: x -= step r; s_slot = grad = -step r   (2 read, 3 written)
__device__ __forceinline__ void phase4(const Args& a, uint32_t nb3, uint32_t nb, uint32_t b, double* sh, double* cf)
{
	totals(a.part3, 2, (int) nb3, cf, sh);
	if (cf[1] > 0 || !(cf[0] < 1e300)) return;
	const uint32_t packs = a.n / 2, stride = nb * kBlock, last = packs - 1;
	for (uint32_t pp = b * kBlock + threadIdx.x; pp < packs; pp += stride) {
		const uint32_t p = last - pp;
		const d2 r = *reinterpret_cast<const d2*>(a.g + (size_t) p * 2);
		d2 xv = *reinterpret_cast<const d2*>(a.x + (size_t) p * 2);
		const d2 sv = {-a.step * r.x, -a.step * r.y};
		xv.x += sv.x; xv.y += sv.y;
		*reinterpret_cast<d2*>(a.x + (size_t) p * 2) = xv;
		*reinterpret_cast<d2*>(a.sslot + (size_t) p * 2) = sv;
		*reinterpret_cast<d2*>(a.g + (size_t) p * 2) = sv;
	}
}

__global__ void __launch_bounds__(kBlock) k_p1(Args a) { __shared__ double sh[kWaves]; phase1(a, gridDim.x, blockIdx.x, sh); }
__global__ void __launch_bounds__(kBlock) k_p2(Args a, uint32_t nb1) { __shared__ double sh[kWaves]; __shared__ double cf[2 * K]; phase2(a, nb1, gridDim.x, blockIdx.x, sh, cf); }
__global__ void __launch_bounds__(kBlock) k_p3(Args a, uint32_t nb2) { __shared__ double sh[kWaves]; __shared__ double cf[2 * K]; phase3(a, nb2, gridDim.x, blockIdx.x, sh, cf); }
__global__ void __launch_bounds__(kBlock) k_p4(Args a, uint32_t nb3) { __shared__ double sh[kWaves]; __shared__ double cf[2 * K]; phase4(a, nb3, gridDim.x, blockIdx.x, sh, cf); }

// every workgroup of the grid has arrived -- and what it wrote before is visible device-wide -- when this returns.  The counter
// only grows (round r of the launch is complete at r * gridDim.x); zeroed by the host before the launch.
__device__ __forceinline__ void grid_barrier(unsigned* ticket, unsigned round)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();                                                // release: this XCD's L2 written back
		atomicAdd(ticket, 1u);
		const unsigned target = round * gridDim.x;
		while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
		__threadfence();                                                // acquire: nothing stale is read from this XCD's L2
	}
	__syncthreads();
}

// phases 2 - 4 run on the first `nb234` workgroups only when sub != 0 (the product's 1-per-CU grids inside a 3-per-CU launch)
__global__ void __launch_bounds__(kBlock) k_fused(Args a, uint32_t nb2, uint32_t nb3, uint32_t nb4)
{
	__shared__ double sh[kWaves];
	__shared__ double cf[2 * K];
	const uint32_t b = blockIdx.x;
	phase1(a, gridDim.x, b, sh);
	grid_barrier(a.ticket, 1);
	if (b < nb2) phase2(a, gridDim.x, nb2, b, sh, cf);
	grid_barrier(a.ticket, 2);
	if (b < nb3) phase3(a, nb2, nb3, b, sh, cf);
	grid_barrier(a.ticket, 3);
	if (b < nb4) phase4(a, nb3, nb4, b, sh, cf);
}

int main(int argc, char** argv)
{
	const uint32_t n = argc > 1 ? (uint32_t) atof(argv[1]) : 10000000u;
	hipStream_t s; CK(hipStreamCreate(&s));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	Args a{};
	double *S, *Y, *yn, *g, *gp, *x, *ss, *p1, *p2, *p3, *g0; unsigned* tk;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&Y, (size_t) K * n * 8));
	for (double** v : {&yn, &g, &gp, &x, &ss, &g0}) CK(hipMalloc(v, (size_t) n * 8));
	CK(hipMalloc(&p1, (size_t) 2 * K * kMaxGrid * 8)); CK(hipMalloc(&p2, (size_t) K * kMaxGrid * 8)); CK(hipMalloc(&p3, (size_t) 2 * kMaxGrid * 8)); CK(hipMalloc(&tk, 4));
	std::vector<double> h((size_t) n);
	for (size_t i = 0; i < n; i++) h[i] = 1e-3 * (double) ((i * 2654435761u) % 1000) - 0.5;
	for (int j = 0; j < K; j++) { CK(hipMemcpy(S + (size_t) j * n, h.data(), (size_t) n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(Y + (size_t) j * n, h.data(), (size_t) n * 8, hipMemcpyHostToDevice)); }
	for (double* v : {yn, g0, x}) CK(hipMemcpy(v, h.data(), (size_t) n * 8, hipMemcpyHostToDevice));
	a.S = S; a.Y = Y; a.ynew = yn; a.g = g; a.gprev = gp; a.x = x; a.sslot = ss; a.part1 = p1; a.part2 = p2; a.part3 = p3; a.ticket = tk; a.n = n; a.step = 1e-12;
	int fused_per_cu = 0;
	CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fused_per_cu, k_fused, kBlock, 0));
	printf("---- n = %u, m = %d, %d CUs; the fused kernel fits %d workgroups per CU\n", n, K, cus, fused_per_cu);
	const double bytes = (13.0 + 12.0 + 12.0 + 5.0) * n * 8;
	double check[8]; int nck = 0;
	auto bench = [&](const char* what, auto&& step) {
		CK(hipMemcpyAsync(g, g0, (size_t) n * 8, hipMemcpyDeviceToDevice, s));
		step();
		CK(hipStreamSynchronize(s));
		double first[2];
		CK(hipMemcpy(first, g, 16, hipMemcpyDeviceToHost));
		if (nck < 8) check[nck++] = first[0];
		for (int i = 0; i < 20; i++) step();
		CK(hipStreamSynchronize(s));
		const int reps = 300;
		double best = 1e30, sum = 0;
		for (int rep = 0; rep < 5; rep++) {
			auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < reps; i++) { step(); CK(hipStreamSynchronize(s)); }     // the ABI is synchronous: one wait per call
			const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps * 1e6;
			best = us < best ? us : best; sum += us;
		}
		printf("  %-86s best %8.2f us  mean %8.2f us  %6.0f GB/s\n", what, best, sum / 5, bytes / best / 1e3);
	};
	auto four = [&](int g1, int g2, int g3, int g4) {
		hipLaunchKernelGGL(k_p1, dim3(g1), dim3(kBlock), 0, s, a);
		hipLaunchKernelGGL(k_p2, dim3(g2), dim3(kBlock), 0, s, a, (uint32_t) g1);
		hipLaunchKernelGGL(k_p3, dim3(g3), dim3(kBlock), 0, s, a, (uint32_t) g2);
		hipLaunchKernelGGL(k_p4, dim3(g4), dim3(kBlock), 0, s, a, (uint32_t) g3);
	};
	auto fused = [&](int grid, int g2, int g3, int g4) {
		CK(hipMemsetAsync(tk, 0, 4, s));
		uint32_t n2 = (uint32_t) g2, n3 = (uint32_t) g3, n4 = (uint32_t) g4;
		void* args[] = {&a, &n2, &n3, &n4};
		CK(hipLaunchCooperativeKernel((const void*) k_fused, dim3(grid), dim3(kBlock), args, 0, s));
	};
	bench("A  four launches, the product's grids (3 / 1 / 1 / 2 workgroups per CU)", [&] { four(3 * cus, cus, cus, 2 * cus); });
	for (int per = 1; per <= 3 && per <= fused_per_cu; per++) {
		char w[160];
		snprintf(w, sizeof w, "A' four launches, every grid %d per CU", per);
		bench(w, [&] { four(per * cus, per * cus, per * cus, per * cus); });
		snprintf(w, sizeof w, "B  ONE cooperative launch, %d per CU, grid barriers between the phases", per);
		bench(w, [&] { fused(per * cus, per * cus, per * cus, per * cus); });
	}
	if (fused_per_cu >= 3) bench("B* ONE cooperative launch of 3 per CU; passes 2, 3 on the first 1 per CU, the update on 2 per CU", [&] { fused(3 * cus, cus, cus, 2 * cus); });
	printf("  first element of the direction after one step, per variant (same grids => same bits):");
	for (int i = 0; i < nck; i++) printf(" %.17g", check[i]);
	printf("\n");
	return 0;
}
