// profiles/src/tune4.hip -- phase ablation of the diagonal-H0 Gram pass (k_gram_h0); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int K = 20, kBlock = 256, kWaves = 4, kTile = 128, kTileLd = 129, kShare = 5;

template <bool ACC, bool LOAD, bool PREFETCH, int ACCMODE>
__global__ void __launch_bounds__(kBlock) k_gram(const double* S, const double* Y, size_t ld_, const double* g, double* G, double* H0, uint32_t n, double* parts)
{
	extern __shared__ double L[];
	const int k = K, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int rowG = 3 * k, rowH = 3 * k + 1, Q = 3 * k + k * (k + 1) / 2;
	int qa[2], qb[2], nq = 0;
	for (int q = threadIdx.x; q < Q; q += kBlock) {
		int ra, rb;
		if (q < 3 * k) { ra = q; rb = rowG; }
		else { int t = q - 3 * k, i = 0; while (t >= k - i) { t -= k - i; i++; } ra = 2 * k + i; rb = k + i + t; }
		qa[nq] = ra; qb[nq] = rb; nq++;
	}
	double acc[2] = {0, 0}, acc2[2] = {0, 0};
	const uint32_t tiles = (n + kTile - 1) / kTile;
	d2 sv[kShare], yv[kShare], gv = {0, 0}, Gv = {1, 1};
	auto fetch = [&](uint32_t tile) {
		const uint32_t i = tile * kTile + 2 * lane;
		const bool in = LOAD && tile < tiles && i + 1 < n;
		#pragma unroll
		for (int u = 0; u < kShare; u++) {
			const int r = wave + u * kWaves;
			sv[u] = in ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(S + (size_t) r * ld_ + i)) : d2{1, 1};
			yv[u] = in ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(Y + (size_t) r * ld_ + i)) : d2{1, 1};
		}
		if (wave == 0) { gv = in ? *reinterpret_cast<const d2*>(g + i) : d2{1, 1}; Gv = in ? *reinterpret_cast<const d2*>(G + i) : d2{1, 1}; }
	};
	if (PREFETCH) fetch(blockIdx.x);
	for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
		const uint32_t i = tile * kTile + 2 * lane;
		if (!PREFETCH) fetch(tile);
		d2 yk[kShare];
		#pragma unroll
		for (int u = 0; u < kShare; u++) {
			const int r = wave + u * kWaves;
			yk[u] = yv[u];
			L[r * kTileLd + 2 * lane] = sv[u].x; L[r * kTileLd + 2 * lane + 1] = sv[u].y;
			L[(k + r) * kTileLd + 2 * lane] = yv[u].x; L[(k + r) * kTileLd + 2 * lane + 1] = yv[u].y;
		}
		if (wave == 0) {
			d2 Gn, h;
			Gn.x = 0.9 * Gv.x + 0.1 * (gv.x * gv.x); Gn.y = 0.9 * Gv.y + 0.1 * (gv.y * gv.y);
			h.x = gv.x / sqrt(Gn.x + 1e-4); h.y = gv.y / sqrt(Gn.y + 1e-4);
			if (LOAD && i + 1 < n) { *reinterpret_cast<d2*>(G + i) = Gn; *reinterpret_cast<d2*>(H0 + i) = h; }
			L[rowG * kTileLd + 2 * lane] = gv.x; L[rowG * kTileLd + 2 * lane + 1] = gv.y;
			L[rowH * kTileLd + 2 * lane] = h.x; L[rowH * kTileLd + 2 * lane + 1] = h.y;
		}
		if (PREFETCH) fetch(tile + gridDim.x);
		__syncthreads();
		const double hx = L[rowH * kTileLd + 2 * lane], hy = L[rowH * kTileLd + 2 * lane + 1];
		#pragma unroll
		for (int u = 0; u < kShare; u++) {
			const int r = wave + u * kWaves;
			L[(2 * k + r) * kTileLd + 2 * lane] = yk[u].x * hx; L[(2 * k + r) * kTileLd + 2 * lane + 1] = yk[u].y * hy;
		}
		__syncthreads();
		if (ACC && nq > 0) {
			const double* A0 = L + qa[0] * kTileLd; const double* B0 = L + qb[0] * kTileLd;
			const double* A1 = L + qa[nq - 1] * kTileLd; const double* B1 = L + qb[nq - 1] * kTileLd;
			if (ACCMODE == 0) {
				#pragma unroll 8
				for (int e = 0; e < kTile; e += 2) {
					acc[0] = fma(A0[e], B0[e], acc[0]); acc2[0] = fma(A0[e + 1], B0[e + 1], acc2[0]);
					if (nq == 2) { acc[1] = fma(A1[e], B1[e], acc[1]); acc2[1] = fma(A1[e + 1], B1[e + 1], acc2[1]); }
				}
			} else {
				// 4 independent chains, unroll 16
				double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
				#pragma unroll 4
				for (int e = 0; e < kTile; e += 4) {
					c0 = fma(A0[e], B0[e], c0); c1 = fma(A0[e + 1], B0[e + 1], c1); c2 = fma(A0[e + 2], B0[e + 2], c2); c3 = fma(A0[e + 3], B0[e + 3], c3);
				}
				acc[0] += (c0 + c1) + (c2 + c3);
				if (nq == 2) { double d0 = 0; for (int e = 0; e < kTile; e++) d0 = fma(A1[e], B1[e], d0); acc[1] += d0; }
			}
		}
		__syncthreads();
	}
	int j = 0;
	for (int q = threadIdx.x; q < Q; q += kBlock) { parts[(size_t) q * 2048 + blockIdx.x] = acc[j] + acc2[j]; j++; }
}

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <class F> double time_ms(F&& launch, int reps = 5)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	launch(0); CK(hipDeviceSynchronize());
	std::vector<float> t;
	for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); launch(i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
	return median(t);
}
int main()
{
	const uint32_t n = 100000000u;
	double *S, *Y, *g, *G, *H0, *parts;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&Y, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8)); CK(hipMalloc(&G, (size_t) n * 8)); CK(hipMalloc(&H0, (size_t) n * 8));
	CK(hipMalloc(&parts, (size_t) 2048 * 400 * 8));
	CK(hipMemset(S, 0, (size_t) K * n * 8)); CK(hipMemset(Y, 0, (size_t) K * n * 8)); CK(hipMemset(g, 0, (size_t) n * 8)); CK(hipMemset(G, 0, (size_t) n * 8));
	const size_t shmem = (size_t) (3 * K + 2) * kTileLd * 8;
#define RUN(ACC, LOAD, PF, AM, GRID) { auto kern = k_gram<ACC, LOAD, PF, AM>; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem)); \
	int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBlock, shmem)); \
	double ms = time_ms([&](int) { hipLaunchKernelGGL(kern, dim3(GRID), dim3(kBlock), shmem, 0, S, Y, (size_t) n, g, G, H0, n, parts); }); \
	printf("acc %d load %d prefetch %d accmode %d grid %-4d occ %d : %.3f ms\n", ACC, LOAD, PF, AM, GRID, occ, ms); }
	for (int grid : {256, 512, 768}) {
		RUN(true, true, true, 0, grid); RUN(true, true, false, 0, grid); RUN(false, true, true, 0, grid); RUN(true, false, true, 0, grid);
		RUN(true, true, true, 1, grid); RUN(true, false, true, 1, grid);
	}
	return 0;
}
