// scratch/tune7.hip -- pass B (combine): cache-policy bits of the single store stream (gfx942/gfx950: sc0, sc1, nt),
// batch depth T and workgroups per CU.  Not part of the product; round-2 tuning.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ldnt(const double* p, size_t i) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)); }
__device__ __forceinline__ d2 ldd(const double* p, size_t i) { return *reinterpret_cast<const d2*>(p + i); }
__device__ __forceinline__ void st(double* p, size_t i, d2 v) { *reinterpret_cast<d2*>(p + i) = v; }
__device__ __forceinline__ double wave_sum(double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }

template <int POL> __device__ __forceinline__ void st_pol(double* p, size_t i, d2 v)
{
	d2* q = reinterpret_cast<d2*>(p + i);
	if (POL == 0) *q = v;
	else if (POL == 1) __builtin_nontemporal_store(v, q);
	else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
	else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(v) : "memory");
	else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
	else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(v) : "memory");
	else if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(q), "v"(v) : "memory");
}
constexpr int K = 20;
constexpr uint32_t kSpinLimit = 1u << 24;

template <int G> __device__ __forceinline__ d2 combine_pack(const double* S, const double* Y, size_t ld_, const double* cf, const double* g, size_t i)
{
	d2 r = ldd(g, i);
	r.x *= cf[0]; r.y *= cf[0];
	#pragma unroll
	for (int j0 = 0; j0 < K; j0 += G) {
		d2 fy[G], fs[G];
		#pragma unroll
		for (int u = 0; u < G; u++) { fy[u] = ldnt(Y + (size_t) (j0 + u) * ld_, i); fs[u] = ldnt(S + (size_t) (j0 + u) * ld_, i); }
		#pragma unroll
		for (int u = 0; u < G; u++) {
			r.x = fma(cf[1 + j0 + u], fy[u].x, r.x); r.y = fma(cf[1 + j0 + u], fy[u].y, r.y);
			r.x = fma(cf[1 + K + j0 + u], fs[u].x, r.x); r.y = fma(cf[1 + K + j0 + u], fs[u].y, r.y);
		}
	}
	return r;
}

// baseline: the product's shape (T packs finished per lane, then stored); MODE 1 = no store at all
template <int BLOCK, int G, int T, int MODE>
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
				d2 r = combine_pack<G>(S, Y, ld_, cf, g, (size_t) p * 2);
				acc = fma(r.x, r.x, acc); acc = fma(r.y, r.y, acc);
				out[t] = r;
			}
		}
		if (MODE >= 0) {
			#pragma unroll
			for (int t = 0; t < T; t++) {
				const uint32_t p = p0 + t * stride;
				if (p < packs) st_pol<MODE>(g, (size_t) p * 2, out[t]);
			}
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

__global__ void k_fill(double* p, size_t n, double scale, uint64_t salt)
{
	for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		uint64_t z = (i + salt) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
		p[i] = scale * ((double) (z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
	}
}

int main(int argc, char** argv)
{
	const uint32_t n = argc > 1 ? (uint32_t) atof(argv[1]) : 100000000u;
	double *S, *Y, *g, *g0, *ref, *parts, *coef; int* err = nullptr;
	CK(hipMalloc(&S, (size_t) K * n * 8)); CK(hipMalloc(&Y, (size_t) K * n * 8)); CK(hipMalloc(&g, (size_t) n * 8));
	CK(hipMalloc(&g0, (size_t) n * 8)); CK(hipMalloc(&ref, (size_t) n * 8));
	CK(hipMalloc(&parts, 4096 * 64 * 8)); CK(hipMalloc(&coef, 64 * 8)); 
	hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, S, (size_t) K * n, 1e-3, 1ull);
	hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, Y, (size_t) K * n, 1e-3, 77ull);
	hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, g0, (size_t) n, 1.0, 123ull);
	double hc[64]; for (int i = 0; i < 64; i++) hc[i] = 0.01 * (i % 7 - 3); hc[0] = 0.9;
	CK(hipMemcpy(coef, hc, sizeof hc, hipMemcpyHostToDevice));
	CK(hipDeviceSynchronize());
	printf("n = %u, K = %d\n", n, K);
	// reference result (one application of the combine to g0)
	CK(hipMemcpy(g, g0, (size_t) n * 8, hipMemcpyDeviceToDevice));
	hipLaunchKernelGGL((k_combine_batched<256, 4, 8, 0>), dim3(512), dim3(256), 0, 0, S, Y, (size_t) n, coef, g, n, parts);
	CK(hipMemcpy(ref, g, (size_t) n * 8, hipMemcpyDeviceToDevice));
	auto check = [&](const char* what) {       // g must equal ref after ONE application to g0
		std::vector<double> a(1 << 16), b(1 << 16);
		size_t bad = 0;
		for (size_t off : {(size_t) 0, (size_t) n / 2, (size_t) n - (1 << 16)}) {
			CK(hipMemcpy(a.data(), g + off, a.size() * 8, hipMemcpyDeviceToHost));
			CK(hipMemcpy(b.data(), ref + off, b.size() * 8, hipMemcpyDeviceToHost));
			for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
		}
		printf("   check %s: %zu mismatches\n", what, bad);
	};
#define RUNB(BLOCK, G, T, GRID, MODE) { double ms = time_ms([&](int i) { hipLaunchKernelGGL((k_combine_batched<BLOCK, G, T, MODE>), dim3(GRID), dim3(BLOCK), 0, 0, S, Y, (size_t) n, coef, g, n, parts); }); \
	printf("batched B%-4d G%-2d T%-2d grid %-5d mode %d : %.3f ms  %.0f GB/s\n", BLOCK, G, T, GRID, MODE, ms, (2.0 * K + 2) * 8.0 * n / ms / 1e6); fflush(stdout); }
	RUNB(256, 4, 8, 512, -1);
	RUNB(256, 4, 8, 512, 0); RUNB(256, 4, 8, 512, 1); RUNB(256, 4, 8, 512, 2); RUNB(256, 4, 8, 512, 3); RUNB(256, 4, 8, 512, 4); RUNB(256, 4, 8, 512, 5); RUNB(256, 4, 8, 512, 6);
	RUNB(256, 4, 16, 512, 0); RUNB(256, 4, 16, 512, 3); RUNB(256, 4, 4, 768, 0); RUNB(256, 4, 4, 768, 3); RUNB(256, 4, 8, 256, 0); RUNB(256, 4, 8, 1024, 0);
	RUNB(256, 2, 8, 512, 0); RUNB(256, 5, 8, 512, 0); RUNB(256, 10, 8, 512, 0); RUNB(512, 4, 8, 256, 0); RUNB(512, 4, 4, 512, 0); RUNB(128, 4, 8, 1024, 0);
	return 0;
}
