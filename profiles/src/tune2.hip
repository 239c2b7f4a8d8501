// profiles/src/tune2.hip -- grid / unroll sweep per kernel SHAPE (reads x writes); not part of the product.
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

struct Ptrs { double* a[6]; };   // a[0..NR-1] read (a[0] default policy, rest nt), a[0..NW-1] written back

template <int NR, int NW, int U>
__global__ void __launch_bounds__(256) k_gen(Ptrs P, uint32_t n, int rev, double* parts)
{
	__shared__ double sh[4];
	const uint32_t packs = n / 2, stride = gridDim.x * 256, last = packs - 1;
	double acc = 0;
	uint32_t p = blockIdx.x * 256 + threadIdx.x;
	for (; p + (U - 1) * stride < packs; p += U * stride) {
		d2 v[U][NR];
		#pragma unroll
		for (int u = 0; u < U; u++) {
			uint32_t i = p + u * stride; if (rev) i = last - i;
			v[u][0] = ld<false>(P.a[0], (size_t) i * 2);
			#pragma unroll
			for (int r = 1; r < NR; r++) v[u][r] = (r < NW) ? ld<false>(P.a[r], (size_t) i * 2) : ld<true>(P.a[r], (size_t) i * 2);
		}
		#pragma unroll
		for (int u = 0; u < U; u++) {
			uint32_t i = p + u * stride; if (rev) i = last - i;
			d2 o = v[u][0];
			#pragma unroll
			for (int r = 1; r < NR; r++) { o.x = fma(1e-3, v[u][r].x, o.x); o.y = fma(1e-3, v[u][r].y, o.y); }
			acc = fma(o.x, o.y, acc);
			#pragma unroll
			for (int w = 0; w < NW; w++) st(P.a[w], (size_t) i * 2, o);
		}
	}
	for (; p < packs; p += stride) {
		uint32_t i = rev ? last - p : p;
		d2 o = ld<false>(P.a[0], (size_t) i * 2);
		#pragma unroll
		for (int r = 1; r < NR; r++) { d2 t = ld<true>(P.a[r], (size_t) i * 2); o.x = fma(1e-3, t.x, o.x); o.y = fma(1e-3, t.y, o.y); }
		acc = fma(o.x, o.y, acc);
		#pragma unroll
		for (int w = 0; w < NW; w++) st(P.a[w], (size_t) i * 2, o);
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) parts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <class F> double time_ms(F&& launch, int reps = 12)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 4; i++) launch(i);
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
	const int rows = 12;
	double* buf; double* parts;
	CK(hipMalloc(&buf, (size_t) rows * n * 8)); CK(hipMalloc(&parts, 1 << 20));
	CK(hipMemset(buf, 0, (size_t) rows * n * 8));
	auto row = [&](int r) { return buf + (size_t) (r % rows) * n; };
	printf("n = %u\n", n);
#define RUN(NR, NW, U, GRID, REV)                                                                                         \
	{                                                                                                                     \
		double ms = time_ms([&](int i) {                                                                                  \
			Ptrs P;                                                                                                       \
			for (int r = 0; r < 6; r++) P.a[r] = (r < NW) ? row(r) : row(NW + ((i * 3 + r) % (rows - NW)));               \
			hipLaunchKernelGGL((k_gen<NR, NW, U>), dim3(GRID), dim3(256), 0, 0, P, n, (REV) ? (i & 1) : 0, parts);          \
		});                                                                                                               \
		printf("R%d W%d U%d grid %-5d rev %d : %.3f ms  %.0f GB/s\n", NR, NW, U, GRID, REV, ms, (NR + NW) * 8.0 * n / ms / 1e6); \
	}
#define SHAPE(NR, NW)                                                         \
	for (int g : {256, 512, 768, 1024, 1536, 2048})                           \
		for (int rev : {0, 1}) {                                              \
			RUN(NR, NW, 1, g, rev); RUN(NR, NW, 2, g, rev); RUN(NR, NW, 4, g, rev); \
		}
	SHAPE(2, 0)
	SHAPE(2, 1)
	SHAPE(3, 1)
	SHAPE(3, 2)
	SHAPE(3, 3)
	return 0;
}
