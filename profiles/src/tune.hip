// profiles/src/tune.hip -- access-pattern sweep for the fused backward kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off profiles/src/tune.hip -o gpurun_out/tune && gpurun_out/tune
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const double* p, size_t i)
{
	return NT ? __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)) : *reinterpret_cast<const d2*>(p + i);
}
template <bool NT> __device__ __forceinline__ void st(double* p, size_t i, d2 v)
{
	if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p + i)); else *reinterpret_cast<d2*>(p + i) = v;
}

__device__ __forceinline__ double wave_sum(double v)
{
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

// mode 0: grid-stride; mode 1: contiguous chunk per block; rev: traverse from the end
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(BLOCK) k_bwd(const double* __restrict__ y, double* q, const double* __restrict__ s,
                                                double alpha, uint32_t n, int mode, int rev, double* parts)
{
	__shared__ double sh[BLOCK / 64];
	const uint32_t packs = n / 2;
	double acc = 0;
	if (mode == 0) {
		const uint32_t stride = gridDim.x * BLOCK;
		uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
		for (; p + (U - 1) * stride < packs; p += U * stride) {
			d2 yy[U], qq[U], ss[U];
			#pragma unroll
			for (int u = 0; u < U; u++) {
				uint32_t i = p + u * stride;
				if (rev) i = packs - 1 - i;
				yy[u] = ld<NTL>(y, (size_t) i * 2); qq[u] = ld<false>(q, (size_t) i * 2); ss[u] = ld<NTL>(s, (size_t) i * 2);
			}
			#pragma unroll
			for (int u = 0; u < U; u++) {
				uint32_t i = p + u * stride;
				if (rev) i = packs - 1 - i;
				d2 o;
				o.x = fma(-alpha, yy[u].x, qq[u].x); o.y = fma(-alpha, yy[u].y, qq[u].y);
				acc = fma(ss[u].x, o.x, acc); acc = fma(ss[u].y, o.y, acc);
				st<NTS>(q, (size_t) i * 2, o);
			}
		}
		for (; p < packs; p += stride) {
			uint32_t i = rev ? packs - 1 - p : p;
			d2 a = ld<NTL>(y, (size_t) i * 2), b = ld<false>(q, (size_t) i * 2), c = ld<NTL>(s, (size_t) i * 2), o;
			o.x = fma(-alpha, a.x, b.x); o.y = fma(-alpha, a.y, b.y);
			acc = fma(c.x, o.x, acc); acc = fma(c.y, o.y, acc);
			st<NTS>(q, (size_t) i * 2, o);
		}
	} else {
		// contiguous chunk per block, block-stride inside the chunk
		const uint32_t per = (packs + gridDim.x - 1) / gridDim.x;
		const uint32_t lo = blockIdx.x * per;
		const uint32_t hi = min(packs, lo + per);
		uint32_t p = lo + threadIdx.x;
		for (; p + (U - 1) * BLOCK < hi; p += U * BLOCK) {
			d2 yy[U], qq[U], ss[U];
			#pragma unroll
			for (int u = 0; u < U; u++) {
				uint32_t i = p + u * BLOCK;
				yy[u] = ld<NTL>(y, (size_t) i * 2); qq[u] = ld<false>(q, (size_t) i * 2); ss[u] = ld<NTL>(s, (size_t) i * 2);
			}
			#pragma unroll
			for (int u = 0; u < U; u++) {
				uint32_t i = p + u * BLOCK;
				d2 o;
				o.x = fma(-alpha, yy[u].x, qq[u].x); o.y = fma(-alpha, yy[u].y, qq[u].y);
				acc = fma(ss[u].x, o.x, acc); acc = fma(ss[u].y, o.y, acc);
				st<NTS>(q, (size_t) i * 2, o);
			}
		}
		for (; p < hi; p += BLOCK) {
			d2 a = ld<NTL>(y, (size_t) p * 2), b = ld<false>(q, (size_t) p * 2), c = ld<NTL>(s, (size_t) p * 2), o;
			o.x = fma(-alpha, a.x, b.x); o.y = fma(-alpha, a.y, b.y);
			acc = fma(c.x, o.x, acc); acc = fma(c.y, o.y, acc);
			st<NTS>(q, (size_t) p * 2, o);
		}
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BLOCK / 64; w++) t += sh[w]; parts[blockIdx.x] = t; }
}

// reference points: plain copy and plain 2-stream dot
template <int BLOCK, int U>
__global__ void __launch_bounds__(BLOCK) k_copy(const double* __restrict__ a, double* __restrict__ b, uint32_t n)
{
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < packs; p += stride) st<false>(b, (size_t) p * 2, ld<false>(a, (size_t) p * 2));
}
template <int BLOCK, int U>
__global__ void __launch_bounds__(BLOCK) k_dot(const double* __restrict__ a, const double* __restrict__ b, uint32_t n, double* parts)
{
	__shared__ double sh[BLOCK / 64];
	const uint32_t packs = n / 2, stride = gridDim.x * BLOCK;
	double acc = 0;
	uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
	for (; p + (U - 1) * stride < packs; p += U * stride) {
		d2 x[U], y[U];
		#pragma unroll
		for (int u = 0; u < U; u++) { x[u] = ld<true>(a, (size_t) (p + u * stride) * 2); y[u] = ld<true>(b, (size_t) (p + u * stride) * 2); }
		#pragma unroll
		for (int u = 0; u < U; u++) { acc = fma(x[u].x, y[u].x, acc); acc = fma(x[u].y, y[u].y, acc); }
	}
	for (; p < packs; p += stride) { d2 x = ld<true>(a, (size_t) p * 2), y = ld<true>(b, (size_t) p * 2); acc = fma(x.x, y.x, acc); acc = fma(x.y, y.y, acc); }
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BLOCK / 64; w++) t += sh[w]; parts[blockIdx.x] = t; }
}

static double median(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <class F> double time_ms(F&& launch, int reps = 12)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int i = 0; i < 3; i++) launch(i);
	CK(hipDeviceSynchronize());
	std::vector<float> t;
	for (int i = 0; i < reps; i++) {
		CK(hipEventRecord(a));
		launch(i);
		CK(hipEventRecord(b));
		CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		t.push_back(ms);
	}
	return median(t);
}

int main(int argc, char** argv)
{
	const uint32_t n = argc > 1 ? (uint32_t) atof(argv[1]) : 100000000u;
	const int m = 4;   // rotate over m different (y, s) rows like the real chain
	double *Y, *S, *q, *parts;
	CK(hipMalloc(&Y, (size_t) m * n * 8)); CK(hipMalloc(&S, (size_t) m * n * 8)); CK(hipMalloc(&q, (size_t) n * 8));
	CK(hipMalloc(&parts, 1 << 20));
	CK(hipMemset(Y, 0, (size_t) m * n * 8)); CK(hipMemset(S, 0, (size_t) m * n * 8)); CK(hipMemset(q, 0, (size_t) n * 8));
	const double bytes4 = 4.0 * n * 8, GB = 1e9;
	printf("n = %u\n", n);
	{
		double ms = time_ms([&](int i) { CK(hipMemcpyAsync(S, Y, (size_t) n * 8, hipMemcpyDeviceToDevice, 0)); });
		printf("hipMemcpy D2D            : %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * n * 8 / ms / 1e6);
		for (int g : {2048, 4096, 8192, 16384}) {
			ms = time_ms([&](int i) { hipLaunchKernelGGL((k_copy<256, 1>), dim3(g), dim3(256), 0, 0, Y, S, n); });
			printf("copy kernel grid %-6d  : %.3f ms  %.0f GB/s (r+w)\n", g, ms, 2.0 * n * 8 / ms / 1e6);
		}
		for (int g : {1024, 2048, 4096, 8192}) {
			ms = time_ms([&](int i) { hipLaunchKernelGGL((k_dot<256, 2>), dim3(g), dim3(256), 0, 0, Y + (size_t) (i % m) * n, S + (size_t) (i % m) * n, n, parts); });
			printf("dot U2 grid %-6d       : %.3f ms  %.0f GB/s (2 reads)\n", g, ms, 2.0 * n * 8 / ms / 1e6);
			ms = time_ms([&](int i) { hipLaunchKernelGGL((k_dot<256, 4>), dim3(g), dim3(256), 0, 0, Y + (size_t) (i % m) * n, S + (size_t) (i % m) * n, n, parts); });
			printf("dot U4 grid %-6d       : %.3f ms  %.0f GB/s (2 reads)\n", g, ms, 2.0 * n * 8 / ms / 1e6);
		}
	}
#define RUN(BLOCK, U, NTL, NTS, GRID, MODE, REV)                                                                             \
	{                                                                                                                        \
		double ms = time_ms([&](int i) {                                                                                     \
			hipLaunchKernelGGL((k_bwd<BLOCK, U, NTL, NTS>), dim3(GRID), dim3(BLOCK), 0, 0, Y + (size_t) (i % m) * n, q,        \
			                   S + (size_t) (i % m) * n, 1e-3, n, MODE, (REV) ? (i & 1) : 0, parts);                            \
		});                                                                                                                  \
		printf("bwd B%-4d U%d ntl%d nts%d grid %-6d mode %d rev %d : %.3f ms  %.0f GB/s\n", BLOCK, U, NTL, NTS, GRID, MODE, REV, ms, \
		       bytes4 / ms / 1e6);                                                                                           \
	}
	for (int g : {128, 192, 256, 320, 384, 512, 640, 768, 1024}) {
		RUN(256, 1, true, false, g, 0, 0);
		RUN(256, 2, true, false, g, 0, 0);
		RUN(256, 4, true, false, g, 0, 0);
		RUN(256, 1, true, true, g, 0, 0);
		RUN(256, 2, true, true, g, 0, 0);
		RUN(512, 1, true, false, g, 0, 0);
		RUN(512, 2, true, false, g, 0, 0);
		RUN(512, 4, true, false, g, 0, 0);
		RUN(512, 2, true, true, g, 0, 0);
		RUN(1024, 1, true, false, g, 0, 0);
		RUN(1024, 2, true, false, g, 0, 0);
		RUN(256, 2, true, false, g, 1, 0);
		RUN(512, 2, true, false, g, 1, 0);
		RUN(256, 2, true, false, g, 0, 1);
		RUN(256, 2, false, false, g, 0, 0);
	}
	(void) GB;
	return 0;
}
