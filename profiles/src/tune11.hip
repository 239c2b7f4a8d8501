// profiles/src/tune11.hip -- what is the read ceiling?  The same 16 GB read as ONE stream, as 4, and as 20 streams of 0.8 GB (pass 1's
// shape), summed into one accumulator per lane: if one stream is no faster than twenty, 6.7 TB/s is what the part gives a streaming
// read and pass 1 is at it.  Not part of the product.  hipcc --offload-arch=gfx950 -O3 tune11.hip -o tune11
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 ldnt(const double* p, size_t i) { return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + i)); }
__device__ __forceinline__ d2 ldd(const double* p, size_t i) { return *reinterpret_cast<const d2*>(p + i); }
constexpr int BLOCK = 256;

// NS streams of `len` doubles each, U packs per stream and lane in flight
template <int NS, int U, bool NT>
__global__ void __launch_bounds__(BLOCK) k_read(const double* base, size_t len, double* out)
{
	double acc = 0;
	const size_t packs = len / 2, stride = (size_t) gridDim.x * BLOCK;
	for (size_t p0 = (size_t) blockIdx.x * BLOCK + threadIdx.x; p0 < packs; p0 += U * stride) {
		d2 v[U][NS];
		#pragma unroll
		for (int u = 0; u < U; u++)
			#pragma unroll
			for (int s = 0; s < NS; s++) {
				const size_t p = p0 + u * stride;
				v[u][s] = p < packs ? (NT ? ldnt(base + (size_t) s * len, p * 2) : ldd(base + (size_t) s * len, p * 2)) : d2{0, 0};
			}
		#pragma unroll
		for (int u = 0; u < U; u++)
			#pragma unroll
			for (int s = 0; s < NS; s++) acc += v[u][s].x + v[u][s].y;
	}
	if (acc == 12345.678) out[0] = acc;          // never true: keeps the loads
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

int main()
{
	const size_t n = 100000000, total = 20 * n;
	double *S, *out;
	CK(hipMalloc(&S, total * 8)); CK(hipMalloc(&out, 64));
	CK(hipMemset(S, 0, total * 8));
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("%d CUs, %.1f GB read per launch\n", cus, total * 8 / 1e9);
#define RR(NS, U, NT, GRID) { double ms = time_ms([&](int) { hipLaunchKernelGGL((k_read<NS, U, NT>), dim3(GRID), dim3(BLOCK), 0, 0, S, total / NS, out); }); \
	printf("read %2d stream(s) U %d nt %d grid %-5d : %.3f ms  %.0f GB/s\n", NS, U, NT, GRID, ms, total * 8.0 / ms / 1e6); fflush(stdout); }
	for (int rep = 0; rep < 2; rep++) {
		RR(1, 8, true, cus); RR(1, 8, true, 2 * cus); RR(1, 8, true, 4 * cus); RR(1, 8, true, 8 * cus); RR(1, 16, true, 2 * cus); RR(1, 4, true, 8 * cus); RR(1, 8, false, 4 * cus);
		RR(4, 4, true, cus); RR(4, 4, true, 2 * cus); RR(4, 4, true, 4 * cus);
		RR(20, 1, true, cus); RR(20, 1, true, 2 * cus); RR(20, 2, true, cus); RR(20, 1, false, cus);
	}
	return 0;
}
