// profiles/src/queue_oom.hip -- what does the HIP runtime do when a stream is first used while the device is (almost) full?
// Fill HBM up to `leave_mb`, then create streams one by one and launch a trivial kernel on each.  Not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[0] += 1; }
int main(int argc, char** argv)
{
	const size_t leave_mb = argc > 1 ? (size_t) atol(argv[1]) : 300;
	const int warm = argc > 2 ? atoi(argv[2]) : 0;          // streams used BEFORE the device is filled
	int* d = nullptr;
	if (hipMalloc((void**) &d, 4096) != hipSuccess) return 1;
	hipStream_t st[12];
	for (int i = 0; i < warm; i++) {
		if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) { printf("warm stream %d: create failed\n", i); return 1; }
		hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[i], d);
		printf("warm stream %d: %s\n", i, hipGetErrorString(hipStreamSynchronize(st[i])));
	}
	size_t free_b = 0, total = 0;
	(void) hipMemGetInfo(&free_b, &total);
	void* ballast = nullptr;
	const size_t want = free_b - (leave_mb << 20);
	printf("free %.1f GB, ballast %.1f GB: %s\n", free_b / 1e9, want / 1e9, hipGetErrorString(hipMalloc(&ballast, want)));
	(void) hipMemGetInfo(&free_b, &total);
	printf("free now %.0f MB\n", free_b / 1048576.0);
	fflush(stdout);
	for (int i = warm; i < 12; i++) {
		hipError_t e = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
		printf("stream %d: create %s", i, hipGetErrorString(e)); fflush(stdout);
		if (e != hipSuccess) { printf("\n"); continue; }
		hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st[i], d);
		e = hipGetLastError();
		printf(", launch %s", hipGetErrorString(e)); fflush(stdout);
		e = hipStreamSynchronize(st[i]);
		(void) hipMemGetInfo(&free_b, &total);
		printf(", sync %s, free %.0f MB\n", hipGetErrorString(e), free_b / 1048576.0); fflush(stdout);
	}
	printf("done\n");
	return 0;
}
