// profiles/src/pcie2.hip -- does a second copy stream raise the rate of ONE direction over PCIe?  0.8 GB in 8 slices, pinned host memory
// (hipHostRegister'ed malloc, like a caller's array), one stream against two alternating streams, H2D and D2H.  Not product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t n = 100000000, bytes = n * 8;
	void* host = nullptr;
	if (posix_memalign(&host, 4096, bytes)) return 1;
	memset(host, 1, bytes);
	CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
	char* dev; CK(hipMalloc((void**) &dev, bytes));
	hipStream_t st[4];
	for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (int dir = 0; dir < 2; dir++)
		for (int streams : {1, 2, 3, 4})
			for (int slices : {8, 16}) {
				double best = 1e9;
				for (int rep = 0; rep < 6; rep++) {
					CK(hipDeviceSynchronize());
					const double t0 = now();
					const size_t per = bytes / slices;
					for (int s = 0; s < slices; s++) {
						char* h = (char*) host + s * per; char* d = dev + s * per;
						if (dir == 0) CK(hipMemcpyAsync(d, h, per, hipMemcpyHostToDevice, st[s % streams]));
						else          CK(hipMemcpyAsync(h, d, per, hipMemcpyDeviceToHost, st[s % streams]));
					}
					for (int s = 0; s < streams; s++) CK(hipStreamSynchronize(st[s]));
					const double t = now() - t0;
					if (t < best) best = t;
				}
				printf("%s, %d stream(s), %2d slices: %.2f ms = %.1f GB/s\n", dir == 0 ? "H2D" : "D2H", streams, slices, 1e3 * best, bytes / best / 1e9);
			}
	return 0;
}
