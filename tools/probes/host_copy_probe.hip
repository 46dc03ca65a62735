// Development probe: host <-> device copy rates on this box for pageable memory, memory pinned in place
// (hipHostRegister; with its own cost), and hipHostMalloc memory: what a host-buffer call could gain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)1 << 30;
    void *d; hipMalloc(&d, n);
    char *pageable = (char *)malloc(n); memset(pageable, 1, n);
    char *pinned; hipHostMalloc((void **)&pinned, n, hipHostMallocDefault); memset(pinned, 2, n);
    hipMemcpy(d, pinned, n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(d, pageable, n, hipMemcpyHostToDevice); printf("H2D pageable      : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipMemcpy(pageable, d, n, hipMemcpyDeviceToHost); printf("D2H pageable      : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipMemcpy(d, pinned, n, hipMemcpyHostToDevice); printf("H2D hipHostMalloc : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipMemcpy(pinned, d, n, hipMemcpyDeviceToHost); printf("D2H hipHostMalloc : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault); double tr = now() - t;
        printf("hipHostRegister 1 GiB: %.1f ms (%s)\n", tr, hipGetErrorString(e));
        t = now(); hipMemcpy(d, pageable, n, hipMemcpyHostToDevice); printf("H2D registered    : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipMemcpy(pageable, d, n, hipMemcpyDeviceToHost); printf("D2H registered    : %.1f GB/s\n", n / (now() - t) / 1e6);
        t = now(); hipHostUnregister(pageable); printf("hipHostUnregister: %.1f ms\n", now() - t);
        t = now(); memcpy(pinned, pageable, n); printf("host memcpy 1 thread: %.1f GB/s\n", n / (now() - t) / 1e6);
    }
    return 0;
}
