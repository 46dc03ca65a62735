// Development probe: what one synchronous round trip to the device costs on this box, by the way the
// host waits (hipStreamSynchronize, polling hipStreamQuery, polling a flag in pinned host memory
// written by the kernel), and with a kernel that reads / writes pinned host memory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty() {}
__global__ void k_flag(volatile unsigned *flag, unsigned v) { *flag = v; __threadfence_system(); }
__global__ void k_touch(const unsigned char *in, unsigned char *out, unsigned n, volatile unsigned *flag, unsigned v)
{
    unsigned acc = 0;
    for (unsigned i = threadIdx.x; i < n; i += 64) { acc += in[i]; out[i] = in[i] ^ 1; }
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *flag = v + (acc & 0); }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    unsigned *flag; hipHostMalloc(&flag, 4096, hipHostMallocDefault); *flag = 0;
    unsigned char *box; hipHostMalloc(&box, 1 << 16, hipHostMallocDefault);
    const int N = 2000;
    for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); hipStreamSynchronize(s); }
    double t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); hipStreamSynchronize(s); }
    printf("empty kernel + hipStreamSynchronize : %.1f us\n", (now() - t) / N);
    t = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); while (hipStreamQuery(s) == hipErrorNotReady) {} }
    printf("empty kernel + hipStreamQuery poll  : %.1f us\n", (now() - t) / N);
    t = now();
    for (int i = 1; i <= N; i++) { hipLaunchKernelGGL(k_flag, 1, 64, 0, s, flag, (unsigned)i); while (*(volatile unsigned *)flag != (unsigned)i) {} }
    printf("flag kernel + poll pinned flag      : %.1f us\n", (now() - t) / N);
    hipStreamSynchronize(s);
    for (unsigned n : {256u, 2304u, 4096u}) {
        t = now();
        for (int i = 1; i <= N; i++) { hipLaunchKernelGGL(k_touch, 1, 64, 0, s, box, box + 32768, n, flag, (unsigned)(N + i)); while (*(volatile unsigned *)flag != (unsigned)(N + i)) {} }
        printf("touch %4u B pinned in/out + poll flag: %.1f us\n", n, (now() - t) / N);
        hipStreamSynchronize(s);
    }
    return 0;
}
