// Development probe: what the copy engines give a PIPELINE of pinned pieces (lzs_pipeline.c): asynchronous copies of
// 24-48 MiB between hipHostMalloc memory and the device, one direction alone, both at once on two streams, and beside a
// kernel that keeps every CU busy -- against the 57 GB/s of one synchronous 1 GiB copy (host_copy_probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void busy(unsigned *p, unsigned iters)
{
    unsigned v = threadIdx.x;
    for (unsigned i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) p[0] = v;
}
int main()
{
    const size_t piece = (size_t)44 << 20;
    const int reps = 24;
    char *d_a, *d_b; hipMalloc((void **)&d_a, piece * 2); hipMalloc((void **)&d_b, piece * 2);
    unsigned *d_flag; hipMalloc((void **)&d_flag, 64);
    hipStream_t s1, s2, s3; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    for (int flags = 0; flags < 2; flags++) {
        char *h_a, *h_b;
        const unsigned f = flags == 0 ? hipHostMallocDefault : hipHostMallocNonCoherent;
        hipHostMalloc((void **)&h_a, piece * 2, f); hipHostMalloc((void **)&h_b, piece * 2, f);
        memset(h_a, 1, piece * 2); memset(h_b, 2, piece * 2);
        printf("== hipHostMalloc flags: %s\n", flags == 0 ? "default" : "non-coherent");
        for (int with_kernel = 0; with_kernel < 2; with_kernel++) {
            if (with_kernel) hipLaunchKernelGGL(busy, dim3(256 * 64), dim3(256), 0, s3, d_flag, 3000000u);      // ~ hundreds of ms of every CU
            double t = now();
            for (int r = 0; r < reps; r++) hipMemcpyAsync(d_a + (r & 1) * piece, h_a + (r & 1) * piece, piece, hipMemcpyHostToDevice, s1);
            hipStreamSynchronize(s1);
            printf("%s H2D alone          : %.1f GB/s\n", with_kernel ? "beside a kernel:" : "idle device:   ", reps * piece / (now() - t) / 1e6);
            t = now();
            for (int r = 0; r < reps; r++) hipMemcpyAsync(h_b + (r & 1) * piece, d_b + (r & 1) * piece, piece, hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s2);
            printf("%s D2H alone          : %.1f GB/s\n", with_kernel ? "beside a kernel:" : "idle device:   ", reps * piece / (now() - t) / 1e6);
            t = now();
            for (int r = 0; r < reps; r++) {
                hipMemcpyAsync(d_a + (r & 1) * piece, h_a + (r & 1) * piece, piece, hipMemcpyHostToDevice, s1);
                hipMemcpyAsync(h_b + (r & 1) * piece, d_b + (r & 1) * piece, piece, hipMemcpyDeviceToHost, s2);
            }
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            printf("%s H2D + D2H at once  : %.1f GB/s each way\n", with_kernel ? "beside a kernel:" : "idle device:   ", reps * piece / (now() - t) / 1e6);
            hipStreamSynchronize(s3);
        }
        // one thread's memcpy into / out of the pinned piece
        char *p = (char *)malloc(piece * 2); memset(p, 3, piece * 2);
        double t = now(); for (int r = 0; r < 8; r++) memcpy(h_a, p, piece * 2); printf("memcpy pageable -> pinned, one thread: %.1f GB/s\n", 8 * 2 * piece / (now() - t) / 1e6);
        t = now(); for (int r = 0; r < 8; r++) memcpy(p, h_b, piece * 2); printf("memcpy pinned -> pageable, one thread: %.1f GB/s\n", 8 * 2 * piece / (now() - t) / 1e6);
        free(p);
        hipHostFree(h_a); hipHostFree(h_b);
    }
    return 0;
}
