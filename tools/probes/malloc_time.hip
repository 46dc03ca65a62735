// Development probe: what hipMalloc / hipFree of a 4 GiB buffer cost, call after call (the stream decoder's table of
// origins: a 238 ms stall was seen in bench.py's second lzs_decompress_stream_device call on some boxes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint32_t *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
int main(int argc, char **argv)
{
    const size_t big = ((size_t)4 << 30) + 64;
    const int other = argc > 1 ? atoi(argv[1]) : 0;          // churn other buffers in between, as a framework's allocator does
    std::vector<void *> keep;
    for (int i = 0; i < other; i++) { void *q; hipMalloc(&q, (size_t)1200 << 20); keep.push_back(q); }
    for (int rep = 0; rep < 8; rep++) {
        void *p = nullptr, *q = nullptr;
        const double t0 = now();
        if (hipMalloc(&p, big) != hipSuccess) { printf("malloc failed\n"); return 1; }
        const double t1 = now();
        hipMalloc(&q, (size_t)1 << 30);
        const double t1b = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (uint32_t *)p, big / 4);
        hipDeviceSynchronize();
        const double t2 = now();
        hipFree(p);
        const double t3 = now();
        hipFree(q);
        const double t4 = now();
        printf("call %d: hipMalloc 4 GiB %.2f ms, hipMalloc 1 GiB %.2f ms, touch %.2f ms, hipFree 4 GiB %.2f ms, hipFree 1 GiB %.2f ms\n", rep, t1 - t0, t1b - t1, t2 - t1b, t3 - t2, t4 - t3);
        if (other && rep == 3) { for (void *k : keep) hipFree(k); keep.clear(); printf("  (the other buffers released)\n"); }
    }
    return 0;
}
