// Diagnostic (development aid): runs the chain compress kernel built with LZS_PROFILE on
// seeded blocks and prints where the wave cycles go.  Not part of the library.
#define LZS_PROFILE 1
#include "../../lzs_compression_amd/csrc/lzs_kernels.hip"
#include <vector>
extern "C" int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);

int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t nb = argc > 2 ? atoi(argv[2]) : 16384, bl = 65536;
    std::vector<uint8_t> h((size_t)nb * bl);
    lzs_workload_fill(h.data(), cls, 0x4C5A5331ull, 0, nb, bl, 32);
    uint8_t *d_in, *d_out; uint32_t *d_len;
    const size_t stride = 73744;
    hipMalloc(&d_in, h.size()); hipMalloc(&d_out, (size_t)nb * stride); hipMalloc(&d_len, nb * 4);
    hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(lzs_prof), zero, sizeof(zero));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(lzs_compress_blocks_kernel, dim3(nb), dim3(64), 0, 0, d_out, stride, 73731u, d_len,
                           (const uint8_t *)d_in, (size_t)bl, (const uint32_t *)nullptr, bl, nb);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long p[16];
        hipMemcpyFromSymbol(p, HIP_SYMBOL(lzs_prof), sizeof(p));
        const double tot = (double)(p[0] + p[1] + p[2] + p[3] + p[4]);
        printf("class %u: %.2f ms (%.2f GB/s)\n", cls, ms, nb * 65536.0 / ms / 1e6);
        printf("  cycles/byte/wave: refill %.1f build %.1f search %.1f parse %.1f other %.1f (sum %.1f)\n",
               p[0] / (nb * 65536.0), p[1] / (nb * 65536.0), p[2] / (nb * 65536.0), p[3] / (nb * 65536.0),
               p[4] / (nb * 65536.0), tot / (nb * 65536.0));
        printf("  pools %llu, search iterations/pool %.1f, busy lanes/iteration %.1f\n", p[7],
               (double)p[5] / p[7], (double)p[6] / p[5]);
        printf("  extended tokens/pool %.2f; parse split, cycles/byte: chunk head %.1f, chase (incl. extended) %.1f, last emit %.1f\n",
               (double)p[8] / p[7], p[9] / (nb * 65536.0), p[10] / (nb * 65536.0), p[11] / (nb * 65536.0));
    }
    return 0;
}
