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
        unsigned long long zero[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(lzs_prof), zero, sizeof(zero));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(lzs_compress_blocks_wg_kernel, dim3(nb), dim3(256), 0, 0, d_out, stride, 73731u, d_len,
                           (const uint8_t *)d_in, (size_t)bl, (const uint32_t *)nullptr, bl, nb, getenv("LZS_CHAIN_FALLBACK") ? 1u : 0u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long p[32];
        hipMemcpyFromSymbol(p, HIP_SYMBOL(lzs_prof), sizeof(p));
        const double nbytes = nb * 65536.0;
        double tot = 0; for (int i = 0; i < 8; i++) tot += (double)p[i];
        printf("class %u: %.2f ms (%.2f GB/s); wave-0 cycles per byte per workgroup:\n", cls, ms, nbytes / ms / 1e6);
        printf("  refill %.1f  build %.1f  search %.1f  wait-for-other-waves %.1f  mark %.1f  pack %.1f  open-match %.1f  other %.1f  (sum %.1f), pools %llu\n",
               p[0] / nbytes, p[1] / nbytes, p[2] / nbytes, p[5] / nbytes, p[3] / nbytes, p[6] / nbytes, p[7] / nbytes, p[4] / nbytes, tot / nbytes, p[13]);
        printf("  wave 0 per pool: refill passes %.2f (positions taken %.1f), step iterations %.2f (busy lanes %.1f)\n",
               (double)p[9] / p[13], (double)p[10] / p[13], (double)p[11] / p[13], (double)p[12] / (p[11] ? p[11] : 1));
        printf("  wave 0 search: cycles per refill pass %.0f, per step iteration %.0f\n", (double)p[14] / (p[9] ? p[9] : 1), (double)p[15] / (p[11] ? p[11] : 1));
        printf("  build: %.0f cycles per batch of 64 (%.2f batches per pool)\n", (double)p[16] / (p[17] ? p[17] : 1), (double)p[17] / p[13]);
        { const double q = (double)p[13];
          printf("  parse+pack cycles per pool: extend %.0f, doubling %.0f, barrier %.0f, exit walk %.0f, node+encode+scan %.0f, barrier %.0f, offsets+bits_or %.0f, barrier %.0f, stores %.0f, barrier %.0f\n",
                 p[20] / q, p[21] / q, p[22] / q, p[23] / q, p[24] / q, p[25] / q, p[26] / q, p[27] / q, p[28] / q, p[29] / q); }
    }
    return 0;
}
