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
        static unsigned long long zero[4][kProfN];
        hipMemcpyToSymbol(HIP_SYMBOL(lzs_prof), zero, sizeof(zero));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(lzs_compress_blocks_wg_kernel, dim3(nb), dim3(256), 0, 0, d_out, stride, 73731u, d_len,
                           (const uint8_t *)d_in, (size_t)bl, (const uint32_t *)nullptr, bl, nb, 0u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        static unsigned long long P[4][kProfN];
        hipMemcpyFromSymbol(P, HIP_SYMBOL(lzs_prof), sizeof(P));
        if (rep == 0) continue;                                  // (the first launch pays for the code load)
        const double nbytes = nb * 65536.0;
        printf("class %u: %.2f ms (%.2f GB/s); cycles per input byte per workgroup, by the wave's role in CHAIN\n", cls, ms, nbytes / ms / 1e6);
        printf("  (role 0 chains the 3-byte buckets, role 1 the 2-byte buckets, roles 2 and 3 wait there; every phase WITHOUT the barriers in it, which are listed apart)\n");
        for (int r = 0; r < 4; r++) {
            const unsigned long long *p = P[r];
            double tot = 0; for (int i = 0; i < 8; i++) tot += (double)p[i];
            double bar = 0; for (int i = 32; i < 42; i++) bar += (double)p[i];
            const double q = (double)p[13];
            const double parse_b = (double)(p[37] + p[38]);
            printf("  role %d: refill %.2f  hash+chain %.2f  search %.2f  extend+parse+pack %.2f  open-match %.2f  loop top %.2f | barriers %.2f (%.0f %% of %.2f)  pools %llu\n", r,
                   (p[0] - p[32]) / nbytes, (p[1] - p[33] - p[34] - p[35]) / nbytes, p[2] / nbytes,
                   (p[3] + p[6] - parse_b) / nbytes, (p[7] - p[39] - p[40] - p[41]) / nbytes, p[4] / nbytes, bar / nbytes, 100.0 * bar / tot, tot / nbytes, p[13]);
            printf("          barrier cycles per pool: after refill %.0f, between build rounds %.0f, hash|chain %.0f, chain|search %.0f, search|parse %.0f, exits published %.0f, chunk sums %.0f, open match %.0f + %.0f + %.0f\n",
                   p[32] / q, p[33] / q, p[34] / q, p[35] / q, p[36] / q, p[37] / q, p[38] / q, p[39] / q, p[40] / q, p[41] / q);
            printf("          search per pool: refill passes %.2f (positions taken %.1f, %.0f cycles a pass), step iterations %.2f (busy lanes %.1f, %.0f cycles an iteration)\n",
                   (double)p[9] / q, (double)p[10] / q, (double)p[14] / (p[9] ? p[9] : 1), (double)p[11] / q, (double)p[12] / (p[11] ? p[11] : 1), (double)p[15] / (p[11] ? p[11] : 1));
            printf("          build: %.0f cycles per batch of 64 incl. its barriers (%.2f batches per pool); cycles per pool: extend %.0f, doubling %.0f, [barrier], exit walk %.0f, tokens+encode+scan %.0f, [barrier], offsets+bits_or %.0f\n",
                   (double)p[16] / (p[17] ? p[17] : 1), (double)p[17] / q, p[20] / q, p[21] / q, p[23] / q, p[24] / q, p[26] / q);
            // (round 6) the trip counts the instruction ledger needs (tools/ledger.py): per wave and pool
            printf("          ledger trips per pool: rounds %.3f, ring refills %.3f, batches hashed+chained %.2f, refill passes %.3f (with the offset-1 length %.3f; %.3f that found nothing to take), step iterations %.3f, "
                   "extend: chunks with a capped match %.3f, compare-loop trips %.3f; parse rounds %.3f, exit hops %.3f, quarters stored %.3f, second PACK passes %.3f\n",
                   1.0, p[42] / q, (double)p[17] / q, (double)p[9] / q, p[18] / q, p[43] / q, (double)p[11] / q, p[19] / q, p[27] / q, p[31] / q, p[28] / q, p[29] / q, p[30] / q);
        }
    }
    return 0;
}
