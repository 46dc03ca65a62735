// Development probe: times the block decoder of a given source snapshot (-DKSRC="\"file\"") on the seeded
// workload compressed by the same snapshot, and compares what comes back with the input byte for byte.  The
// slots are filled with 0xA5 before compressing, so what follows a stream in its slot is not zeros.
#include KSRC
#include <vector>
#include <cstring>
#include <string>
extern "C" int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);
int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t nb = argc > 2 ? atoi(argv[2]) : 16384, bl = 65536;
    std::vector<uint8_t> h((size_t)nb * bl);
    // (the generated input is shared with ab_bench through /tmp: the runs of one session generate it once)
    const std::string cache = "/tmp/lzs_ab_class" + std::to_string(cls) + "_" + std::to_string(nb) + ".bin";
    bool have = false;
    if (FILE *f = fopen(cache.c_str(), "rb")) { have = fread(h.data(), 1, h.size(), f) == h.size(); fclose(f); }
    if (!have) {
        lzs_workload_fill(h.data(), cls, 0x4C5A5331ull, 0, nb, bl, 32);
        if (FILE *f = fopen((cache + ".tmp").c_str(), "wb")) { const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size(); fclose(f); if (ok) rename((cache + ".tmp").c_str(), cache.c_str()); }
    }
    uint8_t *d_in, *d_out, *d_back; uint32_t *d_len, *d_blen;
    const size_t stride = 73744;
    hipMalloc(&d_in, h.size()); hipMalloc(&d_out, (size_t)nb * stride); hipMalloc(&d_len, nb * 4);
    hipMalloc(&d_back, h.size()); hipMalloc(&d_blen, nb * 4);
    hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice);
    hipMemset(d_out, 0xA5, (size_t)nb * stride);
    hipLaunchKernelGGL(lzs_compress_blocks_wg_kernel, dim3(nb), dim3(256), 0, 0, d_out, stride, 73731u, d_len,
                       (const uint8_t *)d_in, (size_t)bl, (const uint32_t *)nullptr, bl, nb, 0u);
    hipDeviceSynchronize();
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 7; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipMemsetAsync(d_back, 0x5A, h.size(), 0);
        hipEventRecord(a);
        lzs_hip_launch_decompress(d_back, bl, bl, d_blen, d_out, stride, d_len, 0, nb, nullptr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    std::vector<uint8_t> back(h.size()); std::vector<uint32_t> bl_(nb);
    hipMemcpy(back.data(), d_back, h.size(), hipMemcpyDeviceToHost);
    hipMemcpy(bl_.data(), d_blen, nb * 4, hipMemcpyDeviceToHost);
    bool ok = memcmp(back.data(), h.data(), h.size()) == 0;
    for (uint32_t b = 0; b < nb; b++) ok = ok && bl_[b] == bl;
    printf("class %u: decode mean %.3f ms best %.3f ms (%.2f GB/s of output), round trip %s\n", cls, sum / 5, best,
           nb * 65536.0 / (sum / 5) / 1e6, ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}
