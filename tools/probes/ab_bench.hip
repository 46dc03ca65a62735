// Development probe: times the wg compress kernel of a given source snapshot (-DKSRC="\"file\"")
// on the seeded workload, so that variants can be compared inside one gpurun call.
// The generated input is kept in /tmp (lzs_ab_class<c>_<nb>.bin) so that the runs of one session share it; AB_NOHASH=1
// skips the FNV of the output (the first run of a variant should print it: a variant that changes the output shows).
#include KSRC
#include <vector>
#include <string>
#include <cstring>
#ifdef AB_VIA_LAUNCHER   // the library's own launcher: the classifier and one variant of the kernel per class (LZS_VARIANT=text|few|lit forces one)
#define AB_LAUNCH(nb, d_out, stride, d_len, d_in, bl) lzs_hip_launch_compress(d_out, stride, 73731u, d_len, d_in, bl, nullptr, bl, nb, nullptr)
#endif
#ifndef AB_LAUNCH       // (a variant whose kernel is launched another way says how: -DAB_LAUNCH=...)
#define AB_LAUNCH(nb, d_out, stride, d_len, d_in, bl) \
    hipLaunchKernelGGL(lzs_compress_blocks_wg_kernel, dim3(nb), dim3(256), 0, 0, d_out, stride, 73731u, d_len, (const uint8_t *)d_in, \
                       (size_t)bl, (const uint32_t *)nullptr, bl, nb, 0u)
#endif
extern "C" int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);
int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t nb = argc > 2 ? atoi(argv[2]) : 16384, bl = 65536;
    std::vector<uint8_t> h((size_t)nb * bl);
    const std::string cache = "/tmp/lzs_ab_class" + std::to_string(cls) + "_" + std::to_string(nb) + ".bin";
    bool have = false;
    if (!getenv("AB_FILE")) if (FILE *f = fopen(cache.c_str(), "rb")) { have = fread(h.data(), 1, h.size(), f) == h.size(); fclose(f); }
    // AB_FILE=path: the blocks are the file's 64 KiB pieces, over and over (real text from the build container, say: is a choice tuned
    // on the seeded classes one for other data of the kind as well?) -- not cached
    if (const char *path = getenv("AB_FILE")) {
        std::vector<uint8_t> f;
        if (FILE *fp = fopen(path, "rb")) { uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + k); fclose(fp); }
        const size_t pieces = f.size() / bl;
        if (!pieces) { fprintf(stderr, "AB_FILE: fewer than 64 KiB\n"); return 2; }
        for (uint32_t b = 0; b < nb; b++) memcpy(h.data() + (size_t)b * bl, f.data() + (b % pieces) * bl, bl);
        have = true;
    }
    if (!have) {
        lzs_workload_fill(h.data(), cls, 0x4C5A5331ull, 0, nb, bl, 32);
        if (FILE *f = fopen((cache + ".tmp").c_str(), "wb")) { const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size(); fclose(f); if (ok) rename((cache + ".tmp").c_str(), cache.c_str()); }
    }
    uint8_t *d_in, *d_out; uint32_t *d_len;
    const size_t stride = 73744;
    hipMalloc(&d_in, h.size()); hipMalloc(&d_out, (size_t)nb * stride); hipMalloc(&d_len, nb * 4);
    hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice);
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 7; rep++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        AB_LAUNCH(nb, d_out, stride, d_len, d_in, bl);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    std::vector<uint32_t> lens(nb);
    hipMemcpy(lens.data(), d_len, nb * 4, hipMemcpyDeviceToHost);
    unsigned long long tot = 0; for (auto v : lens) tot += v;
    // identity of the output: FNV-1a over every block's length and bytes (variants must print the same)
    unsigned long long fnv = 0;
    if (!getenv("AB_NOHASH")) {
        std::vector<uint8_t> outh((size_t)nb * stride);
        hipMemcpy(outh.data(), d_out, outh.size(), hipMemcpyDeviceToHost);
        fnv = 1469598103934665603ull;
        for (uint32_t b = 0; b < nb; b++) {
            fnv = (fnv ^ lens[b]) * 1099511628211ull;
            const uint8_t *q = outh.data() + (size_t)b * stride;
            for (uint32_t i = 0; i < lens[b]; i++) fnv = (fnv ^ q[i]) * 1099511628211ull;
        }
    }
    printf("class %u: mean %.3f ms best %.3f ms (%.2f GB/s), compressed bytes %llu, fnv %016llx\n", cls, sum / 5, best, nb * 65536.0 / (sum / 5) / 1e6, tot, fnv);
    return 0;
}
