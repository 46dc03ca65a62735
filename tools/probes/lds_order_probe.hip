// Probe (development aid, not part of the library): when several lanes of ONE wave
// instruction hit the same LDS address with a returning atomic, in which order are they
// applied?  Checks the hypothesis "ascending lane order" for ds_max_rtn_u32 / ds_wrxchg_rtn
// over many random and adversarial address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const uint32_t *addr, uint32_t *bad_max, uint32_t *bad_xchg, uint32_t npat)
{
    __shared__ uint32_t tab[2048 * 4];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *t = tab + wv * 2048;
    for (uint32_t pat = blockIdx.x * 4 + wv; pat < npat; pat += gridDim.x * 4) {
        const uint32_t raw = addr[pat * 64 + lane];
        const uint32_t a = raw & 2047;
        // every other pattern runs under a partial EXEC mask (as the build does: a wave only
        // inserts the lanes whose buckets it owns)
        const bool active = (pat & 1u) == 0 || ((raw >> 13) & 3u) != 0;
        // expected: nearest lower ACTIVE lane with the same address (+1), else 0
        uint32_t expect = 0;
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t al = __shfl((int)a, (int)l, 64);
            const int act = __shfl((int)active, (int)l, 64);
            if (l < lane && al == a && act) expect = l + 1;
        }
        t[a] = 0;
        __builtin_amdgcn_wave_barrier();
        uint32_t got = expect;
        if (active) got = atomicMax(&t[a], lane + 1);
        __builtin_amdgcn_wave_barrier();
        if (got != expect) atomicAdd(bad_max, 1);
        t[a] = 0;
        __builtin_amdgcn_wave_barrier();
        uint32_t got2 = expect;
        if (active) got2 = atomicExch(&t[a], lane + 1);
        __builtin_amdgcn_wave_barrier();
        if (got2 != expect) atomicAdd(bad_xchg, 1);
    }
}

int main()
{
    const uint32_t npat = 1 << 18;
    uint32_t *h = (uint32_t *)malloc(npat * 64 * 4);
    uint64_t x = 88172645463325252ull;
    for (uint32_t p = 0; p < npat; p++) {
        const uint32_t mode = p % 8;   // number of distinct addresses varies: 1,2,4,8,16,32,64, runs
        for (uint32_t l = 0; l < 64; l++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t v = (uint32_t)(x >> 20);
            const uint32_t maskbits = (uint32_t)(x >> 5) & (3u << 13);   // drives the partial EXEC mask
            // few distinct addresses; same-bank strides (x32, x64 dwords); random; runs
            if (mode < 4) v &= (1u << (mode + 1)) - 1;
            else if (mode == 4) v = (v & 7) * 32;
            else if (mode == 5) v = (v & 15) * 64 + ((v >> 8) & 1);
            else if (mode == 6) v &= 2047;
            else v = (l / 5) * 33;
            h[p * 64 + l] = (v & 2047u) | maskbits;
        }
    }
    uint32_t *d, *bad;
    hipMalloc(&d, npat * 64 * 4);
    hipMalloc(&bad, 8);
    hipMemcpy(d, h, npat * 64 * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, bad, bad + 1, npat);
    uint32_t r[2];
    hipMemcpy(r, bad, 8, hipMemcpyDeviceToHost);
    printf("patterns %u x 64 lanes: atomicMax lane-order violations %u, atomicExch violations %u\n", npat, r[0], r[1]);
    return 0;
}
