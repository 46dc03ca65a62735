// Development probe: single-wave instruction latencies on gfx950 (cycles per instruction in a
// dependent chain), measured with s_memtime around unrolled inline-asm sequences.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void probe(unsigned long long *out, uint32_t seed)
{
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4 + 4) & 4095;   // pointer chain
    __syncthreads();
    uint32_t v = seed + threadIdx.x, w = seed * 3 + 1, x = 5;
    unsigned long long t0, t1;
    int k = 0;
#define MEASURE(name, body) \
    t0 = __builtin_readcyclecounter(); body; t1 = __builtin_readcyclecounter(); \
    if (threadIdx.x == 0) out[k] = t1 - t0; k++;
    // 0: empty
    MEASURE(empty, asm volatile("" ::: "memory"));
    // 1: 64 dependent v_add
    MEASURE(dep_valu, asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w)));
    // 2: 64 independent v_add (4 chains)
    { uint32_t a = v, b = v + 1, c2 = v + 2, d = v + 3;
      MEASURE(ind_valu, asm volatile(REP16("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c2), "+v"(d) : "v"(w)));
      v += a + b + c2 + d; }
    // 3: 64 x (v_cmp -> v_cndmask via vcc), dependent
    MEASURE(cmp_cnd_vcc, asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\nv_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(v) : "v"(w), "v"(x) : "vcc"));
    // 4: 64 x (v_cmp -> s_and -> v_cndmask), dependent through SALU
    MEASURE(cmp_sand_cnd, asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\ns_and_b64 vcc, vcc, exec\nv_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(v) : "v"(w), "v"(x) : "vcc"));
    // 5: 64 dependent SALU
    { uint32_t s = seed;
      MEASURE(dep_salu, asm volatile(REP64("s_add_u32 %0, %0, 7\n") : "+s"(s) :: "scc"));
      v += s; }
    // 6: 64 x (s_and_saveexec + s_or exec) no branch
    MEASURE(saveexec, asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\ns_and_saveexec_b64 s[20:21], vcc\nv_add_u32 %0, %0, 1\ns_or_b64 exec, exec, s[20:21]\n") : "+v"(v) : "v"(w) : "vcc", "s20", "s21"));
    // 7: 64 x not-taken branch
    MEASURE(br_not_taken, asm volatile(REP64("s_cmp_eq_u32 %0, 0x12345\ns_cbranch_scc1 1f\nv_add_u32 %1, %1, 1\n1:\n") :: "s"(seed), "v"(v) : "scc"));
    // 8: 64 x taken branch (skips one instruction)
    MEASURE(br_taken, asm volatile(REP64("s_cmp_lg_u32 %0, 0x12345\ns_cbranch_scc1 1f\nv_add_u32 %1, %1, 1\n1:\n") :: "s"(seed), "v"(v) : "scc"));
    // 9: 64 dependent LDS reads (pointer chase, all lanes same address -> broadcast)
    { uint32_t a = 0;
      MEASURE(lds_chase, asm volatile(REP64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(a)));
      v += a; }
    // 10: 64 dependent LDS reads, random per-lane addresses
    { uint32_t a = (threadIdx.x * 68) & 4095;
      MEASURE(lds_chase_div, asm volatile(REP64("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(a)));
      v += a; }
    // 11: 64 x (v_readfirstlane -> s_add -> v_mov), VALU->SALU->VALU
    MEASURE(rfl_chain, asm volatile(REP64("v_readfirstlane_b32 s20, %0\ns_add_u32 s20, s20, 1\nv_mov_b32 %0, s20\n") : "+v"(v) :: "s20", "scc"));
    // 12: 64 x s_memtime pair cost
    MEASURE(memtime, { unsigned long long z = 0; for (int i = 0; i < 16; i++) z += __builtin_readcyclecounter(); v += (uint32_t)z; });
    // 13: 64 x v_cmp writing sgpr pair then s_cbranch_vccz not taken
    MEASURE(vcc_branch, asm volatile(REP64("v_cmp_ne_u32 vcc, %0, %1\ns_cbranch_vccz 1f\nv_add_u32 %0, %0, 1\n1:\n") : "+v"(v) : "v"(w) : "vcc"));
    // 14: 64 x ballot-like: v_cmp -> s_bcnt1 -> s_cmp -> s_cbranch
    MEASURE(ballot_branch, asm volatile(REP64("v_cmp_ne_u32 vcc, %0, %1\ns_bcnt1_i32_b64 s20, vcc\ns_cmp_lt_u32 s20, 100\ns_cbranch_scc0 1f\nv_add_u32 %0, %0, 1\n1:\n") : "+v"(v) : "v"(w) : "vcc", "s20", "scc"));
    // 15: 64 x ds_bpermute dependent
    MEASURE(bperm, asm volatile(REP64("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(w)));
    if (v == 0x7fffffff) out[31] = v;
}

int main()
{
    unsigned long long *d, h[32] = {0};
    hipMalloc(&d, sizeof h);
    const char *names[] = {"empty (memtime pair)", "dependent v_add", "independent v_add", "v_cmp->v_cndmask (vcc)", "v_cmp->s_and->v_cndmask",
                           "dependent s_add", "cmp+saveexec+add+restore", "cmp+branch not taken+add", "cmp+branch taken", "LDS chase broadcast",
                           "LDS chase divergent", "readfirstlane->s_add->v_mov", "16 x memtime", "v_cmp+vccz branch+add", "ballot+bcnt+cmp+branch+add", "ds_bpermute chain"};
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 12345u + rep);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 16; i++) printf("%-32s %6llu cycles total, %.1f per repeat (minus empty)\n", names[i], h[i], (double)((long long)h[i] - (long long)h[0]) / 64.0);
    return 0;
}
