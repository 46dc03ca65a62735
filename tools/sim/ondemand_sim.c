/* ondemand_sim.c -- development aid (CPU only, no output): what if SEARCH ran the full walk only for the positions a token
 * can start at?  Per pool of 512 positions: guess the token starts from a one-candidate search of every position, search
 * those in full, parse again with what is known, search the starts that turned up new, ... until the parse from the pool's
 * entry to its end stands on fully searched positions only.  Prints the rounds a pool needs and the share of positions
 * that are searched in full (the kernel today: all of them, a third of which start a token).
 *   usage: ondemand_sim [class 0..2] [nblocks]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);
enum { WINDOW = 2047, CAP = 12, N = 65536, POOL = 512 };

static uint32_t lcp(const uint8_t *s, uint32_t a, uint32_t b, uint32_t lim) { uint32_t l = 0; while (l < lim && s[a + l] == s[b + l]) l++; return l; }

int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0, nblocks = argc > 2 ? atoi(argv[2]) : 4;
    uint8_t *s = malloc(N);
    static uint32_t tot[N], tot1[N];         /* bytes a token at p covers: exact, and by the one-candidate search */
    static uint8_t known[N];
    unsigned long long pools = 0, rounds_sum = 0, rounds_max = 0, full = 0, positions = 0, starts = 0, hist[16] = {0};
    unsigned long long first_round = 0, lanes_busy = 0, lane_slots = 0;
    for (unsigned b = 0; b < nblocks; b++) {
        lzs_workload_fill(s, cls, 0x4C5A5331ull, b, 1, N, 1);
        static int32_t last3[1 << 24]; memset(last3, 0xFF, sizeof last3);
        static int32_t last2[1 << 16]; memset(last2, 0xFF, sizeof last2);
        for (uint32_t p = 0; p < N; p++) {
            const uint32_t lim = N - p < CAP ? N - p : CAP, reach = p < WINDOW ? p : WINDOW;
            uint32_t best = 0, off = 0;
            for (uint32_t o = 1; o <= reach && best < lim; o++) { const uint32_t l = lcp(s, p, p - o, lim); if (l > best) { best = l; off = o; } }
            uint32_t t = best < 2 ? 1 : best;
            if (best == CAP) t += lcp(s, p + CAP, p + CAP - off, N - p - CAP);
            tot[p] = t;
            /* one candidate: the nearest earlier position with the same three bytes, else with the same two */
            uint32_t b1 = 0, o1 = 0;
            if (p + 2 < N) { const uint32_t g = s[p] | s[p + 1] << 8 | (uint32_t)s[p + 2] << 16; if (last3[g] >= 0 && p - last3[g] <= WINDOW) { o1 = p - last3[g]; b1 = lcp(s, p, p - o1, lim); } last3[g] = p; }
            if (p + 1 < N) { const uint32_t g = s[p] | s[p + 1] << 8; if (!b1 && last2[g] >= 0 && p - last2[g] <= WINDOW) { o1 = p - last2[g]; b1 = lcp(s, p, p - o1, lim); } last2[g] = p; }
            uint32_t t1 = b1 < 2 ? 1 : b1;
            if (b1 == CAP) t1 += lcp(s, p + CAP, p + CAP - o1, N - p - CAP);
            tot1[p] = t1;
        }
        uint32_t entry = 0;
        for (uint32_t base = 0; base < N; base += POOL) {
            const uint32_t end = base + POOL;
            memset(known + base, 0, POOL);
            if (entry >= end) continue;                                   /* a long match ran over the whole pool */
            unsigned rounds = 0;
            for (;;) {
                /* parse from the entry with what is known (exact) or guessed; collect the starts not searched in full yet */
                uint32_t p = entry, fresh = 0;
                while (p < end) { if (!known[p]) { known[p] = 2; fresh++; } p += (known[p] == 1) ? tot[p] : tot1[p]; }
                if (!fresh) break;
                rounds++;
                if (rounds == 1) first_round += fresh;
                full += fresh; lanes_busy += fresh; lane_slots += 256 * ((fresh + 255) / 256);
                for (uint32_t q = base; q < end; q++) if (known[q] == 2) known[q] = 1;
            }
            uint32_t p = entry; while (p < end) { starts++; p += tot[p]; }
            entry = p;
            pools++; rounds_sum += rounds; if (rounds > rounds_max) rounds_max = rounds; hist[rounds < 15 ? rounds : 15]++;
            positions += POOL;
        }
    }
    printf("class %u, %u blocks: %llu pools; token starts %.1f %% of the positions; searched in full %.1f %% (first round %.1f %%);\n"
           "rounds per pool: mean %.2f, max %llu; lanes with work in a round of 256: %.0f %%\nrounds histogram:", cls, nblocks, pools,
           100.0 * starts / positions, 100.0 * full / positions, 100.0 * first_round / positions, (double)rounds_sum / pools, rounds_max,
           100.0 * lanes_busy / lane_slots);
    for (int i = 0; i < 16; i++) printf(" %d:%llu", i, hist[i]);
    printf("\n");
    return 0;
}
