/* parse_exits_sim.c -- development aid (CPU only): VERDICT r05 item 1(a), "PARSE on demand".  The kernel resolves, for EVERY one
 * of a 64-position chunk's entry positions, where the greedy token chain (lzs-compression.c:301-447) leaves the chunk -- six
 * rounds of pointer doubling -- although the chain enters the chunk at one position only.  Chains of an LZ parse re-synchronise;
 * this counts, on the seeded blocks, (1) how many DISTINCT exits a chunk's 64 entries lead to, (2) how many distinct exits the
 * entries lead to that the chunk before can produce at all (its own possible exits, transitively from any entry of the pool's
 * first chunk), (3) how many tokens the chain takes through a chunk -- the dependent steps a walk "from the <= 4 candidate
 * entries only" would need where doubling needs 6 -- and (4) after how many tokens two chains entering at different
 * positions of a chunk have merged.
 *   build: gcc -O2 tools/sim/parse_exits_sim.c -Llzs_compression_amd -llzs_workload -Loracle -llzs_oracle -Wl,-rpath,$PWD/lzs_compression_amd -Wl,-rpath,$PWD/oracle -o tools/sim/parse_exits_sim
 *   usage: parse_exits_sim [class 0..2] [nblocks]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);
size_t lzs_oracle_trace(const uint8_t *in, size_t n, uint32_t *rec, size_t max_tok);   /* the checker: the real parse */
enum { N = 65536, CHUNK = 64, POOL = 512 };

int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0, nblocks = argc > 2 ? atoi(argv[2]) : 8;
    uint8_t *s = malloc(N + 16);
    static uint32_t step[N + 1];            /* bytes the token that starts at p covers (whatever the parse before p was) */
    static uint32_t rec[3 * N];
    unsigned long long chunks = 0, h_all[66] = {0}, h_poss[66] = {0}, tok_sum = 0, tok_max = 0, merge_sum = 0, merge_n = 0, merge_never = 0, h_merge[66] = {0};
    unsigned long long poss_entries_sum = 0;
    for (unsigned b = 0; b < nblocks; b++) {
        lzs_workload_fill(s, cls, 0x4C5A5331ull, b, 1, N, 1);
        /* step(p): the oracle's own parse started at p gives the token there (the search is a pure function of (input, p):
           SURVEY.md App. A.2) -- one trace per position would be quadratic, so: trace of the suffix parse from every p is
           not needed; the token AT p only depends on bytes before p (the window) and after, so run the trace on the whole
           block from 0 for the real starts and use a brute-force matcher of our own for the others */
        for (uint32_t p = 0; p < N; p++) {
            const uint32_t lim = N - p < 12 ? N - p : 12, reach = p < 2047 ? p : 2047;
            uint32_t best = 0, boff = 0;
            if (lim >= 2)
                for (uint32_t d = 1; d <= reach; d++) {
                    if (s[p - d] != s[p] || s[p - d + 1] != s[p + 1]) continue;
                    uint32_t l = 2; while (l < lim && s[p - d + l] == s[p + l]) l++;
                    if (l > best) { best = l; boff = d; if (l == lim) break; }
                }
            uint32_t len = best < 2 ? 1 : best;
            if (best >= 8) { len = 8; for (;;) { const uint32_t c = p + len; uint32_t lim2 = N - c < 15 ? N - c : 15, e = 0; while (e < lim2 && s[c + e] == s[c + e - boff]) e++; len += e; if (e != 15) break; } }
            step[p] = len;
        }
        const size_t nt = lzs_oracle_trace(s, N, rec, N);
        for (size_t t = 0; t < nt; t++)
            if (step[rec[3 * t]] != rec[3 * t + 2]) { fprintf(stderr, "block %u: token at %u: %u bytes here, %u in the oracle's parse\n", b, rec[3 * t], step[rec[3 * t]], rec[3 * t + 2]); return 1; }
        for (uint32_t P = 0; P + POOL <= N; P += POOL) {
            uint8_t poss[POOL + 4096]; memset(poss, 0, sizeof poss);     /* pool-relative positions a chain from any entry of chunk 0 can stand on */
            for (uint32_t e = 0; e < CHUNK; e++) poss[e] = 1;
            for (uint32_t c0 = 0; c0 < POOL; c0 += CHUNK) {
                uint32_t exits_all[CHUNK], na = 0, exits_p[CHUNK], np = 0, npe = 0;
                for (uint32_t e = 0; e < CHUNK; e++) {
                    uint32_t q = P + c0 + e, toks = 0;
                    while (q < P + c0 + CHUNK && q < N) { q += step[q]; toks++; }
                    uint32_t k; for (k = 0; k < na && exits_all[k] != q; k++) {} if (k == na) exits_all[na++] = q;
                    if (poss[c0 + e]) {
                        npe++;
                        for (k = 0; k < np && exits_p[k] != q; k++) {} if (k == np) exits_p[np++] = q;
                        if (q - P < sizeof poss) poss[q - P] = 1;
                        tok_sum += toks; if (toks > tok_max) tok_max = toks;
                    }
                    /* every position a possible chain stands on inside the chunk is a possible entry of nothing else: only exits matter */
                }
                /* tokens until the chains from entries e and e + 1 stand on the same position */
                for (uint32_t e = 0; e + 1 < CHUNK; e += 7) {
                    uint32_t a = P + c0 + e, bq = a + 1, toks = 0;
                    while (a != bq && a < N && bq < N && toks < 64) { if (a < bq) a += step[a]; else bq += step[bq]; toks++; }
                    if (a == bq) { merge_sum += toks; merge_n++; h_merge[toks < 65 ? toks : 65]++; } else merge_never++;
                }
                h_all[na]++; h_poss[np]++; poss_entries_sum += npe; chunks++;
            }
        }
    }
    printf("class %u, %u blocks of 64 KiB, %llu chunks of 64 positions (pools of 512; every token of the oracle's parse has the length this model gives it)\n", cls, nblocks, chunks);
    printf("distinct exits over ALL 64 entries of a chunk:      ");
    for (int k = 1; k <= 12; k++) printf(" %d: %.1f%%", k, 100.0 * h_all[k] / chunks);
    printf("\ndistinct exits over the entries the pool can produce:");
    for (int k = 1; k <= 12; k++) printf(" %d: %.1f%%", k, 100.0 * h_poss[k] / chunks);
    printf("\npossible entries per chunk (mean): %.1f; tokens a chain takes through a chunk: mean %.1f, max %llu (the dependent steps of a walk; doubling: 6 rounds for all 64 entries)\n",
           (double)poss_entries_sum / chunks, (double)tok_sum / poss_entries_sum, tok_max);
    printf("two chains entering one position apart have merged after (tokens taken, both chains together): mean %.1f;", (double)merge_sum / merge_n);
    unsigned long long acc = 0; for (int k = 0; k <= 65; k++) { acc += h_merge[k]; if (k == 2 || k == 4 || k == 8 || k == 16 || k == 32) printf(" <= %d: %.1f%%", k, 100.0 * acc / (merge_n + merge_never)); }
    printf("; never within 64: %.2f%%\n", 100.0 * merge_never / (merge_n + merge_never));
    return 0;
}
