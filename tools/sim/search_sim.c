/* search_sim.c -- development aid: CPU model of the wg kernel's SEARCH phase (chain walk
 * lengths and the lane scheduling), used to evaluate design variants without a GPU.
 * Not part of the library, computes no compressed output.
 *   usage: search_sim [class] [nblocks] [H3 bits] [H2 bits] [refill_min] [use4] [rolling]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);

enum { WINDOW = 2047, CAP = 12, NOLINK = 0xFFFF };
static uint32_t H3B = 10, H2B = 10;
static int use4 = 0;

static uint32_t gram(const uint8_t *s, uint32_t p, uint32_t k, uint32_t n)
{ uint32_t g = 0; for (uint32_t i = 0; i < k; i++) g |= (uint32_t)(p + i < n ? s[p + i] : 0) << (8 * i); return g; }
static uint32_t H3N = 0;
static uint32_t h3(uint32_t g) { return H3N ? (uint32_t)(((uint64_t)(g * 0x9E3779B1u) * H3N) >> 32) : (g * 0x9E3779B1u) >> (32 - H3B); }
static uint32_t h2(uint32_t g) { return ((g * 40503u) >> 6) & ((1u << H2B) - 1); }
static uint32_t h4(uint32_t g) { return (g * 0x9E3779B1u) >> (32 - H3B); }

static uint32_t lcp(const uint8_t *s, uint32_t a, uint32_t b, uint32_t lim)
{ uint32_t l = 0; while (l < lim && s[a + l] == s[b + l]) l++; return l; }

/* steps[p] = step iterations position p occupies a lane for (0 = instant) */
static uint16_t seg3[65536], seg2[65536];
static void walks(const uint8_t *s, uint32_t n, uint16_t *steps, uint64_t *tot3, uint64_t *tot2)
{
    uint32_t *head3 = malloc(4u << H3B), *head2 = malloc(4u << H2B), *head4 = malloc(4u << H3B);
    uint16_t *l3 = malloc(2 * n), *l2 = malloc(2 * n), *l4 = malloc(2 * n);
    memset(head3, 0xFF, 4u << H3B); memset(head2, 0xFF, 4u << H2B); memset(head4, 0xFF, 4u << H3B);
    /* build everything first: links are distances to the previous entry of the bucket */
    for (uint32_t p = 0; p < n; p++) {
        int deep = p >= 1 && p + 13 <= n && s[p - 1] == s[p];
        for (uint32_t k = 1; deep && k < 13; k++) if (s[p + k] != s[p]) deep = 0;
        l3[p] = l2[p] = l4[p] = NOLINK;
        if (!deep && p + 2 < n) { uint32_t h = h3(gram(s, p, 3, n)); uint32_t o = head3[h]; head3[h] = p; if (o != ~0u && p - o < NOLINK) l3[p] = p - o; }
        if (!deep && p + 1 < n) { uint32_t h = h2(gram(s, p, 2, n)); uint32_t o = head2[h]; head2[h] = p; if (o != ~0u && p - o < NOLINK) l2[p] = p - o; }
        if (!deep && p + 3 < n) { uint32_t h = h4(gram(s, p, 4, n)); uint32_t o = head4[h]; head4[h] = p; if (o != ~0u && p - o < NOLINK) l4[p] = p - o; }
    }
    for (uint32_t p = 0; p < n; p++) {
        const uint32_t lim = n - p < CAP ? n - p : CAP, reach = p < WINDOW ? p : WINDOW;
        uint32_t len1 = p >= 1 ? lcp(s, p, p - 1, lim) : 0;
        const int seeded = len1 >= 2, capped = seeded && len1 == lim;
        const int walk3 = lim >= 3 && !capped && l3[p] <= reach;
        const int walk2 = lim >= 2 && !capped && !seeded && l2[p] <= reach;
        const int instant = !walk3 && !walk2 && !(seeded && len1 == CAP);
        uint32_t st = 0, st3 = 0;
        if (!instant) {
            int three = walk3, four = 0;
            uint32_t cum = 0, dist = walk3 ? l3[p] : (walk2 ? l2[p] : NOLINK);
            uint32_t best = seeded ? len1 : 0, beat = walk3 ? (seeded ? len1 : 2) : 1, stop = walk3 ? lim : 2;
            for (;;) {
                st++;
                if (three) (*tot3)++; else (*tot2)++;
                const uint32_t cum2 = cum + dist;
                const int inwin = cum2 <= reach;
                uint32_t len = 0, nd = NOLINK;
                if (inwin) {
                    const uint32_t q = p - cum2;
                    len = lcp(s, p, q, lim);
                    nd = three ? (four ? l4[q] : l3[q]) : l2[q];
                    if (len > beat) { best = len; beat = len; if (use4 && three && !four && len >= 4) { four = 1; nd = l4[q]; } }
                }
                const int ended = !inwin || len >= stop;
                if (ended && three && best < 2) { st3 = st; three = 0; four = 0; cum = 0; dist = walk2 ? l2[p] : NOLINK; beat = 1; stop = 2; if (!walk2) break; continue; }
                if (ended) break;
                cum = cum2; dist = nd;
            }
        }
        steps[p] = (uint16_t)st; seg3[p] = (uint16_t)(st3 ? st3 : st); seg2[p] = (uint16_t)(st3 ? st - st3 : 0);
    }
    free(head3); free(head2); free(head4); free(l3); free(l2); free(l4);
}

int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t nb = argc > 2 ? atoi(argv[2]) : 64, bl = 65536;
    if (argc > 3) { H3B = atoi(argv[3]); if (H3B > 20) { H3N = H3B; H3B = 11; } }
    if (argc > 4) H2B = atoi(argv[4]);
    const uint32_t refill_min = argc > 5 ? atoi(argv[5]) : 32;
    use4 = argc > 6 ? atoi(argv[6]) : 0;
    const int rolling = argc > 7 ? atoi(argv[7]) : 0;
    const uint32_t poolsz = argc > 8 ? atoi(argv[8]) : 512;
    const int halves = argc > 9 ? atoi(argv[9]) : 1;   /* walks per lane */
    const int defer = argc > 10 ? atoi(argv[10]) : 0;
    const int per = argc > 11 ? atoi(argv[11]) : 1;   /* candidates per step */  /* 2-byte chain restarts at the next refill pass */
    uint8_t *buf = malloc((size_t)nb * bl);
    lzs_workload_fill(buf, cls, 0x4C5A5331ull, 0, nb, bl, 8);
    uint16_t *steps = malloc(2 * bl);
    uint64_t tot3 = 0, tot2 = 0, totsteps = 0, npos = 0, ninst = 0, iters = 0, busy_lane_steps = 0, passes = 0, pools = 0;
    uint64_t hist[8] = {0}, maxsum = 0;
    for (uint32_t b = 0; b < nb; b++) {
        walks(buf + (size_t)b * bl, bl, steps, &tot3, &tot2);
        if (per > 1) for (uint32_t p = 0; p < bl; p++) { steps[p] = (steps[p] + per - 1) / per; }
        for (uint32_t p = 0; p < bl; p++) {
            totsteps += steps[p]; npos++; ninst += steps[p] == 0;
            const uint32_t s = steps[p];
            hist[s == 0 ? 0 : s <= 2 ? 1 : s <= 4 ? 2 : s <= 8 ? 3 : s <= 16 ? 4 : s <= 32 ? 5 : s <= 64 ? 6 : 7] += s ? s : 1;
        }
        /* scheduling: 4 waves x 64 lanes, lock-step round robin over waves */
        static uint32_t rem[4][256], own[4][256], pend2[4][256];
        memset(rem, 0, sizeof rem); memset(pend2, 0, sizeof pend2);
        uint32_t nextp = 0;
        const uint32_t pool = rolling == 1 ? bl : poolsz;
        const uint32_t lag = rolling >= 2 ? rolling - 1 : 0;
        for (uint32_t P = 0; P < bl; P += pool) {
            const uint32_t pend = P + pool, k = P / pool;
            int done[4] = {0, 0, 0, 0}, pool_done[4] = {0, 0, 0, 0};
            uint32_t mx = 0; for (uint32_t p = P; p < pend; p++) if (steps[p] > mx) mx = steps[p];
            maxsum += mx; pools++;
            nextp = P;
            const int last = pend >= bl;
            while (!(done[0] && done[1] && done[2] && done[3])) {
                for (int w = 0; w < 4; w++) {
                    if (done[w]) continue;
                    for (int h = 0; h < halves; h++) {
                        uint32_t nidle = 0, npend = 0; for (int l = 64 * h; l < 64 * h + 64; l++) { nidle += rem[w][l] == 0; npend += rem[w][l] == 0 && pend2[w][l]; }
                        if ((!pool_done[w] && (nidle >= refill_min || nidle == 64)) || (pool_done[w] && npend)) {
                            uint32_t nnew = nidle - npend;
                            uint32_t basep = nextp;
                            if (!pool_done[w]) { nextp += nnew; pool_done[w] = basep + nnew >= pend; } else nnew = 0;
                            uint32_t r = 0;
                            for (int l = 64 * h; l < 64 * h + 64; l++) if (rem[w][l] == 0) {
                                if (pend2[w][l]) { rem[w][l] = pend2[w][l]; pend2[w][l] = 0; continue; }
                                if (!nnew) continue;
                                uint32_t np = basep + r++;
                                if (np < pend) { own[w][l] = k; if (defer) { rem[w][l] = seg3[np]; pend2[w][l] = seg2[np]; } else rem[w][l] = steps[np]; }
                            }
                            passes++;
                        }
                    }
                    uint32_t nbusy = 0, nold = 0;
                    for (int l = 0; l < 64 * halves; l++) { nbusy += rem[w][l] != 0; nold += (rem[w][l] != 0 || pend2[w][l]) && (last || own[w][l] + lag <= k); }
                    if (pool_done[w] && nold == 0) { done[w] = 1; continue; }
                    if (nbusy == 0) continue;
                    iters++; busy_lane_steps += nbusy;
                    for (int l = 0; l < 64 * halves; l++) if (rem[w][l]) rem[w][l]--;
                }
            }
        }
    }
    printf("class %u H3=%u H2=%u refill>=%u use4=%d rolling=%d\n", cls, 1u << H3B, 1u << H2B, refill_min, use4, rolling);
    printf("  steps/position %.2f (3-chain %.2f, 2-chain %.2f), instant %.1f%%\n", (double)totsteps / npos, (double)tot3 / npos, (double)tot2 / npos, 100.0 * ninst / npos);
    printf("  all 4 waves per 512 positions: refill passes %.2f, step iterations %.2f (busy lanes %.1f), mean longest walk per pool %.1f\n",
           (double)passes / (npos / 512), (double)iters / (npos / 512), (double)busy_lane_steps / (iters ? iters : 1), (double)maxsum / pools);
    printf("  share of lane-steps by walk length: inst %.1f%% 1-2 %.1f%% 3-4 %.1f%% 5-8 %.1f%% 9-16 %.1f%% 17-32 %.1f%% 33-64 %.1f%% >64 %.1f%%\n",
           100.0 * hist[0] / totsteps, 100.0 * hist[1] / totsteps, 100.0 * hist[2] / totsteps, 100.0 * hist[3] / totsteps,
           100.0 * hist[4] / totsteps, 100.0 * hist[5] / totsteps, 100.0 * hist[6] / totsteps, 100.0 * hist[7] / totsteps);
    return 0;
}
