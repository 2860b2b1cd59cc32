// Two LDS images with their own layouts -- image R for the row phase (its exchanges are wave-local and in place; the ownership
// change at the end of the column phase writes INTO it), image C for the column phase -- instead of one layout for all four
// line roles (lds_layout_search.c): same barriers, no new hazards, 91 KB of LDS.  Same bank / time model and use counts.
// Result for P = 72 (./lds_two_image_search 72 8 9 120): best single layout 5150 cycles per propagation at (1, 65, 73); best
// image R 2222 at (1, 65, 73) + best image C 2746 at (9, 1, 81) = 4968: -3.5 % of a pipe that is 63 % busy.  Not built.
// build: gcc -O2 -o lds_two_image_search lds_two_image_search.c
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static int N, R1, R2, G, LPW, NW;
static int posx(int x, int PA, int PB) { return (x / R2) * PA + (x % R2) * PB; }
static int group_cycles(const int* e, const int* act, int lo, int hi, int mod) {
    int seen[64][64]; int ns[64];
    memset(ns, 0, sizeof ns);
    for (int l = lo; l < hi; ++l) {
        if (!act[l]) continue;
        int b = ((e[l] % mod) + mod) % mod, dup = 0;
        for (int j = 0; j < ns[b]; ++j) if (seen[b][j] == e[l]) dup = 1;
        if (!dup) seen[b][ns[b]++] = e[l];
    }
    int m = 0; for (int b = 0; b < mod; ++b) if (ns[b] > m) m = ns[b];
    return m;
}
// cycles of pattern pat (0 ROW_P1, 1 ROW_P2, 2 COL_P1, 3 COL_P2) as loads (rd) and stores (wr), time model
static void pat_cycles(int pat, int PA, int PB, int Q, long* rd, long* wr) {
    *rd = 0; *wr = 0;
    for (int w = 0; w < NW; ++w) {
        int R = (pat == 0 || pat == 2) ? R1 : R2;
        for (int k = 0; k < R; ++k) {
            int e[64], act[64];
            for (int l = 0; l < 64; ++l) {
                int li = l / G, t = l % G, line = w * LPW + li;
                act[l] = (li < LPW) && (line < N) && ((pat == 0 || pat == 2) ? (t < R2) : (t < R1));
                if (!act[l]) { e[l] = 0; continue; }
                int idx = (pat == 0 || pat == 2) ? (k * R2 + t) : (t * R2 + k);
                e[l] = (pat < 2) ? line * Q + posx(idx, PA, PB) : idx * Q + posx(line, PA, PB);
            }
            int r_ = group_cycles(e, act, 0, 32, 32) + group_cycles(e, act, 32, 64, 32);
            int w_ = 0;
            for (int g = 0; g < 4; ++g) w_ += group_cycles(e, act, 16 * g, 16 * g + 16, 16);
            if (w_ < 6) w_ = 6;
            if (r_ < 2) r_ = 2;
            *rd += r_; *wr += w_;
        }
    }
}
int main(int argc, char** argv) {
    N = atoi(argv[1]); R1 = atoi(argv[2]); R2 = atoi(argv[3]); int Qmax = atoi(argv[4]);
    G = R1 > R2 ? R1 : R2; LPW = 64 / G; NW = (N + LPW - 1) / LPW;
    static char used[1 << 20];
    long bestR = 1 << 30, bestC = 1 << 30, bestOne = 1 << 30; int bR[3] = {0}, bC[3] = {0}, bO[3] = {0};
    for (int Q = N; Q <= Qmax; ++Q)
        for (int PA = 1; PA <= Q; ++PA)
            for (int PB = 1; PB <= Q; ++PB) {
                int mx = 0, ok = 1;
                for (int x = 0; x < N; ++x) { int p = posx(x, PA, PB); if (p > mx) mx = p; }
                long fld = (long)(N - 1) * Q + mx + 1;
                if (fld >= (1 << 20) || fld * 8 > 70000) continue;
                memset(used, 0, fld);
                for (int y = 0; y < N && ok; ++y) for (int x = 0; x < N; ++x) { int a = y * Q + posx(x, PA, PB); if (used[a]) { ok = 0; break; } used[a] = 1; }
                if (!ok) continue;
                long rd[4], wr[4];
                for (int p = 0; p < 4; ++p) pat_cycles(p, PA, PB, Q, &rd[p], &wr[p]);
                long sR = rd[0] + wr[0] + 2 * rd[1] + wr[1] + wr[2];          // image R
                long sC = 2 * rd[2] + wr[2] + rd[3] + wr[3] + wr[1];          // image C
                long sOne = rd[0] + wr[0] + 2 * rd[1] + 2 * wr[1] + 2 * rd[2] + 2 * wr[2] + rd[3] + wr[3];
                if (sR < bestR) { bestR = sR; bR[0] = PA; bR[1] = PB; bR[2] = Q; }
                if (sC < bestC) { bestC = sC; bC[0] = PA; bC[1] = PB; bC[2] = Q; }
                if (sOne < bestOne) { bestOne = sOne; bO[0] = PA; bO[1] = PB; bO[2] = Q; }
            }
    long fl_rd = 0, fl_wr = 0;
    printf("best single layout: %ld at (PA=%d PB=%d Q=%d)\n", bestOne, bO[0], bO[1], bO[2]);
    printf("best image R: %ld at (%d %d %d); best image C: %ld at (%d %d %d); sum %ld\n", bestR, bR[0], bR[1], bR[2], bestC, bC[0], bC[1], bC[2], bestR + bestC);
    (void)fl_rd; (void)fl_wr;
    return 0;
}
