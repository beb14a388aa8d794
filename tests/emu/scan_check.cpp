// scan_check.cpp -- properties the wavefront kernels rely on, checked on the host against the product's own header (tests only):
//   1. win_join / pair_join (kernels.h) are ASSOCIATIVE on the states the scan builds -- what allows WaveScan to fold the events of a
//      pair in any grouping (lanes, rounds, wavefront summaries) -- and a fold of single-event states equals the sequential rule of
//      Test_UM + Intersect_UM's carry: furthest end, its first event in (l, j) order, second furthest end;
//   2. a 64-lane segmented Hillis-Steele scan with the kernel's update rule (take the value `d` lanes up unless a pair opened in
//      between) gives every lane the fold of its pair's events so far, across rounds with the carry of the last lane;
//   3. xcd_item maps the xcd_grid(n) workgroups of a launch onto the items 0..n-1 exactly once each.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../parsnp_amd/csrc/engine/kernels.h"
using namespace pm;

static bool same(const WinState& a, const WinState& b) { return a.e1 == b.e1 && (a.e1 == 0 || (a.e2 == b.e2 && a.wl == b.wl && a.wj == b.wj && a.wr == b.wr)); }
static bool same(const PairState& a, const PairState& b) { return same(a.s[0], b.s[0]) && same(a.s[1], b.s[1]); }

// the sequential rule, written from the reference's description (not from win_join): per strand
struct Seq { int32_t e1 = 0, e2 = 0, wl = 0, wj = 0, wr = 0; bool have = false;
    void push(int32_t l, int32_t j, int32_t end, int32_t rp) {
        const bool take = !have || end > e1 || (end == e1 && (l < wl || (l == wl && j < wj)));
        if (take) { if (have && e1 > e2) e2 = e1; e1 = end; wl = l; wj = j; wr = rp; have = true; }
        else if (end > e2) e2 = end;
    } };

int main() {
    std::mt19937_64 rng(12345);
    long checked = 0;
    for (int it = 0; it < 20000; it++) {
        const int n = 1 + (int)(rng() % 200);
        const uint64_t lmask = (1ull << 20) - 1; const int lbits = 20;
        std::vector<uint64_t> key(n), val(n); std::vector<int32_t> rp(n);
        // events in the order of the sort key: pairs ascending, l ascending, strand; few distinct ends so that ties happen
        uint64_t pair = rng() % 3; int32_t l = 0;
        for (int i = 0; i < n; i++) {
            if (rng() % 17 == 0) { pair += 1 + rng() % 2; l = 0; }
            l += (int32_t)(rng() % 3);
            const int strand = (int)(rng() % 2);
            const int32_t len = 1 + (int32_t)(rng() % 6), j = (int32_t)(rng() % 5);
            key[i] = (((pair << lbits) | (uint64_t)l) << 1) | (uint64_t)strand;
            val[i] = ((uint64_t)j << 32) | (uint32_t)len; rp[i] = (int32_t)(rng() % 4);
        }
        // (keys with equal (pair, l) must have strand 0 before 1: sort adjacent swaps)
        for (int i = 1; i < n; i++) if (key[i] < key[i - 1]) { std::swap(key[i], key[i - 1]); std::swap(val[i], val[i - 1]); std::swap(rp[i], rp[i - 1]); if (i > 1) i -= 2; }
        // 1. sequential rule vs left fold of win_join vs a random grouping
        std::vector<PairState> want(n);
        { Seq s[2]; uint64_t cur = ~0ull;
          for (int i = 0; i < n; i++) {
              const uint64_t p = key[i] >> (lbits + 1);
              if (p != cur) { s[0] = Seq(); s[1] = Seq(); cur = p; }
              const int32_t li = (int32_t)((key[i] >> 1) & lmask);
              s[key[i] & 1].push(li, (int32_t)(val[i] >> 32), li + (int32_t)(val[i] & 0xffffffffu), rp[i]);
              for (int sd = 0; sd < 2; sd++) want[i].s[sd] = s[sd].have ? WinState{s[sd].e1, s[sd].e2, s[sd].wl, s[sd].wj, s[sd].wr} : WinState{0, 0, 0, 0, 0};
          } }
        // 2. the 64-lane segmented scan, round by round with the carry of the last lane
        std::vector<PairState> got(n);
        PairState carry; carry.s[0] = WinState{0, 0, 0, 0, 0}; carry.s[1] = carry.s[0];
        for (int base = 0; base < n; base += 64) {
            PairState x[64]; int h[64];
            for (int t = 0; t < 64; t++) {
                const int i = base + t;
                x[t].s[0] = WinState{0, 0, 0, 0, 0}; x[t].s[1] = x[t].s[0]; h[t] = 0;
                if (i < n) { x[t] = pair_of_event(key[i], val[i], lmask, rp[i]); h[t] = i == 0 || (key[i - 1] >> (lbits + 1)) != (key[i] >> (lbits + 1)); }
            }
            for (int d = 1; d < 64; d <<= 1) {
                PairState y[64]; int hy[64];
                for (int t = 0; t < 64; t++) { y[t] = t >= d ? x[t - d] : x[t]; hy[t] = t >= d ? h[t - d] : h[t]; }      // (__shfl_up)
                for (int t = 0; t < 64; t++) if (t >= d && !h[t]) { x[t] = pair_join(y[t], x[t]); h[t] = hy[t]; }
            }
            for (int t = 0; t < 64; t++) if (!h[t]) x[t] = pair_join(carry, x[t]);
            const int last = (n - base < 64 ? n - base : 64) - 1;
            for (int t = 0; t <= last; t++) got[base + t] = x[t];
            carry = x[last];
        }
        for (int i = 0; i < n; i++) {
            if (!same(want[i], got[i])) { printf("scan differs from the sequential rule: iteration %d event %d\n", it, i); return 1; }
            checked++;
        }
        // associativity on three random consecutive groups of one pair
        for (int rep = 0; rep < 8; rep++) {
            int a = (int)(rng() % n), b = a + (int)(rng() % 6), c = b + (int)(rng() % 6), d = c + (int)(rng() % 6);
            if (d > n) continue;
            auto fold = [&](int lo, int hi) { PairState s; s.s[0] = WinState{0, 0, 0, 0, 0}; s.s[1] = s.s[0]; for (int i = lo; i < hi; i++) s = pair_join(s, pair_of_event(key[i], val[i], lmask, rp[i])); return s; };
            const PairState A = fold(a, b), B = fold(b, c), C = fold(c, d);
            if (!same(pair_join(pair_join(A, B), C), pair_join(A, pair_join(B, C)))) { printf("pair_join is not associative: iteration %d\n", it); return 1; }
        }
    }
    // 3. xcd_item is a bijection of the launch onto the items
    for (int64_t n = 1; n < 3000; n += (n < 70 ? 1 : 37)) {
        std::vector<int> seen((size_t)n, 0);
        for (int64_t w = 0; w < xcd_grid(n); w++) { const int64_t i = xcd_item(w, n); if (i < 0) { printf("negative item\n"); return 1; } if (i < n) seen[(size_t)i]++; }
        for (int64_t i = 0; i < n; i++) if (seen[(size_t)i] != 1) { printf("xcd_item: item %ld of %ld taken %d times\n", (long)i, (long)n, seen[(size_t)i]); return 1; }
    }
    printf("ok: %ld event states\n", checked);
    return 0;
}
