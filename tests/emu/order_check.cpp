// order_check.cpp -- the pieces of round 5's order check (store_kernels.h: ForeignBound and the kernels around it), checked on the
// host against the product's own header (tests only):
//   1. img_bits64 / rec_bits64 give the bits a base-by-base loop gives (intervals across word borders; img_bits64 also before the
//      genome's start and past its end);
//   2. trim_on_masks -- Aligner::trim on 64-bit masks, what the check decides a noted candidate with -- leaves exactly the
//      (shift, length) that settle_row leaves on the layout image the masks were read from (random marks, up to 70 genomes, 5 to
//      64 bases), and both equal a genome-by-genome restatement of trim() (TMum::trimleft / trimright: src/TMum.cpp:104-148);
//   3. the bound argument: with marks m_lo <= m <= m_hi in every genome the intervals left nest, I_hi within I within I_lo;
//   4. order_key orders regions by (reference start, generation), the first seed (generation 0) before everything.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../parsnp_amd/csrc/engine/store_kernels.h"
using namespace pm;

// Aligner::trim as the reference walks it: genome after genome, marked bases off the start (the start moves in every genome),
// then off the end
static void trim_plain(const std::vector<std::vector<uint8_t>>& bits, const std::vector<int64_t>& st, int32_t lon, int32_t* pdl, int32_t* plen) {
    int32_t dl = 0, len = lon;
    for (size_t j = 0; j < bits.size() && len > 0; j++) {
        while (len > 0 && (size_t)(st[j] + dl) < bits[j].size() && bits[j][(size_t)(st[j] + dl)]) { dl++; len--; }
        while (len > 0 && (size_t)(st[j] + dl + len - 1) < bits[j].size() && bits[j][(size_t)(st[j] + dl + len - 1)]) len--;
    }
    *pdl = dl; *plen = len;
}

int main() {
    std::mt19937_64 rng(20260928);
    long checked = 0;
    for (int it = 0; it < 3000; it++) {
        const int n = 2 + (int)(rng() % 69);
        std::vector<int64_t> word_off(n + 1), nbits(n);
        std::vector<std::vector<uint8_t>> bits(n);
        int64_t words = 0;
        const int density = (int)(rng() % 4);      // sparse marks ... mostly marked
        for (int j = 0; j < n; j++) {
            const int64_t glen = 200 + (int64_t)(rng() % 900);
            nbits[j] = glen + 1;
            word_off[j] = words; words += (nbits[j] + 63) / 64;
            bits[j].assign((size_t)nbits[j], 0);
            for (int64_t p = 0; p < glen; ) {      // runs of marked and unmarked bases
                const int64_t run = 1 + (int64_t)(rng() % (density == 0 ? 40 : 12));
                const bool mark = density == 3 ? rng() % 4 != 0 : density == 0 ? rng() % 6 == 0 : rng() % 2 == 0;
                for (int64_t q = p; q < p + run && q < glen; q++) bits[j][(size_t)q] = mark;
                p += run;
            }
            bits[j][(size_t)glen] = 1;      // the sentinel
        }
        word_off[n] = words;
        std::vector<uint64_t> image((size_t)words, 0);
        std::vector<uint8_t> rec((size_t)words, 0);
        for (int j = 0; j < n; j++)
            for (int64_t p = 0; p < nbits[j]; p++) if (bits[j][(size_t)p]) image[(size_t)(word_off[j] + (p >> 6))] |= 1ull << (p & 63);
        for (int64_t w = 0; w < words; w++) rec[(size_t)w] = rng() % 3 == 0;
        Layout L{image.data(), word_off.data(), nbits.data(), 0};
        // 1. the mask readers
        for (int t = 0; t < 200; t++) {
            const int j = (int)(rng() % n);
            const int32_t len = 1 + (int32_t)(rng() % 64);
            const int64_t a = (int64_t)(rng() % (uint64_t)(nbits[j] + 40)) - 20;
            uint64_t want = 0, want_rec = 0;
            for (int32_t x = 0; x < len; x++) {
                const int64_t p = a + x;
                if (p >= 0 && p < nbits[j] && bits[j][(size_t)p]) want |= 1ull << x;
                if (p >= 0 && p < nbits[j] && rec[(size_t)(word_off[j] + (p >> 6))]) want_rec |= 1ull << x;
            }
            if (img_bits64(L, j, a, len) != want) { printf("img_bits64 differs: genome %d a %ld len %d\n", j, (long)a, len); return 1; }
            // (rec_bits64 is only asked about intervals inside the genome: a noted candidate lies inside every genome)
            if (a >= 0 && a + len <= nbits[j] && rec_bits64(rec.data(), word_off[j], nbits[j], a, len) != want_rec) { printf("rec_bits64 differs: genome %d a %ld len %d\n", j, (long)a, len); return 1; }
            checked++;
        }
        // 2. trimming on masks = trimming on the image = the reference's loop
        const int rows = 40;
        std::vector<int32_t> start((size_t)rows * n), lon(rows), shift(rows, 0), len_(rows, 0);
        std::vector<uint8_t> strand((size_t)rows * n, 1), state(rows, 0);
        std::vector<uint32_t> flags(rows, 0);
        for (int c = 0; c < rows; c++) {
            lon[c] = 5 + (int32_t)(rng() % 60);
            for (int j = 0; j < n; j++) start[(size_t)c * n + j] = (int32_t)(rng() % (uint64_t)(nbits[j] - 1 - lon[c]));
        }
        Store S{start.data(), strand.data(), lon.data(), flags.data(), shift.data(), len_.data(), state.data(), n};
        Packed P{nullptr, nullptr, nullptr};
        for (int c = 0; c < rows; c++) {
            int32_t dl0, len0, dl1, len1, dl2, len2;
            (void)settle_row(S, L, P, c, true, &dl0, &len0);
            std::vector<uint64_t> M(n);
            std::vector<int64_t> st(n);
            for (int j = 0; j < n; j++) { st[j] = start[(size_t)c * n + j]; M[j] = img_bits64(L, j, st[j], lon[c]); }
            trim_on_masks(n, lon[c], [&](int j) -> uint64_t { return M[j]; }, &dl1, &len1);
            trim_plain(bits, st, lon[c], &dl2, &len2);
            if (len2 <= 0) { if (len0 > 0 || len1 > 0) { printf("trimmed away by the loop, not by the kernels: row %d (%d %d / %d %d)\n", c, dl0, len0, dl1, len1); return 1; } }
            else if (dl0 != dl2 || len0 != len2 || dl1 != dl2 || len1 != len2) { printf("trim differs: row %d image (%d, %d) masks (%d, %d) loop (%d, %d)\n", c, dl0, len0, dl1, len1, dl2, len2); return 1; }
            // 3. nested marks -> nested intervals
            std::vector<uint64_t> lo(n), hi(n);
            for (int j = 0; j < n; j++) { lo[j] = M[j] & rng() & rng(); hi[j] = M[j] | (rng() & rng() & (lon[c] == 64 ? ~0ull : ((1ull << lon[c]) - 1))); }
            int32_t dlo, llo, dhi, lhi;
            trim_on_masks(n, lon[c], [&](int j) -> uint64_t { return lo[j]; }, &dlo, &llo);
            trim_on_masks(n, lon[c], [&](int j) -> uint64_t { return hi[j]; }, &dhi, &lhi);
            if (len1 > 0 && !(dlo <= dl1 && dl1 + len1 <= dlo + llo)) { printf("the interval is not inside the lower bound's: row %d\n", c); return 1; }
            if (lhi > 0 && !(len1 > 0 && dl1 <= dhi && dhi + lhi <= dl1 + len1)) { printf("the upper bound's interval is not inside the interval: row %d\n", c); return 1; }
            checked++;
        }
    }
    // 4. the order key
    for (int t = 0; t < 200000; t++) {
        const int64_t a = (int64_t)(rng() % 6000000), b = (int64_t)(rng() % 3 == 0 ? a : rng() % 6000000);
        const int32_t ga = (int32_t)(rng() % 40), gb = (int32_t)(rng() % 40);
        const int64_t ka = order_key(a, ga), kb = order_key(b, gb);
        if (ga == 0 && ka != -1) { printf("the first seed's key is not -1\n"); return 1; }
        if (ga > 0 && gb > 0 && ((a < b || (a == b && ga < gb)) != (ka < kb))) { printf("order_key does not order (%ld, %d) and (%ld, %d)\n", (long)a, ga, (long)b, gb); return 1; }
        if (ga == 0 && gb > 0 && !(ka < kb)) { printf("the first seed does not come first\n"); return 1; }
        checked++;
    }
    printf("ok %ld checks\n", checked);
    return 0;
}
