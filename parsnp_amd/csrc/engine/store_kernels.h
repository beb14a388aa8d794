// store_kernels.h -- device code of the RESIDENT route: what the reference does with a candidate list AFTER csgmum has
// produced it -- the second half of Aligner::setMums1 (validation, src/parsnp.cpp:1713-1841), Aligner::trim (:1399-1477),
// determineRegion (:1199-1290), the work-list generations of doWork (:173-317), the pairwise test of setFinalClusters
// (:2596-2700) and setInterClusterRegions (:2389-2460) -- as kernels over data that never leaves the device:
//
//   MUM store     the rows of every candidate list of the run (int32 start per genome + strand byte, as CompactCandidates
//                 wrote them, never rewritten), plus three scalars per row that the validation owns: `shift` (bases trimmed on
//                 the left: the start moves right in EVERY genome, TMum::trimleft), `len` (current length) and `state`.
//                 Rows [0, A) are the anchor table of the anchor call; every later search appends its candidates.
//   layout image  one bit per base of every genome + a sentinel (the reference's mumlayout, :3181-3186), 64-bit words,
//                 genome after genome; marks are atomic ORs, reads are L2-coherent loads (a wavefront must see the marks of
//                 the candidate it accepted a moment ago, and those were made by atomics that bypass its L1).
//   region store  request rows (int64 start / length per genome) of the seed regions and of every child region.
//
// One WAVEFRONT per work item (row, cluster, pair of LCBs); the lanes take the genomes.  The bodies are written ONCE for both
// executions: `lanes_for` hands lane t the genomes t, t + 64, ... on the device and runs all genomes in order in the host
// emulation (tests/emu: one thread plays every lane), and the reductions (`wave_min_i32`, `wave_or_u32`, ...) combine the
// lanes' partial values on the device and are the identity on the host, where the one thread has accumulated everything.
// A variable that a `lanes_for` body assigns is therefore per lane on the device and shared on the host; it may only be
// read after a reduction or a broadcast has made it uniform.
#pragma once
#include "kernels.h"

namespace pm {

// ------------------------------------------------------------------------------------------ lanes
#if defined(__HIP_DEVICE_COMPILE__)
template <class F> __device__ inline void lanes_for(int from, int n, F f) {
    for (int j = (from & ~63) + (int)__lane_id(); j < n; j += 64) if (j >= from) f(j);
}
__device__ inline int32_t wave_min_i32(int32_t x) { for (int d = 32; d >= 1; d >>= 1) { const int32_t y = __shfl_xor(x, d, 64); if (y < x) x = y; } return x; }
__device__ inline int32_t wave_max_i32(int32_t x) { for (int d = 32; d >= 1; d >>= 1) { const int32_t y = __shfl_xor(x, d, 64); if (y > x) x = y; } return x; }
__device__ inline uint32_t wave_or_u32(uint32_t x) { for (int d = 32; d >= 1; d >>= 1) x |= (uint32_t)__shfl_xor((int)x, d, 64); return x; }
__device__ inline int32_t wave_bcast_i32(int32_t x, int lane) { return __shfl(x, lane, 64); }
__device__ inline uint64_t wave_or_u64(uint64_t x) { return ((uint64_t)wave_or_u32((uint32_t)(x >> 32)) << 32) | wave_or_u32((uint32_t)x); }
__device__ inline uint64_t wave_sum_u64(uint64_t x) {
    for (int d = 32; d >= 1; d >>= 1) x += ((uint64_t)(uint32_t)__shfl_xor((int)(x >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64);
    return x;
}
__device__ inline bool wave_leader() { return __lane_id() == 0; }
__device__ inline void wave_sync() { __syncthreads(); }      // (one wavefront per workgroup: orders its LDS traffic)
#define PM_WAVE_SHARED __shared__
// loads that see what atomics of this or another wavefront have written (agent scope: served by the L2)
__device__ inline uint64_t load_coherent64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int32_t load_coherent32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint8_t load_coherent8(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void store_coherent32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void store_coherent8(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t atomic_fetch_or64(uint64_t* p, uint64_t v) { return (uint64_t)atomicOr((unsigned long long*)p, (unsigned long long)v); }
__device__ inline void atomic_and64(uint64_t* p, uint64_t v) { atomicAnd((unsigned long long*)p, (unsigned long long)v); }
#else
template <class F> inline void lanes_for(int from, int n, F f) { for (int j = from; j < n; j++) f(j); }
inline int32_t wave_min_i32(int32_t x) { return x; }
inline int32_t wave_max_i32(int32_t x) { return x; }
inline uint32_t wave_or_u32(uint32_t x) { return x; }
inline int32_t wave_bcast_i32(int32_t x, int) { return x; }
inline uint64_t wave_or_u64(uint64_t x) { return x; }
inline uint64_t wave_sum_u64(uint64_t x) { return x; }
inline bool wave_leader() { return true; }
inline void wave_sync() {}
#define PM_WAVE_SHARED
inline uint64_t load_coherent64(const uint64_t* p) { return *p; }
inline int32_t load_coherent32(const int32_t* p) { return *p; }
inline uint8_t load_coherent8(const uint8_t* p) { return *p; }
inline void store_coherent32(int32_t* p, int32_t v) { *p = v; }
inline void store_coherent8(uint8_t* p, uint8_t v) { *p = v; }
inline uint64_t atomic_fetch_or64(uint64_t* p, uint64_t v) { const uint64_t o = *p; *p = o | v; return o; }
inline void atomic_and64(uint64_t* p, uint64_t v) { *p &= v; }
#endif

// LDS atomics of the wavefront kernels (plain updates in the emulation, where one thread plays every lane)
PM_HD void lds_min32(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
PM_HD int32_t lds_add32(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    const int32_t o = *p; *p = o + v; return o;
#endif
}
// ------------------------------------------------------------------------------------------ Master.EP from segments
// Master.EP of a chunk of 256 reference positions = the pointwise minimum over the query genomes of EP_g, and EP_g is a STEP
// function: the furthest end over the genome's events that start at or before k (`emax` of the last such event), which only
// rises with k.  MasterEP (kernels.h) lets every lane test every staged event of every genome against its four positions --
// 6 500 vector instructions per chunk at 200 genomes, the events of a chunk being ~2 per genome.  Here the lanes take the GENOMES:
// a genome's function is the list of its segments (value, end), and because the values of one genome never fall,
//     EP_g[k] = min { value of s : s a segment of g that ends after k },
// so the minimum over all genomes is the same expression over ALL segments: every segment drops its value at its last position
// (an LDS atomicMin on a 256-entry array, ~3 per genome) and a suffix minimum over the array gives every position the smallest
// value of the segments that end after it.  One body for the device and for the CPU suite's emulation (lanes_for / wave_sync).
struct MasterEPSeg {
    const RegionInfo* R; int64_t nregions;
    int32_t ngen; const uint64_t* key; const int64_t* lo; const int32_t* emax; int lbits; int32_t* epm;
    const int64_t* cbase; const int32_t* coarse;
    int32_t g_first, g_last;   // sharded run: the min over the other genomes arrives by all-reduce
    const uint8_t* grouped;    // [region] != 0: GroupedPairEvents has written the region's Master.EP; nullptr: none
    int64_t nchunks;           // the launch is xcd_grid(nchunks) wavefronts: neighbouring chunks (the same lines of events) on one L2
    PM_HD void wave(int64_t w0) const {
        const int64_t w = xcd_item(w0, nchunks);
        if (w >= nchunks) return;
        const int32_t nq = ngen - 1;
        const int64_t r = region_of_chunk(cbase, nregions, w);
        if (grouped && grouped[r]) return;
        const RegionInfo& ri = R[r];
        const int64_t b = w - (cbase[r] - r);
        const int32_t k0 = (int32_t)(b << kCoarseShift);
        const uint64_t lmask = (1ull << lbits) - 1;
        const int32_t* row = coarse + cbase[r] * nq + b * nq;
        const int ga = g_first - 1, gb = g_last - 1;
        PM_WAVE_SHARED int32_t A[kChunkPos];      // A[e]: the smallest value of a segment whose last position is e
        PM_WAVE_SHARED int32_t part[2][64];
        const int32_t top = ri.nR;
        lanes_for(0, kChunkPos, [&](int e) { A[e] = top; });
        wave_sync();
        lanes_for(ga, gb, [&](int g) {
            const int64_t first = lo[r * nq + g];
            const int64_t a = first + row[g], e = first + row[nq + g];
            int32_t v = a > first ? emax[a - 1] : 0;      // the genome's value where the chunk begins
            for (int64_t i = a; i < e; i++) {
                const int32_t l = (int32_t)((key[i] >> 1) & lmask) - k0;      // the segment before event i ends at l - 1
                if (l >= 1) lds_min32(&A[l - 1], v);
                v = emax[i];
            }
            lds_min32(&A[kChunkPos - 1], v);      // the last segment runs to the end of the chunk
        });
        wave_sync();
        // suffix minimum: inside every lane's four entries, then over the lanes' minima (six doubling steps through LDS)
        lanes_for(0, 64, [&](int t) {
            int32_t m = A[4 * t + 3];
            for (int u = 2; u >= 0; u--) { const int32_t x = A[4 * t + u]; if (x < m) m = x; A[4 * t + u] = m; }
            part[0][t] = m;
        });
        wave_sync();
        int cur = 0;
        for (int d = 1; d < 64; d <<= 1) {
            lanes_for(0, 64, [&](int t) {
                int32_t x = part[cur][t];
                if (t + d < 64) { const int32_t y = part[cur][t + d]; if (y < x) x = y; }
                part[cur ^ 1][t] = x;
            });
            wave_sync();
            cur ^= 1;
        }
        lanes_for(0, 64, [&](int t) {
            const int32_t above = t < 63 ? part[cur][t + 1] : 0x7fffffff;
            for (int u = 0; u < 4; u++) {
                const int32_t k = k0 + 4 * t + u;
                if (k >= ri.nR) break;
                const int32_t x = A[4 * t + u];
                epm[ri.posbase + k] = x < above ? x : above;
            }
        });
    }
};

// ------------------------------------------------------------------------------------------ the stores
// state of a store row (what Aligner::validate_parallel keeps per candidate)
constexpr uint8_t kStBuilt = 1;      // the reference constructs a TMum for it (no kRowBad, :1723)
constexpr uint8_t kStOk = 2;         // inside every genome (`ok`, no kRowOutside)
constexpr uint8_t kStFlagged = 4;    // (anchor list) overlaps an earlier row of the list: settled against the marks of the others
constexpr uint8_t kStTangled = 8;    // ... and overlaps another flagged row: settled in list order
constexpr uint8_t kStAccepted = 16;  // a MUM of the run
struct Store {
    const int32_t* start; const uint8_t* strand; const int32_t* lon; const uint32_t* flags;
    int32_t* shift; int32_t* len; uint8_t* state;
    int32_t ngen;
};
struct Layout {
    uint64_t* image; const int64_t* word_off; const int64_t* nbits;
    int coherent;      // != 0: reads see marks made during the running kernel (served by the L2); 0: the image does not change under the reader
};
PM_HD uint64_t img_ld(const Layout& L, const uint64_t* p) { return L.coherent ? load_coherent64(p) : *p; }

// consecutive marked bases at pos, pos + 1, ... of genome j, at most maxlen (pos >= 0; bases at or past nbits read as unmarked)
PM_HD int32_t img_run_up(const Layout& L, int j, int64_t pos, int32_t maxlen) {
    const uint64_t* w = L.image + L.word_off[j];
    const int64_t nb = L.nbits[j];
    int32_t n = 0;
    while (n < maxlen) {
        const int64_t p = pos + n;
        if (p >= nb) break;
        const uint64_t x = ~(img_ld(L, w + (p >> 6)) >> (p & 63));      // (the bits shifted in at the top read as unmarked)
        const int avail = 64 - (int)(p & 63);
        int c = x ? ctz64(x) : 64;
        if (c > avail) c = avail;
        n += c;
        if (c < avail) break;
    }
    return n < maxlen ? n : maxlen;
}
// consecutive marked bases at pos, pos - 1, ..., at most maxlen
PM_HD int32_t img_run_down(const Layout& L, int j, int64_t pos, int32_t maxlen) {
    const uint64_t* w = L.image + L.word_off[j];
    int32_t n = 0;
    while (n < maxlen) {
        const int64_t p = pos - n;
        if (p < 0) break;
        const uint64_t x = ~(img_ld(L, w + (p >> 6)) << (63 - (int)(p & 63)));
        const int avail = (int)(p & 63) + 1;
        int c = x ? clz64(x) : 64;
        if (c > avail) c = avail;
        n += c;
        if (c < avail) break;
    }
    return n < maxlen ? n : maxlen;
}
// smallest marked position >= from (the sentinel guarantees one for from <= genome length), nbits when there is none
PM_HD int64_t img_next_set(const Layout& L, int j, int64_t from) {
    const uint64_t* w = L.image + L.word_off[j];
    const int64_t nb = L.nbits[j];
    if (from < 0) from = 0;
    if (from >= nb) return nb;
    int64_t wi = from >> 6;
    uint64_t x = img_ld(L, w + wi) & (~0ull << (from & 63));
    const int64_t nw = (nb + 63) / 64;
    while (!x) { if (++wi >= nw) return nb; x = img_ld(L, w + wi); }
    return wi * 64 + ctz64(x);
}
// largest marked position <= from, or -1
PM_HD int64_t img_prev_set(const Layout& L, int j, int64_t from) {
    const uint64_t* w = L.image + L.word_off[j];
    const int64_t nb = L.nbits[j];
    if (from < 0) return -1;
    if (from >= nb) from = nb - 1;
    int64_t wi = from >> 6;
    const int hi = (int)(from & 63);
    uint64_t x = img_ld(L, w + wi) & (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1));
    while (!x) { if (wi == 0) return -1; x = img_ld(L, w + --wi); }
    return wi * 64 + 63 - clz64(x);
}
// any marked base in [a, b)?  (reads the words of the range only)
PM_HD bool img_any(const Layout& L, int j, int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > L.nbits[j]) b = L.nbits[j];
    const uint64_t* w = L.image + L.word_off[j];
    while (a < b) {
        const int lo = (int)(a & 63);
        const int64_t span = (64 - lo) < (b - a) ? (64 - lo) : (b - a);
        if (img_ld(L, w + (a >> 6)) & ((span == 64 ? ~0ull : ((1ull << span) - 1)) << lo)) return true;
        a += span;
    }
    return false;
}
PM_HD void img_set_range(const Layout& L, int j, int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > L.nbits[j]) b = L.nbits[j];
    uint64_t* w = L.image + L.word_off[j];
    while (a < b) {
        const int lo = (int)(a & 63);
        const int64_t span = (64 - lo) < (b - a) ? (64 - lo) : (b - a);
        atomic_or64(&w[a >> 6], (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo);
        a += span;
    }
}
PM_HD void img_clear_range(const Layout& L, int j, int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > L.nbits[j]) b = L.nbits[j];
    uint64_t* w = L.image + L.word_off[j];
    while (a < b) {
        const int lo = (int)(a & 63);
        const int64_t span = (64 - lo) < (b - a) ? (64 - lo) : (b - a);
        atomic_and64(&w[a >> 6], ~((span == 64 ? ~0ull : ((1ull << span) - 1)) << lo));
        a += span;
    }
}

// ------------------------------------------------------------------------------------------ settle
// The per-candidate block of setMums1 (src/parsnp.cpp:1781-1833) for store row c, by all lanes of the wavefront: the length
// tests, Aligner::trim against the layout (:1399-1477; TMum::trimleft / trimright, TMum.cpp:104-148: every trim shortens the
// MUM in ALL genomes) and the check that reverse-strand members spell the reverse complement of the reference member
// (:1791-1825).  Returns whether the reference keeps the MUM; *pdl / *plen: bases trimmed on the left / final length.
// Does NOT mark the layout.
//
// trim() walks the genomes in order: genome j gives up marked bases at its start (the start moves right everywhere), then at
// its end, and genome j + 1 sees the MUM as genome j left it.  Almost no genome trims anything, so instead of one genome
// after the other the lanes look at all remaining genomes under the current (shift, length), the FIRST genome that would
// trim does so -- the ones before it see nothing marked, exactly what the loop would have found -- and the rest are looked
// at again: one round per genome that trims.
PM_HD bool settle_row(const Store& S, const Layout& L, const Packed& P, int64_t c, bool trim, int32_t* pdl, int32_t* plen) {
    const int n = S.ngen;
    const uint32_t f = S.flags[c];
    int32_t dl = 0, len = S.lon[c];
    *pdl = 0; *plen = len;
    if ((f & (kRowBad | kRowOutside)) || len < 5) return false;
    const int32_t* st = S.start + c * n;
    int from = 0;
    while (trim && len > 0) {
        int32_t first = 0x7fffffff, tl = 0, tr = 0;
        lanes_for(from, n, [&](int j) {
            if (j >= first) return;
            const int64_t s = (int64_t)st[j] + dl;
            const int32_t l = img_run_up(L, j, s, len);
            const int32_t r = l < len ? img_run_down(L, j, s + len - 1, len - l) : 0;
            if ((l | r) != 0) { first = j; tl = l; tr = r; }
        });
        const int32_t F = wave_min_i32(first);
        if (F == 0x7fffffff) break;
        tl = wave_bcast_i32(tl, F & 63); tr = wave_bcast_i32(tr, F & 63);
        dl += tl; len -= tl + tr; from = F + 1;
    }
    *pdl = dl; *plen = len;
    if (len < 2 || n <= 1) return false;
    if (!S.strand[c * n]) return false;
    if (f & kRowReverse) {
        // genome j's bases [l1, l1 + len) reverse-complemented = the stored reverse strand of j from glen - l1 - len on
        uint32_t bad = 0;
        const int64_t r0 = P.goff[0] + (int64_t)st[0] + dl;
        lanes_for(0, n, [&](int j) {
            if (S.strand[c * n + j]) return;
            const int64_t l1 = (int64_t)st[j] + dl;
            if (lce_fwd(P, P.goff[2 * j + 1] + (P.glen[j] - l1 - len), r0, len) != len) bad = 1;
        });
        if (wave_or_u32(bad)) return false;
    }
    return true;
}

// One wavefront per row of the anchor table: state bits from the engine's verdict flags, and the rows that overlap nothing
// earlier (no kRowDirty) settled at once -- nothing they touch is marked before them, so there is nothing to trim
// (Aligner::validate_parallel's clean candidates).
struct SettleClean {
    Store S; Packed P;
    PM_HD void wave(int64_t c) const {
        const uint32_t f = S.flags[c];
        uint8_t st = 0;
        if (!(f & kRowBad)) st = (uint8_t)(kStBuilt | ((f & kRowOutside) ? 0 : kStOk) | ((f & kRowDirty) ? kStFlagged : 0));
        int32_t dl = 0, len = S.lon[c];
        if ((st & (kStBuilt | kStOk | kStFlagged)) == (kStBuilt | kStOk)) {
            Layout none{nullptr, nullptr, nullptr, 0};
            if (settle_row(S, none, P, c, false, &dl, &len)) st |= kStAccepted;
            dl = 0; len = S.lon[c];
        }
        if (wave_leader()) { S.state[c] = st; S.shift[c] = 0; S.len[c] = len; }
    }
};
// tid = (row - row0, genome): the ranges of the rows whose state satisfies (state & mask) == want, set in the image
struct StoreMark {
    Store S; Layout L; int64_t row0; uint8_t mask, want;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = row0 + tid / S.ngen; const int j = (int)(tid % S.ngen);
        if ((S.state[c] & mask) != want) return;
        const int64_t a = (int64_t)S.start[c * S.ngen + j] + S.shift[c];
        img_set_range(L, j, a, a + S.len[c]);
    }
};
// The same marks for a list whose accepted clean rows lie in list order in every genome, one after the other without overlap
// (no such row carries kRowEarly: the caller checks) -- every population sample.  12 million ranges of ~80 bases cost ~28 million
// atomic ORs above (0.95 ms at 200 x 5 Mb); here a lane owns a genome and walks kMarkRows consecutive rows, gathering the bits
// of the word it is in and writing each word ONCE when it moves on: plain stores, except the first and the last word of the
// walk, which the neighbouring walks may share (atomic OR).  One wavefront per (block of rows, 64 genomes); the row entries of
// a wavefront's lanes are adjacent (coalesced), eight rows in flight.  The sentinel bits are set after this kernel.
constexpr int kMarkRows = 256;
struct StoreMarkOrdered {
    Store S; Layout L; int64_t rows;
    PM_HD void wave(int64_t w) const {
        const int n = S.ngen;
        const int64_t groups = (n + 63) / 64, blk = w / groups;
        const int g0 = (int)(w % groups) * 64;
        const int64_t c0 = blk * kMarkRows, c1 = c0 + kMarkRows < rows ? c0 + kMarkRows : rows;
        lanes_for(g0, g0 + 64 < n ? g0 + 64 : n, [&](int j) {
            uint64_t* img = L.image + L.word_off[j];
            const int64_t nb = L.nbits[j];
            int64_t cur = -1; uint64_t bits = 0; bool first = true;
            auto flush = [&](bool last) {
                if (cur < 0) return;
                if (first || last) atomic_or64(&img[cur], bits); else img[cur] = bits;
                first = false;
            };
            for (int64_t cb = c0; cb < c1; cb += 8) {
                int32_t a[8], l[8]; uint8_t st[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int64_t c = cb + u < c1 ? cb + u : c1 - 1;
                    a[u] = S.start[c * n + j]; l[u] = S.len[c]; st[u] = S.state[c];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (cb + u >= c1 || (st[u] & (kStAccepted | kStFlagged)) != kStAccepted) continue;
                    int64_t x = a[u], e = x + l[u];
                    if (x < 0) x = 0;
                    if (e > nb) e = nb;
                    while (x < e) {
                        const int64_t wi = x >> 6;
                        if (wi != cur) { flush(false); cur = wi; bits = 0; }
                        const int lo = (int)(x & 63);
                        const int64_t span = (64 - lo) < (e - x) ? (64 - lo) : (e - x);
                        bits |= (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
                        x += span;
                    }
                }
            }
            flush(true);
        });
    }
};
// tid = (i, genome): the rows listed in rows[] taken out of the image (filterRandom1 :415-418, filterRandomClustersSimple1 :460-466)
struct StoreUnmark {
    Store S; Layout L; const int32_t* rows;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = rows[tid / S.ngen]; const int j = (int)(tid % S.ngen);
        const int64_t a = (int64_t)S.start[c * S.ngen + j] + S.shift[c];
        img_clear_range(L, j, a, a + S.len[c]);
    }
};

// Which flagged rows overlap ANOTHER flagged row in some genome ("tangled": their outcome depends on the list order)?  Two
// scratch images of the layout's shape: every flagged row ORs its ranges into `once`; bits that were there already are ORed
// into `twice` -- the overlap itself, which both rows of an overlapping pair cover -- and a second pass looks for `twice` bits
// under a row's own ranges.  tid = (flagged index, genome) / one wavefront per flagged row.
struct CollideMark {
    Store S; const int32_t* list; Layout once; uint64_t* twice;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = list[tid / S.ngen]; const int j = (int)(tid % S.ngen);
        int64_t a = S.start[c * S.ngen + j], b = a + S.lon[c];
        if (a < 0) a = 0;
        if (b > once.nbits[j]) b = once.nbits[j];
        const int64_t base = once.word_off[j];
        while (a < b) {
            const int lo = (int)(a & 63);
            const int64_t span = (64 - lo) < (b - a) ? (64 - lo) : (b - a);
            const uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
            const uint64_t old = atomic_fetch_or64(&once.image[base + (a >> 6)], mask);
            if (old & mask) atomic_or64(&twice[base + (a >> 6)], old & mask);
            a += span;
        }
    }
};
struct CollideTest {
    Store S; const int32_t* list; Layout twice;
    PM_HD void wave(int64_t i) const {
        const int64_t c = list[i];
        const int n = S.ngen;
        uint32_t hit = 0;
        lanes_for(0, n, [&](int j) {
            const int64_t a = S.start[c * n + j];
            if (img_any(twice, j, a, a + S.lon[c])) hit = 1;
        });
        if (wave_or_u32(hit) && wave_leader()) S.state[c] |= kStTangled;
    }
};
// tid = flagged index: how many of the flagged rows are tangled (store_settle's second look at a list with many flagged rows)
struct CountTangled {
    Store S; const int32_t* list; uint64_t* count;
    PM_HD void operator()(int64_t i) const { if (S.state[list[i]] & kStTangled) atomic_add64(count, 1); }
};
// tid = (flagged index, genome): the two scratch images back to all zero -- the words under the flagged rows' ranges are the
// only ones that were written (clearing 2 x 126 MB per step for 850 rows' worth of bits costs 0.3 ms)
struct CollideClear {
    Store S; const int32_t* list; Layout once; uint64_t* twice;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = list[tid / S.ngen]; const int j = (int)(tid % S.ngen);
        int64_t a = S.start[c * S.ngen + j], b = a + S.lon[c];
        if (a < 0) a = 0;
        if (b > once.nbits[j]) b = once.nbits[j];
        const int64_t base = once.word_off[j];
        for (int64_t w = a >> 6; a < b && w <= ((b - 1) >> 6); w++) { once.image[base + w] = 0; twice[base + w] = 0; }
    }
};
// the flagged rows that meet no other flagged row: each against the marks of the clean rows, all at once (they commute)
struct SettleFlagged {
    Store S; Layout L; Packed P; const int32_t* list;
    PM_HD void wave(int64_t i) const {
        const int64_t c = list[i];
        if (S.state[c] & kStTangled) return;
        int32_t dl, len;
        const bool acc = settle_row(S, L, P, c, true, &dl, &len);
        if (acc) lanes_for(0, S.ngen, [&](int j) { const int64_t a = (int64_t)S.start[c * S.ngen + j] + dl; img_set_range(L, j, a, a + len); });
        if (wave_leader()) { S.shift[c] = dl; S.len[c] = len; if (acc) S.state[c] |= kStAccepted; }
    }
};
// ... and the tangled ones in LIST ORDER: each sees the marks of those before it (:1836-1839).  One wavefront working through them
// one after the other (SettleTangled) is 13 us per row -- nothing for the 18 tangled rows of a population sample, 40 ms for the
// 3 000 of a set with 10 % of every genome rearranged.  But list order only matters between rows that MEET: a tangled row may be
// settled as soon as every earlier tangled row that shares a base with it has been.  Rounds of three launches, one wavefront per
// flagged row, the lanes over the genomes:
//   TangleOwner   every tangled row not settled yet writes its list index into every 64-base WORD of its ranges (`owner`: one
//                 int32 per image word, kept all zero between calls; the smallest index wins, stored as 0x7fffffff - index);
//   TangleSettle  a row that finds its own index in all of its words has no unsettled earlier row near it: settled and marked now.
//                 The rows of one round share no word, so they neither read nor write one another's bits;
//   TangleClear   the words written in this round back to zero.
// A word is coarser than a base: two rows in one word that do not overlap are settled one round apart, never differently.  What
// kTangleRounds rounds leave (chains of rows overlapping one another) SettleTangled finishes in list order.
constexpr int kTangleRounds = 6;
struct TangleOwner {
    Store S; const int32_t* list; const int64_t* word_off; const int64_t* nbits; int32_t* owner; const uint8_t* t_done; const uint64_t* remaining;
    PM_HD void wave(int64_t i) const {
        if (*remaining == 0) return;
        const int64_t c = list[i];
        if (!(S.state[c] & kStTangled) || t_done[i]) return;
        const int n = S.ngen;
        const int32_t mine = 0x7fffffff - (int32_t)i;
        lanes_for(0, n, [&](int j) {
            int64_t a = S.start[c * n + j], b = a + S.lon[c];
            if (a < 0) a = 0;
            if (b > nbits[j]) b = nbits[j];
            if (a >= b) return;
            for (int64_t w = a >> 6; w <= ((b - 1) >> 6); w++) atomic_max32(&owner[word_off[j] + w], mine);
        });
    }
};
struct TangleSettle {
    Store S; Layout L; Packed P; const int32_t* list; const int32_t* owner; uint8_t* t_done; uint64_t* remaining;
    PM_HD void wave(int64_t i) const {
        if (load_coherent64(remaining) == 0) return;
        const int64_t c = list[i];
        if (!(S.state[c] & kStTangled) || t_done[i]) return;
        const int n = S.ngen;
        const int32_t mine = 0x7fffffff - (int32_t)i;
        uint32_t wait = 0;
        lanes_for(0, n, [&](int j) {
            int64_t a = S.start[c * n + j], b = a + S.lon[c];
            if (a < 0) a = 0;
            if (b > L.nbits[j]) b = L.nbits[j];
            if (a >= b) return;
            for (int64_t w = a >> 6; w <= ((b - 1) >> 6); w++) if (owner[L.word_off[j] + w] != mine) wait = 1;
        });
        if (wave_or_u32(wait)) return;
        int32_t dl, len;
        const bool acc = settle_row(S, L, P, c, true, &dl, &len);
        if (acc) lanes_for(0, n, [&](int j) { const int64_t a = (int64_t)S.start[c * n + j] + dl; img_set_range(L, j, a, a + len); });
        if (wave_leader()) { S.shift[c] = dl; S.len[c] = len; if (acc) S.state[c] |= kStAccepted; t_done[i] = 1; atomic_add64(remaining, ~0ull); }
    }
};
struct TangleClear {
    Store S; const int32_t* list; const int64_t* word_off; const int64_t* nbits; int32_t* owner; uint8_t* t_done;
    const uint64_t* remaining; uint64_t* left_after;      // [0] of the launch notes how many tangled rows the round left (the caller sizes the next step's rounds by it)
    PM_HD void wave(int64_t i) const {
        if (i == 0 && wave_leader()) *left_after = *remaining;
        const int64_t c = list[i];
        if (!(S.state[c] & kStTangled) || t_done[i] == 2) return;
        const int n = S.ngen;
        lanes_for(0, n, [&](int j) {
            int64_t a = S.start[c * n + j], b = a + S.lon[c];
            if (a < 0) a = 0;
            if (b > nbits[j]) b = nbits[j];
            if (a >= b) return;
            for (int64_t w = a >> 6; w <= ((b - 1) >> 6); w++) owner[word_off[j] + w] = 0;
        });
        if (wave_leader() && t_done[i] == 1) t_done[i] = 2;
    }
};
struct SettleTangled {
    Store S; Layout L; Packed P; const int32_t* list; int64_t count;
    const uint8_t* t_done; const uint64_t* remaining;      // != nullptr: the rounds before this launch have settled the rows with t_done[i] != 0, *remaining are left
    PM_HD void wave(int64_t) const {
        if (remaining && *remaining == 0) return;
        for (int64_t base = 0; base < count; base += 64) {
            // which of the next 64 flagged rows are tangled: one load per lane instead of a dependent load per row
            uint64_t mask = 0;
            lanes_for(0, 64, [&](int t) { if (base + t < count && (S.state[list[base + t]] & kStTangled) && !(t_done && t_done[base + t])) mask |= 1ull << t; });
            mask = wave_or_u64(mask);
            while (mask) {
                const int t = ctz64(mask);
                mask &= mask - 1;
                const int64_t c = list[base + t];
                int32_t dl, len;
                const bool acc = settle_row(S, L, P, c, true, &dl, &len);
                if (acc) lanes_for(0, S.ngen, [&](int j) { const int64_t a = (int64_t)S.start[c * S.ngen + j] + dl; img_set_range(L, j, a, a + len); });
                if (wave_leader()) { S.shift[c] = dl; S.len[c] = len; if (acc) S.state[c] |= kStAccepted; }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------ regions
// determineRegion (src/parsnp.cpp:1199-1290) for one side of one MUM, genome j: the walk over the layout to the previous /
// next marked base.  left: [p + 1, start - 1) with p the previous marked base (none: the region begins at 1, :1222-1226);
// right: from one base after the MUM's end to the base before the next marked one (or the sentinel at the genome's end).
// The request is (start, end - start) -- TRegion's length = end - start (LCR.cpp:29).
PM_HD void region_side(const Store& S, const Layout& L, const Packed& P, int64_t c, int32_t dl, int32_t len, int side, int j, int64_t* a, int64_t* b) {
    const int64_t s = (int64_t)S.start[c * S.ngen + j] + dl;
    if (side == 0) {
        int64_t p = img_prev_set(L, j, s - 1);
        if (p < 0) p = 0;
        *a = p + 1; *b = s - 1;
    } else {
        const int64_t nxt = s + len + 1, size = P.glen[j];
        const int64_t p = nxt >= size ? nxt : img_next_set(L, j, nxt);
        *a = nxt; *b = p - 1;
    }
}
// what the host learns about a region of the store
struct RegInfo {
    int64_t key;            // order in which the reference would push it (the caller sorts by it); -1: dropped (equal to a region still waiting)
    int64_t ref_start, ref_len;
    int32_t slength;        // shortest length over the genomes (TRegion::slength: chooses the minimum MUM length)
    int32_t parent;         // store row of the MUM it lies next to
};
// shortest region length over the genomes (and the reference column), by the whole wavefront
PM_HD int32_t region_extent(const Store& S, const Layout& L, const Packed& P, int64_t c, int32_t dl, int32_t len, int side, int64_t* ref_start, int64_t* ref_len) {
    int32_t smin = 0x7fffffff, r0a = 0, r0l = 0;
    lanes_for(0, S.ngen, [&](int j) {
        int64_t a, b;
        region_side(S, L, P, c, dl, len, side, j, &a, &b);
        const int32_t ln = (int32_t)(b - a);
        if (ln < smin) smin = ln;
        if (j == 0) { r0a = (int32_t)a; r0l = ln; }
    });
    smin = wave_min_i32(smin);
    *ref_start = wave_bcast_i32(r0a, 0); *ref_len = wave_bcast_i32(r0l, 0);
    return smin;
}
// One wavefront per accepted anchor (acc[i] = its row): both neighbour regions, kept when longer than q in every genome
// (setInitialClusters :2150-2172); kept regions are appended to the region store, key = 2 i + side = the reference's push order.
struct SeedWalk {
    Store S; Layout L; Packed P; const int32_t* acc; int32_t q;
    int64_t* rg_start; int64_t* rg_len; RegInfo* info; uint64_t* count; uint64_t cap;
    int64_t nacc;      // the launch is xcd_grid(nacc) wavefronts: neighbouring anchors walk the same words of the image
    PM_HD void wave(int64_t w) const {
        const int64_t i = xcd_item(w, nacc);
        if (i >= nacc) return;
        const int64_t c = acc[i];
        const int32_t dl = S.shift[c], len = S.len[c];
        const int n = S.ngen;
        for (int side = 0; side < 2; side++) {
            int64_t rs, rl;
            const int32_t smin = region_extent(S, L, P, c, dl, len, side, &rs, &rl);
            if (smin <= q) continue;
            int32_t slot = 0;
            if (wave_leader()) slot = (int32_t)atomic_add64(count, 1);
            slot = wave_bcast_i32(slot, 0);
            if ((uint64_t)slot >= cap) continue;           // (the caller sees count > cap and repeats with room)
            lanes_for(0, n, [&](int j) {
                int64_t a, b;
                region_side(S, L, P, c, dl, len, side, j, &a, &b);
                rg_start[(int64_t)slot * n + j] = a; rg_len[(int64_t)slot * n + j] = b - a;
            });
            if (wave_leader()) info[slot] = RegInfo{2 * i + side, rs, rl, smin, (int32_t)c};
        }
    }
};
// The same seed regions without a host round trip between the validation of the anchor list and the walks, and in the
// reference's PUSH ORDER without a sort: the accepted anchors are listed on the device (AnchorList, from an exclusive scan of
// their flags), SeedCount walks both sides of every accepted anchor and notes which of them are kept, an exclusive scan of the
// counts gives every anchor the slot of its first region, and SeedPlace walks the kept sides again (one side in fifteen at
// 200 x 5 Mb) and writes their rows there.  The launches cover the table's rows (the number of accepted anchors is only known
// to the device): wavefront w takes item xcd_item(w, rows) and leaves when that is past the accepted count.
// tid = row of the anchor table: acc[pos] = row, for the accepted ones
struct AnchorList {
    Store S; const int64_t* pos; int32_t* acc;
    PM_HD void operator()(int64_t c) const { if (S.state[c] & kStAccepted) acc[pos[c]] = (int32_t)c; }
};
struct SeedCount {
    Store S; Layout L; Packed P; const int32_t* acc; const int64_t* nacc; int32_t q; int64_t* cnt; uint8_t* keep; int64_t rows;
    PM_HD void wave(int64_t w) const {
        const int64_t i = xcd_item(w, rows);
        if (i >= rows) return;
        if (i >= *nacc) { if (wave_leader()) { cnt[i] = 0; keep[i] = 0; } return; }
        const int64_t c = acc[i];
        const int32_t dl = S.shift[c], len = S.len[c];
        uint8_t k = 0;
        for (int side = 0; side < 2; side++) {
            int64_t rs, rl;
            if (region_extent(S, L, P, c, dl, len, side, &rs, &rl) > q) k |= (uint8_t)(1 << side);
        }
        if (wave_leader()) { cnt[i] = (k & 1) + (k >> 1); keep[i] = k; }
    }
};
struct SeedPlace {
    Store S; Layout L; Packed P; const int32_t* acc; const int64_t* nacc; const int64_t* off; const uint8_t* keep;
    int64_t* rg_start; int64_t* rg_len; RegInfo* info; uint64_t cap; int64_t rows;
    PM_HD void wave(int64_t w) const {
        const int64_t i = xcd_item(w, rows);
        if (i >= rows || i >= *nacc || !keep[i]) return;
        const int64_t c = acc[i];
        const int32_t dl = S.shift[c], len = S.len[c];
        const int n = S.ngen;
        int64_t slot = off[i];
        for (int side = 0; side < 2; side++) {
            if (!(keep[i] & (1 << side))) continue;
            if ((uint64_t)slot >= cap) return;           // (the caller sees the total past the capacity and repeats with room)
            int64_t rs, rl;
            const int32_t smin = region_extent(S, L, P, c, dl, len, side, &rs, &rl);
            lanes_for(0, n, [&](int j) {
                int64_t a, b;
                region_side(S, L, P, c, dl, len, side, j, &a, &b);
                rg_start[slot * n + j] = a; rg_len[slot * n + j] = b - a;
            });
            if (wave_leader()) info[slot] = RegInfo{2 * i + side, rs, rl, smin, (int32_t)c};
            slot++;
        }
    }
};
// The algorithmic bytes of a search whose request rows the host never saw (bench.py's roofline; Aligner::run_batch sums the same
// over rows it holds): per (region, query genome) with piece length m and reference window n,
//   out[0] += 4 x SURVEY 8d's m/4 + 16 m + 16 n,   out[1] += 2 x (m/2 + 8 B per sampled K-mer -- one 64-B index request per leader, none
//   for pairs that fit 128 bases -- and n/2 once per region),   out[2] += 2 x m/2.
// One wavefront per kAlgPairs pairs, one atomic per wavefront and counter on one of kAlgSets sets of counters, a 64-byte line
// apart (the host adds the sets up): three hot addresses made this kernel 42 us of a step at 4 096 pairs per wavefront.
constexpr int kAlgPairs = 256;
constexpr int kAlgSets = 64;
struct AlgBytes {
    const RegionInfo* R; const int64_t* lens; int32_t ngen; int64_t npairs; uint64_t* out;
    PM_HD void wave(int64_t w) const {
        uint64_t a = 0, k = 0, q = 0;
        lanes_for(0, kAlgPairs, [&](int t) {
            const int64_t pair = w * kAlgPairs + t;
            if (pair >= npairs) return;
            const int64_t r = pair / (ngen - 1); const int g = (int)(pair % (ngen - 1)) + 1;
            const RegionInfo& ri = R[r];
            const int64_t m = lens[r * ngen + g], n = ri.nR;
            a += (uint64_t)(65 * m + 64 * n);
            k += (uint64_t)m + (g == 1 ? (uint64_t)n : 0ull);
            if (!(m <= 128 && n <= 128) && m >= ri.K && n >= ri.K) k += 16ull * (uint64_t)((m - ri.K) / ri.stride + 1);
            q += (uint64_t)m;
        });
        a = wave_sum_u64(a); k = wave_sum_u64(k); q = wave_sum_u64(q);
        uint64_t* set = out + (size_t)(w & (kAlgSets - 1)) * 8;
        if (wave_leader()) { atomic_add64(set, a); atomic_add64(set + 1, k); atomic_add64(set + 2, q); }
    }
};
// tid = (i, genome): request rows of the listed regions, in list order, where the search reads them
struct GatherRegions {
    const int32_t* ids; int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; int64_t* starts; int64_t* lens;
    PM_HD void operator()(int64_t tid) const {
        const int64_t r = ids[tid / ngen]; const int j = (int)(tid % ngen);
        starts[tid] = rg_start[r * ngen + j]; lens[tid] = rg_len[r * ngen + j];
    }
};
// one wavefront per pair of regions: equal in every genome (TRegion operator==, LCR.cpp:48-58)?
struct RegionsEqual {
    const int32_t* a; const int32_t* b; int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; uint8_t* same;
    PM_HD void wave(int64_t i) const {
        const int64_t x = a[i], y = b[i];
        uint32_t diff = 0;
        lanes_for(0, ngen, [&](int j) { if (rg_start[x * ngen + j] != rg_start[y * ngen + j] || rg_len[x * ngen + j] != rg_len[y * ngen + j]) diff = 1; });
        diff = wave_or_u32(diff);
        if (wave_leader()) same[i] = diff ? 0 : 1;
    }
};

// ------------------------------------------------------------------------------------------ events of small regions, once per distinct piece
// The recursion's regions are tiny (mean 46 bp at 200 x 5 Mb) and many: 8 000 regions x 200 genomes = 1.6 M (region, genome)
// pairs, each scanned diagonal by diagonal by SmallPairEvents (0.7 ms of vector arithmetic) and their 3.7 M events gathered and
// radix-sorted (0.33 ms).  But the genomes of a population carry a handful of DISTINCT strings between two anchors (2^k for k
// segregating sites), and everything downstream only needs the events of a pair contiguous and in reference order, not a
// globally sorted array.  So: one wavefront per region whose sides all fit 128 bases, lane t holding the genomes
// [1 + t * per, 1 + (t + 1) * per) --
//   1. round by round every lane loads one genome's piece (three bit planes, masked to its length) and looks it up among the
//      distinct pieces seen so far (at most kGrpPieces, kept in LDS); the lanes whose piece is new elect one of them at a time,
//      which joins the table, until every lane knows its piece's number;
//   2. the distinct pieces are scanned once per strand (pair_diagonals), their events kept sorted by reference position in LDS;
//   3. every lane copies the events of its genomes' pieces, keyed with their own pair, to one block of the grouped event
//      array (the head of the array the sorted events of the other pairs are appended to): a pair's events are contiguous and
//      ordered, and `glo[pair]` is where they start -- no gather, no sort.
// A region with a side longer than 128 bases, more than kGrpPieces distinct pieces, or more than kGrpEvents events of one
// (piece, strand) leaves its flag at 0 and writes nothing: SmallPairEvents, launched after this kernel, takes it.
#ifndef PM_GRP_EVENTS
#define PM_GRP_EVENTS 8      // (16 until round 6: 4 KB of LDS per wavefront less, 14 instead of 10 wavefronts per CU; no region of the three bench workloads holds more)
#endif
constexpr int kGrpEvents = PM_GRP_EVENTS;
constexpr int kGrpPieces = 32;
constexpr int kGrpGenomes = 1024;      // (the piece numbers of one region's genomes sit in LDS)
struct GroupedPairEvents {
    Packed P; const RegionInfo* R; const int64_t* starts; const int64_t* lens; int32_t ngen; const int32_t* rep;
    uint64_t* ev_key; uint64_t* ev_val; uint64_t* block_count; uint64_t ev_cap; int lbits; int64_t* glo; int32_t g_first, g_last;
    uint8_t* flag;      // [region] = 1: its events are in the grouped array (0: not a small region, or more pieces / events than fit)
    EventAtK* st; int32_t* emax;      // what WaveScan would leave for these events (kernels.h), written with them
    int32_t* epm;                     // Master.EP of the region's positions (MasterEP, kernels.h): min over the genomes of this rank
    int64_t nreg;                     // the launch is xcd_grid(nreg) wavefronts: neighbouring regions read the same lines of every genome
    PM_HD void wave(int64_t w) const {
        const int64_t r = xcd_item(w, nreg);
        if (r >= nreg) return;
        const int32_t nq = ngen - 1;
        const int per = (nq + 63) / 64;
        const RegionInfo& ri = R[r];
        const int32_t nR = ri.nR, L = ri.minlen;
        const int64_t rbase = P.goff[0] + ri.ref_pos;
        uint32_t big = nR > 128 ? 1u : 0u;      // every pair must fit 128 bases on both sides (small_pair)
        lanes_for(1, ngen, [&](int g) { if (lens[r * ngen + g] > 128) big = 1; });
        big = wave_or_u32(big);
        if (big) { if (wave_leader()) flag[r] = 0; return; }
        PM_WAVE_SHARED uint64_t sh_pl[64][6];         // the lanes' pieces of the running round
        PM_WAVE_SHARED int32_t sh_m[64];              // ... their lengths (-1: no genome)
        PM_WAVE_SHARED uint8_t sh_open[64];           // ... not numbered yet
        PM_WAVE_SHARED uint64_t rp_pl[kGrpPieces][6]; // the distinct pieces
        PM_WAVE_SHARED int32_t rp_m[kGrpPieces];
        PM_WAVE_SHARED int32_t rp_g[kGrpPieces];      // a genome that holds the piece
        PM_WAVE_SHARED uint8_t sh_piece[kGrpGenomes]; // genome -> number of its piece
        PM_WAVE_SHARED uint32_t sh_ev[2 * kGrpPieces][kGrpEvents];
        PM_WAVE_SHARED int32_t sh_cnt[2 * kGrpPieces];
        PM_WAVE_SHARED uint32_t sh_st[2 * kGrpPieces][kGrpEvents];      // the strand's state after each of its events: e1 | up << 8 | (spb + 128) << 16
        PM_WAVE_SHARED uint8_t sh_used[kGrpPieces];                      // a genome of this rank holds the piece
        PM_WAVE_SHARED int32_t sh_tot[64];
        PM_WAVE_SHARED int32_t sh_elect;
        PM_WAVE_SHARED int32_t sh_bad;
        PM_WAVE_SHARED uint64_t sh_base;
        if (wave_leader()) sh_bad = 0;
        lanes_for(0, kGrpPieces, [&](int i) { sh_used[i] = 0; });
        int npieces = 0;      // (the same in every lane)
        // 1. number the pieces
        for (int k = 0; k < per; k++) {
            wave_sync();
            lanes_for(0, 64, [&](int t) {
                const int g = 1 + t * per + k;
                int32_t m = -1;
                if (g < ngen) {
                    m = (int32_t)lens[r * ngen + g];
                    const int64_t qpos = P.goff[2 * g] + starts[r * ngen + g];
                    const W128 v = W128::ones(m);
                    for (int pl = 0; pl < 3; pl++) { const W128 x = W128::load(P.blk, qpos, pl) & v; sh_pl[t][2 * pl] = x.lo; sh_pl[t][2 * pl + 1] = x.hi; }
                }
                sh_m[t] = m;
                uint8_t open = m >= 0;
                for (int i = 0; i < npieces && open; i++) {
                    if (rp_m[i] != m) continue;
                    bool same = true;
                    for (int x = 0; x < 6 && same; x++) same = rp_pl[i][x] == sh_pl[t][x];
                    if (same) { sh_piece[g - 1] = (uint8_t)i; open = 0; }
                }
                sh_open[t] = open;
            });
            for (;;) {      // the lanes with a piece that is not in the table: the lowest of them adds its own
                wave_sync();
                if (wave_leader()) sh_elect = 64;
                wave_sync();
                lanes_for(0, 64, [&](int t) { if (sh_open[t]) lds_min32(&sh_elect, t); });
                wave_sync();
                const int e = sh_elect;
                if (e == 64) break;
                if (npieces == kGrpPieces) { if (wave_leader()) sh_bad = 1; break; }
                lanes_for(0, 64, [&](int t) {
                    if (t != e) return;
                    for (int x = 0; x < 6; x++) rp_pl[npieces][x] = sh_pl[t][x];
                    rp_m[npieces] = sh_m[t]; rp_g[npieces] = 1 + t * per + k;
                });
                wave_sync();
                lanes_for(0, 64, [&](int t) {
                    if (!sh_open[t] || rp_m[npieces] != sh_m[t]) return;
                    bool same = true;
                    for (int x = 0; x < 6 && same; x++) same = rp_pl[npieces][x] == sh_pl[t][x];
                    if (same) { sh_piece[t * per + k] = (uint8_t)npieces; sh_open[t] = 0; }
                });
                npieces++;
            }
            wave_sync();
            if (sh_bad) break;
        }
        // 2. the events of every (piece, strand): the 64 lanes split the (piece, strand) tasks and, within a task, its diagonals
        const int tasks = 2 * npieces;
        int share = 1;
        while (share * 2 * tasks <= 64) share *= 2;      // lanes per task
        if (!sh_bad) {
            lanes_for(0, tasks, [&](int wi) { sh_cnt[wi] = 0; });
            wave_sync();
            lanes_for(0, tasks * share, [&](int t) {
                const int wi = t / share, sub = t % share;
                const int g = rp_g[wi >> 1], strand = wi & 1;
                const int32_t m = rp_m[wi >> 1];
                if (m < L || nR < L) return;
                const int64_t qs = starts[r * ngen + g];
                const int64_t qbase = strand ? P.goff[2 * g + 1] + (P.glen[g] - qs - m) : P.goff[2 * g] + qs;
                auto emit = [&](int32_t l0, int32_t j0, int32_t len) {
                    if (len <= rep[ri.posbase + l0]) return;             // not unique in R
                    const int32_t at = lds_add32(&sh_cnt[wi], 1);
                    if (at < kGrpEvents) sh_ev[wi][at] = (uint32_t)l0 | ((uint32_t)j0 << 8) | ((uint32_t)len << 16);
                };
                if (nR <= 64 && m <= 64) pair_diagonals<W64>(P, rbase, qbase, nR, m, L, emit, sub, share);
                else pair_diagonals<W128>(P, rbase, qbase, nR, m, L, emit, sub, share);
            });
            wave_sync();
            lanes_for(0, tasks, [&](int wi) {      // in order of the reference position (a few entries: by insertion)
                const int n = sh_cnt[wi];
                if (n > kGrpEvents) { sh_bad = 1; return; }
                for (int i = 1; i < n; i++) {
                    const uint32_t e = sh_ev[wi][i];
                    int at = i;
                    while (at > 0 && (sh_ev[wi][at - 1] & 0xffu) > (e & 0xffu)) { sh_ev[wi][at] = sh_ev[wi][at - 1]; at--; }
                    sh_ev[wi][at] = e;
                }
            });
        }
        wave_sync();
        if (sh_bad) { if (wave_leader()) flag[r] = 0; return; }      // SmallPairEvents takes the region
        // 2b. the running state of every (piece, strand) after each of its events, as win_join keeps it (kernels.h: furthest
        // end e1 and its event w -- first in (l, j) order on equal ends -- and the second furthest end e2), resolved as WaveScan
        // leaves it: EP = e1, UP = max(l_w + rep'[l_w], e2), SP - k = j_w - l_w.  All of them at most 128: one word per event.
        lanes_for(0, tasks, [&](int wi) {
            const int n = sh_cnt[wi];
            int32_t e1 = 0, e2 = 0, wl = 0, wj = 0, up = 0; bool have = false;
            for (int i = 0; i < n; i++) {
                const uint32_t e = sh_ev[wi][i];
                const int32_t l = (int32_t)(e & 0xffu), j = (int32_t)((e >> 8) & 0xffu), end = l + (int32_t)(e >> 16);
                const bool take = !have || end > e1 || (end == e1 && (l < wl || (l == wl && j < wj)));
                if (take) { if (have && e1 > e2) e2 = e1; e1 = end; wl = l; wj = j; up = l + rep[ri.posbase + l]; have = true; }
                else if (end > e2) e2 = end;
                sh_st[wi][i] = (uint32_t)e1 | ((uint32_t)(e2 > up ? e2 : up) << 8) | ((uint32_t)(wj - wl + 128) << 16);
            }
        });
        // 3. the block: every lane's share, then the events of its genomes
        lanes_for(0, 64, [&](int t) {
            int c = 0;
            for (int k = 0; k < per; k++) {
                const int g = 1 + t * per + k;
                if (g < ngen && g >= g_first && g < g_last) { const int i = sh_piece[g - 1]; c += sh_cnt[2 * i] + sh_cnt[2 * i + 1]; sh_used[i] = 1; }
            }
            sh_tot[t] = c;
        });
        wave_sync();
        // Master.EP of the region (MasterEP's fold): a genome's EP at k = the furthest end over its events that start at or before
        // k, 0 without one; the minimum over the genomes of this rank = over the pieces they hold
        lanes_for(0, nR, [&](int k) {
            int32_t ep = nR;
            for (int i = 0; i < npieces; i++) {
                if (!sh_used[i]) continue;
                int32_t v = 0;
                for (int sd = 0; sd < 2; sd++) {
                    const int wi = 2 * i + sd, n = sh_cnt[wi];
                    int32_t e = 0;
                    for (int x = 0; x < n && (int32_t)(sh_ev[wi][x] & 0xffu) <= k; x++) e = (int32_t)(sh_st[wi][x] & 0xffu);
                    if (e > v) v = e;
                }
                if (v < ep) ep = v;
            }
            epm[ri.posbase + k] = ep;
        });
        if (wave_leader()) {
            int total = 0;
            for (int t = 0; t < 64; t++) { const int c = sh_tot[t]; sh_tot[t] = total; total += c; }
            sh_base = total ? atomic_add64(block_count, (uint64_t)total) : 0;
            flag[r] = 1;
        }
        wave_sync();
        lanes_for(0, 64, [&](int t) {
            uint64_t at0 = sh_base + (uint64_t)sh_tot[t];
            for (int k = 0; k < per; k++) {
                const int g = 1 + t * per + k;
                if (g >= ngen) break;
                const int64_t pair = r * nq + (g - 1);
                glo[pair] = (int64_t)at0;
                if (g < g_first || g >= g_last) continue;
                const int i = sh_piece[g - 1];
                const uint32_t* a = sh_ev[2 * i]; const uint32_t* b = sh_ev[2 * i + 1];
                const int na = sh_cnt[2 * i], nb = sh_cnt[2 * i + 1], c = na + nb;
                if (at0 + (uint64_t)c <= ev_cap) {
                    int ia = 0, ib = 0;
                    for (int x = 0; x < c; x++) {      // merged by (reference position, strand): the order of the sort key
                        const bool fwd = ib >= nb || (ia < na && (a[ia] & 0xffu) <= (b[ib] & 0xffu));
                        const uint32_t e = fwd ? a[ia++] : b[ib++];
                        ev_key[at0 + x] = ((((uint64_t)pair << lbits) | (uint64_t)(e & 0xffu)) << 1) | (fwd ? 0ull : 1ull);
                        ev_val[at0 + x] = ((uint64_t)((e >> 8) & 0xffu) << 32) | (uint64_t)(e >> 16);
                        const uint32_t sf = ia ? sh_st[2 * i][ia - 1] : 0, sr = ib ? sh_st[2 * i + 1][ib - 1] : 0;      // (e1 = 0: no event yet)
                        EventAtK o;
                        o.s[0] = sf ? StrandAtK{(int32_t)(sf & 0xffu), (int32_t)((sf >> 8) & 0xffu), (int32_t)(sf >> 16) - 128} : StrandAtK{0, 0, 0};
                        o.s[1] = sr ? StrandAtK{(int32_t)(sr & 0xffu), (int32_t)((sr >> 8) & 0xffu), (int32_t)(sr >> 16) - 128} : StrandAtK{0, 0, 0};
                        st[at0 + x] = o;
                        emax[at0 + x] = o.s[0].e1 > o.s[1].e1 ? o.s[0].e1 : o.s[1].e1;
                    }
                }
                at0 += (uint64_t)c;
            }
        });
    }
};
// tid = pair: the events of a flagged region's pair start at glo[pair] of the grouped array
struct GroupedBounds {
    const uint8_t* flag; const int64_t* glo; int32_t nq; int64_t* lo;
    PM_HD void operator()(int64_t pair) const { if (flag[pair / nq]) lo[pair] = glo[pair]; }
};

// ------------------------------------------------------------------------------------------ one generation of the recursion
// doWork (src/parsnp.cpp:173-317) pops the region with the smallest reference start, validates its candidates in order
// (setMums1, second half), pushes the neighbour regions of every new MUM that are longer than q (:215-254) and sorts.  The
// caller has found the waiting regions to fall into clusters that are disjoint in EVERY genome (Aligner::extend_generations):
// nothing found in one cluster can touch, trim or bound anything of another, so the clusters run side by side -- one
// wavefront each, its regions in the reference's order -- and the children form the next generation.
// What would make the order observable is reported, not decided here (`trouble`): a reverse-strand member that passes the
// sequence check outside its region (bit 1), a region with more candidates than the key holds (bit 2).  A child that sorts before
// a region still waiting in its cluster is no trouble since round 6: the cluster stops there (`done`), and what waits goes back to
// the caller's work list together with the children -- the next generation sorts them as the reference's list would (:291-292).
// Are the clusters of a generation pairwise disjoint in EVERY genome (Aligner::disjoint_clusters)?  Collinear genomes hold them
// in reference order: then it is "every cluster starts after its predecessor ends, with a base between" in every genome
// (ClustersDisjoint: one wavefront per cluster from the second on).  A genome in which that fails for some pair -- an inverted
// block holds its clusters in the opposite order -- raises *unsure: nothing of the generation is validated (ClusterValidate looks
// at the word), and the engine puts the exact question, cluster by cluster (ClusterExtents and the kernels after it).
struct ClustersDisjoint {
    int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first; uint64_t* unsure;
    int64_t stage_first;      // the first cluster of the call's second stage (0: one stage): it runs after its predecessor, not beside it
    int force;                // (tests) raise *unsure whatever the clusters look like
    PM_HD void wave(int64_t w) const {
        const int64_t cl = w + 1;
        if (force) { if (wave_leader()) atomic_or64(unsure, 1ull); return; }
        if (cl == stage_first) return;
        const int64_t p0 = cluster_first[cl - 1], p1 = cluster_first[cl], c1 = cluster_first[cl + 1];
        uint32_t bad = 0;
        lanes_for(1, ngen, [&](int j) {
            int64_t hi = -1, lo = (int64_t)1 << 62;
            for (int64_t x = p0; x < p1; x++) { const int64_t r = now_region[x]; const int64_t e = rg_start[r * ngen + j] + rg_len[r * ngen + j]; if (e > hi) hi = e; }
            for (int64_t x = p1; x < c1; x++) { const int64_t r = now_region[x]; const int64_t a = rg_start[r * ngen + j]; if (a < lo) lo = a; }
            if (lo <= hi + 1) bad = 1;      // touching counts too
        });
        if (wave_or_u32(bad) && wave_leader()) atomic_or64(unsure, 1ull);
    }
};
// Where some genome holds the clusters of a generation in another order, the exact question is put, and since round 6 its answer
// is not "all or nothing" but a verdict PER CLUSTER.  The reference (doWork :173-317) always pops the waiting region with the
// smallest reference start, and the children of a region lie inside it in every genome, so a cluster X that comes before a cluster
// Y on the reference is finished -- with everything it leads to -- before Y begins.  If X and Y meet in some genome (the two
// breakpoint regions of an inversion are the same gap of the inverted genome) Y must see X's marks: Y is DEFERRED, it stays on
// the work list and runs in a later generation, when no cluster before it meets it any more.  A cluster that meets nothing
// earlier runs now; the clusters that run together are pairwise disjoint, which is all ClusterValidate needs.
//   ClusterExtents   one wavefront per cluster: what it touches in every genome (cluster_pieces: the union of its regions, each with two
//                    bases of margin) ORed into the scratch image `once`, bits that were there already into `twice`;
//   ClusterInvolved  a `twice` bit under the own extent = the cluster meets another one; such a cluster writes its number into
//                    every 64-base WORD of its extents (`owner`: one int32 per image word, the smallest number wins);
//   ClusterDefer     an involved cluster that finds a smaller number than its own in a word of its extents is deferred.  (A word
//                    is coarser than a bit: two involved clusters that share a word but no base defer the later one without
//                    need -- a generation more, never a different result);
//   ClusterExtents with mark = 0 wipes the words it wrote (the scratch arrays are kept all zero between calls).
// owner[] holds 0x7fffffff - cluster so that zero means "nobody" and the smallest number is the largest value.
// What a cluster touches in genome j: the union of its regions' intervals [start, start + len + 2) (the + 2: clusters that touch, or have
// no base between, meet as well).  Not the hull: where a genome is rearranged the regions of one cluster can lie megabases apart --
// the seed region before an inverted block and the one behind it are neighbours on the reference -- and a hull would cover, and
// collide with, everything between them.  The lane sorts the cluster's intervals in its genome and hands every maximal run of
// overlapping ones to f(a, b) once, so that a cluster never meets ITSELF; a cluster of more than kPieces regions (dense repeats)
// is taken by its hull.
constexpr int kPieces = 8;
template <class F> PM_HD void cluster_pieces(int32_t ngen, const int64_t* rg_start, const int64_t* rg_len, const int32_t* now_region, int64_t p0, int64_t p1, int j, int64_t nbits, F f) {
    const int cnt = (int)(p1 - p0);
    auto clip = [&](int64_t a, int64_t b) { if (a < 0) a = 0; if (b > nbits) b = nbits; if (a < b) f(a, b); };
    if (cnt > kPieces) {
        int64_t hi = -1, lo = (int64_t)1 << 62;
        for (int64_t x = p0; x < p1; x++) {
            const int64_t r = now_region[x]; const int64_t s = rg_start[r * ngen + j], e = s + rg_len[r * ngen + j];
            if (s < lo) lo = s;
            if (e > hi) hi = e;
        }
        clip(lo, hi + 2);
        return;
    }
    int64_t a[kPieces], b[kPieces];
    for (int i = 0; i < cnt; i++) {      // insertion sort by start
        const int64_t r = now_region[p0 + i]; const int64_t s = rg_start[r * ngen + j], e = s + rg_len[r * ngen + j] + 2;
        int k = i;
        while (k > 0 && a[k - 1] > s) { a[k] = a[k - 1]; b[k] = b[k - 1]; k--; }
        a[k] = s; b[k] = e;
    }
    int64_t ca = a[0], cb = b[0];
    for (int i = 1; i < cnt; i++) {
        if (a[i] <= cb) { if (b[i] > cb) cb = b[i]; }
        else { clip(ca, cb); ca = a[i]; cb = b[i]; }
    }
    clip(ca, cb);
}
struct ClusterExtents {
    int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first;
    Layout once; uint64_t* twice; int32_t* owner; int64_t cl0; int mark;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        lanes_for(1, ngen, [&](int j) {
            const int64_t base = once.word_off[j];
            cluster_pieces(ngen, rg_start, rg_len, now_region, p0, p1, j, once.nbits[j], [&](int64_t a, int64_t b) {
                while (a < b) {
                    const int f = (int)(a & 63);
                    const int64_t span = (64 - f) < (b - a) ? (64 - f) : (b - a);
                    const uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << f;
                    if (mark) {
                        const uint64_t old = atomic_fetch_or64(&once.image[base + (a >> 6)], mask);
                        if (old & mask) atomic_or64(&twice[base + (a >> 6)], old & mask);
                    } else { once.image[base + (a >> 6)] = 0; twice[base + (a >> 6)] = 0; owner[base + (a >> 6)] = 0; }
                    a += span;
                }
            });
        });
    }
};
struct ClusterInvolved {
    int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first;
    Layout twice; int32_t* owner; uint8_t* involved; int64_t cl0;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        uint32_t hit = 0;
        lanes_for(1, ngen, [&](int j) {
            cluster_pieces(ngen, rg_start, rg_len, now_region, p0, p1, j, twice.nbits[j], [&](int64_t a, int64_t b) { if (!hit && img_any(twice, j, a, b)) hit = 1; });
        });
        hit = wave_or_u32(hit);
        if (wave_leader()) involved[cl] = hit ? 1 : 0;
        if (!hit) return;
        const int32_t mine = 0x7fffffff - (int32_t)cl;
        lanes_for(1, ngen, [&](int j) {
            cluster_pieces(ngen, rg_start, rg_len, now_region, p0, p1, j, twice.nbits[j], [&](int64_t a, int64_t b) {
                for (int64_t wd = a >> 6; wd <= ((b - 1) >> 6); wd++) atomic_max32(&owner[twice.word_off[j] + wd], mine);
            });
        });
    }
};
struct ClusterDefer {
    int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first;
    const int64_t* word_off; const int64_t* nbits; const int32_t* owner; const uint8_t* involved; uint8_t* defer; int64_t cl0;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        if (!involved[cl]) { if (wave_leader()) defer[cl] = 0; return; }
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        const int32_t mine = 0x7fffffff - (int32_t)cl;
        uint32_t earlier = 0;
        lanes_for(1, ngen, [&](int j) {
            cluster_pieces(ngen, rg_start, rg_len, now_region, p0, p1, j, nbits[j], [&](int64_t a, int64_t b) {
                for (int64_t wd = a >> 6; wd <= ((b - 1) >> 6) && !earlier; wd++) if (owner[word_off[j] + wd] > mine) earlier = 1;
            });
        });
        earlier = wave_or_u32(earlier);
        if (wave_leader()) defer[cl] = earlier ? 1 : 0;
    }
};
// ... and what its CANDIDATES touch outside their regions.  A reverse-strand member is flipped against the whole genome
// (TMum.cpp:33-35) and usually lands somewhere else in it, where the candidate's trimming READS the marks.  Beside a cluster that
// may MARK there in the same generation that read is a race (what is marked depends on how far the other wavefront has come:
// 50 x 5 Mb rearranged left the route in one run of twelve).  Marks come from candidates, so:
//   ReaderMark   every cluster ORs the members outside their region of its reverse-strand candidates (one base of margin) into the
//                scratch image `once` and writes its number into their words (`owner`, the smallest wins): the READERS, few;
//   MarkerLook   every cluster looks under EVERY member of every candidate it can build (the places it may mark; reads only): bits
//                of a reader there = the two clusters meet.  A reader with a smaller number: this cluster waits.  Else it leaves its
//                own number in the word (`owner2`) ...
//   ReaderLook   ... where the reader finds it: a marker with a smaller number, the reader waits.
//   ReaderMark with mark = 0 wipes the words (the scratch arrays are kept all zero between calls).
// (Taking the REGIONS a member outside could fall into serialised a rearranged set -- a candidate of 500 genomes has hundreds of
// such members, 55 -> 73 ms per step --, and putting all candidates' members through the atomics of ClusterExtents cost 4.5 ms.)
template <class F> PM_HD void cluster_reader_pieces(const Store& S, const int64_t* now_row0, const int32_t* now_cnt, int32_t ngen, const int64_t* rg_start, const int64_t* rg_len,
                                                    const int32_t* now_region, int64_t p0, int64_t p1, int j, int64_t nbits, F f) {
    for (int64_t x = p0; x < p1; x++) {
        const int64_t r = now_region[x];
        const int64_t rs = rg_start[r * ngen + j], re = rs + rg_len[r * ngen + j];
        const int64_t row0 = now_row0[x];
        for (int64_t c = row0; c < row0 + now_cnt[x]; c++) {
            const uint32_t fl = S.flags[c];
            if (!(fl & kRowReverse) || (fl & (kRowBad | kRowOutside)) || S.strand[c * ngen + j]) continue;
            int64_t a = S.start[c * ngen + j], b = a + S.lon[c];
            if (!(a < rs - 1 || b > re + 1)) continue;
            a -= 1; b += 1;
            if (a < 0) a = 0;
            if (b > nbits) b = nbits;
            if (a < b) f(a, b);
        }
    }
}
struct ReaderMark {
    Store S; const int64_t* now_row0; const int32_t* now_cnt; int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first;
    Layout once; int32_t* owner; int32_t* owner2; int64_t cl0; int mark;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        const int32_t mine = 0x7fffffff - (int32_t)cl;
        lanes_for(1, ngen, [&](int j) {
            const int64_t base = once.word_off[j];
            cluster_reader_pieces(S, now_row0, now_cnt, ngen, rg_start, rg_len, now_region, p0, p1, j, once.nbits[j], [&](int64_t a, int64_t b) {
                while (a < b) {
                    const int f = (int)(a & 63);
                    const int64_t span = (64 - f) < (b - a) ? (64 - f) : (b - a);
                    const int64_t wd = base + (a >> 6);
                    if (mark) { atomic_or64(&once.image[wd], (span == 64 ? ~0ull : ((1ull << span) - 1)) << f); atomic_max32(&owner[wd], mine); }
                    else { once.image[wd] = 0; owner[wd] = 0; owner2[wd] = 0; }
                    a += span;
                }
            });
        });
    }
};
struct MarkerLook {
    Store S; const int64_t* now_row0; const int32_t* now_cnt; int32_t ngen; const int64_t* cluster_first;
    Layout once; const int32_t* owner; int32_t* owner2; uint8_t* defer; int64_t cl0;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        const int32_t mine = 0x7fffffff - (int32_t)cl;
        uint32_t earlier = 0;
        lanes_for(1, ngen, [&](int j) {
            const int64_t base = once.word_off[j], nb = once.nbits[j];
            for (int64_t x = p0; x < p1; x++) {
                const int64_t row0 = now_row0[x];
                for (int64_t c = row0; c < row0 + now_cnt[x]; c++) {
                    if (S.flags[c] & (kRowBad | kRowOutside)) continue;
                    int64_t a = (int64_t)S.start[c * ngen + j] - 1, b = a + S.lon[c] + 2;
                    if (a < 0) a = 0;
                    if (b > nb) b = nb;
                    while (a < b) {
                        const int f = (int)(a & 63);
                        const int64_t span = (64 - f) < (b - a) ? (64 - f) : (b - a);
                        const int64_t wd = base + (a >> 6);
                        if (once.image[wd] & ((span == 64 ? ~0ull : ((1ull << span) - 1)) << f)) {
                            const int32_t o = owner[wd];
                            if (o > mine) earlier = 1;
                            else if (o < mine) atomic_max32(&owner2[wd], mine);
                        }
                        a += span;
                    }
                }
            }
        });
        if (wave_or_u32(earlier) && wave_leader()) defer[cl] = 1;
    }
};
struct ReaderLook {
    Store S; const int64_t* now_row0; const int32_t* now_cnt; int32_t ngen; const int64_t* rg_start; const int64_t* rg_len; const int32_t* now_region; const int64_t* cluster_first;
    const int64_t* word_off; const int64_t* nbits; const int32_t* owner2; uint8_t* defer; int64_t cl0;
    PM_HD void wave(int64_t w) const {
        const int64_t cl = cl0 + w;
        const int64_t p0 = cluster_first[cl], p1 = cluster_first[cl + 1];
        const int32_t mine = 0x7fffffff - (int32_t)cl;
        uint32_t earlier = 0;
        lanes_for(1, ngen, [&](int j) {
            cluster_reader_pieces(S, now_row0, now_cnt, ngen, rg_start, rg_len, now_region, p0, p1, j, nbits[j], [&](int64_t a, int64_t b) {
                for (int64_t wd = a >> 6; wd <= ((b - 1) >> 6) && !earlier; wd++) if (owner2[word_off[j] + wd] > mine) earlier = 1;
            });
        });
        if (wave_or_u32(earlier) && wave_leader()) defer[cl] = 1;
    }
};
// A reverse-strand member is flipped against the WHOLE genome (TMum.cpp:33-35): inside a sub-region it usually lands far outside
// the region, and the validation of its candidate -- the trimming -- then READS layout bits in another cluster's territory.  What
// is marked there at that moment depends on how far the other clusters have come: in the reference's sequential order
// (doWork :173-317: the first pushed seed, then always the waiting region with the smallest reference start) it is the anchors
// plus the MUMs of every region processed BEFORE this one; in a generation scheme it is whatever the other wavefronts happened to
// have marked.  Almost always that decides nothing -- in a collinear population 3 % of the recursion's candidates carry such a
// member and fail the sequence check whatever they read -- but not always (fuzz campaign of round 5, seed 7059: the reference trims
// such a candidate to 2 bases and accepts it; a region 20 kb further on, which the reference processes later, had already marked
// its MUM here and the candidate was trimmed away).  So:
//   * ClusterValidate NOTES every such candidate (ForeignRead) with the marks it saw in EVERY genome's interval -- inside its
//     region those are the reference's, cluster by cluster (the clusters are disjoint and an accepted member outside its region
//     ends the route);
//   * every row of the recursion carries the ORDER KEY of its region, (reference start, generation), -1 for the first seed:
//     region Y is processed before region X exactly when key(Y) < key(X) -- a child starts at or after its parent, so everything
//     that leads to Y has a smaller start than X too, and of two regions with one start the older generation went first (two
//     of one generation: the route is left before this matters);
//   * when the recursion is over, the resolve (ForeignBound and the kernels around it) decides every noted candidate AGAIN the way the reference's order had it: in the
//     genomes of its outside members the marks are the anchors' plus those of the recursion's MUMs with a smaller key or of
//     earlier candidates of the same region (found by a scan over the recursion's accepted rows), elsewhere the noted ones;
//     Aligner::trim (:1399-1477) on these masks, then the sequence check.  A verdict, shift or length that differs from what the
//     generation scheme stored means the order shows: the route is left.
struct ForeignRead { int32_t row, region, row0, pad; int64_t key; };
// marks of [a, a + len) of genome j, len <= 64: bit t = base a + t (bases outside the genome read as unmarked, as the runs do)
PM_HD uint64_t img_bits64(const Layout& L, int j, int64_t a, int32_t len) {
    const uint64_t* w = L.image + L.word_off[j];
    const int64_t nb = L.nbits[j];
    uint64_t m = 0;
    int32_t t = a < 0 ? (int32_t)(-a) : 0;
    while (t < len) {
        const int64_t p = a + t;
        if (p >= nb) break;
        const int lo = (int)(p & 63);
        int span = 64 - lo;
        if (span > len - t) span = len - t;
        const uint64_t x = (img_ld(L, w + (p >> 6)) >> lo) & (span == 64 ? ~0ull : ((1ull << span) - 1));
        m |= x << t;
        t += span;
    }
    return m;
}
PM_HD int64_t order_key(int64_t ref_start, int32_t generation) { return generation <= 0 ? -1 : ref_start * 4096 + (generation < 4095 ? generation : 4095); }
// The resolve as launches (engine_core.h: order_check_launch).  Which of the marks in an outside member's interval belong to the
// recursion, and to which MUM, only a scan over the recursion's accepted rows can tell, and a candidate that is reverse in one
// query genome of a population is reverse in all of them: 500 noted candidates x 200 genomes x 17 000 rows at 200 x 5 Mb.  So
//   ClusterValidate  sets, for every MUM it accepts, one BYTE per word of the layout image that the MUM marks (`rec`, cleared
//                  with the image): a coarse picture of the recursion's marks alone, written with plain stores;
//   ForeignBound   per noted candidate, BOUNDS on what the reference's order saw in the genomes of its outside members: at
//                  least the final marks outside the image words the recursion touched (those are anchors' marks, in place
//                  before any region is processed), at most the final marks.  Trimming is monotone -- more marks, or a smaller
//                  interval to start from, never leave more -- so the interval the reference kept lies between the two trimmed
//                  with the bounds.  Both equal, or the larger one shorter than 2 bases (almost every candidate: its outside
//                  members lie among anchors and are trimmed away whatever else is marked): decided on the spot.  The rest goes
//                  onto a short hit list;
//   ForeignScan    for the hits, the recursion's accepted rows against the candidate's intervals, all genomes side by side
//                  (the rows stream through once per hit): the marks the recursion owns there, and which of them come from a
//                  region that the reference's order processes earlier;
//   ForeignDecideHits   the hits decided with exactly those.
constexpr int kScanRows = 16;         // rows of the store per work item of ForeignScan
// the image words of genome j (from word_off) that bases [a, b) touch get their byte
PM_HD void rec_set(uint8_t* rec, int64_t word_off, int64_t nbits, int64_t a, int64_t b) {
    if (a < 0) a = 0;
    if (b > nbits) b = nbits;
    if (a >= b) return;
    for (int64_t w = a >> 6; w <= ((b - 1) >> 6); w++) rec[word_off + w] = 1;
}
// the bases of [a, a + len), len <= 64, that lie in an image word the recursion has marked: bit t = base a + t
PM_HD uint64_t rec_bits64(const uint8_t* rec, int64_t word_off, int64_t nbits, int64_t a, int32_t len) {
    uint64_t m = 0;
    int32_t t = a < 0 ? (int32_t)(-a) : 0;
    while (t < len) {
        const int64_t p = a + t;
        if (p >= nbits) break;
        int32_t span = 64 - (int32_t)(p & 63);
        if (span > len - t) span = len - t;
        if (rec[word_off + (p >> 6)]) m |= (span == 64 ? ~0ull : ((1ull << span) - 1)) << t;
        t += span;
    }
    return m;
}
// Aligner::trim (:1399-1477) of a candidate of `lon` bases on masks: mask(j) = the marks of genome j's interval, bit t = base t
// of the candidate.  settle_row's rounds: the first genome that trims does so, the rest look again.
template <class MaskOf> PM_HD void trim_on_masks(int n, int32_t lon, MaskOf mask, int32_t* pdl, int32_t* plen) {
    int32_t dl = 0, len = lon;
    int from = 0;
    while (len > 0) {
        int32_t first = 0x7fffffff, tl = 0, tr = 0;
        lanes_for(from, n, [&](int j) {
            if (j >= first) return;
            const uint64_t x = (mask(j) >> dl) & (len == 64 ? ~0ull : ((1ull << len) - 1));
            if (!x) return;
            int32_t l = ~x ? ctz64(~x) : 64;
            if (l > len) l = len;
            int32_t r = 0;
            if (l < len) {
                const uint64_t y = ~(x << (64 - len));      // (the last base of the window at bit 63)
                r = y ? clz64(y) : 64;
                if (r > len - l) r = len - l;
            }
            if ((l | r) != 0) { first = j; tl = l; tr = r; }
        });
        const int32_t F = wave_min_i32(first);
        if (F == 0x7fffffff) break;
        tl = wave_bcast_i32(tl, F & 63); tr = wave_bcast_i32(tr, F & 63);
        dl += tl; len -= tl + tr; from = F + 1;
    }
    *pdl = dl; *plen = len;
}
// the rest of settle_row for the trimmed candidate, against what the generations stored: true = verdict, shift or length differ
PM_HD bool verdict_differs(const Store& S, const Packed& P, int64_t c, int32_t dl, int32_t len) {
    const int n = S.ngen;
    const int32_t* st = S.start + c * n;
    bool acc = len >= 2 && n > 1 && S.strand[c * n] != 0;
    if (acc) {
        uint32_t bad = 0;
        const int64_t r0 = P.goff[0] + (int64_t)st[0] + dl;
        lanes_for(0, n, [&](int j) {
            if (S.strand[c * n + j]) return;
            const int64_t l1 = (int64_t)st[j] + dl;
            if (lce_fwd(P, P.goff[2 * j + 1] + (P.glen[j] - l1 - len), r0, len) != len) bad = 1;
        });
        acc = wave_or_u32(bad) == 0;
    }
    const bool was = (S.state[c] & kStAccepted) != 0;
    return acc != was || (acc && (dl != S.shift[c] || len != S.len[c]));
}
PM_HD bool member_outside(const Store& S, int64_t c, int j, const int64_t* rs, const int64_t* rl) {
    if (S.strand[c * S.ngen + j]) return false;
    const int64_t a = S.start[c * S.ngen + j];
    return a < rs[j] - 1 || a + S.lon[c] > rs[j] + rl[j] + 1;
}
// one wavefront per noted candidate
struct ForeignBound {
    Store S; Layout L; Packed P;
    const int64_t* rg_start; const int64_t* rg_len;
    const ForeignRead* list; int64_t count; const uint64_t* masks; const uint8_t* rec;
    int32_t* hits; uint64_t* hit_count; uint64_t hit_cap; uint64_t* hit_owned; uint64_t* hit_present;
    uint32_t* trouble; uint32_t bit;
    PM_HD void wave(int64_t i) const {
        if (i >= count) return;
        const ForeignRead e = list[i];
        const int n = S.ngen;
        const int64_t c = e.row;
        const int32_t lon = S.lon[c];
        const int64_t* rs = rg_start + (int64_t)e.region * n; const int64_t* rl = rg_len + (int64_t)e.region * n;
        const uint64_t* M = masks + (uint64_t)i * (uint64_t)n;
        int32_t dl_lo, len_lo, dl_hi, len_hi;
        trim_on_masks(n, lon, [&](int j) -> uint64_t {      // at least: the final marks outside the words the recursion touched
            if (!member_outside(S, c, j, rs, rl)) return M[j];
            const int64_t a = S.start[c * n + j];
            return img_bits64(L, j, a, lon) & ~rec_bits64(rec, L.word_off[j], L.nbits[j], a, lon);
        }, &dl_lo, &len_lo);
        if (len_lo >= 2) {
            trim_on_masks(n, lon, [&](int j) -> uint64_t {      // at most: the final marks
                return member_outside(S, c, j, rs, rl) ? img_bits64(L, j, S.start[c * n + j], lon) : M[j];
            }, &dl_hi, &len_hi);
            if (dl_hi != dl_lo || len_hi != len_lo) {      // the order decides: onto the hit list
                int32_t at = 0;
                if (wave_leader()) at = (int32_t)atomic_add64(hit_count, 1);
                at = wave_bcast_i32(at, 0);
                if ((uint64_t)at >= hit_cap) { if (wave_leader()) atomic_or32(trouble, bit); return; }      // (more than the scan is launched for: the host route)
                if (wave_leader()) hits[at] = (int32_t)i;
                lanes_for(0, n, [&](int j) { hit_owned[(uint64_t)at * (uint64_t)n + j] = 0; hit_present[(uint64_t)at * (uint64_t)n + j] = 0; });
                return;
            }
        }
        if (verdict_differs(S, P, c, dl_lo, len_lo) && wave_leader()) atomic_or32(trouble, bit);
    }
};
// Work item x = (hit x / chunks, chunk x % chunks of kScanRows recursion rows), the items dealt round robin to a fixed number
// of wavefronts (the number of hits is only known on the device): every lane takes genomes, holds the candidate's interval there
// and runs over the chunk's rows -- which of them overlap it (owned), which of those the reference's order had in place (present).
struct ForeignScan {
    Store S; const int64_t* rg_start; const int64_t* rg_len; const int64_t* key; const ForeignRead* list;
    const int32_t* hits; const uint64_t* hit_count; uint64_t hit_cap; uint64_t* hit_owned; uint64_t* hit_present;
    int64_t rows_first, rows_end, chunks, waves;
    PM_HD void wave(int64_t w) const {
        const uint64_t nh = *hit_count < hit_cap ? *hit_count : hit_cap;
        const int n = S.ngen;
        for (int64_t x = w; x < (int64_t)nh * chunks; x += waves) {
            const int64_t slot = x / chunks, chunk = x % chunks;
            const ForeignRead e = list[hits[slot]];
            const int64_t c = e.row;
            const int32_t lon = S.lon[c];
            const int64_t* rs = rg_start + (int64_t)e.region * n; const int64_t* rl = rg_len + (int64_t)e.region * n;
            const int64_t r_lo = rows_first + chunk * kScanRows, r_hi = r_lo + kScanRows < rows_end ? r_lo + kScanRows : rows_end;
            lanes_for(0, n, [&](int j) {
                if (!member_outside(S, c, j, rs, rl)) return;
                const int64_t a = S.start[c * n + j];
                uint64_t owned = 0, present = 0;
                for (int64_t r = r_lo; r < r_hi; r++) {
                    if (!(S.state[r] & kStAccepted)) continue;
                    const int64_t ar = (int64_t)S.start[r * n + j] + S.shift[r];
                    const int64_t lo = ar > a ? ar : a, hi = ar + S.len[r] < a + lon ? ar + S.len[r] : a + lon;
                    if (lo >= hi) continue;
                    const uint64_t bits = ((hi - lo) == 64 ? ~0ull : ((1ull << (hi - lo)) - 1)) << (lo - a);
                    owned |= bits;
                    if (key[r] < e.key || (r >= e.row0 && r < c)) present |= bits;
                }
                if (owned) atomic_or64(&hit_owned[(uint64_t)slot * (uint64_t)n + j], owned);
                if (present) atomic_or64(&hit_present[(uint64_t)slot * (uint64_t)n + j], present);
            });
        }
    }
};
struct ForeignDecideHits {
    Store S; Layout L; Packed P;
    const int64_t* rg_start; const int64_t* rg_len;
    const ForeignRead* list; const uint64_t* masks; const int32_t* hits; const uint64_t* hit_count; uint64_t hit_cap;
    const uint64_t* hit_owned; const uint64_t* hit_present;
    uint32_t* trouble; uint32_t bit;
    PM_HD void wave(int64_t slot) const {
        const uint64_t nh = *hit_count < hit_cap ? *hit_count : hit_cap;
        if ((uint64_t)slot >= nh) return;
        const int32_t i = hits[slot];
        const ForeignRead e = list[i];
        const int n = S.ngen;
        const int64_t c = e.row;
        const int32_t lon = S.lon[c];
        const int64_t* rs = rg_start + (int64_t)e.region * n; const int64_t* rl = rg_len + (int64_t)e.region * n;
        const uint64_t* M = masks + (uint64_t)i * (uint64_t)n;
        int32_t dl, len;
        trim_on_masks(n, lon, [&](int j) -> uint64_t {      // the anchors' marks + the recursion's that were in place
            if (!member_outside(S, c, j, rs, rl)) return M[j];
            return (img_bits64(L, j, S.start[c * n + j], lon) & ~hit_owned[(uint64_t)slot * (uint64_t)n + j]) | hit_present[(uint64_t)slot * (uint64_t)n + j];
        }, &dl, &len);
        if (verdict_differs(S, P, c, dl, len) && wave_leader()) atomic_or32(trouble, bit);
    }
};
// An ACCEPTED reverse-strand member outside its region (fuzz and the rearranged sets: a few bases left by the trimming that pass the
// sequence check somewhere else in the genome) WRITES marks into another cluster's territory.  That is harmless exactly when nobody
// whom the reference's order serves differently can see the mark: the gap between two marks it falls into is dead ground -- no
// region of the store covers it -- or every region r that does cover it in that genome is on the right side of the order:
//   * r was walked out (region_extent) by a MUM that the reference processes before this one, or by one of the own cluster (which
//     one wavefront works through in the reference's order): its extent is what the reference's is;
//   * r has been processed, in an earlier generation and with a smaller order key (it, and whatever of its rows lay there, came
//     first in the reference as well), or in this cluster;
//   * or r still waits and sorts after this region (it will see the mark, as in the reference).
// A candidate of another region that READS there through an outside member of its own is the order check's business (ForeignRead).
// ClusterValidate lists the writes (OutsideWrite), OutsideWriteCheck runs over (write, 64 regions of the store) items behind the
// generation's validation and raises trouble bit 1 where none of that holds: the route is left, as it was for EVERY such write
// until round 6.  rg_pkey[r]: 0 = region r waits, else the order key it was processed with + 3.
struct OutsideWrite { int32_t row, region, x0, x1; int64_t key; };
constexpr int kOutsideChunk = 64;
struct OutsideWriteCheck {
    Store S; const int64_t* rg_start; const int64_t* rg_len; const RegInfo* info; const int64_t* rg_pkey; const uint64_t* rg_count; uint64_t rg_cap;
    const int32_t* now_region; const OutsideWrite* list; const uint64_t* count; uint64_t cap; const int64_t* row_key; int64_t anchor_rows;
    uint32_t* trouble; int64_t waves;
    PM_HD void wave(int64_t w) const {
        const int64_t nev = (int64_t)(*count < cap ? *count : cap);
        if (nev == 0) return;
        const int n = S.ngen;
        const int64_t nreg = (int64_t)(*rg_count < rg_cap ? *rg_count : rg_cap);
        const int64_t chunks = (nreg + kOutsideChunk - 1) / kOutsideChunk;
        for (int64_t x = w; x < nev * chunks; x += waves) {
            const OutsideWrite e = list[x / chunks];
            const int64_t r_lo = (x % chunks) * kOutsideChunk, r_hi = r_lo + kOutsideChunk < nreg ? r_lo + kOutsideChunk : nreg;
            const int64_t c = e.row;
            const int32_t dl = S.shift[c], len = S.len[c];
            const int64_t* Rs = rg_start + (int64_t)e.region * n; const int64_t* Rl = rg_len + (int64_t)e.region * n;
            const int64_t ref0 = Rs[0];
            // the order keys of the own cluster's regions: (reference start, this generation)
            auto own_cluster = [&](int64_t k) {
                if (k == e.key) return true;
                if (k < 0 || e.key < 0 || ((k ^ e.key) & 4095)) return false;
                for (int32_t y = e.x0; y < e.x1; y++) if (rg_start[(int64_t)now_region[y] * n] == (k >> 12)) return true;
                return false;
            };
            auto wrong_side = [&](int64_t k) {      // k: the key something was processed with -- not before this region, or beside it
                if (own_cluster(k)) return false;
                return k > e.key || (k >= 0 && e.key >= 0 && ((k ^ e.key) & 4095) == 0);
            };
            uint32_t bad = 0;
            lanes_for(0, n, [&](int j) {
                if (bad || S.strand[c * n + j]) return;
                const int64_t a = (int64_t)S.start[c * n + j] + dl;
                if (!(a < Rs[j] - 1 || a + len > Rs[j] + Rl[j] + 1)) return;
                const int64_t lo = a - 1, hi = a + len + 1;
                for (int64_t r = r_lo; r < r_hi && !bad; r++) {
                    if (r == e.region) continue;
                    const int64_t s = rg_start[r * n + j], t = s + rg_len[r * n + j];
                    if (s >= hi || t <= lo) continue;
                    const RegInfo ri = info[r];
                    if (ri.key < 0) continue;                                   // (dropped: equal to a region of its cluster)
                    const int64_t pk = ri.parent < anchor_rows ? -2 : row_key[ri.parent];      // (an anchor's walks precede every region)
                    if (pk >= -1 && wrong_side(pk)) { bad = 1; break; }
                    const int64_t mine = rg_pkey[r];
                    if (mine == 0) { if (ri.ref_start <= ref0) bad = 1; }
                    else if (wrong_side(mine - 3)) bad = 1;
                }
            });
            if (wave_or_u32(bad) && wave_leader()) atomic_or32(trouble, 2u);
        }
    }
};
struct ClusterValidate {
    Store S; Layout L; Packed P;
    int64_t* rg_start; int64_t* rg_len; RegInfo* info; uint64_t* rg_count; uint64_t rg_cap;
    const int32_t* now_region; const int64_t* now_row0; const int32_t* now_cnt; const int64_t* cluster_first;
    int32_t q; uint32_t* trouble;
    int64_t ncl;      // the launch is xcd_grid(ncl) wavefronts: neighbouring clusters read and mark neighbouring words of the image
    int64_t cl0;      // ... for the clusters [cl0, cl0 + ncl) of the list
    const uint64_t* gate;      // != nullptr: the second stage of a call -- it only runs if the first left *gate at 0 (StageGate)
    const uint64_t* unsure;    // != nullptr: nothing runs while the collinear test of the clusters has not passed (ClustersDisjoint)
    ForeignRead* foreign; uint64_t* foreign_count; uint64_t foreign_cap; uint64_t* foreign_masks;      // candidates with a member outside their region (ForeignBound)
    int64_t* row_key; int32_t generation;      // the order key of every row decided here (order_key)
    uint8_t* rec;                              // one byte per image word: the recursion has marked there
    const uint8_t* defer;                      // != nullptr: [cluster] != 0 -- it meets an earlier cluster in some genome and waits (ClusterDefer)
    int32_t* done;                             // [cluster]: how many of its regions were processed (the rest stay on the caller's work list)
    int64_t* rg_pkey; OutsideWrite* outw; uint64_t* outw_count; uint64_t outw_cap;      // accepted members outside their region (OutsideWriteCheck)
    PM_HD void wave(int64_t w) const {
        if ((gate && *gate) || (unsure && *unsure && !defer)) return;      // (the collinear test failed and nobody has said which clusters may run)
        const int64_t cl = cl0 + xcd_item(w, ncl);
        if (cl >= cl0 + ncl) return;
        if (defer && defer[cl]) return;      // (done[cl] stays 0)
        const int n = S.ngen;
        int64_t pending_min = -1;
        const int64_t x0 = cluster_first[cl], x1 = cluster_first[cl + 1];
        if (wave_leader()) done[cl] = (int32_t)(x1 - x0);
        for (int64_t x = x0; x < x1; x++) {
            const int64_t rid = now_region[x];
            const int64_t* rs = rg_start + rid * n; const int64_t* rl = rg_len + rid * n;
            // a child of a region processed here sorts before this one (or ties with it): the reference would take the child first, and
            // a child needs its search.  The cluster stops; what waits stays on the caller's work list, with the children (:291-292)
            if (pending_min >= 0 && pending_min <= rs[0]) { if (wave_leader()) done[cl] = (int32_t)(x - x0); return; }
            const int64_t row0 = now_row0[x]; const int32_t cnt = now_cnt[x];
            if (cnt >= (1 << 22)) { if (wave_leader()) atomic_or32(trouble, 4u); return; }
            const int64_t okey = order_key(rs[0], generation);
            if (wave_leader()) rg_pkey[rid] = okey + 3;
            for (int64_t c = row0; c < row0 + cnt; c++) {
                const uint32_t f = S.flags[c];
                int32_t dl, len;
                if (wave_leader()) row_key[c] = okey;
                if ((f & kRowReverse) && !(f & (kRowBad | kRowOutside)) && S.lon[c] >= 5) {
                    const int32_t lon = S.lon[c];
                    uint32_t outside = 0;
                    lanes_for(0, n, [&](int j) {
                        if (S.strand[c * n + j]) return;
                        const int64_t a = S.start[c * n + j];
                        if (a < rs[j] - 1 || a + lon > rs[j] + rl[j] + 1) outside = 1;
                    });
                    if (wave_or_u32(outside)) {      // noted for the resolve, with what it sees now
                        int32_t at = 0;
                        if (wave_leader()) at = (int32_t)atomic_add64(foreign_count, 1);
                        at = wave_bcast_i32(at, 0);
                        if ((uint64_t)at >= foreign_cap || lon > 64) { if (wave_leader()) atomic_or32(trouble, 4u); }      // (the host route decides)
                        else {
                            if (wave_leader()) foreign[at] = ForeignRead{(int32_t)c, (int32_t)rid, (int32_t)row0, 0, okey};
                            lanes_for(0, n, [&](int j) { foreign_masks[(uint64_t)at * (uint64_t)n + j] = img_bits64(L, j, S.start[c * n + j], lon); });
                        }
                    }
                }
                const bool acc = settle_row(S, L, P, c, true, &dl, &len);
                if (acc && (f & kRowReverse)) {
                    // a reverse member is flipped against the WHOLE genome (TMum.cpp:33-35): it can pass the sequence check while
                    // lying outside this region's interval, and its marks could then meet another cluster's
                    uint32_t out = 0;
                    lanes_for(0, n, [&](int j) {
                        if (S.strand[c * n + j]) return;
                        const int64_t a = (int64_t)S.start[c * n + j] + dl;
                        if (a < rs[j] - 1 || a + len > rs[j] + rl[j] + 1) out = 1;
                    });
                    if (wave_or_u32(out)) {      // listed: harmless unless somebody on the wrong side of the order can see the marks
                        int32_t at = 0;
                        if (wave_leader()) at = (int32_t)atomic_add64(outw_count, 1);
                        at = wave_bcast_i32(at, 0);
                        if ((uint64_t)at >= outw_cap) { if (wave_leader()) atomic_or32(trouble, 2u); }
                        else if (wave_leader()) outw[at] = OutsideWrite{(int32_t)c, (int32_t)rid, (int32_t)x0, (int32_t)x1, okey};
                    }
                }
                if (acc) lanes_for(0, n, [&](int j) { const int64_t a = (int64_t)S.start[c * n + j] + dl; img_set_range(L, j, a, a + len); rec_set(rec, L.word_off[j], L.nbits[j], a, a + len); });
                if (wave_leader()) {
                    store_coherent32(&S.shift[c], dl); store_coherent32(&S.len[c], len);
                    store_coherent8(&S.state[c], (uint8_t)(((f & kRowBad) ? 0 : kStBuilt) | ((f & (kRowBad | kRowOutside)) ? 0 : kStOk) | (acc ? kStAccepted : 0)));
                }
            }
            // children: left of a new MUM, right of it, in candidate order (:215-254); one that equals a region still waiting in
            // this cluster is dropped, as the work list drops a region equal to one it holds (:294-306)
            int32_t seq = 0;
            for (int64_t c = row0; c < row0 + cnt; c++) {
                if (!(load_coherent8(&S.state[c]) & kStAccepted)) continue;
                const int32_t dl = load_coherent32(&S.shift[c]), len = load_coherent32(&S.len[c]);
                for (int side = 0; side < 2; side++) {
                    int64_t ks, kl;
                    const int32_t smin = region_extent(S, L, P, c, dl, len, side, &ks, &kl);
                    if (smin <= q) continue;
                    int32_t slot = 0;
                    if (wave_leader()) slot = (int32_t)atomic_add64(rg_count, 1);
                    slot = wave_bcast_i32(slot, 0);
                    if ((uint64_t)slot >= rg_cap) continue;
                    lanes_for(0, n, [&](int j) {
                        int64_t a, b;
                        region_side(S, L, P, c, dl, len, side, j, &a, &b);
                        rg_start[(int64_t)slot * n + j] = a; rg_len[(int64_t)slot * n + j] = b - a;
                    });
                    bool dup = false;
                    for (int64_t y = x + 1; y < x1 && !dup; y++) {
                        const int64_t o = now_region[y];
                        uint32_t diff = 0;
                        lanes_for(0, n, [&](int j) {
                            if (rg_start[(int64_t)slot * n + j] != rg_start[o * n + j] || rg_len[(int64_t)slot * n + j] != rg_len[o * n + j]) diff = 1;
                        });
                        dup = wave_or_u32(diff) == 0;
                    }
                    if (wave_leader()) { info[slot] = RegInfo{dup ? -1 : ((x << 24) | (int64_t)seq), ks, kl, smin, (int32_t)c}; rg_pkey[slot] = 0; }
                    seq++;
                    if (!dup && (pending_min < 0 || ks < pending_min)) pending_min = ks;
                }
            }
        }
    }
};

// between the two stages of a validation call (one thread): did the first stage leave a child region or trouble behind?  Then the
// second stage -- formed by the caller on the assumption that it would not -- must not run.  head: [0] region counter, [1] trouble
// word, [2] the gate
struct StageGate {
    uint64_t* head; uint64_t regions_before; int force;      // force: (tests) close the gate whatever the first stage did
    PM_HD void operator()(int64_t) const { head[2] = (force || head[0] != regions_before || (uint32_t)head[1] != 0 || head[3] != 0) ? 1 : 0; }
};

// ------------------------------------------------------------------------------------------ chaining
// The test of one MUM against the open chain's last MUM (setFinalClusters :2596-2700) for the pairs (cur[i], back[i]):
// one wavefront per pair.  With every member of both on the forward strand -- all but inversions -- the test is a reduction:
// every genome's gap (next start - chain end) inside [0, d], and the smallest and largest gap for the ratio test, which the
// caller applies in the reference's float arithmetic.  A pair with a reverse member (the loop's strand rules and its
// `max_gap = fgap` at :2608-2611 depend on the genome order) is handed back: verdict 2, the caller judges it from the rows.
struct JudgePairs {
    Store S; const int32_t* cur; const int32_t* back; int32_t d; int32_t* min_gap; int32_t* max_gap; uint8_t* verdict;
    PM_HD void wave(int64_t i) const {
        const int64_t a = cur[i], b = back[i];
        const int n = S.ngen;
        if ((S.flags[a] | S.flags[b]) & kRowReverse) { if (wave_leader()) verdict[i] = 2; return; }
        const int64_t sa = S.shift[a], eb = (int64_t)S.shift[b] + S.len[b];
        int32_t mn = 0x7fffffff, mx = -0x7fffffff;
        uint32_t bad = 0;
        lanes_for(0, n, [&](int j) {
            const int64_t g = ((int64_t)S.start[a * n + j] + sa) - ((int64_t)S.start[b * n + j] + eb);
            if (g < 0 || g > d) bad = 1;
            const int32_t gi = g < -0x7fffffff ? -0x7fffffff : g > 0x7fffffff ? 0x7fffffff : (int32_t)g;
            if (gi < mn) mn = gi;
            if (gi > mx) mx = gi;
        });
        bad = wave_or_u32(bad); mn = wave_min_i32(mn); mx = wave_max_i32(mx);
        if (wave_leader()) { verdict[i] = bad ? 1 : 0; min_gap[i] = mn; max_gap[i] = mx; }
    }
};
// setInterClusterRegions (src/parsnp.cpp:2389-2460) for the pairs of consecutive LCBs (last MUM of the first, first MUM of
// the next): where the two do not overlap in any genome, the filler from the end of the first to the next marked base.
// add[i]: 1 = a filler is made (its rows in out_start / out_end), 0 = none, 2 = the reference's bookkeeping would overrun
// (a genome whose scan does not run after one whose scan did, :2419-2433): the caller stops as the reference would.
struct FillBetween {
    Store S; Layout L; Packed P; const int32_t* last_of; const int32_t* first_of_next; uint8_t* add; int64_t* out_start; int64_t* out_end;
    PM_HD void wave(int64_t i) const {
        const int64_t ct = last_of[i], nx = first_of_next[i];
        const int n = S.ngen;
        const int64_t ce = (int64_t)S.shift[ct] + S.len[ct], ns = S.shift[nx];
        uint32_t overlap = 0, small = 0;
        int32_t first_scan = 0x7fffffff, last_noscan = -1;
        lanes_for(0, n, [&](int j) {
            const int64_t e = (int64_t)S.start[ct * n + j] + ce;
            if (((int64_t)S.start[nx * n + j] + ns) - e <= 0) overlap = 1;
            const int64_t stop = P.glen[j];
            int64_t end;
            if (e + 1 <= stop) { end = img_next_set(L, j, e + 1) - 1; if (j < first_scan) first_scan = j; }
            else { end = stop - 1; if (j > last_noscan) last_noscan = j; }
            out_start[i * n + j] = e; out_end[i * n + j] = end + 1;
            if (end + 1 - e < 5) small = 1;
        });
        overlap = wave_or_u32(overlap); small = wave_or_u32(small);
        first_scan = wave_min_i32(first_scan); last_noscan = wave_max_i32(last_noscan);
        if (wave_leader()) add[i] = overlap ? 0 : (first_scan < last_noscan ? 2 : (small ? 0 : 1));
    }
};
// ------------------------------------------------------------------------------------------ phases C-D on the store
// The list logic of filterRandom1's sort (:338), setFinalClusters (:2563-2719), filterRandomClustersSimple1 (:433-497), the second
// setFinalClusters (:3261-3268) and setInterClusterRegions (:2389-2460) for the case in which it is ORDER-FREE: all reference
// starts of the accepted MUMs differ (one sorted order; the unstable std::sort of the reference cannot show), the ratio test
// has two outcomes (diag_diff <= 1: a MUM joins the open chain or closes it, so the chain's last MUM is always the list
// predecessor and the chains are the maximal runs between "close" verdicts), every MUM is longer than the filter length.
// Then the sorted list is a radix sort of (reference start, row), a chain boundary is a flag per consecutive pair, an LCB's
// length a segmented sum, the dissolved LCBs (length <= c, never the last: :447) a flag per LCB, the second chaining pass the
// same flags over the compacted list, and the fillers a test per consecutive pair of final LCBs.  The host receives the
// sorted rows of the final MUM list, a head flag per MUM and seven counters; what is not order-free (a tie) is reported in the
// trouble word and the caller runs its own list logic instead -- nothing on the device has changed by then.
constexpr uint8_t kChJoin = 0, kChClose = 1;
constexpr uint64_t kChainTie = 1, kChainOverrun = 2, kChainOrder = 4;      // (kChainOrder: the order check, ForeignBound and the kernels around it, queued ahead of the chain kernels)
// the header of a chain call (int64 words in device memory)
enum { kChN1 = 0, kChLcb1, kChLcbDissolved, kChMumDissolved, kChN2, kChLcb2, kChFill, kChTrouble, kChWords };
// tid = store row (one past the end: 0): accepted?
struct ChainFlag {
    Store S; int64_t rows; int64_t* flag;
    PM_HD void operator()(int64_t c) const { flag[c] = c < rows && (S.state[c] & kStAccepted) ? 1 : 0; }
};
// tid = store row: the sort's input (reference start with the trim applied, row), compacted
struct ChainKeys {
    Store S; const int64_t* pos; uint64_t* key; uint64_t* val; int64_t cap;
    PM_HD void operator()(int64_t c) const {
        if (!(S.state[c] & kStAccepted) || pos[c] >= cap) return;      // (more accepted rows than the caller's list holds: the call fails on the count)
        key[pos[c]] = (uint64_t)(uint32_t)(S.start[c * S.ngen] + S.shift[c]); val[pos[c]] = (uint64_t)c;
    }
};
// One wavefront per list position x: the test of setFinalClusters (:2596-2700) of MUM row[x] against row[x - 1], to the verdict.
// All-forward pairs are a reduction over the genomes (JudgePairs) and the ratio test of :2693 in the reference's float / double
// mix; a pair with a reverse member is left to ChainJudgeReverse (the strand rules of :2604-2625 and `max_gap = fgap` at :2608-2611
// depend on the genome order: one lane walks the genomes).  *count: the list's length (the launch covers its capacity).
struct ChainJudge {
    Store S; const uint64_t* key; const uint64_t* row; const int64_t* count; int32_t d; float diag_diff; uint8_t* verdict; uint64_t* trouble;
    int force_tie;      // (tests) report a tie although there is none: the caller's own list logic takes over
    PM_HD void wave(int64_t x) const {
        if (x >= *count) return;
        if (x == 0) { if (wave_leader()) { verdict[0] = kChClose; if (force_tie) atomic_or64(trouble, kChainTie); } return; }
        const int64_t a = (int64_t)row[x], b = (int64_t)row[x - 1];
        const int n = S.ngen;
        if (key[x] == key[x - 1] && wave_leader()) atomic_or64(trouble, kChainTie);
        const int64_t sa = S.shift[a], la = S.len[a], sb = S.shift[b], lb = S.len[b];
        uint8_t out = kChClose;
        if (!((S.flags[a] | S.flags[b]) & kRowReverse)) {
            int32_t mn = 0x7fffffff, mx = -0x7fffffff;
            uint32_t bad = 0;
            lanes_for(0, n, [&](int j) {
                const int64_t g = ((int64_t)S.start[a * n + j] + sa) - ((int64_t)S.start[b * n + j] + sb + lb);
                if (g < 0 || g > d) bad = 1;
                const int32_t gi = g < -0x7fffffff ? -0x7fffffff : g > 0x7fffffff ? 0x7fffffff : (int32_t)g;
                if (gi < mn) mn = gi;
                if (gi > mx) mx = gi;
            });
            bad = wave_or_u32(bad); mn = wave_min_i32(mn); mx = wave_max_i32(mx);
            if (!bad) {
                // every gap in [0, d]: the loop leaves max_gap = the largest gap (from 0) and min_gap = the smallest (from d + 10)
                float max_gap = 0, min_gap = (float)(d + 10);
                if ((float)mx > max_gap) max_gap = (float)mx;
                if ((float)mn < min_gap) min_gap = (float)mn;
                if (min_gap == 0) min_gap = 1;
                if (max_gap == 0) max_gap = 1;
                out = min_gap / max_gap >= 1.0 - diag_diff ? kChJoin : kChClose;
            }
        } else return;      // (a reverse member: ChainJudgeReverse)
        if (wave_leader()) verdict[x] = out;
    }
};
// tid = list position x: the same test for a pair with a reverse-strand member, genome by genome -- the strand rules of :2604-2625 and
// the `max_gap = fgap` of :2608-2611 make the loop's outcome depend on the order of the genomes, so ONE lane walks them; but the pairs
// do not depend on one another, so every lane of the launch has a pair of its own (a lane of a wavefront per pair until round 6: 7 ms for
// the 60 000 pairs of 500 rearranged genomes, and 63 lanes idle).  A lane streams its two rows, 64-byte lines of 16 genomes each.
struct ChainJudgeReverse {
    Store S; const uint64_t* row; const int64_t* count; int32_t d; float diag_diff; uint8_t* verdict;
    PM_HD void operator()(int64_t x) const {
        if (x < 1 || x >= *count) return;
        const int64_t a = (int64_t)row[x], b = (int64_t)row[x - 1];
        if (!((S.flags[a] | S.flags[b]) & kRowReverse)) return;
        const int n = S.ngen;
        const int64_t sa = S.shift[a], la = S.len[a], sb = S.shift[b], lb = S.len[b];
        uint8_t out = kChClose;
        bool addmum = true;
        float max_gap = 0, min_gap = (float)(d + 10);
        for (int k = 0; k < n; k++) {
            const int64_t ns = (int64_t)S.start[a * n + k] + sa, bs = (int64_t)S.start[b * n + k] + sb;
            const int64_t fgap = ns - (bs + lb);        // forward: next start - chain end
            const int64_t rgap = bs - (ns + la);        // reverse: previous MUM start - next end
            const bool f = S.strand[a * n + k] != 0;
            if (f && fgap > max_gap) max_gap = (float)fgap;
            else if (!f && rgap > max_gap) max_gap = (float)fgap;       // sic (:2608-2611)
            if (f && fgap < min_gap) min_gap = (float)fgap;
            else if (!f && rgap < min_gap) min_gap = (float)rgap;
            if (S.strand[a * n + k] != S.strand[b * n + k]) addmum = false;
            else if (f && fgap < 0) addmum = false;
            else if (!f && fgap >= 0) addmum = false;
            else if (f && fgap > d) addmum = false;
            else if (!f && rgap > d) addmum = false;
            if (!addmum) break;
        }
        if (addmum) {
            if (min_gap == 0) min_gap = 1;
            if (max_gap == 0) max_gap = 1;
            out = min_gap / max_gap >= 1.0 - diag_diff ? kChJoin : kChClose;
        }
        verdict[x] = out;
    }
};
// tid = list position (capacity + 1 of them: the scan's closing word): does a chain begin here?
struct ChainHeads {
    const uint8_t* verdict; const int64_t* count; int64_t* head;
    PM_HD void operator()(int64_t x) const { head[x] = x < *count && verdict[x] == kChClose ? 1 : 0; }
};
// tid = list position: the MUM's length into its LCB's sum (lcb = number of heads up to and including x, less one)
struct ChainLcbSum {
    Store S; const uint64_t* row; const int64_t* count; const int64_t* hpos; uint64_t* lcb_len;
    PM_HD void operator()(int64_t x) const {
        if (x >= *count) return;
        atomic_add64(&lcb_len[hpos[x + 1] - 1], (uint64_t)(int64_t)S.len[(int64_t)row[x]]);
    }
};
// tid = list position (capacity + 1): filterRandomClustersSimple1 (:433-497) -- an LCB whose MUM lengths sum to <= c is dissolved,
// the last one is never examined (:447); survive[x] for the compaction, the two counters of the log
struct ChainDissolve {
    const int64_t* count; const int64_t* hpos; const int64_t* head; const uint64_t* lcb_len; int64_t c; int64_t* survive; int64_t* hdr;
    PM_HD void operator()(int64_t x) const {
        const int64_t n = *count;
        if (x >= n) { survive[x] = 0; return; }
        const int64_t nl = hpos[n], id = hpos[x + 1] - 1;
        if (x == 0) { hdr[kChN1] = n; hdr[kChLcb1] = nl; }
        const bool dis = hdr[kChTrouble] == 0 && id != nl - 1 && (int64_t)lcb_len[id] <= c;
        survive[x] = dis ? 0 : 1;
        if (dis) { atomic_add64((uint64_t*)&hdr[kChMumDissolved], 1); if (head[x]) atomic_add64((uint64_t*)&hdr[kChLcbDissolved], 1); }
    }
};
// one wavefront per list position: a dissolved MUM leaves the layout (:460-466)
struct ChainUnmark {
    Store S; Layout L; const uint64_t* row; const int64_t* count; const int64_t* survive;
    PM_HD void wave(int64_t x) const {
        if (x >= *count || survive[x]) return;
        const int64_t c = (int64_t)row[x];
        const int64_t sh = S.shift[c], len = S.len[c];
        lanes_for(0, S.ngen, [&](int j) { const int64_t a = (int64_t)S.start[c * S.ngen + j] + sh; img_clear_range(L, j, a, a + len); });
    }
};
// tid = list position: the surviving MUMs, in order
struct ChainCompact {
    const int64_t* count; const int64_t* survive; const int64_t* spos; const uint64_t* key; const uint64_t* row; uint64_t* key2; uint64_t* row2; int64_t* hdr;
    PM_HD void operator()(int64_t x) const {
        const int64_t n = *count;
        if (x == 0) hdr[kChN2] = spos[n];
        if (x >= n || !survive[x]) return;
        key2[spos[x]] = key[x]; row2[spos[x]] = row[x];
    }
};
// One wavefront per list position of the final list: where an LCB begins (x >= 1), setInterClusterRegions (:2389-2460) for the
// LCB that ends at x - 1 and the one that begins at x -- FillBetween without the rows, which nothing downstream reads (a
// filler is never printed; it shifts the numbers of the LCBs behind it, :2452-2457).
struct ChainFill {
    Store S; Layout L; Packed P; const uint64_t* row; const int64_t* count; const int64_t* head; int64_t* hdr;
    PM_HD void wave(int64_t w) const {
        const int64_t x = w + 1;
        if (x >= *count || !head[x]) return;
        const int64_t ct = (int64_t)row[x - 1], nx = (int64_t)row[x];
        const int n = S.ngen;
        const int64_t ce = (int64_t)S.shift[ct] + S.len[ct], ns = S.shift[nx];
        uint32_t overlap = 0, small = 0;
        int32_t first_scan = 0x7fffffff, last_noscan = -1;
        lanes_for(0, n, [&](int j) {
            const int64_t e = (int64_t)S.start[ct * n + j] + ce;
            if (((int64_t)S.start[nx * n + j] + ns) - e <= 0) overlap = 1;
            const int64_t stop = P.glen[j];
            int64_t end;
            if (e + 1 <= stop) { end = img_next_set(L, j, e + 1) - 1; if (j < first_scan) first_scan = j; }
            else { end = stop - 1; if (j > last_noscan) last_noscan = j; }
            if (end + 1 - e < 5) small = 1;
        });
        overlap = wave_or_u32(overlap); small = wave_or_u32(small);
        first_scan = wave_min_i32(first_scan); last_noscan = wave_max_i32(last_noscan);
        if (!wave_leader() || overlap) return;
        if (first_scan < last_noscan) atomic_or64((uint64_t*)&hdr[kChTrouble], kChainOverrun);
        else if (!small) atomic_add64((uint64_t*)&hdr[kChFill], 1);
    }
};
// tid = list position: what the host receives -- the store row and the head flag of every MUM of the final list
struct ChainOut {
    const uint64_t* row; const int64_t* count; const int64_t* head; const int64_t* hpos; int32_t* out_row; uint8_t* out_head; int64_t* hdr;
    PM_HD void operator()(int64_t x) const {
        const int64_t n = *count;
        if (x == 0) hdr[kChLcb2] = hpos[n];
        if (x >= n) return;
        out_row[x] = (int32_t)row[x]; out_head[x] = (uint8_t)head[x];
    }
};

// tid = (k, genome): the rows of the fillers that are made (which[k] = their pair), packed for one copy to the host
struct FillGather {
    const int32_t* which; int32_t ngen; const int64_t* in_start; const int64_t* in_end; int64_t* out;      // out: [n_made][2][ngen]
    PM_HD void operator()(int64_t tid) const {
        const int64_t k = tid / ngen; const int j = (int)(tid % ngen);
        const int64_t i = which[k];
        out[(2 * k) * ngen + j] = in_start[i * ngen + j]; out[(2 * k + 1) * ngen + j] = in_end[i * ngen + j];
    }
};
// tid = (i, genome): the listed rows as the host wants them -- start with the trim applied, strand byte
struct StoreRowsOut {
    Store S; const int32_t* rows; int32_t* out_start; uint8_t* out_strand; int raw;      // raw: as the search delivered them
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = rows[tid / S.ngen]; const int j = (int)(tid % S.ngen);
        out_start[tid] = S.start[c * S.ngen + j] + (raw ? 0 : S.shift[c]);
        out_strand[tid] = S.strand[c * S.ngen + j];
    }
};
// tid = row - row0: per-row scalars for the host in one record
struct RowInfo { int32_t start0, len, shift; uint32_t state_flags; };      // state_flags: state | flags << 8
struct StoreInfoOut {
    Store S; int64_t row0; RowInfo* out;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = row0 + tid;
        out[tid] = RowInfo{S.start[c * S.ngen] + S.shift[c], S.len[c], S.shift[c], (uint32_t)S.state[c] | (S.flags[c] << 8)};
    }
};

}  // namespace pm
