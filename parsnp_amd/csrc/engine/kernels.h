// kernels.h -- device code of the multi-MUM engine, written as per-thread functors.
//
// Every kernel is a struct with `operator()(int64_t tid)`: one call is the work of one GPU thread.  The HIP
// build (engine_hip.hip) launches them as 256-thread workgroups on gfx950; tests/emu instantiates the same functors
// in a sequential loop on the host so that the kernel LOGIC can be checked without a GPU (test infrastructure only
// -- the product has no host execution path).  Cross-thread communication is through global atomics, plus three
// wavefront-level steps on the device (event slot reservation by ballot / shuffle scan, the leader's index hit broadcast to
// its followers) whose host versions compute the same values one thread at a time, so both executions give the same result.
//
// Data layout in HBM (see DESIGN.md "Data layout"):
//   blk[]  one 16-byte block per 32 bases: { uint64 b2 : 2 bits per base (A=0 C=1 G=2 T=3, N stored as 0),
//                                             uint32 nm : 1 bit per base (1 = N), uint32 pad }
//          -- 'N' matches 'N' (src/csgmum/csg.c:13-25); both planes of a 32-base window come from two adjacent
//          16-byte loads, i.e. normally one 64-byte sector.
//   every genome is stored twice (forward, reverse complement), each strand starting on a block boundary with
//   64 guard bases either side, so a 32-base window can be read at any offset without bounds checks.
//
// Reference semantics being computed: SURVEY.md 3.3 (= Find_UM / Intersect_UM / Merge_Master / extraction of
// src/csgmum/mum.c and src/parsnp.cpp:1570-1695), file:line citations at each functor.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ inline
#else
#define PM_HD inline
#endif

namespace pm {

// ------------------------------------------------------------------------------------------ portable intrinsics
PM_HD int ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
PM_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
PM_HD int ctz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}
PM_HD int clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return __builtin_clz(x);
#endif
}
PM_HD uint64_t atomic_cas64(uint64_t* p, uint64_t expect, uint64_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)atomicCAS((unsigned long long*)p, (unsigned long long)expect, (unsigned long long)val);
#else
    uint64_t old = *p; if (old == expect) *p = val; return old;
#endif
}
PM_HD int32_t atomic_exch32(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicExch(p, v);
#else
    int32_t o = *p; *p = v; return o;
#endif
}
PM_HD uint64_t atomic_add64(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)atomicAdd((unsigned long long*)p, (unsigned long long)v);
#else
    uint64_t o = *p; *p = o + v; return o;
#endif
}
// Reserve n consecutive slots of a global append buffer.  On the device every lane of the (full, converged) wavefront
// calls this once: the 64 counts are prefix-summed with cross-lane shuffles and ONE atomic per wavefront is issued
// (a single hot counter otherwise costs 60 % of SeedExtend).  The host emulation has no wavefronts: plain add.
PM_HD uint64_t wave_reserve(uint64_t* counter, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = (int)__lane_id();
    uint32_t x = n;
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = (uint32_t)__shfl_up((int)x, d, 64); if (lane >= d) x += y; }
    const uint32_t total = (uint32_t)__shfl((int)x, 63, 64);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd((unsigned long long*)counter, (unsigned long long)total);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 63, 64) << 32) | (uint32_t)__shfl((int)(base & 0xffffffffu), 63, 64);
    return (uint64_t)base + (x - n);
#else
    uint64_t o = *counter; *counter = o + n; return o;
#endif
}
// the same for at most one slot per lane: ballot + popcount instead of a 6-step shuffle scan
PM_HD uint64_t wave_reserve01(uint64_t* counter, bool want) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int first = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if ((int)__lane_id() == first) base = atomicAdd((unsigned long long*)counter, (unsigned long long)__popcll(mask));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, first);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), first);
    return (((uint64_t)hi << 32) | lo) + before;
#else
    uint64_t o = *counter; *counter = o + (want ? 1 : 0); return o;
#endif
}
// the same for at most TWO slots per lane (SeedExtend's two samples): two ballots instead of a 6-step shuffle scan
PM_HD uint64_t wave_reserve2(uint64_t* counter, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m1 = __ballot(n >= 1), m2 = __ballot(n >= 2);
    if ((m1 | m2) == 0) return 0;
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)) +
                            __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
    const int first = __ffsll((long long)m1) - 1;
    unsigned long long base = 0;
    if ((int)__lane_id() == first) base = atomicAdd((unsigned long long*)counter, (unsigned long long)(__popcll(m1) + __popcll(m2)));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, first);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), first);
    return (((uint64_t)hi << 32) | lo) + before;
#else
    uint64_t o = *counter; *counter = o + n; return o;
#endif
}
PM_HD void atomic_max32(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}
PM_HD void atomic_min32(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
PM_HD void atomic_or32(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
PM_HD void atomic_or64(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr((unsigned long long*)p, (unsigned long long)v);
#else
    *p |= v;
#endif
}

// ------------------------------------------------------------------------------------------ wavefront steps
// The kernels below are launched one WAVEFRONT per work item (64-thread workgroups, `wave(w)` instead of
// `operator()(tid)`); lanes cooperate through shuffles and LDS.  The host emulation (tests only) runs `wave(w)` as one
// sequential loop that computes the same values.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline bool wave_leader_k() { return __lane_id() == 0; }
__device__ inline int32_t wave_incl_sum(int32_t x) {
    const int lane = (int)__lane_id();
    for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    return x;
}
__device__ inline int64_t wave_incl_sum64(int64_t x) {
    const int lane = (int)__lane_id();
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t y = ((int64_t)__shfl_up((int)(x >> 32), d, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}
__device__ inline int32_t wave_incl_min(int32_t x) {
    const int lane = (int)__lane_id();
    for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(x, d, 64); if (lane >= d && y < x) x = y; }
    return x;
}
__device__ inline int32_t wave_all_max(int32_t x) {
    for (int d = 32; d >= 1; d >>= 1) { const int32_t y = __shfl_xor(x, d, 64); if (y > x) x = y; }
    return x;
}
__device__ inline uint32_t wave_all_or(uint32_t x) {
    for (int d = 32; d >= 1; d >>= 1) x |= (uint32_t)__shfl_xor((int)x, d, 64);
    return x;
}
#else
inline bool wave_leader_k() { return true; }
#endif

// Workgroups are handed to the 8 XCDs of the chip round robin by their number, and every XCD has its own L2.  Work items whose
// NEIGHBOURS read the same cache lines are therefore numbered so that a contiguous eighth of them lands on one XCD: workgroup w
// takes item (w mod 8) * ceil(n / 8) + w / 8 of a launch of xcd_grid(n) workgroups (the last ones may find nothing).
constexpr int kXcds = 8;
PM_HD int64_t xcd_grid(int64_t n) { return (n + kXcds - 1) / kXcds * kXcds; }
PM_HD int64_t xcd_item(int64_t w, int64_t n) { return (w % kXcds) * ((n + kXcds - 1) / kXcds) + w / kXcds; }

// ------------------------------------------------------------------------------------------ shared structures
struct alignas(16) SeqBlock { uint64_t b2; uint32_t nm; uint32_t pad; };   // 32 bases
struct Packed {            // the resident genomes
    const SeqBlock* blk;
    const int64_t* goff;   // [2*g + strand] global base offset of the strand
    const int64_t* glen;   // [g] genome length
};

struct RegionInfo {        // one per region of a batch
    int64_t ref_pos;       // start of the reference substring in genome 0
    int32_t nR;            // its length
    int32_t K;             // seed length  = min(max(minsize,1), 16)
    int32_t stride;        // query sampling step = max(minsize,1) - K + 1
    int32_t minlen;        // max(minsize, 1): shortest event that can matter
    int32_t minsize;       // as requested (candidate test, parsnp.cpp:1663)
    uint32_t tmask;        // hash-table slice size - 1 (power of two, >= 1.5 nR)
    int64_t tbase;         // hash-table slice start
    int64_t posbase;       // start of this region in the per-reference-position arrays
    int64_t fbase;         // first uint32 word of this region's presence filter
    uint32_t fmask;        // filter bits - 1 (power of two, >= 8 nR): one hashed bit per K-mer of the reference substring
    uint32_t pad_;
};

constexpr uint64_t kEmpty = ~0ull;
#ifndef PM_PER
#define PM_PER 2
#endif
constexpr int kPer = PM_PER;           // SeedExtend: adjacent query samples per lane (they share the lane's sequence windows)
constexpr int kUnitSamples = 64 * kPer;   // query samples per work unit
#ifndef PM_LEAD
#define PM_LEAD 4
#endif
// SeedExtend: one lane in kLead (a leader) probes the index, the others follow its hit.  (8 until round 6: after an indel the samples up
// to the next leader are off their leader's diagonal and go to SeedRest one by one -- 5.5 M of the anchor call's 100 M samples at
// 200 x 5 Mb; with a leader every 4 lanes 3.3 M, with one every 2 lanes 2.0 M: event search 1.94 / 1.82 / 1.87 ms per step)
constexpr int kLead = PM_LEAD;
// (A lane whose first K-mer is not at the predicted position trying one base to either side in the windows it holds was measured
// in round 4 and removed in round 5: SeedRest's queue halves, 0.50 -> 0.29 ms, but SeedExtend -- bound by instruction issue -- pays
// 129 more vector instructions in nearly every wavefront, 1.78 -> 1.99 ms.  History: up to commit 682a045.)
#ifndef PM_KMAX
#define PM_KMAX 16
#endif
constexpr int kMaxK = PM_KMAX;         // longest seed: K = min(minsize, kMaxK), query sampling step minsize - K + 1 (a tag holds 16 bases)
constexpr int kSlices = 1024;          // the event buffer is appended through this many independent counters
constexpr int kSliceStride = 8;        // uint64 words between two counters: one 64-byte line each

// error bits raised by kernels
constexpr uint32_t kErrWork = 1u;  // per-thread work budget exceeded (degenerate repeat structure)
constexpr uint32_t kErrRows = 2u;  // a request row derived on the device lies outside its genome, or is not the region its reference column describes

// 32 bases starting at global base position p: 2-bit plane and N-mask plane
PM_HD void window(const SeqBlock* blk, int64_t p, uint64_t* bits, uint32_t* mask) {
    int64_t w = p >> 5;
    int sh = (int)(p & 31);
    // both blocks always (guard blocks exist), no branch: for sh = 0 the second term shifts out completely
    SeqBlock lo = blk[w];
    SeqBlock hi = blk[w + 1];
    *bits = (lo.b2 >> (2 * sh)) | ((hi.b2 << 1) << (63 - 2 * sh));
    *mask = (lo.nm >> sh) | ((hi.nm << 1) << (31 - sh));
}
// the same from two blocks already in registers, and matching of two windows
struct Win { uint64_t b; uint32_t m; };
PM_HD Win funnel(const SeqBlock& lo, const SeqBlock& hi, int sh) {
    Win w;
    w.b = (lo.b2 >> (2 * sh)) | ((hi.b2 << 1) << (63 - 2 * sh));
    w.m = (lo.nm >> sh) | ((hi.nm << 1) << (31 - sh));
    return w;
}
// window starting t bases (0 <= t < 64) into block b1, out of the consecutive blocks b1, b2, b3
PM_HD Win funnel3(const SeqBlock& b1, const SeqBlock& b2, const SeqBlock& b3, int t) {
    const bool up = t >= 32;
    SeqBlock lo, hi;
    lo.b2 = up ? b2.b2 : b1.b2; lo.nm = up ? b2.nm : b1.nm; lo.pad = 0;
    hi.b2 = up ? b3.b2 : b2.b2; hi.nm = up ? b3.nm : b2.nm; hi.pad = 0;
    return funnel(lo, hi, t & 31);
}
PM_HD int match_fwd(const Win& a, const Win& b) {      // equal bases from the first base of both windows, 0..32
    const uint64_t x = a.b ^ b.b; const uint32_t mx = a.m ^ b.m;
    const uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;
    int c = d ? (ctz64(d) >> 1) : 32;
    if (mx) { const int cm = ctz32(mx); if (cm < c) c = cm; }
    return c;
}
PM_HD int match_bwd(const Win& a, const Win& b) {      // equal bases from the last base of both windows backwards, 0..32
    const uint64_t x = a.b ^ b.b; const uint32_t mx = a.m ^ b.m;
    const uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;
    int c = d ? (clz64(d) >> 1) : 32;
    if (mx) { const int cm = clz32(mx); if (cm < c) c = cm; }
    return c;
}
// differences of two 32-base windows: bit 2t of x / bit t of m set = base t differs (an N only equals an N)
struct Diff { uint64_t x; uint32_t m; };
PM_HD Diff diff_of(const Win& a, const Win& b) { const uint64_t x = a.b ^ b.b; return Diff{(x | (x >> 1)) & 0x5555555555555555ull, a.m ^ b.m}; }
PM_HD int eq_up(const Diff& d, int t) {        // equal bases from base t (0..32) upwards, at most 32 - t
    if (t >= 32) return 0;
    const uint64_t x = d.x >> (2 * t); const uint32_t mm = d.m >> t;
    int c = x ? (ctz64(x) >> 1) : 32 - t;
    if (mm) { const int cm = ctz32(mm); if (cm < c) c = cm; }
    return c;
}
PM_HD int eq_down(const Diff& d, int t) {      // equal bases from base t - 1 (t = 0..32) downwards, at most t
    if (t <= 0) return 0;
    const uint64_t x = d.x << (64 - 2 * t); const uint32_t mm = d.m << (32 - t);
    int c = x ? (clz64(x) >> 1) : t;
    if (mm) { const int cm = clz32(mm); if (cm < c) c = cm; }
    return c;
}
// number of equal bases going right from (a, b), at most maxlen
PM_HD int32_t lce_fwd(const Packed& P, int64_t a, int64_t b, int32_t maxlen) {
    int32_t n = 0;
    while (n < maxlen) {
        uint64_t xa, xb; uint32_t ma, mb;
        window(P.blk, a + n, &xa, &ma);
        window(P.blk, b + n, &xb, &mb);
        uint64_t x = xa ^ xb; uint32_t mx = ma ^ mb;
        uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;
        int c = d ? (ctz64(d) >> 1) : 32;
        if (mx) { int cm = ctz32(mx); if (cm < c) c = cm; }
        n += c;
        if (c < 32) break;
    }
    return n < maxlen ? n : maxlen;
}
// the same, 64 bases per round: three consecutive blocks of each sequence are fetched together (one memory latency per
// 64 bases instead of one per 32)
PM_HD int32_t lce_fwd64(const Packed& P, int64_t a, int64_t b, int32_t maxlen) {
    int32_t n = 0;
    while (n < maxlen) {
        const int64_t pa = a + n, pb = b + n;
        const SeqBlock* ba = P.blk + (pa >> 5); const SeqBlock* bb = P.blk + (pb >> 5);
        const int sa = (int)(pa & 31), sb = (int)(pb & 31);
        const SeqBlock a0 = ba[0], a1 = ba[1], a2 = ba[2], b0 = bb[0], b1 = bb[1], b2 = bb[2];
        int c = match_fwd(funnel(a0, a1, sa), funnel(b0, b1, sb));
        if (c == 32) c += match_fwd(funnel(a1, a2, sa), funnel(b1, b2, sb));
        n += c;
        if (c < 64) break;
    }
    return n < maxlen ? n : maxlen;
}
// number of equal bases going left from (a-1, b-1), at most maxlen
PM_HD int32_t lce_bwd(const Packed& P, int64_t a, int64_t b, int32_t maxlen) {
    int32_t n = 0;
    while (n < maxlen) {
        uint64_t xa, xb; uint32_t ma, mb;
        window(P.blk, a - n - 32, &xa, &ma);
        window(P.blk, b - n - 32, &xb, &mb);
        uint64_t x = xa ^ xb; uint32_t mx = ma ^ mb;
        uint64_t d = (x | (x >> 1)) & 0x5555555555555555ull;
        int c = d ? (clz64(d) >> 1) : 32;
        if (mx) { int cm = clz32(mx); if (cm < c) c = cm; }
        n += c;
        if (c < 32) break;
    }
    return n < maxlen ? n : maxlen;
}
// K-mer at p as a 48-bit tag: 2K base bits | K mask bits << 32
PM_HD uint64_t kmer_tag(const Packed& P, int64_t p, int K) {
    uint64_t b; uint32_t m;
    window(P.blk, p, &b, &m);
    if (K < 32) b &= (1ull << (2 * K)) - 1;
    if (K < 32) m &= (uint32_t)((1ull << K) - 1);
    return b | ((uint64_t)m << 32);
}
PM_HD uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
// hash of a 48-bit K-mer tag.  low 32 bits: slot index, bits 35..: filter bit, high 32 bits: fingerprint.  Built from two
// 32-bit finalisers (the kernel is VALU bound and 64-bit multiplies cost ~10 vector ops each on gfx950).
PM_HD uint64_t hash_tag(uint64_t t) {
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t a = fmix32(lo ^ (hi * 0x9e3779b1u));
    const uint32_t b = fmix32(a ^ lo ^ 0x68bc21ebu) + hi;
    return ((uint64_t)b << 32) | a;
}
// The index is keyed by CANONICAL K-mers: the smaller of a K-mer's tag and the tag of its reverse complement.  One probe
// per sampled query K-mer then serves both strands of the query -- a hit whose reference K-mer equals the query K-mer is a
// forward-strand seed, one that equals its reverse complement a reverse-strand seed (a palindromic K-mer is both) -- and the
// reverse strand of a query is never streamed (it was 37 % of SeedExtend: nothing but filter probes that miss).
PM_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
// reverse complement of a K-mer tag, 1 <= K <= 16 (2K base bits | K mask bits << 32; an N stays an N, its base bits 0)
PM_HD uint64_t rc_tag(uint64_t tag, int K) {
    const uint32_t b = (uint32_t)tag, m = (uint32_t)(tag >> 32);
    uint32_t rb = brev32(b);                                                   // base order reversed, the two bits of a base swapped
    rb = ((rb >> 1) & 0x55555555u) | ((rb & 0x55555555u) << 1);                // ... and swapped back
    rb = (~rb) >> (32 - 2 * K);                                                // complement (3 - code), the K bases moved down
    uint32_t rm = 0;
    if (m) {
        rm = brev32(m) >> (32 - K);
        uint32_t x = rm;                                                       // one mask bit -> the two bits of its base
        x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
        rb &= ~(x * 3u);
    }
    return (uint64_t)rb | ((uint64_t)rm << 32);
}
PM_HD uint64_t canonical_tag(uint64_t tag, int K) { const uint64_t rc = rc_tag(tag, K); return rc < tag ? rc : tag; }
// largest r in [0, count) with base[r] <= x (base ascending, base[0] <= x)
PM_HD int64_t upper_slot(const int64_t* base, int64_t count, int64_t x) {
    int64_t lo = 0, hi = count;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (base[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// ------------------------------------------------------------------------------------------ genome packing
// tid = output word of one strand. ascii: the genome as uploaded ('A','C','G','T', anything else = N).
struct PackStrand {
    const uint8_t* ascii; int64_t L; int strand; SeqBlock* blk; int64_t word0;   // word0: first block of the strand
    PM_HD void operator()(int64_t tid) const {
        uint64_t bits = 0; uint32_t mask = 0;
        for (int i = 0; i < 32; i++) {
            int64_t p = tid * 32 + i;
            if (p >= L) break;
            uint8_t c = strand ? ascii[L - 1 - p] : ascii[p];
            uint32_t code;
            switch (c) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: code = 4; }
            if (code == 4) mask |= 1u << i;
            else bits |= (uint64_t)(strand ? 3 - code : code) << (2 * i);
        }
        blk[word0 + tid] = SeqBlock{bits, mask, 0};
    }
};

// ------------------------------------------------------------------------------------------ reference index
// Replaces new_CSG/build_CSG/find_leaves (src/csgmum/csg.c:105-575): an open-addressing hash of the reference
// substring's CANONICAL K-mers.  One 8-byte slot per distinct canonical K-mer:  [63:32] fingerprint | [31]
// more-than-one-occurrence flag | [30:0] head position; further occurrences (in either orientation) hang off next[].  A
// fingerprint hit is confirmed against the K-mer at the head position (the caller reads those reference bases anyway).
constexpr uint64_t kMulti = 1ull << 31;
PM_HD int32_t slot_head(uint64_t s) { return (int32_t)(s & 0x7fffffffu); }

// the two filter bits of a K-mer inside its filter word (the second from hash bits the word index does not use)
PM_HD uint32_t filter_mask(uint64_t hv, uint32_t bit) { return (1u << (bit & 31)) | (1u << ((uint32_t)(hv >> 58) & 31)); }

// tid = flat reference position over the batch.
struct IndexInsert {
    Packed P; const RegionInfo* R; int64_t nregions; const int64_t* posbase;   // posbase[nregions+1]
    uint64_t* slots; int32_t* next; uint32_t* filter;
    int32_t* home;      // [position]: the slot of its K-mer in the region's slice (-1: no K-mer starts there) -- RepeatLength reads its slot
                        // from there instead of hashing and probing again (it is rep[], which RepeatLength then overwrites)
    PM_HD void operator()(int64_t tid) const {
        int64_t r = upper_slot(posbase, nregions, tid);
        const RegionInfo& ri = R[r];
        int32_t l = (int32_t)(tid - ri.posbase);
        next[tid] = -1;
        if (l + ri.K > ri.nR) { home[tid] = -1; return; }
        const int64_t base = P.goff[0] + ri.ref_pos;
        const uint64_t tag = canonical_tag(kmer_tag(P, base + l, ri.K), ri.K);
        const uint64_t hv = hash_tag(tag);
        const uint64_t fp = hv & 0xffffffff00000000ull;
        uint32_t h = (uint32_t)hv & ri.tmask;
        {   // presence filter: most query K-mers of a non-matching strand are rejected by one word that lives in L2.  Two
            // hashed bits of the SAME 32-bit word per K-mer (a blocked Bloom filter: still one load per probe): at 8 filter
            // bits per reference position a foreign K-mer passes with 5 % instead of the 12 % of one bit
            const uint32_t bit = (uint32_t)(hv >> 35) & ri.fmask;
            atomic_or32(&filter[ri.fbase + (bit >> 5)], filter_mask(hv, bit));
        }
        for (;;) {
            uint64_t* slot = &slots[ri.tbase + h];
            uint64_t seen = *slot;      // (a look first: the CAS at once was measured, 0.69 instead of 0.47 ms for the 5 Mb reference)
            if (seen == kEmpty) {
                seen = atomic_cas64(slot, kEmpty, fp | (uint64_t)l);
                if (seen == kEmpty) { home[tid] = (int32_t)h; return; }      // first occurrence of this K-mer
            }
            if ((seen & 0xffffffff00000000ull) == fp && canonical_tag(kmer_tag(P, base + slot_head(seen), ri.K), ri.K) == tag) {
                for (;;) {                                          // push onto this K-mer's chain
                    next[tid] = slot_head(seen);
                    uint64_t prev = atomic_cas64(slot, seen, fp | kMulti | (uint64_t)l);
                    if (prev == seen) { home[tid] = (int32_t)h; return; }
                    seen = prev;                                    // another occurrence got in first: same K-mer, new head
                }
            }
            h = (h + 1) & ri.tmask;
        }
    }
};
// -> slot value of the canonical K-mer `tag` in region ri (confirmed against the reference K-mer at the head position), or kEmpty
PM_HD uint64_t index_lookup(const Packed& P, const RegionInfo& ri, const uint64_t* slots, const uint32_t* filter, uint64_t tag) {
    const uint64_t hv = hash_tag(tag);
    const uint64_t fp = hv & 0xffffffff00000000ull;
    uint32_t h = (uint32_t)hv & ri.tmask;
    const int64_t base = P.goff[0] + ri.ref_pos;
    const uint32_t bit = (uint32_t)(hv >> 35) & ri.fmask;
    { const uint32_t fm = filter_mask(hv, bit); if ((filter[ri.fbase + (bit >> 5)] & fm) != fm) return kEmpty; }
    for (;;) {
        const uint64_t seen = slots[ri.tbase + h];
        if (seen == kEmpty) return kEmpty;
        if ((seen & 0xffffffff00000000ull) == fp && canonical_tag(kmer_tag(P, base + slot_head(seen), ri.K), ri.K) == tag) return seen;
        h = (h + 1) & ri.tmask;
    }
}

// -> first slot of the probe sequence that carries the fingerprint of `tag` (NOT confirmed against the reference K-mer:
// the caller reads those bases anyway and falls back to index_lookup on the 2^-32 mismatch), or kEmpty
PM_HD uint64_t index_probe(const RegionInfo& ri, const uint64_t* slots, const uint32_t* filter, uint64_t tag) {
    const uint64_t hv = hash_tag(tag);
    const uint64_t fp = hv & 0xffffffff00000000ull;
    uint32_t h = (uint32_t)hv & ri.tmask;
    const uint32_t bit = (uint32_t)(hv >> 35) & ri.fmask;
    { const uint32_t fm = filter_mask(hv, bit); if ((filter[ri.fbase + (bit >> 5)] & fm) != fm) return kEmpty; }
    for (;;) {
        const uint64_t seen = slots[ri.tbase + h];
        if (seen == kEmpty || (seen & 0xffffffff00000000ull) == fp) return seen;
        h = (h + 1) & ri.tmask;
    }
}

// rep'[l]: longest prefix of R[l..) that occurs at another position of R if that is >= K, else 0.
// (= the uniqueness point pos_label-l of mum.c:219-224 wherever it can influence the output; SURVEY 3.3-1,-7.)
// run[l]: number of consecutive positions from l that hold the same symbol as l, inside the region.  Lets RepeatLength
// compare two suffixes that start inside single-symbol runs (N padding, homopolymers) in O(1) instead of base by base:
// a run of r identical symbols puts r positions on one K-mer chain, and r^2 comparisons of O(r) bases each are what used
// to exhaust the work budget.  tid = flat reference position.
struct RunLength {
    Packed P; const RegionInfo* R; int64_t nregions; const int64_t* posbase; int32_t* run;
    PM_HD void operator()(int64_t tid) const {
        int64_t r = upper_slot(posbase, nregions, tid);
        const RegionInfo& ri = R[r];
        const int32_t l = (int32_t)(tid - ri.posbase);
        const int64_t base = P.goff[0] + ri.ref_pos;
        run[tid] = 1 + lce_fwd(P, base + l, base + l + 1, ri.nR - l - 1);   // the sequence against itself shifted by one base
    }
};
struct RepeatLength {
    Packed P; const RegionInfo* R; int64_t nregions; const int64_t* posbase;
    const uint64_t* slots; const uint32_t* filter; const int32_t* next; const int32_t* run; int32_t* rep; uint32_t* repeated; uint32_t* err; int64_t budget;
    // repeated: one bit per flat reference position, set when the canonical K-mer starting there occurs elsewhere in R, in
    // either orientation (its index chain has more than one entry); n/8 bytes, L2 resident -- SeedExtend's followers read
    // it instead of probing the index
    PM_HD void operator()(int64_t tid) const {
        int64_t r = upper_slot(posbase, nregions, tid);
        const RegionInfo& ri = R[r];
        int32_t l = (int32_t)(tid - ri.posbase);
        int32_t best = 0;
        bool shared = false;      // the canonical K-mer at l occurs elsewhere in R (in either orientation)
        const int32_t h = rep[tid];      // (IndexInsert left the K-mer's slot here)
        if (l + ri.K <= ri.nR && h >= 0) {
            int64_t base = P.goff[0] + ri.ref_pos;
            const uint64_t slot = slots[ri.tbase + h];
            if (slot & kMulti) {
                shared = true;
                int64_t work = 0;
                const int32_t r1 = run[tid];
                const bool single = r1 >= ri.K;          // the K-mer is one symbol K times: its chain holds the runs of that symbol and of its complement
                for (int32_t o = slot_head(slot); o >= 0; o = next[ri.posbase + o]) {
                    if (o == l) continue;
                    const int32_t cap = ri.nR - (l > o ? l : o);
                    int32_t len, cost;
                    if (single) {
                        // both suffixes begin with a run: different run lengths part where the shorter run ends
                        const int32_t r2 = run[ri.posbase + o];
                        if (lce_fwd(P, base + l, base + o, 1) == 0) { len = 0; cost = 1; }       // the complementary symbol
                        else if (r1 != r2) { len = r1 < r2 ? r1 : r2; cost = 1; }
                        else { const int32_t more = lce_fwd(P, base + l + r1, base + o + r1, cap - r1); len = r1 + more; cost = 1 + (more >> 5); }
                        if (len > cap) len = cap;
                    } else {
                        // from the K-mer's first base: an occurrence in the other orientation parts inside the K-mer and does not count
                        len = lce_fwd(P, base + l, base + o, cap);
                        cost = 1 + (len >> 5);
                        if (len < ri.K) len = 0;
                    }
                    if (len > best) best = len;
                    work += cost;
                    if (work > budget) { atomic_or32(err, kErrWork); break; }
                }
            }
        }
        rep[tid] = best;
        if (shared) atomic_or32(&repeated[tid >> 5], 1u << (tid & 31));
    }
};

// ------------------------------------------------------------------------------------------ work units
// A (region, query piece) pair whose two sides both fit 128 bases -- practically all of the recursion's pairs -- is
// compared diagonal by diagonal in registers (SmallPairEvents) instead of going through the K-mer index.
PM_HD bool small_pair(int64_t nR, int64_t m) { return nR <= 128 && m <= 128; }
// units of one (region, query genome) pair: ceil(samples / 64) -- the samples of the FORWARD strand of the query piece; a
// seed of the reverse strand is found through the same probe (canonical K-mers)
struct CountUnits {
    const RegionInfo* R; const int64_t* lens; int32_t ngen; int64_t* count;   // lens[r*ngen + g]
    int32_t g_first, g_last;   // query genomes [g_first, g_last) live on this GPU (all of them unless the run is sharded)
    int no_small;              // every pair through SeedExtend (calcmumi mode)
    PM_HD void operator()(int64_t pair) const {
        int64_t r = pair / (ngen - 1); int g = (int)(pair % (ngen - 1)) + 1;
        const RegionInfo& ri = R[r];
        int64_t m = lens[r * ngen + g];
        int64_t ns = (m >= ri.K && ri.nR >= ri.K && g >= g_first && g < g_last) ? (m - ri.K) / ri.stride + 1 : 0;
        if (small_pair(ri.nR, m) && !no_small) ns = 0;          // handled by SmallPairEvents, without index probes
        count[pair] = (ns + kUnitSamples - 1) / kUnitSamples;
    }
};
// everything a SeedExtend wavefront needs to know about its unit, in one 32-byte record (one scalar load)
struct alignas(32) UnitRec {
    int64_t qbase;      // global base offset of the query piece on the forward strand
    int64_t qbase_r;    // ... and of the same piece, mirrored, on the stored reverse complement
    int32_t region;     // index into RegionInfo
    int32_t pair;       // region * (ngen-1) + (query genome - 1): the event key prefix
    int32_t m;          // length of the query piece
    int32_t chunk;      // which kUnitSamples samples of the piece
};
// ------------------------------------------------------------------------------------------ request rows held by the device
// tid = (region, genome): request rows the host never saw (regions of the region store, store_kernels.h), checked as the host
// checks rows it is handed (pm_multi_mum_batch: "region outside its genome"); the reference column must be the one the caller
// sized the index from
struct CheckRows {
    const RegionInfo* R; const int64_t* starts; const int64_t* lens; int32_t ngen; const int64_t* glen; uint32_t* err;
    PM_HD void operator()(int64_t tid) const {
        const int64_t r = tid / ngen; const int j = (int)(tid % ngen);
        const int64_t st = starts[tid], ln = lens[tid];
        bool bad = st < 0 || ln < 0 || st + ln > glen[j];
        if (j == 0) bad = bad || st != R[r].ref_pos || ln != (int64_t)R[r].nR;
        if (bad) atomic_or32(err, kErrRows);
    }
};

// ------------------------------------------------------------------------------------------ layout image
// The reference keeps one bit per base of every genome, set under every accepted MUM (mumlayout, src/parsnp.cpp:3181-3186; marked
// :1836-1839): here an image in device memory (store_kernels.h), 64-bit words, bit i of a genome = bit (i & 63) of its word i >> 6;
// genome j occupies words [word_off[j], word_off[j+1]).
// tid = genome: the bit past the last base (what a scan to the right stops at)
struct LayoutSentinel {
    const int64_t* word_off; const int64_t* nbits; uint64_t* image;
    PM_HD void operator()(int64_t tid) const {
        const int64_t nb = nbits[tid];
        if (nb > 0) atomic_or64(&image[word_off[tid] + ((nb - 1) >> 6)], 1ull << ((nb - 1) & 63));
    }
};
// tid = work unit: which (pair, chunk) it is (off[] = exclusive prefix of the per-pair unit counts)
struct FillUnits {
    Packed P; const int64_t* starts; const int64_t* lens; int32_t ngen;
    const int64_t* off; const int64_t* count; int64_t npairs; UnitRec* units;
    PM_HD void operator()(int64_t tid) const {
        if (tid >= off[npairs]) return;                  // (a launch over a capacity: past the units there are)
        int64_t pair = upper_slot(off, npairs, tid);     // the last pair whose first unit is <= tid owns it
        const int64_t r = pair / (ngen - 1); const int g = (int)(pair % (ngen - 1)) + 1;
        const int64_t m = lens[r * ngen + g], qs = starts[r * ngen + g];
        UnitRec rec;
        rec.qbase = P.goff[2 * g] + qs;
        rec.qbase_r = P.goff[2 * g + 1] + (P.glen[g] - qs - m);
        rec.region = (int32_t)r; rec.pair = (int32_t)pair; rec.m = (int32_t)m;
        // (pair by pair: laying the anchor call's units out chunk by chunk, so that the wavefronts in flight look at one stretch
        // of the reference in many genomes, was measured -- the same 1.78 ms, and 2.26 instead of 1.75 GB fetched per launch:
        // neighbouring units of ONE genome share their window blocks, and that sharing is lost)
        rec.chunk = (int32_t)(tid - off[pair]);
        units[tid] = rec;
    }
};

// ------------------------------------------------------------------------------------------ seed & extend
// Replaces Find_UM (src/csgmum/mum.c:177-250): enumerates every R-unique maximal exact match of length >= minlen
// between BOTH strands of one query piece and the reference substring.  tid = unit*64 + lane; lane t takes the kPer
// ADJACENT samples (chunk*64 + t)*kPer .. +kPer-1 of the piece's forward strand; sample s sits at query offset s*stride.
// A match of length >= minlen = stride + K - 1 contains >= 1 whole sampled K-mer whichever strand it lies on; a forward
// match is reported from its first sampled K-mer only, a reverse match from its last one (= the first in the reverse
// strand's own direction).
//
// The launch is bound by the LATENCY of its chain of dependent memory round trips (unit record, query blocks, index
// probe, reference blocks, arms, append) at the hardware's 8 wavefronts per SIMD -- not by bytes and, since one probe
// serves 2 * kLead * kPer K-mers, no longer by the request rate.  Adjacent samples are stride <= 16 bases apart, so the
// 128-base windows a lane loads around its first sample (query and reference) hold its other samples as well: kPer
// samples per lane means 1/kPer as many wavefronts walking that chain for the same work.
//
// What SeedExtend itself does is the part all lanes share: windows, tags, the leaders' probes, confirmation at the
// predicted positions, and the arms of a confirmed forward seed.  The rest is rare per sample but not per wavefront -- a
// sample whose K-mer crosses a difference asks the index itself (15 % of the samples at 1 % divergence; one in twenty
// passes the presence filter), a chain of repeated K-mers wants walking, a seed lies on the reverse strand -- and a
// wavefront of 128 samples almost always holds one such lane, for which all 64 used to execute ~600 instructions of slot
// walks and memory arms.  Those samples are now QUEUED (16 bytes each, one reservation per wavefront) and worked off by
// SeedRest, one lane per queued sample in full wavefronts.
struct RestItem { int32_t unit; int32_t sample; int32_t l; int32_t pad; };      // l >= 0: a reverse-strand seed confirmed at reference position l; l < 0: probe the index
struct SeedExtend {
    Packed P; const RegionInfo* R; const UnitRec* units;
    const uint64_t* slots; const uint32_t* filter; const int32_t* next; const int32_t* rep; const uint32_t* repeated;
    uint64_t* ev_key; uint64_t* ev_val; uint64_t* ev_counters; uint64_t slice_cap; int lbits; uint32_t* err; int64_t budget;
    RestItem* queue; uint64_t* queue_count; uint64_t queue_cap;      // samples handed to SeedRest: kSlices sub-queues of queue_cap items
    const int64_t* nunits_live;      // the units there are (the launch may cover a capacity: a search of store regions does not wait for the count)
    PM_HD void operator()(int64_t tid) const {
        int64_t unit = tid >> 6; int lane = (int)(tid & 63);
#if defined(__HIP_DEVICE_COMPILE__)
        // the unit is the same for the 64 lanes of a wavefront: say so, and its record (pair, region, lengths, offsets)
        // is fetched with scalar loads instead of 64-lane vector loads of one address
        unit = (int64_t)__builtin_amdgcn_readfirstlane((int)unit);
#endif
        if (unit >= *nunits_live) return;
        const UnitRec rec = units[unit];
        const int32_t pair = rec.pair;
        const RegionInfo& ri = R[rec.region];
        const int64_t m = rec.m;
        const int64_t qbase = rec.qbase;
        const int64_t rbase = P.goff[0] + ri.ref_pos;
        const int K = ri.K;
        const int32_t stride = ri.stride;
        // events are appended to one of kSlices sub-buffers (all four wavefronts of a workgroup use the same one): a
        // single counter serialises the ~10^6 wavefront reservations of a recursion batch
        const uint64_t slice = (uint64_t)((unit >> 2) & (kSlices - 1));
        uint64_t* ev_count = ev_counters + slice * kSliceStride;
        const uint64_t ev_cap = slice_cap;
        uint64_t* const key_out = ev_key + slice * slice_cap;
        uint64_t* const val_out = ev_val + slice * slice_cap;
        // the first kPer events of the lane wait in registers and are written with ONE reservation per wavefront; more
        // (repeated or palindromic K-mers) are rare and take their own slots
        uint64_t bk[kPer], bv[kPer];
        int nb = 0;
#pragma unroll
        for (int u = 0; u < kPer; u++) { bk[u] = kEmpty; bv[u] = 0; }
        auto emit = [&](int strand, int32_t l0, int64_t j0, int32_t len) {
            const uint64_t ek = ((((uint64_t)pair << lbits) | (uint64_t)l0) << 1) | (uint64_t)strand;
            const uint64_t evv = ((uint64_t)j0 << 32) | (uint32_t)len;
            bool kept = false;
#pragma unroll
            for (int u = 0; u < kPer; u++) if (!kept && nb == u) { bk[u] = ek; bv[u] = evv; kept = true; }
            if (kept) nb++;
            else { const uint64_t at = atomic_add64(ev_count, 1); if (at < ev_cap) { key_out[at] = ek; val_out[at] = evv; } }
        };
        const uint64_t kbits = K < 32 ? (1ull << (2 * K)) - 1 : ~0ull;
        const uint32_t kmask = K < 32 ? (uint32_t)((1ull << K) - 1) : ~0u;
        auto tag_of = [&](const Win& w) { return (w.b & kbits) | ((uint64_t)(w.m & kmask) << 32); };
        const int32_t s0 = (rec.chunk * 64 + lane) * kPer;
        const int64_t j0 = (int64_t)s0 * stride;
        // Adjacent samples are `stride` bases apart.  With (kPer - 1) * stride + K <= 32 all K-mers of the lane start inside
        // ONE 32-base window -- window 1 of the 96 bases [j0 - 32, j0 + 64) that four 16-byte blocks hold
        const bool regs = (kPer - 1) * stride + K <= 32;
        const int64_t qp0 = qbase + j0;
        const int shq = (int)(qp0 & 31);
        Win qw0 = Win{0, 0}, qw1 = qw0, qw2 = qw0;
        if (j0 + K <= m && regs) {
            const SeqBlock* qb = P.blk + (qp0 >> 5) - 1;
            const SeqBlock q0 = qb[0], q1 = qb[1], q2 = qb[2], q3 = qb[3];
            qw0 = funnel(q0, q1, shq); qw1 = funnel(q1, q2, shq); qw2 = funnel(q2, q3, shq);
        }
        bool valid[kPer];
#pragma unroll
        for (int u = 0; u < kPer; u++) valid[u] = j0 + (int64_t)u * stride + K <= m;
        auto tag_at = [&](int u) -> uint64_t {      // the u-th K-mer of the lane
            if (!regs) return kmer_tag(P, qp0 + (int64_t)u * stride, K);
            const int off = u * stride;
            return ((qw1.b >> (2 * off)) & kbits) | ((uint64_t)((qw1.m >> off) & kmask) << 32);
        };
        // every sample's K-mer and the K-mer of the reverse strand over the same bases, ONCE: the leader's probe, the strand
        // tests and the presence-filter checks below all want one or both (five reverse complements per lane otherwise, ~100
        // of the kernel's 1 300 vector instructions per wavefront -- and the launch is bound by instruction issue)
        uint64_t tgs[kPer], gts[kPer];
#pragma unroll
        for (int u = 0; u < kPer; u++) { tgs[u] = valid[u] ? tag_at(u) : 0; gts[u] = rc_tag(tgs[u], K); }
        auto ctag_at = [&](int u) -> uint64_t { return gts[u] < tgs[u] ? gts[u] : tgs[u]; };
        // Index probes are scarce (random requests at the fabric's request rate, two dependent round trips).  Consecutive
        // lanes hold consecutive samples, and inside a forward match the K-mer of sample s+t sits t*stride bases after the
        // K-mer of sample s.  So only every kLead-th lane (a leader) probes the index, for its first sample; every other sample
        // first looks where a leader's hit predicts its K-mer -- the lane's own leader, else the leader of the group before
        // or after (a leader's K-mer crosses a difference in one sample out of seven at 1 % divergence).  If the reference
        // K-mer there equals its own and its canonical form occurs nowhere else in R, that position is all a probe would have
        // returned.  Otherwise the sample probes for itself.
        const int sub = lane & (kLead - 1);
        const bool follow = regs && stride <= K && m >= (int64_t)kUnitSamples * stride;   // short query pieces (recursion): every sample probes
        uint64_t slot = kEmpty;      // the leader's probe of its first sample
        int32_t base = -1;           // predicted reference position of the lane's FIRST sample
        bool own_probe = false;      // ... which is the head of this lane's own slot
        if (follow) {
            const int g0 = lane & ~(kLead - 1);
#if defined(__HIP_DEVICE_COMPILE__)
            if (valid[0] && sub == 0) slot = index_probe(ri, slots, filter, ctag_at(0));
            const int32_t mine = (sub == 0 && slot != kEmpty && !(slot & kMulti)) ? slot_head(slot) : -1;
            const int32_t own = __shfl(mine, g0, 64);
            int32_t before = __shfl(mine, (g0 + 64 - kLead) & 63, 64), after = __shfl(mine, (g0 + kLead) & 63, 64);
            if (g0 == 0) before = -1;
            if (g0 == 64 - kLead) after = -1;
#else
            // host emulation (one thread at a time): the leaders' probes are recomputed by each lane
            auto probe_at = [&](int64_t jj) -> int32_t {
                if (jj < 0 || jj + K > m) return -1;
                const uint64_t ls = index_probe(ri, slots, filter, canonical_tag(kmer_tag(P, qbase + jj, K), K));
                return (ls != kEmpty && !(ls & kMulti)) ? slot_head(ls) : -1;
            };
            if (valid[0] && sub == 0) slot = index_probe(ri, slots, filter, ctag_at(0));
            const int32_t own = sub == 0 ? ((slot != kEmpty && !(slot & kMulti)) ? slot_head(slot) : -1) : probe_at(j0 - (int64_t)sub * kPer * stride);
            int32_t before = -1, after = -1;
            if (own < 0 && g0 > 0) before = probe_at(j0 - (int64_t)(kLead + sub) * kPer * stride);
            if (own < 0 && before < 0 && g0 < 64 - kLead) after = probe_at(j0 + (int64_t)(kLead - sub) * kPer * stride);
#endif
            if (own >= 0) { base = own + sub * kPer * stride; own_probe = sub == 0; }
            else if (before >= 0) base = before + (kLead + sub) * kPer * stride;
            else if (after >= (kLead - sub) * kPer * stride) base = after - (kLead - sub) * kPer * stride;
        }
        // Sample u is looked for at base + u*stride: the 96 reference bases [base - 32, base + 64) are fetched by all lanes
        // in ONE round and compared with the query's base by base (three 32-base difference words).  Everything the windows
        // can say -- is the K-mer there, how far the match runs to the left, where the first difference to the right lies --
        // is read off those words for all the lane's samples at once, so that the eight blocks are dead before the rare,
        // divergent rest (probes, long arms, chains) begins: the kernel's registers decide how many wavefronts hide each
        // other's memory latency.
        enum : uint8_t { kNone = 0, kProbe = 1, kFwd = 2, kRev = 4 };      // what to do with a sample after the window phase
        uint8_t todo[kPer]; int32_t left[kPer], right[kPer];
        bool any_fast = false;       // the lane holds reference windows: [base - 32, base + 64) against the query's [j0 - 32, j0 + 64)
        int32_t first_diff = 64;     // ... and this is the first base of [j0, j0 + 64) that differs from the reference on that diagonal (64: none)
        {
            bool fast[kPer];
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                fast[u] = follow && valid[u] && base >= 0 && base + u * stride + K <= ri.nR;
                if (u == 0 && sub == 0 && follow) fast[0] = fast[0] && own_probe;      // a leader's first sample: its own probe says where (or that there is nothing)
                any_fast = any_fast || fast[u];
            }
            Diff d0 = Diff{0, 0}, d1 = d0, d2 = d0;
            Win rw1 = Win{0, 0};
            uint64_t rwords = 0;
            const int64_t fpos0 = ri.posbase + (base >= 0 ? base : 0);
            int rbit0 = (int)(fpos0 & 31);      // bit of the lane's first sample in `rwords`
            if (any_fast) {
                const int64_t rp = rbase + base;
                const SeqBlock* rb = P.blk + (rp >> 5) - 1;
                const uint32_t rep0 = repeated[fpos0 >> 5], rep1 = repeated[(fpos0 >> 5) + 1];      // (one more word than positions exists)
                const SeqBlock r0 = rb[0], r1 = rb[1], r2 = rb[2], r3 = rb[3];
                const int shr = (int)(rp & 31);
                const Win rw0 = funnel(r0, r1, shr), rw2 = funnel(r2, r3, shr);
                rw1 = funnel(r1, r2, shr);
                d0 = diff_of(qw0, rw0); d1 = diff_of(qw1, rw1); d2 = diff_of(qw2, rw2);
                rwords = (uint64_t)rep0 | ((uint64_t)rep1 << 32);
                first_diff = eq_up(d1, 0);
                if (first_diff == 32) first_diff += eq_up(d2, 0);
            }
            bool fwd_here = false;       // the previous sample of this lane was confirmed as a forward seed on this diagonal
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                todo[u] = kNone; left[u] = 0; right[u] = 0;
                if (!valid[u]) { fwd_here = false; continue; }
                if (follow && u == 0 && sub == 0 && !own_probe) { todo[0] = kProbe; fwd_here = false; continue; }      // (handled from `slot` below)
                if (!fast[u]) { todo[u] = kProbe; fwd_here = false; continue; }
                const int off = u * stride;
                const bool same = ((d1.x >> (2 * off)) & kbits) == 0 && ((d1.m >> off) & kmask) == 0;      // the reference K-mer there is the sample's
                const bool predicted = !(u == 0 && own_probe);
                const bool shared = ((rwords >> ((rbit0 + off) & 63)) & 1u) != 0;      // its canonical form occurs elsewhere in R
                const uint64_t tg = tgs[u];
                const uint64_t gat = gts[u];                       // the K-mer of the reverse strand that covers the same bases
                bool rev = same && gat == tg;                      // a palindrome seeds both strands
                if (!same && !predicted) rev = (((rw1.b >> (2 * off)) & kbits) | ((uint64_t)((rw1.m >> off) & kmask) << 32)) == gat;   // the probe found the K-mer on the other strand
                if (predicted ? !(same && !shared) : !(same || rev)) { todo[u] = kProbe; fwd_here = false; continue; }
                // forward seed: after a confirmed forward seed of the same lane on the same diagonal the `stride` bases before
                // this K-mer lie inside that K-mer (stride <= K) -- the match belongs to the earlier sample
                if (same) {
                    if (!fwd_here) {
                        int lf = eq_down(d1, off);
                        if (lf == off) lf += eq_down(d0, 32);
                        // only `stride` bases matter -- a longer left arm means an earlier sample owns the match -- and the arm
                        // ends where the query piece or the reference substring begins
                        const int64_t ju = j0 + off; const int32_t lu = base + off;
                        const int32_t lim = (int32_t)(ju < lu ? ju : lu);
                        if (lf > lim) lf = lim;
                        left[u] = lf;
                        if (lf < stride) {
                            const int t = off + K;
                            int rt = eq_up(d1, t);
                            if (rt == 32 - t) rt += eq_up(d2, 0);
                            right[u] = rt;                          // = 64 - t: equal as far as the windows reach
                            todo[u] |= kFwd;
                        }
                    }
                    fwd_here = true;
                } else fwd_here = false;
                if (rev) todo[u] |= kRev;
            }
        }
        // the samples for SeedRest: at most one item per sample
        int32_t qs[kPer], ql[kPer];
        int nqueued = 0;
        auto hand_over = [&](int32_t sample, int32_t l) {
            bool kept = false;
#pragma unroll
            for (int u = 0; u < kPer; u++) if (!kept && nqueued == u) { qs[u] = sample; ql[u] = l; kept = true; }
            nqueued++;
        };
#pragma unroll
        for (int u = 0; u < kPer; u++) { qs[u] = 0; ql[u] = -1; }
        // Forward seeds, step 1: everything but the part of a right arm that leaves the lane's windows.  (After a confirmed forward
        // seed the lane's next sample on the diagonal is never a first one, so at most ONE sample of a lane has such an arm.)
        int32_t f_l[kPer], f_maxr[kPer], f_rt[kPer], f_rep[kPer];
        int pend = -1;               // the sample whose right arm is equal as far as the windows reach, and may go on
#pragma unroll
        for (int u = 0; u < kPer; u++) {
            f_l[u] = 0; f_maxr[u] = 0; f_rt[u] = 0; f_rep[u] = 0;
            if (todo[u] == kNone) continue;
            const int64_t j = j0 + (int64_t)u * stride;
            if (todo[u] & kProbe) {
                // a leader's first sample whose probe did not give one unrepeated, confirmed position holds a slot (a chain, the
                // other strand, another K-mer with the same fingerprint) or nothing; any other sample asks the presence filter --
                // a K-mer it does not know is not in the index -- and only what passes goes on to the slot table
                bool go;
                if (follow && u == 0 && sub == 0) go = slot != kEmpty;
                else {
                    const uint64_t hv = hash_tag(ctag_at(u));
                    const uint32_t bit = (uint32_t)(hv >> 35) & ri.fmask;
                    const uint32_t fm = filter_mask(hv, bit);
                    go = (filter[ri.fbase + (bit >> 5)] & fm) == fm;
                }
                if (go) hand_over(s0 + u, -1);
                continue;
            }
            const int32_t l = base + u * stride;
            if (todo[u] & kFwd) {
                f_l[u] = l;
                f_rep[u] = rep[ri.posbase + l - left[u]];          // in flight while the right arm is settled
                const int64_t mr = m - j - K; const int32_t rr = ri.nR - l - K;
                f_maxr[u] = (int32_t)(mr < rr ? mr : rr);
                const int32_t reach = 64 - u * stride - K;          // bases after the K-mer that the windows hold
                f_rt[u] = right[u];
                if (right[u] >= reach && f_maxr[u] > reach) { f_rt[u] = reach; pend = u; }
            }
            if (todo[u] & kRev) hand_over(s0 + u, l);
        }
        // Step 2: the arms that go on.  The windows of the lanes of a wavefront that lie on ONE diagonal tile the unit's stretch of
        // the query, 20 t ... 20 t + 64 for lane t at stride 10 -- so where a lane's arm ends is something the lanes after it have
        // already compared: a segmented suffix minimum over the lanes (segments = runs of lanes on the same diagonal; a lane reports
        // the first difference of ITS kPer * stride bases, the last lane of a run that of all its 64) gives every lane the first
        // difference at or after the start of its successor's stretch, six shuffle steps for all arms of the wavefront.  An arm
        // that outruns its run (the unit ends, an indel) is finished from memory by the WHOLE wavefront, a lane per 32 bases with
        // coalesced loads -- instead of 64 bases per round and lane (every round six scattered loads for all 64 lanes, ~40 of
        // the kernel's 56 vector memory instructions per wavefront: rounds 3-5).
        int32_t more = 0;            // equal bases after the windows' end
#if defined(__HIP_DEVICE_COMPILE__)
        if (follow) {
            constexpr int32_t kOpenEnd = 1 << 20, kNoDiff = 0x7fffffff;
            const int32_t own = kPer * stride;
            const int32_t pj = lane * own;                                     // the lane's j0, from the unit's first sample
            const int32_t dg = any_fast ? base - (int32_t)j0 : (int32_t)0x80000000 + lane;      // (no windows: a run of its own)
            const int32_t dg_next = __shfl_down(dg, 1, 64);
            const bool run_end = lane == 63 || dg_next != dg;
            int32_t v = first_diff < (run_end ? 64 : own) ? pj + first_diff : run_end ? kOpenEnd + pj + 64 : kNoDiff;
            int closed = run_end ? 1 : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t ov = __shfl_down(v, d, 64); const int oc = __shfl_down(closed, d, 64);
                if (!closed && lane + d < 64) { if (ov < v) v = ov; closed = oc; }
            }
            // everything from the sample's K-mer to the end of the lane's windows is equal, so the first difference at or after the
            // successor's stretch is the arm's end
            int32_t nxt = __shfl_down(v, 1, 64);
            if (run_end) nxt = kOpenEnd + pj + 64;
            int32_t cont = -1, capleft = 0;      // where (query offset in the piece) an arm goes on from memory, and how far at most
            if (pend >= 0) {
                const int32_t room = f_maxr[pend] - f_rt[pend];                // > 0
                if (nxt < kOpenEnd) more = nxt - (pj + 64);
                else {
                    more = nxt - kOpenEnd - (pj + 64);                         // equal as far as the run's windows reach
                    if (more < room) { cont = (int32_t)(j0 - pj) + (nxt - kOpenEnd); capleft = room - more; }
                }
            }
            unsigned long long need = __ballot(cont >= 0);
            while (need) {
                const int src = __ffsll((long long)need) - 1;
                need &= need - 1;
                const int32_t c0 = __builtin_amdgcn_readlane(cont, src), dgs = __builtin_amdgcn_readlane(dg, src), cap = __builtin_amdgcn_readlane(capleft, src);
                int32_t n = 0;
                for (;;) {
                    const int32_t o = n + lane * 32;
                    int c = 32;
                    if (o < cap) {
                        Win wa, wb;
                        window(P.blk, qbase + c0 + o, &wa.b, &wa.m);
                        window(P.blk, rbase + c0 + dgs + o, &wb.b, &wb.m);
                        c = match_fwd(wa, wb);
                    }
                    const unsigned long long stop = __ballot(c < 32);
                    if (stop) { const int f = __ffsll((long long)stop) - 1; n += f * 32 + __builtin_amdgcn_readlane(c, f); break; }
                    n += 64 * 32;
                    if (n >= cap) break;
                }
                if (lane == src) more += n;
            }
        }
#else
        if (pend >= 0) {
            const int64_t j = j0 + (int64_t)pend * stride;
            const int32_t reach = 64 - pend * stride - K;
            more = lce_fwd64(P, qbase + j + K + reach, rbase + f_l[pend] + K + reach, f_maxr[pend] - reach);
        }
#endif
        // Step 3: the events
#pragma unroll
        for (int u = 0; u < kPer; u++) {
            if (!(todo[u] & kFwd)) continue;
            const int64_t j = j0 + (int64_t)u * stride;
            const int32_t lf = left[u];
            int32_t rt = f_rt[u] + (u == pend ? more : 0);
            if (rt > f_maxr[u]) rt = f_maxr[u];
            const int32_t len = lf + K + rt;
            if (len >= ri.minlen && len > f_rep[u]) emit(0, f_l[u] - lf, j - lf, len);      // len <= rep': not unique in R
        }
        {
            // (a sub-queue per workgroup, as for the events: one counter for the launch's ~10^6 wavefronts is a 20 ms queue)
            uint64_t qat = kPer == 2 ? wave_reserve2(queue_count + slice * kSliceStride, (uint32_t)nqueued) : wave_reserve(queue_count + slice * kSliceStride, (uint32_t)nqueued);
            RestItem* const sub = queue + slice * queue_cap;
#pragma unroll
            for (int u = 0; u < kPer; u++)
                if (u < nqueued) {
                    if (qat < queue_cap) sub[qat] = RestItem{(int32_t)unit, qs[u], ql[u], 0};
                    qat++;
                }
        }
        uint64_t at = kPer == 1 ? wave_reserve01(ev_count, nb != 0) : kPer == 2 ? wave_reserve2(ev_count, (uint32_t)nb) : wave_reserve(ev_count, (uint32_t)nb);
#pragma unroll
        for (int u = 0; u < kPer; u++)
            if (u < nb) { if (at < ev_cap) { key_out[at] = bk[u]; val_out[at] = bv[u]; } at++; }
    }
};

// tid = queued sample (kernels.h: RestItem).  Everything here comes straight from memory: the K-mer, the slot walk, both arms.
struct SeedRest {
    // the queue: kSlices sub-queues of queue_cap items, each with its own counter (one 64-byte line apart, like the event buffers)
    Packed P; const RegionInfo* R; const UnitRec* units; const RestItem* queue; const uint64_t* queue_count; uint64_t queue_cap;
    const uint64_t* slots; const uint32_t* filter; const int32_t* next; const int32_t* rep;
    uint64_t* ev_key; uint64_t* ev_val; uint64_t* ev_counters; uint64_t slice_cap; int lbits; uint32_t* err; int64_t budget;
    PM_HD void operator()(int64_t tid) const {
        const uint64_t sub = (uint64_t)tid / queue_cap, idx = (uint64_t)tid % queue_cap;
        const uint64_t have = queue_count[sub * kSliceStride];
        const bool live = idx < (have < queue_cap ? have : queue_cap);
        const RestItem it = live ? queue[tid] : RestItem{0, 0, -1, 0};
        const UnitRec rec = units[it.unit];
        const int32_t pair = rec.pair;
        const RegionInfo& ri = R[rec.region];
        const int64_t m = rec.m;
        const int64_t qbase = rec.qbase;
        const int64_t rbase = P.goff[0] + ri.ref_pos;
        const int K = ri.K;
        const int32_t stride = ri.stride;
        int64_t work = 0;
        const uint64_t slice = (uint64_t)((tid >> 8) & (kSlices - 1));      // one sub-buffer per workgroup
        uint64_t* ev_count = ev_counters + slice * kSliceStride;
        const uint64_t ev_cap = slice_cap;
        uint64_t* const key_out = ev_key + slice * slice_cap;
        uint64_t* const val_out = ev_val + slice * slice_cap;
        uint64_t bk = kEmpty, bv = 0;      // the lane's first event waits for the wavefront's one reservation
        auto emit = [&](int strand, int32_t l0, int64_t j0, int32_t len) {
            const uint64_t ek = ((((uint64_t)pair << lbits) | (uint64_t)l0) << 1) | (uint64_t)strand;
            const uint64_t evv = ((uint64_t)j0 << 32) | (uint32_t)len;
            if (bk == kEmpty) { bk = ek; bv = evv; }
            else { const uint64_t at = atomic_add64(ev_count, 1); if (at < ev_cap) { key_out[at] = ek; val_out[at] = evv; } }
        };
        // a seed on the reverse strand: the sampled bases are the K-mer at jr = m - K - j of the mirrored piece, whose left arm
        // runs where the forward strand's right arm would (inversions and spurious hits only: arms straight from memory)
        auto reverse_seed = [&](int64_t j, int32_t l) {
            const int64_t jr = m - K - j;
            const int64_t qpr = rec.qbase_r + jr;
            int32_t lim = (int32_t)(jr < l ? jr : l);
            if (lim > stride) lim = stride;
            const int32_t left = lce_bwd(P, qpr, rbase + l, lim);
            if (left >= stride) return;
            const int32_t rep_l0 = rep[ri.posbase + l - left];
            const int32_t rr = ri.nR - l - K;
            const int32_t maxr = (int32_t)(j < rr ? j : rr);               // m - jr - K = j bases follow the K-mer on the mirrored piece
            const int32_t right = lce_fwd(P, qpr + K, rbase + l + K, maxr);
            const int32_t len = left + K + right;
            if (len >= ri.minlen && len > rep_l0) emit(1, l - left, jr - left, len);
        };
        // a seed on the forward strand with both arms from memory
        auto forward_seed = [&](int64_t j, int32_t l) {
            int32_t lim = (int32_t)(j < l ? j : l);
            if (lim > stride) lim = stride;
            const int32_t left = lce_bwd(P, qbase + j, rbase + l, lim);     // only `stride` bases matter: a longer left arm means an earlier sample owns the match
            if (left >= stride) return;
            const int32_t rep_l0 = rep[ri.posbase + l - left];
            const int64_t mr = m - j - K; const int32_t rr = ri.nR - l - K;
            const int32_t maxr = (int32_t)(mr < rr ? mr : rr);
            const int32_t right = lce_fwd64(P, qbase + j + K, rbase + l + K, maxr);
            const int32_t len = left + K + right;
            if (len >= ri.minlen && len > rep_l0) emit(0, l - left, j - left, len);      // len <= rep': not unique in R
        };
        // every reference position of a slot's chain against one sample: the entries share the canonical K-mer, not the orientation
        auto walk = [&](int64_t j, uint64_t tag, uint64_t gat, uint64_t slot, int32_t skip, uint64_t head_rt) {
            const bool multi = (slot & kMulti) != 0;
            const int32_t head = slot_head(slot);
            for (int32_t l = head; l >= 0; l = multi ? next[ri.posbase + l] : -1) {
                if (++work > budget) { atomic_or32(err, kErrWork); return; }
                if (l == skip) continue;                    // (already handled from the registers)
                const uint64_t rt = l == head ? head_rt : kmer_tag(P, rbase + l, K);      // (the head's K-mer was read to confirm the slot)
                if (rt == tag) forward_seed(j, l);
                if (rt == gat) reverse_seed(j, l);
            }
        };
        // a sample without a usable prediction asks the index itself
        auto probe_sample = [&](int64_t j, uint64_t tag, uint64_t gat) {
            const uint64_t ctag = gat < tag ? gat : tag;
            uint64_t slot = index_probe(ri, slots, filter, ctag);
            if (slot == kEmpty) return;
            uint64_t rt = kmer_tag(P, rbase + slot_head(slot), K);
            if (rt != tag && rt != gat) {                   // another K-mer with the same 32-bit fingerprint (2^-32): the confirmed lookup
                slot = index_lookup(P, ri, slots, filter, ctag);
                if (slot == kEmpty) return;
                rt = kmer_tag(P, rbase + slot_head(slot), K);
            }
            walk(j, tag, gat, slot, -1, rt);
        };

        if (live) {
            const int64_t j = (int64_t)it.sample * stride;
            if (it.l >= 0) reverse_seed(j, it.l);
            else {
                const uint64_t tg = kmer_tag(P, qbase + j, K);
                probe_sample(j, tg, rc_tag(tg, K));
            }
        }
        const uint64_t at = wave_reserve01(ev_count, bk != kEmpty);
        if (bk != kEmpty && at < ev_cap) { key_out[at] = bk; val_out[at] = bv; }
    }
};

// ------------------------------------------------------------------------------------------ small pairs
PM_HD uint64_t even_bits(uint64_t x) {      // bits 0,2,4,... of x packed into the low 32 bits
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return x;
}
// 64 bases starting at global base position p as three one-bit-per-base planes (low base bit, high base bit, N)
PM_HD void planes64(const SeqBlock* blk, int64_t p, uint64_t* b0, uint64_t* b1, uint64_t* nm) {
    const SeqBlock* b = blk + (p >> 5);
    const int sh = (int)(p & 31);
    const SeqBlock x0 = b[0], x1 = b[1], x2 = b[2];
    const uint64_t lo = (x0.b2 >> (2 * sh)) | ((x1.b2 << 1) << (63 - 2 * sh));
    const uint64_t hi = (x1.b2 >> (2 * sh)) | ((x2.b2 << 1) << (63 - 2 * sh));
    *b0 = even_bits(lo) | (even_bits(hi) << 32);
    *b1 = even_bits(lo >> 1) | (even_bits(hi >> 1) << 32);
    const uint64_t m01 = (uint64_t)x0.nm | ((uint64_t)x1.nm << 32);
    *nm = (m01 >> sh) | (((uint64_t)x2.nm << 1) << (63 - sh));
}
// 64- or 128-base bit planes with the few operations the diagonal scan needs
struct W64 {
    uint64_t v;
    static PM_HD W64 load(const SeqBlock* blk, int64_t p, int plane) {
        uint64_t b0, b1, nm; planes64(blk, p, &b0, &b1, &nm);
        W64 w; w.v = plane == 0 ? b0 : plane == 1 ? b1 : nm; return w;
    }
    static PM_HD W64 ones(int n) { W64 w; w.v = n >= 64 ? ~0ull : ((1ull << n) - 1); return w; }
    PM_HD W64 shr(int k) const { W64 w; w.v = v >> k; return w; }
    PM_HD W64 operator^(const W64& o) const { W64 w; w.v = v ^ o.v; return w; }
    PM_HD W64 operator|(const W64& o) const { W64 w; w.v = v | o.v; return w; }
    PM_HD W64 operator&(const W64& o) const { W64 w; w.v = v & o.v; return w; }
    PM_HD W64 operator~() const { W64 w; w.v = ~v; return w; }
    PM_HD bool any() const { return v != 0; }
    PM_HD int low() const { return ctz64(v); }                      // index of the lowest one (any() must hold)
    PM_HD int run_at(int s) const { const uint64_t t = ~(v >> s); return t ? ctz64(t) : 64 - s; }   // length of the run of ones starting at s
    PM_HD W64 clear_below(int k) const { W64 w; w.v = k >= 64 ? 0 : v & (~0ull << k); return w; }
    static constexpr int kBits = 64;
};
struct W128 {
    uint64_t lo, hi;
    static PM_HD W128 load(const SeqBlock* blk, int64_t p, int plane) {
        uint64_t a0, a1, an, c0, c1, cn; planes64(blk, p, &a0, &a1, &an); planes64(blk, p + 64, &c0, &c1, &cn);
        W128 w; w.lo = plane == 0 ? a0 : plane == 1 ? a1 : an; w.hi = plane == 0 ? c0 : plane == 1 ? c1 : cn; return w;
    }
    static PM_HD W128 ones(int n) {
        W128 w;
        w.lo = n >= 64 ? ~0ull : ((1ull << n) - 1);
        w.hi = n >= 128 ? ~0ull : n > 64 ? ((1ull << (n - 64)) - 1) : 0;
        return w;
    }
    PM_HD W128 shr(int k) const {
        W128 w;
        if (k >= 64) { w.lo = k >= 128 ? 0 : hi >> (k - 64); w.hi = 0; }
        else { w.lo = (lo >> k) | ((hi << 1) << (63 - k)); w.hi = hi >> k; }
        return w;
    }
    PM_HD W128 operator^(const W128& o) const { W128 w; w.lo = lo ^ o.lo; w.hi = hi ^ o.hi; return w; }
    PM_HD W128 operator|(const W128& o) const { W128 w; w.lo = lo | o.lo; w.hi = hi | o.hi; return w; }
    PM_HD W128 operator&(const W128& o) const { W128 w; w.lo = lo & o.lo; w.hi = hi & o.hi; return w; }
    PM_HD W128 operator~() const { W128 w; w.lo = ~lo; w.hi = ~hi; return w; }
    PM_HD bool any() const { return (lo | hi) != 0; }
    PM_HD int low() const { return lo ? ctz64(lo) : 64 + ctz64(hi); }
    PM_HD int run_at(int s) const { const W128 t = ~shr(s); return t.any() ? t.low() : 128 - s; }
    PM_HD W128 clear_below(int k) const {
        W128 w;
        if (k >= 128) { w.lo = 0; w.hi = 0; }
        else if (k >= 64) { w.lo = 0; w.hi = hi & (~0ull << (k - 64)); }
        else { w.lo = lo & (~0ull << k); w.hi = hi; }
        return w;
    }
    static constexpr int kBits = 128;
};

// The same events as SeedExtend -- every maximal exact match of length >= minlen between the query piece and the
// reference substring that is unique in R (len > rep') -- for pairs whose sides both fit 128 bases, found without the
// index: both sides sit in registers as bit planes, every diagonal is one shift + XOR, and a run of equal bases is a run
// of ones.  tid = (pair, strand); one lane does all diagonals of its pair (a few thousand bit operations), so the 3 million
// such pairs of a recursion batch are 50 000 wavefronts instead of 3 million nearly empty SeedExtend units.
// every maximal exact match of length >= L between reference bases [rpos, rpos + nR) and query bases [qpos, qpos + m), both
// sides at most W::kBits long: emit(l0, j0, len).  Both sides as three bit planes; a diagonal is one shift + XOR.
// (first, step: this caller's share of the diagonals -- every step-th, from the first-th -- when several lanes split a pair)
template <class W, class Emit>
PM_HD void pair_diagonals(const Packed& P, int64_t rpos, int64_t qpos, int32_t nR, int32_t m, int32_t L, Emit emit, int32_t first = 0, int32_t step = 1) {
    const W r0 = W::load(P.blk, rpos, 0), r1 = W::load(P.blk, rpos, 1), rn = W::load(P.blk, rpos, 2);
    const W q0 = W::load(P.blk, qpos, 0), q1 = W::load(P.blk, qpos, 1), qn = W::load(P.blk, qpos, 2);
    const W vr = W::ones(nR), vq = W::ones(m);
    for (int32_t d = -(m - L) + first; d <= nR - L; d += step) {        // diagonal: reference position = query position + d
        W eq;
        if (d >= 0) eq = ~((r0.shr(d) ^ q0) | (r1.shr(d) ^ q1) | (rn.shr(d) ^ qn)) & vr.shr(d) & vq;      // bit t: query t, reference t + d
        else eq = ~((r0 ^ q0.shr(-d)) | (r1 ^ q1.shr(-d)) | (rn ^ qn.shr(-d))) & vr & vq.shr(-d);         // bit t: reference t, query t - d
        W x = eq;                                            // any run of at least L ones?
        int k = 1;
        while (2 * k <= L) { x = x & x.shr(k); k *= 2; }
        if (L > k) x = x & x.shr(L - k);
        if (!x.any()) continue;
        while (eq.any()) {
            const int s = eq.low();
            const int len = eq.run_at(s);
            if (len >= L) emit(d >= 0 ? s + d : s, d >= 0 ? s : s - d, len);      // (l0, j0, len)
            eq = eq.clear_below(s + len);
        }
    }
}
struct SmallPairEvents {
    Packed P; const RegionInfo* R; const int64_t* starts; const int64_t* lens; int32_t ngen;
    const int32_t* rep;
    uint64_t* ev_key; uint64_t* ev_val; uint64_t* ev_counters; uint64_t slice_cap; int lbits;
    int32_t g_first, g_last;
    const uint8_t* grouped;      // [region] != 0: the region's pairs are GroupedPairEvents' (store_kernels.h); nullptr: none

    template <class W, class Emit>
    PM_HD void scan(int64_t rpos, int64_t qpos, int32_t nR, int32_t m, int32_t L, Emit emit) const { pair_diagonals<W>(P, rpos, qpos, nR, m, L, emit); }

    PM_HD void operator()(int64_t tid) const {
        const int64_t pair = tid >> 1; const int strand = (int)(tid & 1);
        const int32_t nq = ngen - 1;
        const int64_t r = pair / nq; const int g = (int)(pair % nq) + 1;
        const RegionInfo& ri = R[r];
        const bool mine = !(grouped && grouped[r]);       // (the lanes of a grouped region leave without touching its rows)
        const int64_t m = mine ? lens[r * ngen + g] : 0, qs = mine ? starts[r * ngen + g] : 0;
        const int32_t nR = ri.nR, L = ri.minlen;
        const uint64_t slice = (uint64_t)((tid >> 8) & (kSlices - 1));      // one sub-buffer per workgroup, as in SeedExtend
        uint64_t* ev_count = ev_counters + slice * kSliceStride;
        uint64_t* const key_out = ev_key + slice * slice_cap;
        uint64_t* const val_out = ev_val + slice * slice_cap;
        uint64_t first_key = kEmpty, first_val = 0;
        if (mine && small_pair(nR, m) && m >= L && nR >= L && g >= g_first && g < g_last) {
            const int64_t qbase = strand ? P.goff[2 * g + 1] + (P.glen[g] - qs - m) : P.goff[2 * g] + qs;
            const int64_t rbase = P.goff[0] + ri.ref_pos;
            auto emit = [&](int32_t l0, int32_t j0, int32_t len) {
                if (len <= rep[ri.posbase + l0]) return;             // not unique in R
                const uint64_t ek = ((((uint64_t)pair << lbits) | (uint64_t)l0) << 1) | (uint64_t)strand;
                const uint64_t evv = ((uint64_t)j0 << 32) | (uint32_t)len;
                if (first_key == kEmpty) { first_key = ek; first_val = evv; }
                else { const uint64_t at = atomic_add64(ev_count, 1); if (at < slice_cap) { key_out[at] = ek; val_out[at] = evv; } }
            };
            if (nR <= 64 && m <= 64) scan<W64>(rbase, qbase, nR, (int32_t)m, L, emit);
            else scan<W128>(rbase, qbase, nR, (int32_t)m, L, emit);
        }
        const uint64_t at = wave_reserve01(ev_count, first_key != kEmpty);
        if (first_key != kEmpty && at < slice_cap) { key_out[at] = first_key; val_out[at] = first_val; }
    }
};

// exclusive prefix of the kSlices event counters (one wavefront; no host round trip between the event search and the gather)
struct SliceOffsets {
    const uint64_t* counters; int64_t* off;     // off[kSlices + 1]
    PM_HD void wave(int64_t) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = (int)__lane_id();
        constexpr int per = kSlices / 64;
        int64_t c[per]; int64_t sum = 0;
        for (int i = 0; i < per; i++) { c[i] = (int64_t)counters[(size_t)(lane * per + i) * kSliceStride]; sum += c[i]; }
        int64_t run = wave_incl_sum64(sum) - sum;
        for (int i = 0; i < per; i++) { off[lane * per + i] = run; run += c[i]; }
        if (lane == 63) off[kSlices] = run;
#else
        int64_t run = 0;
        for (int i = 0; i < kSlices; i++) { off[i] = run; run += (int64_t)counters[(size_t)i * kSliceStride]; }
        off[kSlices] = run;
#endif
    }
};
// gather the kSlices sub-buffers into one contiguous array.  tid = output event; off[kSlices+1] = prefix of the counts
struct CompactEvents {
    const uint64_t* key_in; const uint64_t* val_in; const int64_t* off; uint64_t slice_cap; uint64_t* key_out; uint64_t* val_out;
    PM_HD void operator()(int64_t tid) const {
        int64_t sl = upper_slot(off, kSlices, tid);
        int64_t src = sl * (int64_t)slice_cap + (tid - off[sl]);
        key_out[tid] = key_in[src]; val_out[tid] = val_in[src];
    }
};

// ------------------------------------------------------------------------------------------ the events in order, without a sort
// The readers of the events want a pair's events contiguous and in reference order, and a table of how many of them lie before
// every 256-position block (coarse[]).  A radix sort of 64-bit keys delivered the first (four passes over 16 bytes per event,
// 0.55 ms for the anchor call's 7.7 M events at 200 x 5 Mb) and a pass over the sorted keys the second.  But the table IS a
// counting sort's histogram: a pair's events of one block -- two on average, bounded by what 256 reference positions can start --
// form a BUCKET, the buckets numbered pair by pair, block by block.  So (round 6):
//   EventBucketCount  one atomic add per event on its bucket's counter (the counters of a unit's events are neighbours);
//   exclusive scan    over the 3.9 M counters: where every bucket begins in the final array;
//   EventPlace        every event to its bucket (a second atomic gives it a place there), 16 bytes written once;
//   EventOrder        a thread per bucket puts its few events in key order (insertion; a long bucket -- a degenerate repeat -- by
//                     Shell's gaps);
//   CoarseFromBuckets the block-major table of the readers and the pairs' bounds, from the scanned counters.
// Bucket numbering: region r has blocks_r = cbase[r + 1] - cbase[r] - 1 blocks; its pairs' buckets begin at nq * (cbase[r] - r),
// genome g's at + g * blocks_r.
// tid = pair: its first bucket (the divisions once per pair instead of twice per event)
struct PairBucketBase {
    int32_t nq; const int64_t* cbase; int64_t* pbase;
    PM_HD void operator()(int64_t pair) const {
        const int64_t r = pair / nq; const int64_t g = pair % nq;
        pbase[pair] = (int64_t)nq * (cbase[r] - r) + g * (cbase[r + 1] - cbase[r] - 1);
    }
};
PM_HD int64_t bucket_of_key(uint64_t k, int lbits, const int64_t* pbase) {
    return pbase[k >> (lbits + 1)] + (int64_t)(((k >> 1) & ((1ull << lbits) - 1)) >> 8);
}
// tid = (slice, index in the slice): slice = tid >> shift, over 2^shift >= the fullest slice's count (>= 64: a wavefront reads one
// slice).  Neighbouring entries of a slice are the events of one wavefront of the search in lane order -- runs of one bucket -- so
// the first lane of a run adds for all of it (a third of the atomics; the host emulation adds one by one).
struct EventBucketCount {
    const uint64_t* key_in; const uint64_t* counters; uint64_t slice_cap; int shift; int lbits; const int64_t* pbase; int64_t* count;
    PM_HD void operator()(int64_t tid) const {
        const uint64_t sl = (uint64_t)tid >> shift, idx = (uint64_t)tid & ((1ull << shift) - 1);
        const bool active = idx < counters[sl * kSliceStride];
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = (int)__lane_id();
        const int64_t b = active ? bucket_of_key(key_in[sl * slice_cap + idx], lbits, pbase) : -1;
        const int64_t prev = ((int64_t)__shfl_up((int)(b >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)b, 1, 64);
        const bool head = active && (lane == 0 || prev != b);
        const unsigned long long heads = __ballot(head), act = __ballot(active);
        if (!head) return;
        const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
        const int upto = above ? __ffsll((long long)above) - 1 : __popcll(act);
        atomic_add64((uint64_t*)&count[b], (uint64_t)(upto - lane));
#else
        if (active) atomic_add64((uint64_t*)&count[bucket_of_key(key_in[sl * slice_cap + idx], lbits, pbase)], 1);
#endif
    }
};
struct EventPlace {
    const uint64_t* key_in; const uint64_t* val_in; const uint64_t* counters; uint64_t slice_cap; int shift; int lbits; const int64_t* pbase;
    const int64_t* begin; int64_t* count; uint64_t* key_out; uint64_t* val_out;      // count: what EventBucketCount left (taken down to 0 here)
    PM_HD void operator()(int64_t tid) const {
        const uint64_t sl = (uint64_t)tid >> shift, idx = (uint64_t)tid & ((1ull << shift) - 1);
        const bool active = idx < counters[sl * kSliceStride];
        const uint64_t src = sl * slice_cap + idx;
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = (int)__lane_id();
        const uint64_t k = active ? key_in[src] : 0;
        const int64_t b = active ? bucket_of_key(k, lbits, pbase) : -1;
        const int64_t prev = ((int64_t)__shfl_up((int)(b >> 32), 1, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)b, 1, 64);
        const bool head = active && (lane == 0 || prev != b);
        const unsigned long long heads = __ballot(head), act = __ballot(active);
        int64_t first = 0;      // the run's first place (the places of a bucket are handed out from its end)
        if (head) {
            const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
            const int upto = above ? __ffsll((long long)above) - 1 : __popcll(act);
            const int64_t run = upto - lane;
            first = begin[b] + (int64_t)atomic_add64((uint64_t*)&count[b], (uint64_t)(-run)) - run;
        }
        const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1));      // the heads at or before this lane
        const int mine = below ? 63 - __clzll((long long)below) : 0;
        first = ((int64_t)__shfl((int)(first >> 32), mine, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)first, mine, 64);
        if (!active) return;
        const int64_t at = first + (lane - mine);
        key_out[at] = k; val_out[at] = val_in[src];
#else
        if (!active) return;
        const uint64_t k = key_in[src];
        const int64_t b = bucket_of_key(k, lbits, pbase);
        const int64_t at = begin[b] + (int64_t)atomic_add64((uint64_t*)&count[b], ~0ull) - 1;      // (the last free place of the bucket)
        key_out[at] = k; val_out[at] = val_in[src];
#endif
    }
};
// tid = bucket
struct EventOrder {
    const int64_t* begin; int64_t nbuckets; uint64_t* key; uint64_t* val;
    PM_HD void operator()(int64_t b) const {
        const int64_t a = begin[b], n = begin[b + 1] - a;
        if (n < 2) return;
        uint64_t* k = key + a; uint64_t* v = val + a;
        if (n == 2) {
            const uint64_t k0 = k[0], k1 = k[1];
            if (k1 < k0) { const uint64_t v0 = v[0], v1 = v[1]; k[0] = k1; k[1] = k0; v[0] = v1; v[1] = v0; }
            return;
        }
        if (n <= 8) {      // in registers: the bucket's events once in, a fixed network of 19 exchanges, once out
            uint64_t rk[8], rv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { rk[i] = i < n ? k[i] : ~0ull; rv[i] = i < n ? v[i] : 0; }
#define PM_CX(a, b) { const bool sw = rk[b] < rk[a]; const uint64_t ka = sw ? rk[b] : rk[a], kb = sw ? rk[a] : rk[b], va = sw ? rv[b] : rv[a], vb = sw ? rv[a] : rv[b]; rk[a] = ka; rk[b] = kb; rv[a] = va; rv[b] = vb; }
            PM_CX(0, 1) PM_CX(2, 3) PM_CX(4, 5) PM_CX(6, 7)
            PM_CX(0, 2) PM_CX(1, 3) PM_CX(4, 6) PM_CX(5, 7)
            PM_CX(1, 2) PM_CX(5, 6) PM_CX(0, 4) PM_CX(3, 7)
            PM_CX(1, 5) PM_CX(2, 6)
            PM_CX(1, 4) PM_CX(3, 6)
            PM_CX(2, 4) PM_CX(3, 5)
            PM_CX(3, 4)
#undef PM_CX
#pragma unroll
            for (int i = 0; i < 8; i++) if (i < n) { k[i] = rk[i]; v[i] = rv[i]; }
            return;
        }
        int64_t gap = 1;
        while (gap < n / 3) gap = 3 * gap + 1;
        for (; gap >= 1; gap /= 3)
            for (int64_t i = gap; i < n; i++) {
                const uint64_t kk = k[i], vv = v[i];
                int64_t j = i;
                while (j >= gap && k[j - gap] > kk) { k[j] = k[j - gap]; v[j] = v[j - gap]; j -= gap; }
                k[j] = kk; v[j] = vv;
            }
    }
};
// tid = entry of the block-major table: (row of the batch, genome)
struct CoarseFromBuckets {
    const int64_t* begin; const int64_t* cbase; int64_t nregions; int32_t nq; int64_t base; int32_t* coarse; int64_t* lo; int64_t npairs; int64_t total;
    PM_HD void operator()(int64_t tid) const {
        const int64_t row = tid / nq; const int64_t g = tid % nq;
        int64_t a = 0, z = nregions;
        while (z - a > 1) { const int64_t mid = (a + z) >> 1; if (cbase[mid] <= row) a = mid; else z = mid; }
        const int64_t r = a, b = row - cbase[r], blocks = cbase[r + 1] - cbase[r] - 1;
        const int64_t first = (int64_t)nq * (cbase[r] - r) + g * blocks;
        const int64_t f = begin[first];
        coarse[tid] = (int32_t)(begin[first + b] - f);      // (row `blocks`: the next pair's first bucket = the pair's count)
        if (b == 0) lo[r * nq + g] = base + f;
        if (tid == 0) lo[npairs] = base + total;
    }
};

// ------------------------------------------------------------------------------------------ per-pair scan
// After sorting by (pair, l, strand): first event of every pair
struct PairBounds {
    const uint64_t* key; int64_t nev; int lbits; int64_t npairs; int64_t* lo;   // lo[npairs+1]
    int64_t base;      // the sorted events are key[base, nev) (before them: the grouped events, store_kernels.h)
    PM_HD void operator()(int64_t pair) const {
        uint64_t want = (uint64_t)pair << (lbits + 1);
        int64_t a = base, b = nev;
        while (a < b) { int64_t mid = (a + b) >> 1; if (key[mid] < want) a = mid + 1; else b = mid; }
        lo[pair] = a;
        if (pair == npairs - 1) lo[npairs] = nev;
    }
};
// what the scan leaves per event for the readers (the fold, the MUMi coverage): per strand, over the pair's events up to this
// one, EP = e1, UP = max(l_w + rep[l_w], e2) and SP - k = j_w - l_w of the winner -- resolved here, once per event, instead of
// three dependent loads (winner's key, value, repeat length) per strand, candidate and genome in the fold.  e1 == 0: no event.
struct StrandAtK { int32_t e1, up, spb; };
struct EventAtK { StrandAtK s[2]; };
// ------------------------------------------------------------------------------------------ the scan, a wavefront at a time
// Test_UM (mum.c:27-45) + the forward carry of Intersect_UM (mum.c:125-175) in closed form (SURVEY 3.3-3,-4): per strand, over a
// pair's events with l <= k,
//   e1 = furthest end, w = the event that reaches it (first in (l, j) order on equal ends), e2 = second furthest end;
//   EP[k] = e1, UP[k] = max(l_w + rep'[l_w], e2), SP[k] = j_w + (k - l_w).
// Adding an event to that state is an associative fold -- join(state, the state of the one event) -- so the running state of
// every event is a SEGMENTED INCLUSIVE SCAN over the sorted events, segments = pairs: one lane per event, 64 consecutive events
// per round (coalesced loads, 1.5 KB of states stored per instruction), six shuffle steps per round, the last lane's state
// carried into the next round.  The state carries its winner's (l, j, rep') along, so the ties of equal reach are settled
// without a load and the resolved record (EP, UP, SP - k) falls out of the registers.  A wavefront takes kWaveEvents consecutive
// events; pass 1 leaves the state of every wavefront's trailing pair, pass 2 joins the summaries back to the wavefront its
// first pair starts in (`lo`) and scans.  (Until round 4 a THREAD per 128 consecutive events: 64 lanes reading and writing 64
// places 3 KB apart with every instruction, every step of a thread's loop a trip to memory, and a look back over up to 300
// chunk summaries -- 0.55 ms for the anchor call's 7.7 M events against 0.28.)
#ifndef PM_WAVE_EVENTS
#define PM_WAVE_EVENTS 512
#endif
constexpr int kWaveEvents = PM_WAVE_EVENTS;
struct WinState { int32_t e1, e2, wl, wj, wr; };        // furthest end (0: no event), second furthest, the winner's l, j and rep'[l]
struct PairState { WinState s[2]; };
PM_HD WinState win_join(const WinState& a, const WinState& b) {      // a: earlier events, b: later events of the same pair and strand
    if (a.e1 == 0) return b;
    if (b.e1 == 0) return a;
    const bool a_wins = a.e1 > b.e1 || (a.e1 == b.e1 && (a.wl < b.wl || (a.wl == b.wl && a.wj <= b.wj)));      // equal reach: the earlier (l, j)
    WinState o;
    if (a_wins) { o = a; o.e2 = a.e2 > b.e1 ? a.e2 : b.e1; }
    else { o = b; o.e2 = b.e2 > a.e1 ? b.e2 : a.e1; }
    return o;
}
PM_HD PairState pair_join(const PairState& a, const PairState& b) { PairState o; o.s[0] = win_join(a.s[0], b.s[0]); o.s[1] = win_join(a.s[1], b.s[1]); return o; }
PM_HD PairState pair_of_event(uint64_t k, uint64_t v, uint64_t lmask, int32_t rp) {
    const int32_t l = (int32_t)((k >> 1) & lmask), j = (int32_t)(v >> 32);
    PairState o; o.s[0] = WinState{0, 0, 0, 0, 0}; o.s[1] = o.s[0];
    const WinState w{l + (int32_t)(v & 0xffffffffu), 0, l, j, rp};
    if (k & 1) o.s[1] = w; else o.s[0] = w;
    return o;
}
PM_HD EventAtK resolved(const PairState& p) {
    EventAtK o;
    for (int sd = 0; sd < 2; sd++) {
        const WinState& w = p.s[sd];
        const int32_t up = w.wl + w.wr;
        o.s[sd] = w.e1 ? StrandAtK{w.e1, w.e2 > up ? w.e2 : up, w.wj - w.wl} : StrandAtK{0, 0, 0};
    }
    return o;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline WinState win_shfl_up(const WinState& x, int d) { return WinState{__shfl_up(x.e1, d, 64), __shfl_up(x.e2, d, 64), __shfl_up(x.wl, d, 64), __shfl_up(x.wj, d, 64), __shfl_up(x.wr, d, 64)}; }
__device__ inline WinState win_bcast(const WinState& x, int lane) { return WinState{__shfl(x.e1, lane, 64), __shfl(x.e2, lane, 64), __shfl(x.wl, lane, 64), __shfl(x.wj, lane, 64), __shfl(x.wr, lane, 64)}; }
// segmented inclusive scan over the lanes: `head` = this lane's event opens a pair; returns whether a pair opened at or before this lane
__device__ inline bool pair_scan(PairState& x, bool head) {
    const int lane = (int)__lane_id();
    int h = head ? 1 : 0;
    for (int d = 1; d < 64; d <<= 1) {
        PairState y; y.s[0] = win_shfl_up(x.s[0], d); y.s[1] = win_shfl_up(x.s[1], d);
        const int hy = __shfl_up(h, d, 64);
        if (lane >= d && !h) { x = pair_join(y, x); h = hy; }
    }
    return h != 0;
}
#endif
// what one wavefront does with its kWaveEvents events: `carry` = the state of its first pair before them (nothing if the
// first event opens a pair); store = false: only the trailing pair's state comes back
struct WaveScanCore {
    const uint64_t* key; const uint64_t* val; int64_t nev; int lbits; const RegionInfo* R; int32_t nq; const int32_t* rep;
    PM_HD PairState run(int64_t w, PairState carry, bool store, EventAtK* st, int32_t* emax, bool* opened) const {
        const uint64_t lmask = (1ull << lbits) - 1;
        const int64_t a = w * kWaveEvents, b = a + kWaveEvents < nev ? a + kWaveEvents : nev;
        bool any_head = false;
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = (int)__lane_id();
        for (int64_t base = a; base < b; base += 64) {
            const int64_t i = base + lane;
            const bool in = i < b;
            PairState x; x.s[0] = WinState{0, 0, 0, 0, 0}; x.s[1] = x.s[0];
            bool head = false;
            if (in) {
                const uint64_t k = key[i], v = val[i];
                const uint64_t pair = k >> (lbits + 1);
                head = i == 0 || (key[i - 1] >> (lbits + 1)) != pair;
                x = pair_of_event(k, v, lmask, store ? rep[R[(int64_t)pair / nq].posbase + (int32_t)((k >> 1) & lmask)] : 0);
            }
            const bool h = pair_scan(x, head);
            if (!h) x = pair_join(carry, x);                       // no pair opened at or before this lane: the carried pair goes on
            if (in && store) { st[i] = resolved(x); emax[i] = x.s[0].e1 > x.s[1].e1 ? x.s[0].e1 : x.s[1].e1; }
            const int last = (int)((b - base < 64 ? b - base : 64) - 1);
            carry.s[0] = win_bcast(x.s[0], last); carry.s[1] = win_bcast(x.s[1], last);
            any_head = any_head || __ballot(head) != 0;
        }
#else
        for (int64_t i = a; i < b; i++) {
            const uint64_t k = key[i], v = val[i];
            const uint64_t pair = k >> (lbits + 1);
            const bool head = i == 0 || (key[i - 1] >> (lbits + 1)) != pair;
            const PairState x = pair_of_event(k, v, lmask, store ? rep[R[(int64_t)pair / nq].posbase + (int32_t)((k >> 1) & lmask)] : 0);
            carry = head ? x : pair_join(carry, x);
            any_head = any_head || head;
            if (store) { st[i] = resolved(carry); emax[i] = carry.s[0].e1 > carry.s[1].e1 ? carry.s[0].e1 : carry.s[1].e1; }
        }
#endif
        if (opened) *opened = any_head;
        return carry;
    }
};
// pass 1, one wavefront per kWaveEvents events: the state of its trailing pair over its own events
struct WaveSummary {
    WaveScanCore core; PairState* summary;
    PM_HD void wave(int64_t w) const {
        PairState none; none.s[0] = WinState{0, 0, 0, 0, 0}; none.s[1] = none.s[0];
        PairState t = core.run(w, none, false, nullptr, nullptr, nullptr);
        if (wave_leader_k()) {      // (pass 1 runs without the repeat lengths -- they decide nothing -- and fetches the two winners' at the end)
            const int64_t b = (w + 1) * kWaveEvents < core.nev ? (w + 1) * kWaveEvents : core.nev;
            const int64_t posbase = core.R[(int64_t)(core.key[b - 1] >> (core.lbits + 1)) / core.nq].posbase;
            if (t.s[0].e1) t.s[0].wr = core.rep[posbase + t.s[0].wl];
            if (t.s[1].e1) t.s[1].wr = core.rep[posbase + t.s[1].wl];
            summary[w] = t;
        }
    }
};
// pass 2: the carried state of the wavefront's first pair from the summaries of the wavefronts since that pair began, then the scan
struct WaveScan {
    WaveScanCore core; const PairState* summary; const int64_t* lo; int64_t lo_base; EventAtK* st; int32_t* emax;
    PM_HD void wave(int64_t w) const {
        PairState carry; carry.s[0] = WinState{0, 0, 0, 0, 0}; carry.s[1] = carry.s[0];
        const int64_t a = w * kWaveEvents;
        const uint64_t pair = core.key[a] >> (core.lbits + 1);
        if (a > 0 && (core.key[a - 1] >> (core.lbits + 1)) == pair) {
            const int64_t s0 = (lo[pair] - lo_base) / kWaveEvents;      // (the same in every lane: a loop of scalar loads)
            for (int64_t p = s0; p < w; p++) carry = pair_join(carry, summary[p]);
        }
        (void)core.run(w, carry, true, st, emax, nullptr);
    }
};

// ------------------------------------------------------------------------------------------ MUMi coverage
// calcmumi mode (Aligner::setMumi, src/parsnp.cpp:1977-2063): each query genome ALONE against the reference chunk.
// Master = that genome's strand-merged (UP,EP); a position is accepted when EP exceeds the EP of the last accepted
// position, UP < EP and EP-k < len (:2044); accepted matches of length >= 15 mark [k, EP) (:2050-2056); the result is
// the number of marked positions.  (UP,EP) only change where an event starts, so the scan runs over events.
// tid = pair (region 0, query genome).
struct MumiCoverage {
    const RegionInfo* R; const uint64_t* key; const uint64_t* val; const int64_t* lo; const EventAtK* st; const int32_t* rep;
    int lbits; int32_t ngen; int64_t* covered;
    PM_HD void operator()(int64_t pair) const {
        const RegionInfo& ri = R[pair / (ngen - 1)];
        const uint64_t lmask = (1ull << lbits) - 1;
        int32_t last_ep = 0, cov_end = 0; int64_t cov = 0;
        const int64_t a = lo[pair], b = lo[pair + 1];
        for (int64_t i = a; i < b; i++) {
            int32_t l = (int32_t)((key[i] >> 1) & lmask);
            int32_t lnext = i + 1 < b ? (int32_t)((key[i + 1] >> 1) & lmask) : ri.nR;
            if (lnext == l) continue;                    // the state after the last event starting at l is the one at k = l
            const EventAtK& s = st[i];
            const int32_t ep[2] = {s.s[0].e1, s.s[1].e1}, up[2] = {s.s[0].up, s.s[1].up};      // (no event on a strand: 0, 0)
            int c = ep[0] > ep[1] ? 0 : 1;               // Merge_Master: forward only if strictly better
            int32_t EP = ep[c], UP = up[c];
            if (!(EP > last_ep && UP < EP)) continue;
            int32_t k = l;
            if (!(EP - k < ri.nR)) { k = l + 1; if (k >= lnext || k >= ri.nR) continue; }   // EP-k < len fails only for a full-length match at k = 0
            last_ep = EP;
            if (EP - k >= 15) {
                int32_t from = k > cov_end ? k : cov_end;
                if (EP > from) cov += EP - from;
                if (EP > cov_end) cov_end = EP;
            }
        }
        covered[pair] = cov;
    }
};

// ------------------------------------------------------------------------------------------ Master.EP
// Master[k].EP = min over query genomes of max(EP_fwd[k], EP_rev[k])  (Intersect_UM's min fold mum.c:163 +
// Merge_Master mum.c:92-123; order independent).
//
// Coarse index over the sorted events of every (region, genome) pair: coarse[.., b, g] = number of the pair's events
// with l < b * 256.  The entries of region r are [cbase[r]*nq, cbase[r+1]*nq), BLOCK-major (per block the nq genomes
// side by side: a wavefront with its lanes over the genomes reads them coalesced), blocks_r + 1 rows.
// Written by streaming over the sorted events (tid = event): event i of a pair closes every block row between its
// predecessor's block and its own; the pair's last event closes the remaining rows with the pair's event count.  Rows of
// pairs without events stay at the 0 the table was cleared to.
constexpr int kCoarseShift = 8;
constexpr int kChunkPos = 1 << kCoarseShift;     // reference positions per MasterEP wavefront = one coarse block
struct CoarseFill {
    const uint64_t* key; int64_t nev; int lbits; const int64_t* lo; const RegionInfo* R; const int64_t* cbase; int32_t nq; int32_t* coarse;
    PM_HD void operator()(int64_t i) const {
        const uint64_t lmask = (1ull << lbits) - 1;
        const uint64_t k = key[i];
        const int64_t pair = (int64_t)(k >> (lbits + 1));
        const int64_t r = pair / nq; const int g = (int)(pair % nq);
        const int64_t rows = cbase[r + 1] - cbase[r];              // blocks + 1
        int32_t* col = coarse + cbase[r] * nq + g;                  // row b of this pair: col[b * nq]
        const int64_t rank = i - lo[pair];
        const int64_t b = (int64_t)((k >> 1) & lmask) >> kCoarseShift;
        int64_t bprev = -1;                                         // first event of the pair: rows 0..b are 0 already
        if (rank > 0) bprev = (int64_t)((key[i - 1] >> 1) & lmask) >> kCoarseShift;
        else bprev = b;
        for (int64_t x = bprev + 1; x <= b; x++) col[x * nq] = (int32_t)rank;
        if (i + 1 == nev || (int64_t)(key[i + 1] >> (lbits + 1)) != pair)
            for (int64_t x = b + 1; x < rows; x++) col[x * nq] = (int32_t)(rank + 1);
    }
};
// chunk (= wavefront) w of the batch -> its region: chunks of region r are [cbase[r] - r, cbase[r+1] - (r+1))
PM_HD int64_t region_of_chunk(const int64_t* cbase, int64_t nregions, int64_t w) {
    int64_t lo = 0, hi = nregions;
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (cbase[mid] - mid <= w) lo = mid; else hi = mid; }
    return lo;
}
// One wavefront per chunk of 256 reference positions (lane = 4 consecutive positions).  The events of the chunk are
// staged in LDS, 64 genomes at a time (lanes over the genomes: coarse rows, pair bounds and carry-in values are
// coalesced loads; every event of the batch is read from HBM exactly once); then all lanes walk one genome's entries
// together (LDS broadcast reads) and keep, per position, the value of the last entry that starts at or before it
// -- the genome's EP there -- folding it into the running minimum.  Replaces a thread per 16 positions that walked the
// genomes serially through ~6 dependent global loads each (1.6 ms for the 5 Mb anchor call).
constexpr int kEpCap = 2048;                      // staged entries per round (16 KB of LDS)
struct MasterEP {
    const RegionInfo* R; int64_t nregions;
    int32_t ngen; const uint64_t* key; const int64_t* lo; const int32_t* emax; int lbits; int32_t* epm;
    const int64_t* cbase; const int32_t* coarse;
    int32_t g_first, g_last;   // sharded run: the min over the other genomes arrives by all-reduce
    const uint8_t* grouped;    // [region] != 0: GroupedPairEvents (store_kernels.h) has written the region's Master.EP; nullptr: none
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
#if defined(__HIP_DEVICE_COMPILE__)
        __shared__ uint64_t s_ent[kEpCap];
        __shared__ int32_t s_off[65];
        __shared__ int32_t s_vin[64];
        const int lane = (int)__lane_id();
        const int32_t p0 = lane * 4;               // first of this lane's positions, relative to k0
        int32_t ep[4] = {ri.nR, ri.nR, ri.nR, ri.nR};
        for (int g0 = ga; g0 < gb; g0 += 64) {
            const int g = g0 + lane;
            const bool act = g < gb;
            int64_t a = 0; int32_t cnt = 0, vin = 0;
            if (act) {
                const int64_t first = lo[r * nq + g];
                a = first + row[g]; cnt = row[nq + g] - row[g];
                vin = a > first ? emax[a - 1] : 0;
            }
            const int32_t incl = wave_incl_sum(cnt);
            const int32_t total = __shfl(incl, 63, 64);
            const int here = gb - g0 < 64 ? gb - g0 : 64;
            if (total <= kEpCap) {
                s_off[lane] = incl - cnt; s_vin[lane] = vin;
                if (lane == 63) s_off[64] = total;
                for (int32_t i = 0; i < cnt; i++) {
                    const int32_t l = (int32_t)((key[a + i] >> 1) & lmask) - k0;
                    s_ent[incl - cnt + i] = ((uint64_t)(uint32_t)l << 32) | (uint32_t)emax[a + i];
                }
                __syncthreads();
                for (int gg = 0; gg < here; gg++) {
                    const int32_t o0 = s_off[gg], o1 = s_off[gg + 1];
                    int32_t v0 = s_vin[gg], v1 = v0, v2 = v0, v3 = v0;
                    for (int32_t i = o0; i < o1; i++) {
                        const uint64_t e = s_ent[i];
                        const int32_t l = (int32_t)(e >> 32), v = (int32_t)(uint32_t)e;
                        if (l <= p0) v0 = v;
                        if (l <= p0 + 1) v1 = v;
                        if (l <= p0 + 2) v2 = v;
                        if (l <= p0 + 3) v3 = v;
                    }
                    ep[0] = v0 < ep[0] ? v0 : ep[0]; ep[1] = v1 < ep[1] ? v1 : ep[1];
                    ep[2] = v2 < ep[2] ? v2 : ep[2]; ep[3] = v3 < ep[3] ? v3 : ep[3];
                }
                __syncthreads();
            } else {
                // more events than the staging area holds (degenerate repeats): the same walk straight from global memory
                for (int gg = 0; gg < here; gg++) {
                    const int64_t aa = ((int64_t)__shfl((int)(a >> 32), gg, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)a, gg, 64);
                    const int32_t cc = __shfl(cnt, gg, 64);
                    int32_t v0 = __shfl(vin, gg, 64), v1 = v0, v2 = v0, v3 = v0;
                    for (int32_t i = 0; i < cc; i++) {
                        const int32_t l = (int32_t)((key[aa + i] >> 1) & lmask) - k0, v = emax[aa + i];
                        if (l <= p0) v0 = v;
                        if (l <= p0 + 1) v1 = v;
                        if (l <= p0 + 2) v2 = v;
                        if (l <= p0 + 3) v3 = v;
                    }
                    ep[0] = v0 < ep[0] ? v0 : ep[0]; ep[1] = v1 < ep[1] ? v1 : ep[1];
                    ep[2] = v2 < ep[2] ? v2 : ep[2]; ep[3] = v3 < ep[3] ? v3 : ep[3];
                }
            }
        }
        for (int t = 0; t < 4; t++) if (k0 + p0 + t < ri.nR) epm[ri.posbase + k0 + p0 + t] = ep[t];
#else
        for (int32_t k = k0; k < k0 + kChunkPos && k < ri.nR; k++) {
            int32_t ep = ri.nR;
            for (int g = ga; g < gb; g++) {
                const int64_t first = lo[r * nq + g];
                const int64_t a = first + row[g], e = first + row[nq + g];
                int32_t v = a > first ? emax[a - 1] : 0;
                for (int64_t i = a; i < e; i++) if ((int32_t)((key[i] >> 1) & lmask) <= k) v = emax[i];
                if (v < ep) ep = v;
            }
            epm[ri.posbase + k] = ep;
        }
#endif
    }
};

// Candidate positions: Master[k].EP > Master[k-1].EP and EP-k >= minsize (parsnp.cpp:1657-1663; the UP < EP part
// of the test is applied in the fold).  Two passes without atomics, output in (region, k) order -- what the sort
// after the old one-counter version produced: CandMark (tid = flat reference position) leaves one 64-bit hit mask and one
// count per wavefront, an exclusive scan turns the counts into offsets, CandWrite places the hits.
PM_HD bool cand_hit(const RegionInfo& ri, const int32_t* epm, int32_t l) {
    const int32_t e = epm[ri.posbase + l];
    const int32_t prev = l > 0 ? epm[ri.posbase + l - 1] : 0;
    return e > prev && e - l >= ri.minsize;
}
struct CandMark {
    const RegionInfo* R; int64_t nregions; const int64_t* posbase; int64_t npos; const int32_t* epm;
    uint64_t* wmask; int64_t* wcount;      // per 64 positions
    PM_HD void wave(int64_t w) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int64_t tid = w * 64 + (int64_t)__lane_id();
        bool hit = false;
        if (tid < npos) { const RegionInfo& ri = R[upper_slot(posbase, nregions, tid)]; hit = cand_hit(ri, epm, (int32_t)(tid - ri.posbase)); }
        const unsigned long long m = __ballot(hit);
        if (__lane_id() == 0) { wmask[w] = m; wcount[w] = __popcll(m); }
#else
        uint64_t m = 0;
        for (int t = 0; t < 64 && w * 64 + t < npos; t++) {
            const int64_t tid = w * 64 + t;
            const RegionInfo& ri = R[upper_slot(posbase, nregions, tid)];
            if (cand_hit(ri, epm, (int32_t)(tid - ri.posbase))) m |= 1ull << t;
        }
        wmask[w] = m; wcount[w] = __builtin_popcountll(m);
#endif
    }
};
struct CandWrite {
    const RegionInfo* R; int64_t nregions; const int64_t* posbase; const uint64_t* wmask; const int64_t* woff;
    uint64_t* cand; uint64_t cand_cap;
    PM_HD void operator()(int64_t tid) const {
        const uint64_t m = wmask[tid >> 6];
        const int t = (int)(tid & 63);
        if (!((m >> t) & 1)) return;
        const uint64_t at = (uint64_t)woff[tid >> 6] + (uint64_t)
#if defined(__HIP_DEVICE_COMPILE__)
            __popcll(m & ((1ull << t) - 1));
#else
            __builtin_popcountll(m & ((1ull << t) - 1));
#endif
        if (at >= cand_cap) return;
        const int64_t r = upper_slot(posbase, nregions, tid);
        cand[at] = ((uint64_t)r << 32) | (uint32_t)(tid - R[r].posbase);
    }
};

// ------------------------------------------------------------------------------------------ per-candidate fold
struct GenomeAtK { int32_t epf, upf, spf, epr, upr, spr; };
// both strands of query genome g (0-based) at candidate (r, k): the propagated Pair[k] / SP[k] of Find_UM + Intersect_UM
PM_HD GenomeAtK state_at(const RegionInfo& ri, int64_t r, int32_t k, int g, int32_t nq, const uint64_t* key, const uint64_t* val,
                         const int64_t* lo, const EventAtK* st, const int32_t* rep, int lbits, const int64_t* cbase, const int32_t* coarse) {
    const int64_t pair = r * nq + g;
    const int32_t* row = coarse + cbase[r] * nq + (int64_t)(k >> kCoarseShift) * nq;
    const int64_t first = lo[pair];
    int64_t a = first + row[g], b = first + row[nq + g];
    const uint64_t want = ((((uint64_t)pair << lbits) | (uint64_t)k) << 1) | 1ull;
    while (a < b) { const int64_t mid = (a + b) >> 1; if (key[mid] <= want) a = mid + 1; else b = mid; }
    GenomeAtK o{0, 0, 0, 0, 0, 0};
    if (a > first) {
        const EventAtK s = st[a - 1];
        if (s.s[0].e1) { o.epf = s.s[0].e1; o.upf = s.s[0].up; o.spf = s.s[0].spb + k; }
        if (s.s[1].e1) { o.epr = s.s[1].e1; o.upr = s.s[1].up; o.spr = s.s[1].spb + k; }
    }
    return o;
}
// tid = cand * (ngen-1) + g.  Only a sharded run launches this (the columns of the rank's genomes are exchanged before
// the fold); an unsharded run computes the states inside FoldCandidates.
struct StateAtCandidate {
    const RegionInfo* R; const uint64_t* cand; int32_t ngen; const uint64_t* key; const uint64_t* val; const int64_t* lo;
    const EventAtK* st; const int32_t* rep; int lbits; GenomeAtK* out;
    const int64_t* cbase; const int32_t* coarse;
    PM_HD void operator()(int64_t tid) const {
        const int32_t nq = ngen - 1;
        const int64_t c = tid / nq; const int g = (int)(tid % nq);
        const int64_t r = (int64_t)(cand[c] >> 32); const int32_t k = (int32_t)(cand[c] & 0xffffffffu);
        out[tid] = state_at(R[r], r, k, g, nq, key, val, lo, st, rep, lbits, cbase, coarse);
    }
};
// Sharded run with device collectives (RCCL): the (EP,UP,SP) columns of this rank's genome block, packed [candidate][widest]
// for the all-gather, and the gathered blocks of all ranks spread back into the [candidate][genome] table.
// tid = (candidate, x): x-th genome of the block.  Block r of `world` covers query genomes [q r / world, q (r+1) / world).
struct PackStates {
    const GenomeAtK* at; int32_t nq; int32_t g0; int32_t mine; int32_t widest; GenomeAtK* send;     // g0: first query genome (0-based) of this rank
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = tid / widest; const int x = (int)(tid % widest);
        send[tid] = x < mine ? at[c * nq + g0 + x] : GenomeAtK{0, 0, 0, 0, 0, 0};
    }
};
struct UnpackStates {
    const GenomeAtK* recv; int32_t nq; int32_t world; int32_t widest; int64_t ncand; GenomeAtK* at;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = tid / nq; const int g = (int)(tid % nq);
        int r = (int)(((int64_t)(g + 1) * world - 1) / nq);           // the block that holds query genome g
        while (r > 0 && (int)((int64_t)nq * r / world) > g) r--;
        while (r + 1 < world && (int)((int64_t)nq * (r + 1) / world) <= g) r++;
        const int first = (int)((int64_t)nq * r / world);
        at[tid] = recv[(int64_t)r * ncand * widest + c * widest + (g - first)];
    }
};

// One wavefront per candidate, lanes over the query genomes.  Intersect_UM's (UP=max, EP=min) fold and Merge_Master's
// strand choice run genome after genome in ini order (mum.c:162-163, :92-123; ties -> reverse):
//     fe = min(em, epf), re = min(em, epr);  forward iff fe > re;  em = max(fe, re) = min(em, max(epf, epr));  um = max(um, up of the chosen strand)
// so em before genome g is the exclusive prefix minimum of max(epf, epr) -- one shuffle scan per 64 genomes -- and the
// strand flags and um follow from it independently per lane.  Then the UP < EP test of parsnp.cpp:1657.
// `at` == nullptr: the per-genome states are computed here (never written to memory); else read from `at`.
// Output per candidate and genome: the start of the MUM inside the genome's request window on the chosen strand (sp) and
// the strand, as before; and out_ok / k / lon per candidate.
struct FoldCandidates {
    const RegionInfo* R; const uint64_t* cand; int32_t ngen; const GenomeAtK* at;
    const uint64_t* key; const uint64_t* val; const int64_t* lo; const EventAtK* st; const int32_t* rep; int lbits;
    const int64_t* cbase; const int32_t* coarse;
    int32_t* out_k; int32_t* out_lon; int32_t* out_sp; uint8_t* out_fwd; uint8_t* out_ok;
    int64_t ncand;      // the launch is xcd_grid(ncand) wavefronts; ncand: the candidate CAPACITY of the call ...
    const int64_t* count;      // ... and the number there are (device memory: the call does not wait for it)
    PM_HD void wave(int64_t w) const {
        // neighbouring candidates read neighbouring events of every genome: they go to the SAME L2 (xcd_item)
        const int64_t nc = *count < ncand ? *count : ncand;
        if (w / kXcds >= (nc + kXcds - 1) / kXcds) return;      // (a launch over the capacity: past the live count's last round)
        const int64_t c = xcd_item(w, nc);
        if (c >= nc) return;
        const int32_t nq = ngen - 1;
        const int64_t r = (int64_t)(cand[c] >> 32); const int32_t k = (int32_t)(cand[c] & 0xffffffffu);
        const RegionInfo& ri = R[r];
        int32_t em = ri.nR, um = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = (int)__lane_id();
        // (fetching the states of 4 x 64 genomes stage by stage before folding the first was measured: 0.42 instead of 0.35 ms for
        // the anchor call -- the launch is bound by its scattered requests, not by one wavefront's chain of them)
        for (int g0 = 0; g0 < nq; g0 += 64) {
            const int g = g0 + lane;
            const bool act = g < nq;
            GenomeAtK s{0, 0, 0, 0, 0, 0};
            if (act) s = at ? at[c * nq + g] : state_at(ri, r, k, g, nq, key, val, lo, st, rep, lbits, cbase, coarse);
            const int32_t m = act ? (s.epf > s.epr ? s.epf : s.epr) : 0x7fffffff;
            const int32_t incl = wave_incl_min(m);
            int32_t before = __shfl_up(incl, 1, 64);
            if (lane == 0 || before > em) before = em;
            const int32_t fe = before < s.epf ? before : s.epf, re = before < s.epr ? before : s.epr;
            const bool fwd = fe > re;
            const int32_t u = act ? (fwd ? s.upf : s.upr) : 0;
            if (act) { out_sp[c * nq + g] = fwd ? s.spf : s.spr; out_fwd[c * nq + g] = fwd ? 1 : 0; }
            const int32_t last = __shfl(incl, 63, 64);
            if (last < em) em = last;
            const int32_t umax = wave_all_max(u);
            if (umax > um) um = umax;
        }
        if (lane == 0) { out_k[c] = k; out_lon[c] = em - k; out_ok[c] = (um < em && em - k >= ri.minsize) ? 1 : 0; }
#else
        for (int g = 0; g < nq; g++) {
            const GenomeAtK s = at ? at[c * nq + g] : state_at(ri, r, k, g, nq, key, val, lo, st, rep, lbits, cbase, coarse);
            const int32_t fe = em < s.epf ? em : s.epf, fu = um > s.upf ? um : s.upf;
            const int32_t re = em < s.epr ? em : s.epr, ru = um > s.upr ? um : s.upr;
            if (fe > re) { em = fe; um = fu; out_sp[c * nq + g] = s.spf; out_fwd[c * nq + g] = 1; }
            else { em = re; um = ru; out_sp[c * nq + g] = s.spr; out_fwd[c * nq + g] = 0; }
        }
        out_k[c] = k; out_lon[c] = em - k;
        out_ok[c] = (um < em && em - k >= ri.minsize) ? 1 : 0;
#endif
    }
};

// accepted candidates only, densely, in candidate order: what the host receives.  pos = exclusive prefix of ok.
struct OkCount {
    const uint8_t* ok; const int64_t* ncand; int64_t* cnt;     // launched over the capacity + 1: zeros past the live count close the scan
    PM_HD void operator()(int64_t c) const { cnt[c] = c < *ncand ? (ok[c] ? 1 : 0) : 0; }
};
// tid = (candidate, query genome): the accepted candidates densely, as (sp, strand) per query genome (the documented
// result of pm_multi_mum_batch; a session switched to MUM rows with pm_session_rows gets CompactCandidates instead)
struct CompactSp {
    const uint64_t* cand; const uint8_t* ok; const int64_t* pos; int32_t nq;
    const int32_t* k; const int32_t* lon; const int32_t* sp; const uint8_t* fwd;
    int32_t* out_region; int32_t* out_k; int32_t* out_lon; int32_t* out_sp; uint8_t* out_fwd;
    const int64_t* ncand;
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = tid / nq; const int g = (int)(tid % nq);
        if (c >= *ncand || !ok[c]) return;
        const int64_t w = pos[c];
        out_sp[w * nq + g] = sp[tid]; out_fwd[w * nq + g] = fwd[tid];
        if (g == 0) { out_region[w] = (int32_t)(cand[c] >> 32); out_k[w] = k[c]; out_lon[w] = lon[c]; }
    }
};
// Flag bits of an accepted candidate as the host's validation wants them (src/parsnp.cpp:1717-1780, TMum.cpp:25-60)
constexpr uint32_t kRowBad = 1;        // a start position outside its region window: the reference skips it before building the TMum (:1723)
constexpr uint32_t kRowOutside = 2;    // the MUM would leave a genome (`notgood`)
constexpr uint32_t kRowReverse = 4;    // some member is on the reverse strand
constexpr uint32_t kRowDirty = 8;      // overlaps an earlier candidate of the list in some genome (cheap running-extent test)
constexpr uint32_t kRowEarly = 16;     // starts, in some genome, before the end of an earlier candidate of the list (same running extents): no
                                       // candidate without this bit lies out of list order -- the host's order test for free
// tid = (candidate, genome column 0..ngen-1).  Writes the accepted candidates densely and AS MUM ROWS: per genome the
// start on the genome's forward coordinates exactly as the TMum constructor derives it -- forward: window start + sp,
// reverse: flipped against the WHOLE genome length (TMum.cpp:33-35) -- and the strand byte; column 0 is the reference.
// The host used to rebuild these rows from (sp, fwd) per candidate (`rows`, 1.6 ms of the anchor validation at 200 x 5 Mb).
struct CompactCandidates {
    const uint64_t* cand; const uint8_t* ok; const int64_t* pos; int32_t ngen;
    const int32_t* k; const int32_t* lon; const int32_t* sp; const uint8_t* fwd;
    const int64_t* starts; const int64_t* lens; const int64_t* glen;
    int32_t* out_region; int32_t* out_k; int32_t* out_lon; int32_t* out_start; uint8_t* out_strand; uint32_t* out_flags;   // out_flags zeroed
    const int64_t* ncand;      // the live candidate count (the launch covers the capacity)
    PM_HD void operator()(int64_t tid) const {
        const int64_t c = tid / ngen; const int j = (int)(tid % ngen);
        if (c >= *ncand || !ok[c]) return;
        const int64_t w = pos[c];
        const int64_t r = (int64_t)(cand[c] >> 32);
        const int64_t lo = lon[c];
        const int64_t ws = starts[r * ngen + j], wl = lens[r * ngen + j];
        int64_t st; uint32_t f, bits = 0;
        if (j == 0) {
            st = (int64_t)k[c] + ws; f = 1;
            if ((uint64_t)k[c] + 1 > (uint64_t)(uint32_t)wl) bits |= kRowBad;
            out_region[w] = (int32_t)r; out_k[w] = k[c]; out_lon[w] = lon[c];
        } else {
            const int64_t q = sp[c * (ngen - 1) + j - 1];
            f = fwd[c * (ngen - 1) + j - 1];
            if ((uint64_t)q + 1 > (uint64_t)(uint32_t)wl) bits |= kRowBad;
            const int64_t at = q + ws;
            st = f ? at : glen[j] - (at + lo);
            if (!f) bits |= kRowReverse;
        }
        if (st + lo > glen[j] || st < 0) bits |= kRowOutside;
        out_start[w * ngen + j] = (int32_t)st; out_strand[w * ngen + j] = (uint8_t)f;
        if (bits) atomic_or32(&out_flags[w], bits);
    }
};

// The cheap overlap test of the anchor validation (host: validate_parallel): walking the accepted candidates in order, one
// that lies entirely after, or entirely before, everything earlier in a genome overlaps nothing earlier there; else it
// is "dirty" and takes the ordered host path.  Three passes over the [candidate][genome] start rows, a wavefront per
// (block of 256 candidates, group of 64 genomes), lane = genome (coalesced 256-byte row pieces):
//   DirtyExtent  extent (max end, min start) of each block per genome
//   DirtyPrefix  exclusive prefix of the block extents per genome (one wavefront per genome group, blocks in order)
//   DirtyMark    the walk itself from the block's carry-in
// Only candidates that can mark anything take part (constructed, inside, length >= 5: src/parsnp.cpp:1781).
constexpr int kDirtyBlock = 256;
PM_HD bool row_marks(uint32_t flags, int32_t lon) { return !(flags & (kRowBad | kRowOutside)) && lon >= 5; }
struct DirtyExtent {
    const int32_t* start; const int32_t* lon; const uint32_t* flags; const int64_t* n_live; int32_t ngen; int32_t* bmax; int32_t* bmin;   // [block][ngen]; *n_live: the rows there are (the launch covers the capacity)
    PM_HD void wave(int64_t w) const {
        const int64_t n = *n_live;
        const int64_t groups = (ngen + 63) / 64, blk = w / groups; const int g0 = (int)(w % groups) * 64;
        const int64_t c0 = blk * kDirtyBlock, c1 = c0 + kDirtyBlock < n ? c0 + kDirtyBlock : n;
#if defined(__HIP_DEVICE_COMPILE__)
        const int j = g0 + (int)__lane_id();
        if (j >= ngen) return;
        int32_t mx = -1, mn = 0x7fffffff;
#pragma unroll 8
        for (int64_t c = c0; c < c1; c++) {
            const int32_t a = start[c * ngen + j], l = lon[c];
            if (!row_marks(flags[c], l)) continue;
            if (a + l > mx) mx = a + l;
            if (a < mn) mn = a;
        }
        bmax[blk * ngen + j] = mx; bmin[blk * ngen + j] = mn;
#else
        for (int j = g0; j < g0 + 64 && j < ngen; j++) {
            int32_t mx = -1, mn = 0x7fffffff;
            for (int64_t c = c0; c < c1; c++) {
                const int32_t a = start[c * ngen + j], l = lon[c];
                if (!row_marks(flags[c], l)) continue;
                if (a + l > mx) mx = a + l;
                if (a < mn) mn = a;
            }
            bmax[blk * ngen + j] = mx; bmin[blk * ngen + j] = mn;
        }
#endif
    }
};
struct DirtyPrefix {
    int64_t nblocks; int32_t ngen; int32_t* bmax; int32_t* bmin;       // in place: extents of the blocks BEFORE each block
    PM_HD void wave(int64_t w) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int j0 = (int)w * 64 + (int)__lane_id(), j1 = j0 + 1;
        if (j0 >= ngen) return;
#else
        const int j0 = (int)w * 64, j1 = j0 + 64 < ngen ? j0 + 64 : ngen;
#endif
        for (int j = j0; j < j1; j++) {
            int32_t mx = -1, mn = 0x7fffffff;
            // sixteen blocks per round: their extents are in flight together (four wavefronts cannot hide a load per block: 71 us
            // for the 236 blocks of a 60 000-row list when every block waited for its own)
            for (int64_t b0 = 0; b0 < nblocks; b0 += 16) {
                int32_t x[16], y[16];
#pragma unroll
                for (int u = 0; u < 16; u++) { const int64_t b = b0 + u < nblocks ? b0 + u : nblocks - 1; x[u] = bmax[b * ngen + j]; y[u] = bmin[b * ngen + j]; }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if (b0 + u >= nblocks) continue;
                    bmax[(b0 + u) * ngen + j] = mx; bmin[(b0 + u) * ngen + j] = mn;
                    if (x[u] > mx) mx = x[u];
                    if (y[u] < mn) mn = y[u];
                }
            }
        }
    }
};
struct DirtyMark {
    const int32_t* start; const int32_t* lon; const int64_t* n_live; int32_t ngen; const int32_t* bmax; const int32_t* bmin;
    const uint32_t* flags; uint32_t* dirty;      // dirty[c] (zeroed): set when candidate c overlaps something earlier; flags is only read
    PM_HD void wave(int64_t w) const {
        const int64_t n = *n_live;
        const int64_t groups = (ngen + 63) / 64, blk = w / groups; const int g0 = (int)(w % groups) * 64;
        const int64_t c0 = blk * kDirtyBlock, c1 = c0 + kDirtyBlock < n ? c0 + kDirtyBlock : n;
#if defined(__HIP_DEVICE_COMPILE__)
        const int j = g0 + (int)__lane_id();
        const bool act = j < ngen;
        const int jj = act ? j : 0;
        int32_t mx = act ? bmax[blk * ngen + j] : -1, mn = act ? bmin[blk * ngen + j] : 0x7fffffff;
        // eight candidates per round: their rows are in flight together (one wavefront per SIMD cannot hide a load per step)
        for (int64_t cb = c0; cb < c1; cb += 8) {
            int32_t a[8], l[8]; uint32_t fl[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t c = cb + u < c1 ? cb + u : c1 - 1;
                a[u] = start[c * ngen + jj]; l[u] = lon[c]; fl[u] = flags[c];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (cb + u >= c1 || !row_marks(fl[u], l[u])) continue;      // uniform over the wavefront
                const int32_t b = a[u] + l[u];
                const bool early = act && !(a[u] >= mx);
                const bool hit = early && !(b <= mn);
                if (act) { if (b > mx) mx = b; if (a[u] < mn) mn = a[u]; }
                const unsigned long long any_early = __ballot(early), any_hit = __ballot(hit);      // (both by the whole wavefront)
                if (any_early && __lane_id() == 0) atomic_or32(&dirty[cb + u], kRowEarly | (any_hit ? kRowDirty : 0u));
            }
        }
#else
        for (int j = g0; j < g0 + 64 && j < ngen; j++) {
            int32_t mx = bmax[blk * ngen + j], mn = bmin[blk * ngen + j];
            for (int64_t c = c0; c < c1; c++) {
                const int32_t l = lon[c];
                if (!row_marks(flags[c], l)) continue;
                const int32_t a = start[c * ngen + j], b = a + l;
                if (!(a >= mx)) dirty[c] |= kRowEarly | (!(b <= mn) ? kRowDirty : 0u);
                if (b > mx) mx = b;
                if (a < mn) mn = a;
            }
        }
#endif
    }
};
// tid = candidate
struct DirtyMerge {
    const uint32_t* dirty; uint32_t* flags; const int64_t* n_live;
    PM_HD void operator()(int64_t c) const { if (c < *n_live && dirty[c]) flags[c] |= dirty[c] & (kRowDirty | kRowEarly); }
};

}  // namespace pm
