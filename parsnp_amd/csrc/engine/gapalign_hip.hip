// gapalign_hip.hip -- the inter-MUM gap aligner on the MI355X: pm_gap_align_batch (include/parsnp_mum.h).
//
// Replaces, for a whole run's worth of gaps at once, what the reference's XMFA writer does gap by gap:
// MuscleInterface::CallMuscleFast (src/MuscleInterface.cpp:37-78, called at src/parsnp.cpp:854-855), i.e. the one
// libMUSCLE 3.7 configuration SEQTYPE_DNA / MaxIters 1 / stable / ClustalW weights.  The arithmetic is the restatement
// of parsnp_amd/csrc/host/gapalign.cpp (pinned there against the reference's own libMUSCLE), stage by stage with the
// same float32/float64 mix, the same order of additions and the same tie rules; this file is compiled with
// -ffp-contract=off so that no multiply-add is fused (the host build has no FMA either).
//
// One alignment (n sequences, up to kMaxCols columns) = one wavefront; a launch runs a fixed number of wavefront
// "slots" that pull jobs from a queue (longest first), so the workspace is per slot, not per job:
//   distances    muscle/libMUSCLE/fastdistnuc.cpp:103-262  distinct strings once; 6-mer multiset intersection through an
//                8-bit count table in LDS (46 656 bytes, the reference's own table size)
//   guide tree   upgma2.cpp:133-355                         lanes over the clusters: (value, index) arg-min reductions
//   weights      clwwt.cpp:65-163                           lane per leaf, path sums in double
//   profiles     profilefrommsa.cpp:262-331                 lane per column, sequences in MSA order (the order of the
//                                                           float additions is part of the result)
//   pairwise DP  nwsmall.cpp:447-620                        lane per row of profile A, anti-diagonal sweep: a cell needs
//                                                           its upper / upper-left neighbours from the lane above
//                                                           (shuffles) and its left neighbour from itself; trace-back
//                                                           bits in LDS, bittraceback.cpp:130-209 by one lane
//   merge        aligngivenpath.cpp:124-366                 rows re-spelled through a column map, lanes over columns
// A job the device declines (more than kMaxSeqs sequences, an alignment wider than its row capacity or kMaxCols, an
// empty sequence, MUSCLE's own "quit" conditions) is reported with cols = -1 and stays with the caller's host path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/parsnp_mum.h"

namespace {

constexpr int kMaxSeqs = 512;      // sequences per alignment on the device
constexpr int kMaxCols = 96;       // columns of any intermediate alignment on the device (LDS per wavefront ~25 KB: 6 alignments in flight per CU)
constexpr int kTable = 46656;      // 6^6 (fastdistnuc.cpp:82)
constexpr float kMinusInf = (float)-1e37;
constexpr float kBigDist = (float)1e29;
constexpr float kGapOpen = -400.0f;
constexpr float kGapExtend = 0.0f;
constexpr float kSueff = (float)0.1;
constexpr unsigned kNone = 0xffffffffu;
enum : uint8_t { kMM = 0, kDM = 1, kIM = 2, kXM = 3, kMD = 4, kMI = 8 };

__constant__ float c_matrix[4][4] = {      // nucmx.cpp:8-25: BLASTZ scores + 2*30
    {91.0f + 60.0f, -114.0f + 60.0f, -31.0f + 60.0f, -123.0f + 60.0f},
    {-114.0f + 60.0f, 100.0f + 60.0f, -125.0f + 60.0f, -31.0f + 60.0f},
    {-31.0f + 60.0f, -125.0f + 60.0f, 100.0f + 60.0f, -114.0f + 60.0f},
    {-123.0f + 60.0f, -31.0f + 60.0f, -114.0f + 60.0f, 91.0f + 60.0f},
};
__constant__ uint8_t c_letter[256];         // alpha.cpp:125-166: 0..3 residues, 4..15 wildcards, 16 gap, 255 none

struct Job { int64_t first_seq; int32_t n; int32_t max_cols; int64_t row_off; };

// per-slot workspace (global memory), sized for the widest job of the launch
struct Slot {
    float* dist;            // n(n-1)/2 + 1
    uint16_t* ucommon;      // u x u
    uint16_t* tcode;        // per distinct string: its distinct 6-mers ...
    uint8_t* tcnt;          // ... and their 8-bit multiplicities
    int32_t* ntup;          // [n]
    uint64_t* hash;         // [n]
    int32_t* len;           // [n]
    int32_t* cls;           // [n] representative (first sequence with the same string)
    int32_t* repidx;        // [n] rank of a representative among the representatives
    int32_t* replist;       // [n]
    uint32_t* left; uint32_t* right; uint32_t* parent;   // [2n]
    double* to_parent;      // [2n]
    float* height;          // [n]
    uint32_t* under;        // [2n]
    double* strength;       // [2n]
    float* weight;          // [n]
    int32_t* lo;            // [2n] first row of a node's alignment
    int32_t* ncols;         // [2n]
    int32_t* perm;          // [n] row -> sequence
    float* total;           // [2n] sequential float sum of the weights of a node's rows, in row order
    uint8_t* table;         // [46656] 6-mer counts of one string (fastdistnuc.cpp:82), all zero between uses
};

struct Params {
    const Job* jobs; int64_t njobs; const int64_t* seq_off; const uint8_t* chars;
    uint8_t* out_rows; int32_t* out_cols; unsigned long long* next; uint8_t* ws; int64_t ws_stride; int32_t nmax, cap;
    volatile int32_t* dbg;     // PM_GAP_DEBUG: per slot (job, stage) in host memory the host can read while the kernel runs
    unsigned long long* prof;  // PM_GAP_DEBUG=3: shader clocks per stage, summed over the jobs
};
constexpr int kProfStages = 16;

// every lane's outstanding loads / stores (global and LDS) have completed before any lane goes on: the lanes of the one
// wavefront of a workgroup talk to each other through global workspace and LDS
#define GA_SYNC() do { __builtin_amdgcn_s_waitcnt(0); __threadfence_block(); __syncthreads(); } while (0)
__device__ inline unsigned tri(unsigned a, unsigned b) { return a >= b ? b + (a * (a - 1)) / 2 : a + (b * (b - 1)) / 2; }
__device__ inline bool is_gap(uint8_t c) { return c == '-' || c == '.'; }
// A row in LDS holds one byte per column: MUSCLE's letter code (alpha.cpp:125-166: 0..3 = ACGT, 4..15 wildcards in the
// order MRWSYKVHDBXN), kRowGap for '-', and two flags a profile needs of a gap column: the row's gap starts here / ends
// here (profilefrommsa.cpp:262-331 asks the neighbouring columns).  Upper-case input without 'U' round-trips through it;
// anything else is declined.
constexpr uint8_t kRowGap = 16, kRowCode = 0x1f, kRowStart = 0x40, kRowEnd = 0x80;
__device__ inline bool row_gap(uint8_t b) { return (b & kRowCode) == kRowGap; }
__device__ inline uint8_t row_char(uint8_t b) { const uint8_t c = b & kRowCode; return c == kRowGap ? (uint8_t)'-' : (uint8_t)"ACGTMRWSYKVHDBXN"[c]; }

// A value that every lane of the wavefront holds alike -- the job number, a length read from one address, the result of a reduction
// -- said so: it moves to a scalar register, and the branches and loop bounds that depend on it become SCALAR branches.  Without this
// the compiler must take every `return false` and every loop exit of a job for divergent (it cannot see that a butterfly sum or an
// LDS word is the same in all lanes), structures the job loop as nested exec-mask loops in which "returned" lanes wait for the
// others, and a build without the stage markers hung in that structure on the device (round 3, reproduced in round 5:
// scripts/gap_nomark.sh; the markers' branches happened to break it up).  Uniform by construction; said so, it is also cheaper.
__device__ inline int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ inline unsigned uni(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ inline float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ inline int wave_sum(int x) { for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64); return uni(x); }
// arg-min over (value, index): strictly smaller value wins, equal values -> the lower index (a sequential scan that
// replaces its best only on `<` keeps the first one it met)
__device__ inline void wave_argmin(float& v, unsigned& i) {
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(v, d, 64); const unsigned oi = (unsigned)__shfl_xor((int)i, d, 64);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    v = uni(v); i = uni(i);
}

// LDS of the one wavefront of a workgroup, used phase after phase (the alignment rows themselves -- n rows of `cap`
// bytes, row p = leaf p in the order of the root alignment -- follow it as dynamic LDS: every profile is a sequential
// sum over its rows, 20 000 row visits for 201 sequences, and from global memory each visit is a round trip)
struct __align__(16) Shared {
  union {           // the guide tree is finished (and synchronised on) before the first profile is built
    struct { float mind[kMaxSeqs]; unsigned nearest[kMaxSeqs]; unsigned node[kMaxSeqs]; float height[kMaxSeqs]; } t;      // tree
    struct {
            float fa[4][kMaxCols]; uint8_t orda[kMaxCols]; float opena[kMaxCols], closea[kMaxCols];   // profile A: sorted counts, their letters
            float sb[4][kMaxCols]; float openb[kMaxCols], closeb[kMaxCols];                           // profile B: scores per letter
            float bD[kMaxCols + 2], bM[kMaxCols + 2], bN[kMaxCols + 2]; uint8_t bX[kMaxCols + 2];     // row handed from one 64-row stripe to the next
            uint8_t path[2 * kMaxCols + 2], rev[2 * kMaxCols + 2];
            int16_t mapa[2 * kMaxCols + 2], mapb[2 * kMaxCols + 2];
        float result[3];
        float acc[6][kMaxCols];      // a column's six sums (four letters, gap opened here, gap closed here) before they are gathered
    } p;
  };
    uint8_t letter[256];       // alpha.cpp:125-166 (c_letter, copied: a table in constant memory costs a trip to memory per lane)
    float total_i[kMaxSeqs];   // per internal node: sequential float sum of the weights of its rows, in row order
    uint8_t ncols_i[kMaxSeqs]; // per internal node: columns of its alignment
    uint8_t rowlen[kMaxSeqs];  // length of the sequence in row p
    uint16_t codes[kMaxCols];
    float wrow[kMaxSeqs];      // weight of the sequence in row p (rows = leaves in the order of the root alignment)
    int32_t flag;
};

// ---- profile of the alignment held by rows [lo, lo+ns) (nc columns) -> either the A arrays or the B arrays
// R: the rows in LDS.  Each of a column's six sums runs over the rows one after the other (float: the order is part of the
// result) and the sums do not meet, so each gets a lane of its own: one masked compare and one add per row, eight rows
// fetched at a time.  A row that does not touch a sum adds +0, which changes nothing (the sums are never negative).
// kWild: some sequence of the job holds a wildcard (N, ...); without one the per-row test for it is not compiled in.
template <bool kWild>
__device__ void build_profile(Shared& S, const uint8_t* R, int cap, int lo, int ns, int nc, float total, bool as_a) {
    const int lane = (int)__lane_id();
    // msa2.cpp:418-431 + msa.cpp:369-381: this alignment's weights, rescaled to sum 1.  `total` is the sequential float sum
    // of the weights in MSA order, kept per node: a merged alignment's rows are A's then B's, so its sum continues A's.
    const float f = total != 0 ? 1.0f / total : 1.0f;
    const bool scale = total != 0;
    const int items = nc * 6;
    for (int i0 = 0; i0 < items; i0 += 64) {
        const int item = i0 + lane;
        const bool on = item < items;
        const int c = on ? item / 6 : 0, kind = on ? item - c * 6 : 0;
        const uint32_t cmask = kind < 4 ? (uint32_t)kRowCode : (kind == 4 ? (uint32_t)kRowStart : (uint32_t)kRowEnd);
        const uint32_t cval = kind < 4 ? (uint32_t)kind : cmask;
        const uint8_t* col = R + (size_t)lo * (size_t)cap + c;
        float acc = 0;
        auto fold = [&](uint32_t b, float ws) {
            if (scale) ws *= f;
            float add = (b & cmask) == cval ? ws : 0.0f;
            if (kWild) {
                const uint32_t code = b & kRowCode;
                if (code >= 4 && code < kRowGap && kind < 4)        // a wildcard: 'X' is half G half A, the others a twentieth of each
                    add = code == 14 ? ((kind == 2 || kind == 0) ? ws / 2 : 0.0f) : ws / 20;
            }
            acc += add;
        };
        int s0 = 0;
        for (; s0 + 8 <= ns; s0 += 8) {
            uint32_t b[8]; float w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { b[k] = col[(size_t)(s0 + k) * (size_t)cap]; w[k] = S.wrow[lo + s0 + k]; }
#pragma unroll
            for (int k = 0; k < 8; k++) fold(b[k], w[k]);
        }
        for (; s0 < ns; s0++) fold(col[(size_t)s0 * (size_t)cap], S.wrow[lo + s0]);
        if (on) S.p.acc[kind][c] = acc;
    }
    GA_SYNC();
    for (int c0 = 0; c0 < nc; c0 += 64) {
        const int c = c0 + lane;
        if (c < nc) {
            float cnt[4] = {S.p.acc[0][c], S.p.acc[1][c], S.p.acc[2][c], S.p.acc[3][c]};
            const float start = S.p.acc[4][c], end = S.p.acc[5][c];
            unsigned order[4] = {0, 1, 2, 3};       // profilefrommsa.cpp:180-204: bubble sort, strict <
            bool any = true;
            for (int pass = 0; any && pass < 8; pass++) {     // at most 3 passes move anything
                any = false;
                for (unsigned k = 0; k < 3; k++) {
                    const unsigned a = order[k], b = order[k + 1];
                    if (cnt[a] < cnt[b]) { order[k + 1] = a; order[k] = b; any = true; }
                }
            }
            const float start_occ = (float)(1.0 - start), end_occ = (float)(1.0 - end);
            const float open = start_occ * kGapOpen / 2, close = end_occ * kGapOpen / 2;
            if (as_a) {
                for (unsigned k = 0; k < 4; k++) S.p.fa[k][c] = cnt[order[k]];
                S.p.orda[c] = (uint8_t)(order[0] | (order[1] << 2) | (order[2] << 4) | (order[3] << 6));
                S.p.opena[c] = open; S.p.closea[c] = close;
            } else {
                for (unsigned i = 0; i < 4; i++) {
                    float sum = 0;
                    for (unsigned j = 0; j < 4; j++) sum += cnt[j] * c_matrix[i][j];
                    S.p.sb[i][c] = sum;
                }
                S.p.openb[c] = open; S.p.closeb[c] = close;
            }
        }
    }
    GA_SYNC();
}

// scorepp.cpp:83-95 with all four terms (a count of 0 ends the reference's loop; its term is +-0 here, the zeros come last)
__device__ inline float match_ab(const Shared& S, const float f[4], uint8_t ord, int j) {
    float sc = 0.0f;
    sc += f[0] * S.p.sb[ord & 3][j];
    sc += f[1] * S.p.sb[(ord >> 2) & 3][j];
    sc += f[2] * S.p.sb[(ord >> 4) & 3][j];
    sc += f[3] * S.p.sb[(ord >> 6) & 3][j];
    return sc - 0.0f;
}

// nwsmall.cpp:447-620 + bittraceback.cpp:130-209 -> S.p.path[0..*plen) in forward order; false = the reference gives up
__device__ bool nw_small(Shared& S, uint8_t* TB, int la, int lb, int* plen, bool prof, unsigned long long& prof_sweep, unsigned long long& prof_t0) {
    const int lane = (int)__lane_id();
    const float e = kGapExtend;
    const int stride = lb + 1;
    if (lane == 0) {          // termgaps.cpp:19-33 (TERMGAPS_Half falls through into _Ext): 0, then *= -1
        S.p.opena[0] = 0.0f * -1.0f; if (la > 1) S.p.closea[la - 1] = 0.0f * -1.0f;
        S.p.openb[0] = 0.0f * -1.0f; if (lb > 1) S.p.closeb[lb - 1] = 0.0f * -1.0f;
    }
    for (int x = lane; x <= la; x += 64) TB[x * stride] = 0;
    for (int x = lane; x <= lb; x += 64) TB[x] = 0;
    GA_SYNC();
    const float open_a0 = S.p.opena[0], open_b0 = S.p.openb[0];
    for (int r0 = 0; r0 < la; r0 += 64) {
        const int i = r0 + lane + 1;                    // this lane's row (1-based)
        const bool row_ok = i <= la;
        float fr[4] = {0, 0, 0, 0}, fn[4] = {0, 0, 0, 0}; uint8_t ordr = 0, ordn = 0;
        float open_a = 0, close_a = 0, close_am2 = 0;
        if (row_ok) {
            for (int k = 0; k < 4; k++) fr[k] = S.p.fa[k][i - 1];
            ordr = S.p.orda[i - 1]; open_a = S.p.opena[i - 1]; close_a = S.p.closea[i - 1];
            if (i >= 2) close_am2 = S.p.closea[i - 2];
            if (i < la) { for (int k = 0; k < 4; k++) fn[k] = S.p.fa[k][i]; ordn = S.p.orda[i]; }
        }
        const int rows_here = la - r0 < 64 ? la - r0 : 64;
        // outputs of this lane's last two steps
        float lastD = kMinusInf, lastM = kMinusInf, lastI = kMinusInf;   // D[i][j-1]: unused; M[i][j-1]; I[i][j-1]
        float q1 = 0, q2 = 0; uint8_t x1 = 0, x2 = 0;                     // M[i+1][j+1] produced one / two steps ago
        float outD = kMinusInf, outM = kMinusInf;                         // D[i][j], M[i][j] of the last step (for the lane below)
        const float closea0 = S.p.closea[0];
        for (int t = 0; t < rows_here + lb - 1; t++) {
            const int j = t - lane + 1;
            const bool act = row_ok && j >= 1 && j <= lb;
            // everything this step reads from the profiles depends on (i, j) alone, not on the cells before it: asked for
            // first (indices clamped for the lanes that sit out), so that these reads and the exchange with the lane above
            // are in flight together -- one wait per step for a wavefront that has the SIMD nearly to itself
            const int jc = j < 1 ? 1 : (j > lb ? lb : j);
            const float ob = S.p.openb[jc - 1], cb = S.p.closeb[jc - 1], cb2 = S.p.closeb[jc >= 2 ? jc - 2 : 0];
            const float mc = match_ab(S, fr, ordr, jc - 1);                       // this cell's own match score (column 1 and row 1 start from it)
            const float mn = match_ab(S, fn, ordn, jc < lb ? jc : lb - 1);        // the match score of the cell below and to the right
            // what the lane above produced: at its last step (up) and two steps ago (diagonal)
            float upD = __shfl_up(outD, 1, 64), upM = __shfl_up(outM, 1, 64), dgM = __shfl_up(q2, 1, 64);
            uint8_t dgX = (uint8_t)__shfl_up((int)x2, 1, 64);
            if (lane == 0 && act) {
                if (r0 == 0) { upD = kMinusInf; upM = kMinusInf; }        // row 0: M[0][j] = D[0][j] = -inf for j >= 1
                else { upD = S.p.bD[j]; upM = S.p.bM[j]; dgM = S.p.bN[j]; dgX = S.p.bX[j]; }
            }
            q2 = q1; x2 = x1;
            if (act) {
                float m; uint8_t xm;
                if (j == 1) {
                    if (i == la) {
                        if (la > 1) m = mc + (la - 2) * e + open_a0 + close_am2;
                        else m = mc + open_a0 + closea0;
                        xm = kDM;
                    } else if (i == 1) { m = mc; xm = kMM; }
                    else { m = mc + open_a0 + (i - 2) * e + close_am2; xm = kDM; }
                } else if (i == 1) {
                    m = mc + open_b0 + (j - 2) * e + cb2; xm = kIM;
                } else { m = dgM; xm = dgX; }
                // REC_D
                const float dd = upD + e, md = upM + open_a;
                const bool from_m = !(dd > md);
                const float D = from_m ? md : dd;
                // REC_I
                float iij = j == 1 ? kMinusInf : lastI;
                iij += e;
                const float mi = (j == 1 ? kMinusInf : lastM) + ob;
                const bool open_i = mi >= iij;
                const float I = open_i ? mi : iij;
                TB[i * stride + j] = (uint8_t)(xm | (from_m ? kMD : 0) | (open_i ? kMI : 0));
                if (i < la && j < lb) {
                    const float dm = D + close_a, im = I + cb, mm = m;
                    const bool pm = mm >= dm && mm >= im;
                    const bool pd = !pm && dm >= mm && dm >= im;
                    float nx = mn;
                    nx += pm ? mm : (pd ? dm : im);
                    q1 = nx; x1 = pm ? kMM : (pd ? kDM : kIM);
                }
                lastM = m; lastI = I; outD = D; outM = m;
                if (lane == 63 && i < la) { S.p.bD[j] = D; S.p.bM[j] = m; if (j < lb) { S.p.bN[j + 1] = q1; S.p.bX[j + 1] = x1; } }
                if (i == la && j == lb) { S.p.result[0] = m; S.p.result[1] = D; S.p.result[2] = I; }
            }
        }
        GA_SYNC();
    }
    GA_SYNC();
    if (prof) { const unsigned long long t_ = (unsigned long long)clock64(); prof_sweep += t_ - prof_t0; prof_t0 = t_; }
    bool ok = true;
    if (lane == 0) {
        const float mab = S.p.result[0], dab = S.p.result[1], iab = S.p.result[2];
        float score = mab; int type = 0;                  // 0 'M', 1 'D', 2 'I': the codes of the trace-back bits (kMM, kDM, kIM)
        if (dab > score) { score = dab; type = 1; }
        if (iab > score) { score = iab; type = 2; }
        int a = la, b = lb, n = 0;
        uint8_t* rev = S.p.rev;        // end to start, as the trace-back meets the cells
        for (;;) {                     // one dependent LDS read per cell: the rest is selects, not branches
            if (n >= 2 * kMaxCols + 2) { ok = false; break; }
            rev[n++] = type == 0 ? (uint8_t)'M' : (type == 1 ? (uint8_t)'D' : (uint8_t)'I');
            const uint32_t bits = TB[a * stride + b];
            const uint32_t x = bits & kXM;
            const int next = type == 0 ? (int)x : (type == 1 ? ((bits & kMD) ? 0 : 1) : ((bits & kMI) ? 0 : 2));
            const int need_a = type != 2, need_b = type != 1;
            if ((type == 0 && x == 3) || (need_a && a == 0) || (need_b && b == 0)) { ok = false; break; }
            a -= need_a; b -= need_b;
            if (a == 0 && b == 0) break;
            type = next;
        }
        S.flag = ok ? n : -1;
    }
    GA_SYNC();
    const int n = uni(S.flag);
    for (int x = lane; x < n; x += 64) S.p.path[x] = S.p.rev[n - 1 - x];
    GA_SYNC();
    *plen = n;
    return n >= 0;
}

// -DPM_GAP_NO_MARKERS: the kernel without its stage markers and stage clocks (what hung once in round 3, on that round's kernel;
// measurement builds only: `make -C parsnp_amd/csrc nomark`, scripts/gap_nomark.sh)
#if defined(PM_GAP_NO_MARKERS) || defined(PM_GAP_NO_STAGES)
constexpr bool kGapStages = false;      // the (job, stage) markers of PM_GAP_DEBUG=1|2
#else
constexpr bool kGapStages = true;
#endif
#if defined(PM_GAP_NO_MARKERS) || defined(PM_GAP_NO_CLOCKS)
constexpr bool kGapClocks = false;      // the stage clocks of PM_GAP_DEBUG=3
#else
constexpr bool kGapClocks = true;
#endif
// (-DPM_GAP_STRIP_STAGE=s: ONE marker compiled out -- s = 1 ... 7 the stages, 100 the three markers of the merge loop, 900 the job
// markers of the kernel's loop: which marker's absence the hang of the marker-free build needs, scripts/gap_nomark.sh)
#if !defined(PM_GAP_STRIP_STAGE)
#define PM_GAP_STRIP_STAGE (-1)
#endif
constexpr bool ga_stage_on(int s) { return kGapStages && !(PM_GAP_STRIP_STAGE == s || (PM_GAP_STRIP_STAGE == 100 && s >= 100 && s < 900)); }
#define GA_STAGE(stage_) do { if (ga_stage_on(stage_) && P.dbg && lane == 0) P.dbg[blockIdx.x * 2 + 1] = (stage_); } while (0)
#define GA_STAGE_DYN(stage_) do { if (P.dbg && lane == 0) P.dbg[blockIdx.x * 2 + 1] = (stage_); } while (0)
// PM_GAP_DEBUG=3: the shader clock spent since the previous mark goes to stage k_
// (kept in registers and added to the launch's totals once per job: a shared counter per mark would be what is measured)
#define GA_CLOCK(k_) do { if (kGapClocks && P.prof) { const unsigned long long t_ = (unsigned long long)clock64(); prof_acc[(k_)] += t_ - prof_t0; prof_t0 = t_; } } while (0)
// NOT inlined into the kernel's job loop, on purpose.  Inlined, hipcc (ROCm 7.2) structures the loop and this function's early
// returns into an exec-mask loop whose exit mask is `threadIdx.x == 0` (the condition of the job fetch and of the result store around
// the call): lane 0 leaves it, lanes 1-63 go round the job body again and never leave -- the kernel hangs on the first batch of more
// than one job.  The (job, stage) markers of PM_GAP_DEBUG happened to break that structure up, which is why only a build WITHOUT them
// hung (round 3; reproduced, bisected to the two job markers of the kernel's loop and read off the ISA in round 5: DESIGN.md 9-6,
// scripts/gap_nomark.sh).  As a call the job is one node of the loop's control flow; -DPM_GAP_INLINE restores the old shape for the
// regression check.  tests/test_gpu_gapalign.py::test_marker_free_build runs the marker-free build.
#if !defined(PM_GAP_INLINE)
__attribute__((noinline))
#endif
__device__ bool align_job(Shared& S, uint8_t* R, uint8_t* TB, const Slot& W, const Params& P, const Job& job, int* out_cols) {
    const int lane = (int)__lane_id();
    const int n = job.n, cap = P.cap;
    unsigned long long prof_acc[kProfStages] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t0 = (kGapClocks && P.prof) ? (unsigned long long)clock64() : 0;
    if (n < 2 || n > kMaxSeqs) return false;
    // ---- sequences: lengths, FixAlpha (seq.cpp:331-344) happens when the rows are filled
    int bad = 0, wild = 0;
    for (int i = lane; i < n; i += 64) {
        const int64_t a = P.seq_off[job.first_seq + i], b = P.seq_off[job.first_seq + i + 1];
        const int L = (int)(b - a);
        W.len[i] = L;
        if (L <= 0 || L > cap || L > kMaxCols) bad = 1;
        uint64_t h = 1469598103934665603ull;
        for (int x = 0; x < L; x++) {
            uint8_t ch = P.chars[a + x];
            if (S.letter[ch] >= 16) ch = 'N';
            else if (ch >= 'a' || ch == 'U') bad = 1;
            if (S.letter[ch] >= 4) wild = 1;            // the row coding holds upper-case letters without 'U' (the caller's host path takes the rest)
            h = (h ^ ch) * 1099511628211ull;
        }
        W.hash[i] = h ^ (uint64_t)L;
    }
    if (wave_sum(bad)) return false;
    const bool any_wild = wave_sum(wild) != 0;
    GA_SYNC();
    auto seq_char = [&](int i, int x) -> uint8_t { uint8_t ch = P.chars[P.seq_off[job.first_seq + i] + x]; return S.letter[ch] >= 16 ? (uint8_t)'N' : ch; };

    GA_STAGE(1); GA_CLOCK(0);
    // ---- distinct strings: cls[i] = first sequence spelling the same string
    for (int i = lane; i < n; i += 64) {
        int rep = i;
        const uint64_t h = W.hash[i]; const int L = W.len[i];
        for (int o = 0; o < i; o++) {
            if (W.hash[o] != h || W.len[o] != L) continue;
            bool same = true;
            for (int x = 0; x < L && same; x++) same = seq_char(o, x) == seq_char(i, x);
            if (same) { rep = o; break; }
        }
        W.cls[i] = rep;
    }
    GA_SYNC();
    int u = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool is_rep = i < n && W.cls[i] == i;
        const unsigned long long m = __ballot(is_rep);
        if (is_rep) { const int r = u + __popcll(m & ((1ull << lane) - 1)); W.repidx[i] = r; W.replist[r] = i; }
        u += __popcll(m);
    }
    GA_SYNC();
    GA_STAGE(2); GA_CLOCK(1);
    // ---- per distinct string: its distinct 6-mers with 8-bit (wrapping) multiplicities (fastdistnuc.cpp:82-90)
    // (the count table is all zero here: the kernel clears it once per slot, and every use below clears what it set)
    for (int a = 0; a < u; a++) {
        const int i = uni(W.replist[a]), L = uni(W.len[i]);
        int nt = 0;
        if (L >= 6) {
            for (int p = lane; p < L; p += 64) {
                uint32_t t = 0;
                if (p >= 5) for (int x = p - 5; x <= p; x++) { uint8_t l = S.letter[seq_char(i, x)]; if (l >= 4) l = 4; t = t * 6 + l; }
                S.codes[p] = (uint16_t)t;
            }
            GA_SYNC();
            for (int p0 = 0; p0 < L; p0 += 64) {
                const int p = p0 + lane;
                bool emit = false; int cnt = 0;
                if (p >= 5 && p < L) {
                    const uint16_t mine = S.codes[p];
                    bool first = true;
                    for (int q = 5; q < L; q++) { if (S.codes[q] == mine) { cnt++; if (q < p) first = false; } }
                    emit = first && (cnt & 255) != 0;
                }
                const unsigned long long m = __ballot(emit);
                if (emit) {
                    const int at = nt + __popcll(m & ((1ull << lane) - 1));
                    W.tcode[(size_t)a * kMaxCols + at] = S.codes[p]; W.tcnt[(size_t)a * kMaxCols + at] = (uint8_t)(cnt & 255);
                }
                nt += __popcll(m);
            }
            GA_SYNC();
        }
        if (lane == 0) W.ntup[a] = nt;
    }
    GA_SYNC();
    for (int a = 0; a < u; a++) {
        const int na = uni(W.ntup[a]);
        for (int t = lane; t < na; t += 64) W.table[W.tcode[(size_t)a * kMaxCols + t]] = W.tcnt[(size_t)a * kMaxCols + t];
        GA_SYNC();
        for (int b = 0; b <= a; b++) {
            const int nb = uni(W.ntup[b]);
            int sum = 0;
            for (int t = lane; t < nb; t += 64) {
                const uint8_t c1 = W.table[W.tcode[(size_t)b * kMaxCols + t]], c2 = W.tcnt[(size_t)b * kMaxCols + t];
                sum += c1 < c2 ? c1 : c2;
            }
            sum = wave_sum(sum);
            if (lane == 0) { W.ucommon[(size_t)a * u + b] = (uint16_t)sum; W.ucommon[(size_t)b * u + a] = (uint16_t)sum; }
        }
        GA_SYNC();
        for (int t = lane; t < na; t += 64) W.table[W.tcode[(size_t)a * kMaxCols + t]] = 0;
        GA_SYNC();
    }
    GA_STAGE(3); GA_CLOCK(2);
    // ---- distances (fastdistnuc.cpp:236-262)
    for (int i = 1; i < n; i++) {
        const int ci = W.repidx[W.cls[i]];
        double c11 = W.ucommon[(size_t)ci * u + ci];
        if (c11 == 0) c11 = 1;
        for (int j = lane; j < i; j += 64) {
            const int cj = W.repidx[W.cls[j]];
            double c22 = W.ucommon[(size_t)cj * u + cj];
            if (c22 == 0) c22 = 1;
            const unsigned c12 = W.ucommon[(size_t)ci * u + cj];
            const double d1 = 3.0 * (c11 - c12) / c11;
            const double d2 = 3.0 * (c22 - c12) / c22;
            W.dist[tri((unsigned)i, (unsigned)j)] = (float)(d1 < d2 ? d1 : d2);
        }
    }
    if (lane == 0) W.dist[(size_t)n * (n - 1) / 2] = 0.0f;
    GA_SYNC();

    GA_STAGE(4); GA_CLOCK(3);
    // ---- UPGMB (upgma2.cpp:133-355): 0.1 * average + 0.9 * minimum linkage, stale row minima kept
    const unsigned un = (unsigned)n;
    for (unsigned x = lane; x < un; x += 64) {
        float best = kBigDist; unsigned arg = kNone;
#pragma unroll 8
        for (unsigned j = 0; j < un; j++) {
            if (j == x) continue;
            const float d = W.dist[tri(x, j)];
            if (d < best) { best = d; arg = j; }
        }
        S.t.mind[x] = best; S.t.nearest[x] = arg; S.t.node[x] = x;
    }
    GA_SYNC();
    GA_CLOCK(4);
    for (unsigned k = 0; k + 1 < un; k++) {
        float best = kBigDist; unsigned lmin = kNone;
        for (unsigned j = lane; j < un; j += 64) {
            if (S.t.node[j] == kNone) continue;
            if (S.t.mind[j] < best) { best = S.t.mind[j]; lmin = j; }
        }
        wave_argmin(best, lmin);
        if (lmin == kNone) return false;
        const unsigned rmin = uni(S.t.nearest[lmin]);
        if (rmin == kNone || rmin >= un) return false;
        // the distances of this step are requested together (two per cluster and lane, plus the pair's own): one round
        // trip to the workspace per merge instead of one per 64 clusters
        constexpr int kRounds = kMaxSeqs / 64;
        float dl[kRounds], dr[kRounds];
        const float dlr = W.dist[tri(lmin, rmin)];
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const unsigned j = (unsigned)r * 64 + (unsigned)lane;
            dl[r] = 0; dr[r] = 0;
            if (j < un && j != lmin && j != rmin && S.t.node[j] != kNone) { dl[r] = W.dist[tri(lmin, j)]; dr[r] = W.dist[tri(rmin, j)]; }
        }
        float new_min = kBigDist; unsigned new_nearest = kNone;
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const unsigned j = (unsigned)r * 64 + (unsigned)lane;
            if (j >= un || j == lmin || j == rmin || S.t.node[j] == kNone) continue;
            const float nd = kSueff * ((dl[r] + dr[r]) / 2) + (1 - kSueff) * (dl[r] < dr[r] ? dl[r] : dr[r]);
            if (S.t.nearest[j] == rmin) S.t.nearest[j] = lmin;
            W.dist[tri(lmin, j)] = nd;
            if (nd < new_min) { new_min = nd; new_nearest = j; }
        }
        wave_argmin(new_min, new_nearest);
        if (lane == 0) {
            const float h = dlr / 2;
            const unsigned ul = S.t.node[lmin], ur = S.t.node[rmin];
            const float hl = ul < un ? 0 : S.t.height[ul - un];
            const float hr = ur < un ? 0 : S.t.height[ur - un];
            const unsigned v = un + k;
            W.left[v] = ul; W.right[v] = ur;
            W.parent[ul] = v; W.parent[ur] = v;
            W.to_parent[ul] = (double)(h - hl); W.to_parent[ur] = (double)(h - hr);
            S.t.height[k] = h;
            S.t.node[lmin] = v; S.t.nearest[lmin] = new_nearest; S.t.mind[lmin] = new_min; S.t.node[rmin] = kNone;
        }
        GA_SYNC();
    }
    const unsigned root = 2 * un - 2, nodes = 2 * un - 1;
    GA_STAGE(5); GA_CLOCK(5);
    // ---- ClustalW weights (clwwt.cpp:65-163)
    if (lane == 0) {
        for (unsigned v = 0; v < nodes; v++) W.under[v] = v < un ? 1 : W.under[W.left[v]] + W.under[W.right[v]];
        W.lo[root] = 0;
        for (unsigned v = root; v >= un; v--) { W.lo[W.left[v]] = W.lo[v]; W.lo[W.right[v]] = W.lo[v] + (int32_t)W.under[W.left[v]]; }
        for (unsigned l = 0; l < un; l++) W.perm[W.lo[l]] = (int32_t)l;
    }
    GA_SYNC();
    if (n == 2) { if (lane < 2) W.weight[lane] = 0.5f; }
    else {
        for (unsigned v = lane; v < nodes; v += 64) W.strength[v] = v == root ? 0.0 : W.to_parent[v] / (double)W.under[v];
        GA_SYNC();
        int broken = 0;
        for (unsigned l = lane; l < un; l += 64) {
            double sum = 0;
            unsigned steps = 0;
            for (unsigned v = l; v != root && steps <= nodes; v = W.parent[v], steps++) { if (v >= nodes) { steps = nodes + 1; break; } sum += W.strength[v]; }
            if (steps > nodes) broken = 1;
            if (sum < 0.0001) sum = 1.0;
            W.weight[l] = (float)sum;
        }
        if (wave_sum(broken)) return false;        // not a tree: cannot happen, and must not spin if it does
        GA_SYNC();
        float total = 0.0;
        for (unsigned l = 0; l < un; l++) total += W.weight[l];
        if (total == 0.0) return false;
        GA_SYNC();
        for (unsigned l = lane; l < un; l += 64) W.weight[l] /= total;
    }
    GA_SYNC();
    GA_STAGE(6); GA_CLOCK(6);
    // ---- leaves: one-row alignments (row p of the LDS block = leaf p of the root alignment's order; a lane per row)
    for (int p = lane; p < n; p += 64) {
        const int i = W.perm[p], L = W.len[i];
        uint8_t* row = R + (size_t)p * (size_t)cap;
        const int64_t a0 = P.seq_off[job.first_seq + i];
        for (int x = 0; x < L; x++) { const uint8_t l = S.letter[P.chars[a0 + x]]; row[x] = l >= 16 ? (uint8_t)15 : l; }      // FixAlpha (seq.cpp:331-344): anything else is an 'N'

        S.wrow[p] = W.weight[i]; S.rowlen[p] = (uint8_t)L;
    }
    GA_SYNC();
    GA_STAGE(7); GA_CLOCK(7);
    // ---- progressive alignment (progressivealign.cpp:16-82): children are created before their parent, so ascending
    // node order computes every alignment after its two inputs (the reference's left-first post-order does the same
    // merges in another order)
    // (a node's inputs -- first row and number of rows of each child -- are fixed by now: 64 nodes' worth are fetched at a
    // time, a lane each, and handed round; what a node produces -- columns, weight total -- stays in LDS)
    for (unsigned v0 = un; v0 < nodes; v0 += 64) {
      unsigned my_a = 0, my_b = 0; int my_loa = 0, my_nsa = 0, my_lob = 0, my_nsb = 0;
      if (v0 + (unsigned)lane < nodes) {
          my_a = W.left[v0 + lane]; my_b = W.right[v0 + lane];
          my_loa = W.lo[my_a]; my_nsa = (int)W.under[my_a]; my_lob = W.lo[my_b]; my_nsb = (int)W.under[my_b];
      }
      const unsigned vend = v0 + 64 < nodes ? v0 + 64 : nodes;
      for (unsigned v = v0; v < vend; v++) {
        const int k = (int)(v - v0);
        const unsigned a = uni((unsigned)__shfl((int)my_a, k, 64)), b = uni((unsigned)__shfl((int)my_b, k, 64));
        const int loa = uni(__shfl(my_loa, k, 64)), nsa = uni(__shfl(my_nsa, k, 64)), lob = uni(__shfl(my_lob, k, 64)), nsb = uni(__shfl(my_nsb, k, 64));
        const int la = uni(a < un ? (int)S.rowlen[loa] : (int)S.ncols_i[a - un]);
        const int lb = uni(b < un ? (int)S.rowlen[lob] : (int)S.ncols_i[b - un]);
        const float total_a = a < un ? 0.0f + S.wrow[loa] : S.total_i[a - un];
        const float total_b = b < un ? 0.0f + S.wrow[lob] : S.total_i[b - un];
        if (la <= 0 || lb <= 0 || la > kMaxCols || lb > kMaxCols) return false;
        if (ga_stage_on(100)) GA_STAGE_DYN(100 + (int)(v - un) * 10);
        GA_CLOCK(8);
        if (any_wild) { build_profile<true>(S, R, cap, loa, nsa, la, total_a, true); build_profile<true>(S, R, cap, lob, nsb, lb, total_b, false); }
        else { build_profile<false>(S, R, cap, loa, nsa, la, total_a, true); build_profile<false>(S, R, cap, lob, nsb, lb, total_b, false); }
        int plen = 0;
        if (ga_stage_on(100)) GA_STAGE_DYN(101 + (int)(v - un) * 10); GA_CLOCK(13);
        if (!nw_small(S, TB, la, lb, &plen, kGapClocks && P.prof != nullptr, prof_acc[9], prof_t0)) return false;
        if (ga_stage_on(100)) GA_STAGE_DYN(102 + (int)(v - un) * 10); GA_CLOCK(10);
        if (plen > cap || plen > kMaxCols) return false;
        // aligngivenpath.cpp:124-255: a column of A, of B, or of both
        // (a column's place in A / in B = the number of A / B columns before it: counted with ballots)
        {
            int ca = 0, cb = 0;
            for (int c0 = 0; c0 < plen; c0 += 64) {
                const int c = c0 + lane;
                const uint8_t t = c < plen ? S.p.path[c] : (uint8_t)0;
                const bool in_a = c < plen && t != 'I', in_b = c < plen && t != 'D';
                const unsigned long long ma = __ballot(in_a), mb = __ballot(in_b), below = (1ull << lane) - 1;
                if (c < plen) {
                    S.p.mapa[c] = in_a ? (int16_t)(ca + __popcll(ma & below)) : (int16_t)-1;
                    S.p.mapb[c] = in_b ? (int16_t)(cb + __popcll(mb & below)) : (int16_t)-1;
                }
                ca += __popcll(ma); cb += __popcll(mb);
            }
            GA_SYNC();
            if (ca != la || cb != lb) return false;
        }
        // rows re-spelled in place through the column maps.  An input with no column inserted keeps its rows as they are
        // (the map is the identity): when one sequence joins a large alignment, that is usually the large one.  Every lane
        // reads its (at most two) characters of a row before any lane writes: one wavefront, LDS operations in program order.
        static_assert(kMaxCols <= 128, "two columns per lane");
        {
            const int c1 = lane, c2 = lane + 64;
            const int ma1 = c1 < plen ? S.p.mapa[c1] : -1, ma2 = c2 < plen ? S.p.mapa[c2] : -1;
            const int mb1 = c1 < plen ? S.p.mapb[c1] : -1, mb2 = c2 < plen ? S.p.mapb[c2] : -1;
            auto respell = [&](int p0, int np, int m1, int m2) {
                // four rows at a time: their new codes, written; then, for the gap columns, the codes next to them (now in
                // place) say whether the row's gap starts / ends here (the flags a profile reads, see kRowStart)
                for (int s0 = 0; s0 < np; s0 += 4) {
                    uint8_t v1[4], v2[4], l1[4], r1[4], l2[4], r2[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint8_t* row = R + (size_t)(p0 + (s0 + k < np ? s0 + k : np - 1)) * (size_t)cap;
                        v1[k] = m1 >= 0 ? (uint8_t)(row[m1] & kRowCode) : kRowGap; v2[k] = m2 >= 0 ? (uint8_t)(row[m2] & kRowCode) : kRowGap;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (s0 + k >= np) continue;
                        uint8_t* row = R + (size_t)(p0 + s0 + k) * (size_t)cap;
                        if (c1 < plen) row[c1] = v1[k];
                        if (c2 < plen) row[c2] = v2[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint8_t* row = R + (size_t)(p0 + (s0 + k < np ? s0 + k : np - 1)) * (size_t)cap;
                        l1[k] = c1 > 0 && c1 < plen ? row[c1 - 1] : (uint8_t)0; r1[k] = c1 + 1 < plen ? row[c1 + 1] : (uint8_t)0;
                        l2[k] = c2 < plen ? row[c2 - 1] : (uint8_t)0; r2[k] = c2 + 1 < plen ? row[c2 + 1] : (uint8_t)0;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (s0 + k >= np) continue;
                        uint8_t* row = R + (size_t)(p0 + s0 + k) * (size_t)cap;
                        if (c1 < plen && v1[k] == kRowGap) row[c1] = (uint8_t)(kRowGap | (row_gap(l1[k]) ? 0 : kRowStart) | (row_gap(r1[k]) ? 0 : kRowEnd));
                        if (c2 < plen && v2[k] == kRowGap) row[c2] = (uint8_t)(kRowGap | (row_gap(l2[k]) ? 0 : kRowStart) | (row_gap(r2[k]) ? 0 : kRowEnd));
                    }
                }
            };
            if (plen != la) respell(loa, nsa, ma1, ma2);
            if (plen != lb) respell(lob, nsb, mb1, mb2);
        }
        if (lane == 0) {
            S.ncols_i[v - un] = (uint8_t)plen;
            float t = total_a;                       // the merged alignment's rows: A's, then B's
            for (int x = 0; x < nsb; x++) t += S.wrow[lob + x];
            S.total_i[v - un] = t;
        }
        GA_SYNC();
        GA_CLOCK(11);
      }
    }
    const int nc = n >= 2 ? (int)S.ncols_i[root - un] : 0;
    if (nc > job.max_cols) return false;
    for (int p = lane; p < n; p += 64) {              // a lane per row: its place in the output comes from one load
        const uint8_t* src = R + (size_t)p * (size_t)cap;
        uint8_t* dst = P.out_rows + job.row_off + (int64_t)W.perm[p] * job.max_cols;
        for (int c = 0; c < nc; c++) dst[c] = row_char(src[c]);
    }
    *out_cols = nc;
    GA_CLOCK(12);
    if (kGapClocks && P.prof && lane == 0) for (int k = 0; k < kProfStages; k++) if (prof_acc[k]) atomicAdd(&P.prof[k], prof_acc[k]);
    return true;
}

__device__ Slot carve(uint8_t* base, int nmax, int cap) {
    Slot W;
    size_t off = 0;
    auto take = [&](size_t bytes) { uint8_t* p = base + off; off += (bytes + 15) & ~(size_t)15; return p; };
    const size_t n = (size_t)nmax;
    W.dist = (float*)take(4 * (n * (n - 1) / 2 + 1));
    W.ucommon = (uint16_t*)take(2 * n * n);
    W.tcode = (uint16_t*)take(2 * n * kMaxCols);
    W.tcnt = take(n * kMaxCols);
    W.ntup = (int32_t*)take(4 * n);
    W.hash = (uint64_t*)take(8 * n);
    W.len = (int32_t*)take(4 * n); W.cls = (int32_t*)take(4 * n); W.repidx = (int32_t*)take(4 * n); W.replist = (int32_t*)take(4 * n);
    W.left = (uint32_t*)take(8 * n); W.right = (uint32_t*)take(8 * n); W.parent = (uint32_t*)take(8 * n);
    W.to_parent = (double*)take(16 * n);
    W.height = (float*)take(4 * n);
    W.under = (uint32_t*)take(8 * n);
    W.strength = (double*)take(16 * n);
    W.weight = (float*)take(4 * n);
    W.lo = (int32_t*)take(8 * n); W.ncols = (int32_t*)take(8 * n);
    W.perm = (int32_t*)take(4 * n);
    W.total = (float*)take(8 * n);
    W.table = take(kTable);
    return W;
}
size_t slot_bytes(int nmax, int cap) {
    const size_t n = (size_t)nmax;
    auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
    return r(4 * (n * (n - 1) / 2 + 1)) + r(2 * n * n) + r(2 * n * kMaxCols) + r(n * kMaxCols) + r(4 * n) + r(8 * n) + 4 * r(4 * n) + 3 * r(8 * n) + r(16 * n) +
           r(4 * n) + r(8 * n) + r(16 * n) + r(4 * n) + 2 * r(8 * n) + r(4 * n) + r(8 * n) + r(kTable) + 256;
}

__global__ __launch_bounds__(64) void gap_align_kernel(Params P) {
    __shared__ Shared S;
    extern __shared__ __align__(16) uint8_t rows_lds[];      // nmax rows of cap bytes, then the (cap+1)^2 trace-back bytes of one pairwise DP
    uint8_t* const tb_lds = rows_lds + ((((size_t)P.nmax * (size_t)P.cap) + 15) & ~(size_t)15);
    for (int x = (int)threadIdx.x; x < 256; x += 64) S.letter[x] = c_letter[x];
    const Slot W = carve(P.ws + (size_t)blockIdx.x * (size_t)P.ws_stride, P.nmax, P.cap);
    for (int x = (int)threadIdx.x; x < kTable; x += 64) W.table[x] = 0;
    GA_SYNC();
    for (;;) {
        if (threadIdx.x == 0) S.flag = (int32_t)atomicAdd(P.next, 1ull);
        GA_SYNC();
        const int64_t j = uni(S.flag);
        GA_SYNC();
        if (j >= P.njobs) break;
        const Job job = P.jobs[j];
        if (ga_stage_on(900) && P.dbg && threadIdx.x == 0) { P.dbg[blockIdx.x * 2] = (int32_t)j; P.dbg[blockIdx.x * 2 + 1] = 0; }
        int cols = -1;
        if (!align_job(S, rows_lds, tb_lds, W, P, job, &cols)) cols = -1;
        GA_SYNC();
        if (threadIdx.x == 0) P.out_cols[j] = cols;
        if (ga_stage_on(900) && P.dbg && threadIdx.x == 0) P.dbg[blockIdx.x * 2 + 1] = -1;
    }
}

__global__ void warmup_kernel(int* p) { if (threadIdx.x == 0) *p = 1; }

// the last error, process wide: the XMFA writer makes the call on a side thread and asks for the message on another.
// pm_gap_last_error() hands out a copy that stays valid until the calling thread asks again.
std::mutex g_err_mu;
std::string g_err;
int32_t* g_dbg = nullptr; int64_t g_dbg_slots = 0;
int32_t* g_dbg_dev = nullptr;      // PM_GAP_DEBUG=2: the markers live in device memory (cheap to write), peeked through a copy on another stream
int fail(int code, const std::string& m) { std::lock_guard<std::mutex> lk(g_err_mu); g_err = m; return code; }
// the letter table is uploaded once per device; the flag is set only after the upload succeeded (a failed call must not leave
// later calls of the process aligning with an uninitialised table)
std::mutex g_tables_mu;
bool g_tables_ready[64] = {false};
}  // namespace

extern "C" const char* pm_gap_last_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_err;
    return copy.c_str();
}
// Start the HIP runtime (device discovery, context, code objects: ~0.15 s of a fresh process) -- callable from a side
// thread while the caller still parses its FASTA files, so that pm_session_create finds it running.
extern "C" int pm_warmup(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(PM_ENODEV, "no HIP device available");
    if (device < 0) { const char* e = getenv("PARSNP_DEVICE"); if (e && *e) device = atoi(e); }
    if (device >= 0 && (device >= count || hipSetDevice(device) != hipSuccess)) return fail(PM_ENODEV, "cannot select the requested HIP device");
    // an allocation brings the context up; a stream with one kernel on it brings up a hardware queue and this library's
    // code objects -- the parts of the start-up that a later pm_session_create would otherwise wait for
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return fail(PM_EHIP, "hipMalloc failed during warm-up");
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        hipLaunchKernelGGL(warmup_kernel, dim3(1), dim3(64), 0, st, (int*)p);
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    (void)hipFree(p);
    return PM_OK;
}
// PM_GAP_DEBUG=1: (job, stage) of every slot of the running launch, readable from another thread
extern "C" int64_t pm_gap_debug_peek(int32_t* out, int64_t cap) {
    int64_t n = 0;
    if (g_dbg_dev) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -1;
        n = std::min<int64_t>(cap, 2 * g_dbg_slots);
        if (hipMemcpyAsync(out, g_dbg_dev, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) n = -2;
        (void)hipStreamDestroy(st);
        return n;
    }
    for (int64_t i = 0; g_dbg && i < 2 * g_dbg_slots && i < cap; i++) out[n++] = g_dbg[i];
    return n;
}

extern "C" int pm_gap_align_batch(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                                  const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols) {
    return pm_gap_align_groups(device, n_jobs, n_seqs, seq_off, chars, max_cols, row_off, out_rows, out_bytes, cols, 1, &n_jobs, nullptr, nullptr);
}
// The same batch in groups of consecutive jobs (group g = jobs [group_end[g-1], group_end[g])): everything is uploaded once,
// the groups are aligned one after the other, and when the rows and column counts of a group are in the caller's memory
// `done(ctx, g)` is called (from this thread) -- the caller can start using them while the later groups are still running.
extern "C" int pm_gap_align_groups(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                                   const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols,
                                   int n_groups, const int64_t* group_end, void (*done)(void* ctx, int group), void* ctx) {
    if (n_jobs < 0 || (n_jobs > 0 && (!n_seqs || !seq_off || !chars || !max_cols || !row_off || !out_rows || !cols))) return fail(PM_EINVAL, "bad argument");
    if (n_groups < 1 || !group_end || group_end[n_groups - 1] != n_jobs) return fail(PM_EINVAL, "bad job groups");
    for (int g = 0; g < n_groups; g++) if (group_end[g] < (g ? group_end[g - 1] : 0)) return fail(PM_EINVAL, "bad job groups");
    auto all_done = [&](int from) { if (done) for (int g = from; g < n_groups; g++) done(ctx, g); };
    if (n_jobs == 0) { all_done(0); return PM_OK; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(PM_ENODEV, "no HIP device available; the gap aligner of this library has no CPU path");
    if (device < 0) { const char* e = getenv("PARSNP_DEVICE"); if (e && *e) device = atoi(e); }
    if (device >= 0 && (device >= count || hipSetDevice(device) != hipSuccess)) return fail(PM_ENODEV, "cannot select the requested HIP device");
#define GA_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { release(); return fail(PM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    std::vector<void*> owned;
    hipStream_t stream = nullptr;
    auto release = [&]() { for (void* p : owned) (void)hipFree(p); owned.clear(); if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; } };
    {
        hipError_t table_err = hipSuccess;
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        std::lock_guard<std::mutex> lk(g_tables_mu);
        if (cur_dev < 0 || cur_dev >= 64 || !g_tables_ready[cur_dev]) {
            uint8_t letter[256];
            memset(letter, 255, sizeof letter);
            const char* res = "ACGT";
            for (int i = 0; i < 4; i++) { letter[(uint8_t)res[i]] = (uint8_t)i; letter[(uint8_t)(res[i] + 32)] = (uint8_t)i; }
            letter[(uint8_t)'U'] = letter[(uint8_t)'u'] = 3;
            const char* wild = "MRWSYKVHDBXN";
            for (int i = 0; i < 12; i++) { letter[(uint8_t)wild[i]] = (uint8_t)(4 + i); letter[(uint8_t)(wild[i] + 32)] = (uint8_t)(4 + i); }
            letter[(uint8_t)'-'] = letter[(uint8_t)'.'] = 16;
            table_err = hipMemcpyToSymbol(HIP_SYMBOL(c_letter), letter, 256);
            if (table_err == hipSuccess && cur_dev >= 0 && cur_dev < 64) g_tables_ready[cur_dev] = true;
        }
        GA_CHECK(table_err);
    }
    // jobs the device takes, longest first (the cost of one alignment grows with the square of its width)
    std::vector<Job> jobs; std::vector<int64_t> which; std::vector<int> group_of_job;
    int64_t seq = 0, total_chars = 0;
    int nmax = 2, cap = 1;
    std::vector<int> widest((size_t)n_jobs, 0);
    int grp = 0;
    for (int64_t j = 0; j < n_jobs; j++) {
        while (grp + 1 < n_groups && j >= group_end[grp]) grp++;
        cols[j] = -1;
        const int n = n_seqs[j];
        int w = 0; bool ok = n >= 2 && n <= kMaxSeqs && max_cols[j] >= 1;
        for (int i = 0; i < n && ok; i++) { const int64_t L = seq_off[seq + i + 1] - seq_off[seq + i]; if (L <= 0 || L > kMaxCols) ok = false; else w = std::max<int>(w, (int)L); }
        if (ok && row_off[j] + (int64_t)n * max_cols[j] > out_bytes) ok = false;
        if (ok) { jobs.push_back(Job{seq, n, max_cols[j], row_off[j]}); which.push_back(j); group_of_job.push_back(grp); widest[(size_t)j] = w; nmax = std::max(nmax, n); cap = std::max(cap, std::min<int>(max_cols[j], kMaxCols)); }
        seq += n;
    }
    total_chars = seq_off[seq];
    if (jobs.empty()) { all_done(0); return PM_OK; }
    std::vector<size_t> first_of_group((size_t)n_groups + 1, 0);      // in the sorted job list
    {
        std::vector<size_t> order(jobs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            if (group_of_job[a] != group_of_job[b]) return group_of_job[a] < group_of_job[b];
            return widest[(size_t)which[a]] > widest[(size_t)which[b]];
        });
        std::vector<Job> j2; std::vector<int64_t> w2;
        for (size_t i : order) { j2.push_back(jobs[i]); w2.push_back(which[i]); first_of_group[(size_t)group_of_job[i] + 1]++; }
        for (int g = 0; g < n_groups; g++) first_of_group[(size_t)g + 1] += first_of_group[(size_t)g];
        jobs.swap(j2); which.swap(w2);
    }
    const bool timers = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = now();
    auto lap = [&](const char* what) { if (timers) { const double t = now(); fprintf(stderr, "[gap batch] %-12s %.4f s\n", what, t - tl); tl = t; } };
    hipDeviceProp_t prop;
    if (device < 0) GA_CHECK(hipGetDevice(&device));      // the current device of this thread
    GA_CHECK(hipGetDeviceProperties(&prop, device));
    // LDS of a workgroup: the fixed block plus the alignment rows of the widest job; as many workgroups per CU as fit in 160 KB
    const size_t rows_lds = ((((size_t)nmax * (size_t)cap) + 15) & ~(size_t)15) + ((((size_t)cap + 1) * ((size_t)cap + 1) + 15) & ~(size_t)15);
    const size_t lds = sizeof(Shared) + rows_lds;
    if (lds > 160 * 1024 - 1024) { release(); return fail(PM_ELIMIT, "gap alignment rows do not fit the LDS"); }
    if (lds > 64 * 1024) GA_CHECK(hipFuncSetAttribute((const void*)gap_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 256)));
    const int64_t slots = std::min<int64_t>((int64_t)jobs.size(), (int64_t)prop.multiProcessorCount * per_cu);
    const size_t stride = slot_bytes(nmax, cap);
    GA_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    auto dalloc = [&](size_t bytes, void** p) { hipError_t e = hipMalloc(p, bytes ? bytes : 1); if (e == hipSuccess) owned.push_back(*p); return e; };
    Job* d_jobs; int64_t* d_off; uint8_t* d_chars; uint8_t* d_out; int32_t* d_cols; unsigned long long* d_next; uint8_t* d_ws;
    GA_CHECK(dalloc(sizeof(Job) * jobs.size(), (void**)&d_jobs));
    GA_CHECK(dalloc(8 * (size_t)(seq + 1), (void**)&d_off));
    GA_CHECK(dalloc((size_t)total_chars, (void**)&d_chars));
    GA_CHECK(dalloc((size_t)out_bytes, (void**)&d_out));
    GA_CHECK(dalloc(4 * jobs.size(), (void**)&d_cols));
    GA_CHECK(dalloc(8 * (size_t)n_groups, (void**)&d_next));
    GA_CHECK(dalloc(stride * (size_t)slots, (void**)&d_ws));
    GA_CHECK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice, stream));
    GA_CHECK(hipMemcpyAsync(d_off, seq_off, 8 * (size_t)(seq + 1), hipMemcpyHostToDevice, stream));
    GA_CHECK(hipMemcpyAsync(d_chars, chars, (size_t)total_chars, hipMemcpyHostToDevice, stream));
    GA_CHECK(hipMemsetAsync(d_next, 0, 8 * (size_t)n_groups, stream));
    if (timers) { GA_CHECK(hipStreamSynchronize(stream)); lap("alloc + h2d"); }
    int32_t* dbg = nullptr;
    if (getenv("PM_GAP_DEBUG") && atoi(getenv("PM_GAP_DEBUG")) == 2) {
        if (g_dbg_dev) (void)hipFree(g_dbg_dev);
        g_dbg_dev = nullptr;
        if (hipMalloc((void**)&g_dbg_dev, 8 * (size_t)slots) == hipSuccess) { g_dbg_slots = slots; (void)hipMemsetAsync(g_dbg_dev, 0xff, 8 * (size_t)slots, stream); dbg = g_dbg_dev; }
    } else if (getenv("PM_GAP_DEBUG")) {
        if (!g_dbg || g_dbg_slots < slots) { if (g_dbg) (void)hipHostFree(g_dbg); g_dbg = nullptr; if (hipHostMalloc((void**)&g_dbg, 8 * (size_t)slots, hipHostMallocMapped) == hipSuccess) g_dbg_slots = slots; }
        if (g_dbg) { memset(g_dbg, 0xff, 8 * (size_t)slots); (void)hipHostGetDevicePointer((void**)&dbg, g_dbg, 0); }
    }
    unsigned long long* d_prof = nullptr;
    if (getenv("PM_GAP_DEBUG") && atoi(getenv("PM_GAP_DEBUG")) == 3) {
        GA_CHECK(dalloc(8 * kProfStages, (void**)&d_prof));
        GA_CHECK(hipMemsetAsync(d_prof, 0, 8 * kProfStages, stream));
    }
    if (timers) fprintf(stderr, "[gap batch] %zu jobs in %d group(s), widest %d sequences x %d columns: %zu B of LDS per wavefront, %d per CU, %lld slots\n", jobs.size(), n_groups, nmax, cap, lds, per_cu, (long long)slots);
    std::vector<int32_t> got(jobs.size());
    int signalled = 0;                 // groups whose completion the caller has been told
    for (int g = 0; g < n_groups; g++) {
        const size_t j0 = first_of_group[(size_t)g], j1 = first_of_group[(size_t)g + 1];
        if (j1 > j0) {
            // a group's jobs: their own queue counter, their slice of the job and column arrays; workspace and slots shared
            Params P{d_jobs + j0, (int64_t)(j1 - j0), d_off, d_chars, d_out, d_cols + j0, d_next + g, d_ws, (int64_t)stride, nmax, cap, dbg, d_prof};
            const int64_t gslots = std::min<int64_t>((int64_t)(j1 - j0), slots);
            hipLaunchKernelGGL(gap_align_kernel, dim3((unsigned)gslots), dim3(64), rows_lds, stream, P);
            GA_CHECK(hipGetLastError());
            // the rows of the group: the span of the output its jobs cover (a span may include rows of other groups: the
            // device buffer holds their final bytes if they are done, and they are copied again when they are not)
            int64_t lo = out_bytes, hi = 0;
            for (size_t i = j0; i < j1; i++) { lo = std::min(lo, jobs[i].row_off); hi = std::max(hi, jobs[i].row_off + (int64_t)jobs[i].n * jobs[i].max_cols); }
            GA_CHECK(hipMemcpyAsync(got.data() + j0, d_cols + j0, 4 * (j1 - j0), hipMemcpyDeviceToHost, stream));
            if (hi > lo) GA_CHECK(hipMemcpyAsync(out_rows + lo, d_out + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, stream));
            GA_CHECK(hipStreamSynchronize(stream));
            for (size_t i = j0; i < j1; i++) cols[which[i]] = got[i];
        }
        if (timers) { char what[32]; snprintf(what, sizeof what, "group %d", g + 1); lap(what); }
        if (done) done(ctx, g);
        signalled = g + 1;
    }
    (void)signalled;
    if (d_prof) {
        unsigned long long prof[kProfStages];
        GA_CHECK(hipMemcpy(prof, d_prof, sizeof prof, hipMemcpyDeviceToHost));
        static const char* const names[kProfStages] = {"lengths + hashes", "distinct strings", "6-mers + common counts", "distances", "tree: row minima", "tree: merges", "weights",
                                                       "leaves", "node set-up", "pairwise DP: init + sweep", "DP: trace-back", "path maps + rows + totals", "output", "profiles", "", ""};
        unsigned long long sum = 0;
        for (int k = 0; k < kProfStages; k++) sum += prof[k];
        for (int k = 0; k < kProfStages; k++)
            if (prof[k]) fprintf(stderr, "[gap stages] %-24s %6.2f %%  %12.0f clocks per job\n", names[k], 100.0 * (double)prof[k] / (double)sum, (double)prof[k] / (double)jobs.size());
    }
    release();
    lap("release");
#undef GA_CHECK
    return PM_OK;
}
