// gapalign.cpp -- restatement of the one libMUSCLE 3.7 configuration the reference's XMFA writer uses for the gaps
// between adjacent MUMs (src/MuscleInterface.cpp:43-66: SEQTYPE_DNA, MaxIters 1, stable, ClustalW weights).
// Every stage keeps the reference's float32/float64 mix and its tie rules, because the alignment depends on them:
//   k-mer distance        muscle/libMUSCLE/fastdistnuc.cpp:103-262   (6-mers over {A,C,G,T,other}, 8-bit counts)
//   guide tree            upgma2.cpp:133-355                          (UPGMB: 0.1*avg + 0.9*min, stale row minima kept)
//   sequence weights      clwwt.cpp:65-163, msa2.cpp:403-431          (ClustalW, renormalised inside every profile)
//   progressive alignment progressivealign.cpp:16-82, aligntwomsas.cpp:10-41
//   profiles              profilefrommsa.cpp:262-331, msa2.cpp:19-176 (NUC_SP matrix nucmx.cpp:15-25, gap open -400)
//   pairwise DP           nwsmall.cpp:447-620 (affine, half terminal gaps termgaps.cpp:19-33), bittraceback.cpp:130-209
//   merge                 aligngivenpath.cpp:257-366
#include "gapalign.h"

#include <cstdint>
#include <cstring>
#include <unordered_map>

namespace parsnp {
namespace {

const float kMinusInf = (float)-1e37;    // intmath.h:46
const float kBigDist = (float)1e29;      // distcalc.h:7
const float kGapOpen = -400.0f;          // params.cpp:309-311
const float kGapExtend = 0.0f;           // params.cpp:312-313
const float kSueff = (float)0.1;         // params.cpp:73
const unsigned kNone = 0xffffffffu;

// nucmx.cpp:8-25: BLASTZ scores + 2*30
const float kMatrix[4][4] = {
    {91.0f + 60.0f, -114.0f + 60.0f, -31.0f + 60.0f, -123.0f + 60.0f},
    {-114.0f + 60.0f, 100.0f + 60.0f, -125.0f + 60.0f, -31.0f + 60.0f},
    {-31.0f + 60.0f, -125.0f + 60.0f, 100.0f + 60.0f, -114.0f + 60.0f},
    {-123.0f + 60.0f, -31.0f + 60.0f, -114.0f + 60.0f, 91.0f + 60.0f},
};

// alpha.cpp:125-166 (SetAlphaDNA falls through into SetAlphaRNA): letter codes of alpha.h:47-67
struct Alphabet {
    uint8_t letter[256];     // 0..3 residues, 4..15 wildcards (NX_M..NX_N), 16 gap, 255 not a residue
    Alphabet() {
        memset(letter, 255, sizeof letter);
        const char* res = "ACGT";
        for (int i = 0; i < 4; i++) { letter[(uint8_t)res[i]] = (uint8_t)i; letter[(uint8_t)(res[i] + 32)] = (uint8_t)i; }
        letter[(uint8_t)'U'] = letter[(uint8_t)'u'] = 3;
        const char* wild = "MRWSYKVHDBXN";
        for (int i = 0; i < 12; i++) { letter[(uint8_t)wild[i]] = (uint8_t)(4 + i); letter[(uint8_t)(wild[i] + 32)] = (uint8_t)(4 + i); }
        letter[(uint8_t)'-'] = letter[(uint8_t)'.'] = 16;
    }
};
const Alphabet kAlpha;
inline bool is_gap(char c) { return c == '-' || c == '.'; }

// ---------------------------------------------------------------------------------------------- distance + tree
inline unsigned tri(unsigned a, unsigned b) { return a >= b ? b + (a * (a - 1)) / 2 : a + (b * (b - 1)) / 2; }

struct Tree {
    unsigned leaves = 0;
    std::vector<unsigned> left, right, parent;   // node ids: leaves 0..N-1, internal N..2N-2, root = 2N-2
    std::vector<double> to_parent;               // edge length above a node
    unsigned root() const { return 2 * leaves - 2; }
    bool leaf(unsigned v) const { return v < leaves; }
};

void kmer_distances(const std::vector<std::string>& s, std::vector<float>* dist) {
    const unsigned n = (unsigned)s.size();
    // The number of common 6-mers of two sequences depends on the two strings alone, and the sequences of a gap are
    // mostly copies of a few alleles: counted once per pair of DISTINCT strings, then laid out per pair of sequences.
    std::vector<unsigned> cls(n);
    std::vector<unsigned> rep;                       // first sequence of every distinct string
    {
        std::unordered_map<std::string, unsigned> seen;
        seen.reserve(2 * n);
        for (unsigned i = 0; i < n; i++) {
            auto it = seen.emplace(s[i], (unsigned)rep.size());
            if (it.second) rep.push_back(i);
            cls[i] = it.first->second;
        }
    }
    const unsigned u = (unsigned)rep.size();
    // per distinct string: its distinct 6-mers with their 8-bit (wrapping) multiplicities, in first-occurrence order
    std::vector<std::vector<std::pair<uint32_t, uint8_t>>> tuples(u);
    // (kept per thread and left all-zero after use: a block this size per call would go through mmap/munmap, and with
    // every thread aligning gaps at once those calls serialise the whole process)
    static thread_local std::vector<uint8_t> count(6 * 6 * 6 * 6 * 6 * 6, 0);
    std::vector<uint32_t> seen;
    for (unsigned a = 0; a < u; a++) {
        const std::string& q = s[rep[a]];
        if (q.size() < 5) continue;
        seen.clear();
        uint32_t t = 0;
        for (size_t p = 0; p < q.size(); p++) {
            uint8_t l = kAlpha.letter[(uint8_t)q[p]];
            if (l >= 4) l = 4;
            t = (t * 6 + l) % 46656u;
            if (p >= 5) { if (count[t] == 0) seen.push_back(t); ++count[t]; }   // unsigned char counts wrap (fastdistnuc.cpp:82-90)
        }
        // a count that wrapped back to 0 contributes nothing; a 6-mer listed twice (0 -> 256 -> 0 -> ...) is emitted once
        for (uint32_t x : seen) { if (count[x]) { tuples[a].emplace_back(x, count[x]); count[x] = 0; } }
    }
    std::vector<unsigned> ucommon((size_t)u * u, 0);
    for (unsigned a = 0; a < u; a++) {
        if (s[rep[a]].size() < 5) continue;
        for (auto& tc : tuples[a]) count[tc.first] = tc.second;
        for (unsigned b = 0; b <= a; b++) {
            if (s[rep[b]].size() < 5) continue;
            unsigned sum = 0;
            for (auto& tc : tuples[b]) { uint8_t c1 = count[tc.first]; sum += c1 < tc.second ? c1 : tc.second; }
            ucommon[(size_t)a * u + b] = ucommon[(size_t)b * u + a] = sum;
        }
        for (auto& tc : tuples[a]) count[tc.first] = 0;
    }
    auto common = [&](unsigned i, unsigned j) { return ucommon[(size_t)cls[i] * u + cls[j]]; };
    dist->assign((size_t)n * (n - 1) / 2 + 1, 0.0f);
    for (unsigned i = 0; i < n; i++) {
        double c11 = common(i, i);
        if (c11 == 0) c11 = 1;
        for (unsigned j = 0; j < i; j++) {
            double c22 = common(j, j);
            if (c22 == 0) c22 = 1;
            const unsigned c12 = common(i, j);
            const double d1 = 3.0 * (c11 - c12) / c11;
            const double d2 = 3.0 * (c22 - c12) / c22;
            (*dist)[tri(i, j)] = (float)(d1 < d2 ? d1 : d2);
        }
    }
}

void upgmb(unsigned n, std::vector<float>& dist, Tree* tree) {
    std::vector<unsigned> node(n), nearest(n, kNone);
    std::vector<float> mind(n, kBigDist);
    std::vector<unsigned> left(n - 1), right(n - 1);
    std::vector<float> height(n - 1), llen(n - 1), rlen(n - 1);
    for (unsigned i = 0; i < n; i++) node[i] = i;
    for (unsigned i = 1; i < n; i++) {
        for (unsigned j = 0; j < i; j++) {
            const float d = dist[tri(i, j)];
            if (d < mind[i]) { mind[i] = d; nearest[i] = j; }
            if (d < mind[j]) { mind[j] = d; nearest[j] = i; }
        }
    }
    for (unsigned k = 0; k + 1 < n; k++) {
        unsigned lmin = kNone, rmin = kNone;
        float best = kBigDist;
        for (unsigned j = 0; j < n; j++) {
            if (node[j] == kNone) continue;
            if (mind[j] < best) { best = mind[j]; lmin = j; rmin = nearest[j]; }
        }
        float new_min = kBigDist;
        unsigned new_nearest = kNone;
        for (unsigned j = 0; j < n; j++) {
            if (j == lmin || j == rmin || node[j] == kNone) continue;
            const unsigned vl = tri(lmin, j), vr = tri(rmin, j);
            const float dl = dist[vl], dr = dist[vr];
            const float nd = kSueff * ((dl + dr) / 2) + (1 - kSueff) * (dl < dr ? dl : dr);
            if (nearest[j] == rmin) nearest[j] = lmin;
            dist[vl] = nd;
            if (nd < new_min) { new_min = nd; new_nearest = j; }
        }
        const float dlr = dist[tri(lmin, rmin)];
        const float h = dlr / 2;
        const unsigned ul = node[lmin], ur = node[rmin];
        const float hl = ul < n ? 0 : height[ul - n];
        const float hr = ur < n ? 0 : height[ur - n];
        left[k] = ul; right[k] = ur;
        llen[k] = h - hl; rlen[k] = h - hr;
        height[k] = h;
        node[lmin] = n + k;
        nearest[lmin] = new_nearest;
        mind[lmin] = new_min;
        node[rmin] = kNone;
    }
    tree->leaves = n;
    tree->left.assign(2 * n - 1, kNone);
    tree->right.assign(2 * n - 1, kNone);
    tree->parent.assign(2 * n - 1, kNone);
    tree->to_parent.assign(2 * n - 1, 0.0);
    for (unsigned k = 0; k + 1 < n; k++) {
        const unsigned v = n + k;
        tree->left[v] = left[k]; tree->right[v] = right[k];
        tree->parent[left[k]] = v; tree->parent[right[k]] = v;
        tree->to_parent[left[k]] = llen[k]; tree->to_parent[right[k]] = rlen[k];
    }
}

bool clustalw_weights(const Tree& t, std::vector<float>* w) {
    const unsigned n = t.leaves;
    w->assign(n, 0.0f);
    if (n == 1) { (*w)[0] = 1.0f; return true; }
    if (n == 2) { (*w)[0] = (*w)[1] = 0.5f; return true; }
    const unsigned nodes = 2 * n - 1;
    std::vector<unsigned> under(nodes, 0);
    for (unsigned v = 0; v < nodes; v++) {   // children are created before their parent (upgmb), so ascending order works
        if (t.leaf(v)) under[v] = 1; else under[v] = under[t.left[v]] + under[t.right[v]];
    }
    std::vector<double> strength(nodes, 0.0);
    for (unsigned v = 0; v < nodes; v++) {
        if (v == t.root()) continue;
        strength[v] = t.to_parent[v] / (double)under[v];
    }
    for (unsigned l = 0; l < n; l++) {
        double sum = 0;
        for (unsigned v = l; v != t.root(); v = t.parent[v]) sum += strength[v];
        if (sum < 0.0001) sum = 1.0;
        (*w)[l] = (float)sum;
    }
    float total = 0.0;
    for (unsigned l = 0; l < n; l++) total += (*w)[l];
    if (total == 0.0) return false;   // the reference quits here
    for (unsigned l = 0; l < n; l++) (*w)[l] /= total;
    return true;
}

// ---------------------------------------------------------------------------------------------- profiles
struct Msa {                    // column-major: column c holds the ns characters col[c*ns .. c*ns+ns)
    std::vector<unsigned> ids;  // ns sequence ids, row order
    std::vector<char> col;
    size_t ns = 0, nc = 0;
    size_t cols() const { return nc; }
    char at(size_t s, size_t c) const { return col[c * ns + s]; }
};

struct ProfPos {
    unsigned order[4];
    float counts[4];
    float scores[4];
    float open, close;
};

void sort_counts(const float c[4], unsigned order[4]) {   // profilefrommsa.cpp:180-204 (bubble sort, strict <)
    for (unsigned i = 0; i < 4; i++) order[i] = i;
    bool any = true;
    while (any) {
        any = false;
        for (unsigned k = 0; k < 3; k++) {
            const unsigned a = order[k], b = order[k + 1];
            if (c[a] < c[b]) { order[k + 1] = a; order[k] = b; any = true; }
        }
    }
}

void build_profile(const Msa& m, const std::vector<float>& seq_weight, std::vector<ProfPos>* prof) {
    const size_t ns = m.ns, nc = m.nc;
    // msa2.cpp:418-431 + msa.cpp:369-381: this alignment's weights, rescaled to sum 1
    std::vector<float> w(ns);
    float total = 0;
    for (size_t s = 0; s < ns; s++) { w[s] = seq_weight[m.ids[s]]; total += w[s]; }
    if (total != 0) { const float f = 1.0f / total; for (size_t s = 0; s < ns; s++) w[s] *= f; }
    prof->assign(nc, ProfPos());
    // A column's sums run over the sequences in MSA order, each a chain of dependent float additions (the order is part of
    // the result).  Four columns are summed side by side: four independent chains per pass over the sequences.
    constexpr size_t kBlock = 4;
    for (size_t c0 = 0; c0 < nc; c0 += kBlock) {
        const size_t nb = nc - c0 < kBlock ? nc - c0 : kBlock;
        float cnt[kBlock][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        float start[kBlock] = {0, 0, 0, 0}, end[kBlock] = {0, 0, 0, 0};
        auto add = [&](size_t b, size_t s, float ws) {
            const size_t c = c0 + b;
            const char ch = m.at(s, c);
            if (is_gap(ch)) {
                if (c == 0 || !is_gap(m.at(s, c - 1))) start[b] += ws;
                if (c + 1 == nc || !is_gap(m.at(s, c + 1))) end[b] += ws;
                return;
            }
            const uint8_t l = kAlpha.letter[(uint8_t)ch];
            if (l < 4) cnt[b][l] += ws;
            else if (l == 14) { cnt[b][2] += ws / 2; cnt[b][0] += ws / 2; }   // msa2.cpp:66-71: the amino-acid code AX_R (=14) meets NX_X
            else { const float f = ws / 20; for (unsigned k = 0; k < 4; k++) cnt[b][k] += f; }   // msa2.cpp:76-80
        };
        if (nb == kBlock) {
            for (size_t s = 0; s < ns; s++) { const float ws = w[s]; add(0, s, ws); add(1, s, ws); add(2, s, ws); add(3, s, ws); }
        } else {
            for (size_t s = 0; s < ns; s++) { const float ws = w[s]; for (size_t b = 0; b < nb; b++) add(b, s, ws); }
        }
        for (size_t b = 0; b < nb; b++) {
            ProfPos& pp = (*prof)[c0 + b];
            for (unsigned k = 0; k < 4; k++) pp.counts[k] = cnt[b][k];
            sort_counts(pp.counts, pp.order);
            for (unsigned i = 0; i < 4; i++) {
                float sum = 0;
                for (unsigned j = 0; j < 4; j++) sum += pp.counts[j] * kMatrix[i][j];
                pp.scores[i] = sum;
            }
            const float start_occ = (float)(1.0 - start[b]);
            const float end_occ = (float)(1.0 - end[b]);
            pp.open = start_occ * kGapOpen / 2;
            pp.close = end_occ * kGapOpen / 2;
        }
    }
}

inline float match_score(const ProfPos& a, const ProfPos& b) {   // scorepp.cpp:83-95
    float score = 0;
    for (unsigned k = 0; k < 4; k++) {
        const unsigned l = a.order[k];
        const float f = a.counts[l];
        if (f == 0) break;
        score += f * b.scores[l];
    }
    return score - 0.0f;
}

void term_gaps(std::vector<ProfPos>& p) {   // termgaps.cpp:19-33 (TERMGAPS_Half falls through into _Ext)
    ProfPos& first = p.front();
    ProfPos& last = p.back();
    if (first.open != kMinusInf) first.open = 0;
    if (p.size() > 1 && last.close != kMinusInf) last.close = 0;
    if (first.open != kMinusInf) first.open *= -1;
    if (p.size() > 1 && last.close != kMinusInf) last.close *= -1;
}

// ---------------------------------------------------------------------------------------------- pairwise DP
enum : uint8_t { kMM = 0, kDM = 1, kIM = 2, kXM = 3, kMD = 4, kMI = 8 };   // types.h:30-43

bool nw_small(std::vector<ProfPos>& pa, std::vector<ProfPos>& pb, std::string* path) {
    const unsigned la = (unsigned)pa.size(), lb = (unsigned)pb.size();
    if (la == 0 || lb == 0) return false;
    term_gaps(pa);
    term_gaps(pb);
    const float e = kGapExtend;
    const size_t stride = (size_t)lb + 1;
    std::vector<uint8_t> tb((size_t)(la + 1) * stride, 0);
    std::vector<float> b0(stride), b1(stride), b2(stride), drow(stride);
    float *mcurr = b0.data(), *mnext = b1.data(), *mprev = b2.data();
    auto set_m = [&](unsigned i, unsigned j, uint8_t bit) { uint8_t& t = tb[i * stride + j]; t &= (uint8_t)~kXM; t |= bit; };

    // column-wise copies of profile B: the row of match scores is then four multiply-adds over contiguous floats.  A
    // count of 0 ends match_score()'s loop; here its term is added as +-0, which changes nothing (counts are sorted, so
    // the zeros are last), and the terms keep their order.
    std::vector<float> soa((size_t)lb * 6);
    float* sb[4] = {soa.data(), soa.data() + lb, soa.data() + 2 * (size_t)lb, soa.data() + 3 * (size_t)lb};
    float* openb = soa.data() + 4 * (size_t)lb; float* closeb = soa.data() + 5 * (size_t)lb;
    for (unsigned j = 0; j < lb; j++) {
        for (unsigned l = 0; l < 4; l++) sb[l][j] = pb[j].scores[l];
        openb[j] = pb[j].open; closeb[j] = pb[j].close;
    }
    auto score_row = [&](const ProfPos& a, float* out) {        // out[j + 1] = match_score(a, pb[j]) for 1 <= j < lb
        const float f0 = a.counts[a.order[0]], f1 = a.counts[a.order[1]], f2 = a.counts[a.order[2]], f3 = a.counts[a.order[3]];
        const float *s0 = sb[a.order[0]], *s1 = sb[a.order[1]], *s2 = sb[a.order[2]], *s3 = sb[a.order[3]];
        for (unsigned j = 1; j < lb; j++) {
            float sc = 0.0f;
            sc += f0 * s0[j]; sc += f1 * s1[j]; sc += f2 * s2[j]; sc += f3 * s3[j];
            out[j + 1] = sc - 0.0f;
        }
    };

    float iij = kMinusInf;
    for (unsigned j = 0; j <= lb; j++) drow[j] = kMinusInf;
    mprev[0] = 0;
    for (unsigned j = 1; j <= lb; j++) mprev[j] = kMinusInf;
    mcurr[0] = kMinusInf;
    mcurr[1] = match_score(pa[0], pb[0]);
    set_m(1, 1, kMM);
    for (unsigned j = 2; j <= lb; j++) {
        mcurr[j] = match_score(pa[0], pb[j - 1]) + pb[0].open + (j - 2) * e + pb[j - 2].close;
        set_m(1, j, kIM);
    }

#define REC_D(i, j) { \
        const float dd = drow[j] + e; \
        const float md = mprev[j] + pa[(i) - 1].open; \
        if (dd > md) drow[j] = dd; \
        else { drow[j] = md; tbrow[j] |= kMD; } }
#define REC_I(i, j) { \
        iij += e; \
        const float mi = mcurr[(j) - 1] + pb[(j) - 1].open; \
        if (mi >= iij) { iij = mi; tbrow[j] |= kMI; } }

    for (unsigned i = 1; i < la; i++) {
        uint8_t* tbrow = &tb[i * stride];
        iij = kMinusInf;
        drow[0] = pa[0].open + (i - 1) * e;
        mcurr[0] = kMinusInf;
        if (i == 1) { mcurr[1] = match_score(pa[0], pb[0]); set_m(i, 1, kMM); }
        else { mcurr[1] = match_score(pa[i - 1], pb[0]) + pa[0].open + (i - 2) * e + pa[i - 2].close; set_m(i, 1, kDM); }
        score_row(pa[i], mnext);
        uint8_t* tbnext = &tb[(i + 1) * stride];
        const float open_a = pa[i - 1].open, close_a = pa[i - 1].close;
        for (unsigned j = 1; j < lb; j++) {      // REC_D, REC_I and the three-way choice without branches (same tests, same ties)
            const float dd = drow[j] + e;
            const float md = mprev[j] + open_a;
            const bool from_m = !(dd > md);
            const float dj = from_m ? md : dd;
            drow[j] = dj;
            iij += e;
            const float mi = mcurr[j - 1] + openb[j - 1];
            const bool open_i = mi >= iij;
            iij = open_i ? mi : iij;
            tbrow[j] |= (uint8_t)((from_m ? kMD : 0) | (open_i ? kMI : 0));
            const float dm = dj + close_a;
            const float im = iij + closeb[j - 1];
            const float mm = mcurr[j];
            const bool pm = mm >= dm && mm >= im;
            const bool pd = !pm && dm >= mm && dm >= im;
            mnext[j + 1] += pm ? mm : (pd ? dm : im);
            tbnext[j + 1] = (uint8_t)((tbnext[j + 1] & (uint8_t)~kXM) | (pm ? kMM : (pd ? kDM : kIM)));
        }
        REC_D(i, lb)
        REC_I(i, lb)
        float* t = mprev; mprev = mcurr; mcurr = mnext; mnext = t;
    }

    {
        uint8_t* tbrow = &tb[(size_t)la * stride];
        mcurr[0] = kMinusInf;
        if (la > 1) mcurr[1] = match_score(pa[la - 1], pb[0]) + (la - 2) * e + pa[0].open + pa[la - 2].close;
        else mcurr[1] = match_score(pa[la - 1], pb[0]) + pa[0].open + pa[0].close;
        set_m(la, 1, kDM);
        drow[0] = kMinusInf;
        for (unsigned j = 1; j <= lb; j++) REC_D(la, j)
        iij = kMinusInf;
        for (unsigned j = 1; j <= lb; j++) REC_I(la, j)
    }
#undef REC_D
#undef REC_I

    const float mab = mcurr[lb], dab = drow[lb], iab = iij;
    float score = mab;
    char type = 'M';
    if (dab > score) { score = dab; type = 'D'; }
    if (iab > score) { score = iab; type = 'I'; }

    // bittraceback.cpp:130-209
    std::string rev;
    unsigned a = la, b = lb;
    for (;;) {
        rev.push_back(type);
        const uint8_t bits = tb[a * stride + b];
        char next;
        if (type == 'M') {
            const uint8_t x = bits & kXM;
            if (x == kMM) next = 'M'; else if (x == kDM) next = 'D'; else if (x == kIM) next = 'I'; else return false;
            if (a == 0 || b == 0) return false;
            --a; --b;
        } else if (type == 'D') {
            next = (bits & kMD) ? 'M' : 'D';
            if (a == 0) return false;
            --a;
        } else {
            next = (bits & kMI) ? 'M' : 'I';
            if (b == 0) return false;
            --b;
        }
        if (a == 0 && b == 0) break;
        type = next;
    }
    path->assign(rev.rbegin(), rev.rend());
    return true;
}

bool align_two(const Msa& a, const Msa& b, const std::vector<float>& seq_weight, Msa* out) {
    std::vector<ProfPos> pa, pb;
    build_profile(a, seq_weight, &pa);
    build_profile(b, seq_weight, &pb);
    std::string path;
    if (!nw_small(pa, pb, &path)) return false;
    const size_t na = a.ns, nb = b.ns, ns = na + nb;
    out->ids = a.ids;
    out->ids.insert(out->ids.end(), b.ids.begin(), b.ids.end());
    out->ns = ns; out->nc = path.size();
    out->col.assign(ns * path.size(), '-');
    size_t ca = 0, cb = 0;
    for (size_t c = 0; c < path.size(); c++) {   // aligngivenpath.cpp:124-255: a column of A, of B, or of both, stacked
        const char t = path[c];
        char* dst = &out->col[c * ns];
        if (t != 'I') { if (ca >= a.nc) return false; memcpy(dst, &a.col[ca * na], na); ca++; }
        if (t != 'D') { if (cb >= b.nc) return false; memcpy(dst + na, &b.col[cb * nb], nb); cb++; }
    }
    return ca == a.cols() && cb == b.cols();
}

}  // namespace

bool gap_align(const std::vector<std::string>& seqs, std::vector<std::string>* rows) {
    const unsigned n = (unsigned)seqs.size();
    if (n < 2) return false;
    std::vector<std::string> s(seqs);
    for (auto& q : s) {
        if (q.empty()) return false;
        for (auto& ch : q) if (kAlpha.letter[(uint8_t)ch] >= 16) ch = 'N';   // seq.cpp:331-344 (Seq::FixAlpha)
    }
    std::vector<float> dist;
    kmer_distances(s, &dist);
    Tree tree;
    upgmb(n, dist, &tree);
    std::vector<float> weight;
    if (!clustalw_weights(tree, &weight)) return false;

    // progressivealign.cpp:31-72: left-first post-order over the guide tree
    std::vector<Msa> at(2 * n - 1);
    unsigned v = tree.root();
    while (!tree.leaf(v)) v = tree.left[v];
    for (;;) {
        if (tree.leaf(v)) {
            at[v].ids.assign(1, v);
            at[v].col.assign(s[v].begin(), s[v].end());
            at[v].ns = 1; at[v].nc = s[v].size();
        } else {
            Msa& l = at[tree.left[v]];
            Msa& r = at[tree.right[v]];
            if (!align_two(l, r, weight, &at[v])) return false;
            std::vector<char>().swap(l.col);
            std::vector<char>().swap(r.col);
        }
        if (v == tree.root()) break;
        const unsigned p = tree.parent[v];
        if (tree.right[p] == v) { v = p; continue; }
        v = tree.right[p];
        while (!tree.leaf(v)) v = tree.left[v];
    }
    const Msa& fin = at[tree.root()];
    std::vector<std::string> out(n);
    for (size_t k = 0; k < fin.ids.size(); k++) {
        std::string& row = out[fin.ids[k]];
        row.resize(fin.nc);
        for (size_t c = 0; c < fin.nc; c++) row[c] = fin.at(k, c);
    }
    rows->swap(out);
    return true;
}

}  // namespace parsnp

// C entry for the parity tests (tests/test_gapalign.py): sequences joined by '\n' in, aligned rows joined by '\n' out.
extern "C" long parsnp_gap_align(const char* joined, char* out, long cap) {
    std::vector<std::string> seqs;
    std::string cur;
    for (const char* p = joined; *p; p++) { if (*p == '\n') { seqs.push_back(cur); cur.clear(); } else cur.push_back(*p); }
    if (!cur.empty()) seqs.push_back(cur);
    std::vector<std::string> rows;
    if (!parsnp::gap_align(seqs, &rows)) return -1;
    std::string res;
    for (auto& r : rows) { res += r; res.push_back('\n'); }
    if ((long)res.size() + 1 > cap) return -2;
    memcpy(out, res.c_str(), res.size() + 1);
    return (long)res.size();
}
