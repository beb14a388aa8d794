/* oracle/mum_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference hot path (marbl/parsnp, src/csgmum + Aligner::setMums1) used as the
 * CHECKER for the HIP implementation.  It is never linked into, imported by or executed from the product
 * (parsnp_amd/, include/, the parsnp_core replacement); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may touch it.
 *
 * It does NOT rebuild the reference's compressed suffix graph (src/csgmum/csg.c:448-575).  It restates the
 * OBSERVABLE behaviour of the path in a data-structure independent form on top of a plain
 * suffix array:
 *   rep[l]             longest repeated prefix of R[l..)                 <-> uniqueness point / pos_label, mum.c:219-224
 *   events             R-unique maximal exact matches (j,l,len)          <-> Find_UM, mum.c:177-250
 *   Test_UM            per-position best/runner-up in query order        <-> mum.c:27-45
 *   carry              forward propagation along the reference           <-> Intersect_UM, mum.c:125-175
 *   strand fold        forward vs reverse, ties to reverse, in ini order <-> Merge_Master, mum.c:92-123
 *   extraction         multi-MUM candidates from Master                  <-> src/parsnp.cpp:1633-1695
 *
 * PARITY PIN: tests/test_oracle_vs_reference.py checks every function here against the reference's own
 * code (oracle/_ref/libcsgmum_ref.so, built from /root/reference by oracle/Makefile) on seeded random
 * cases and on the MERS example, and tests/golden/ holds vectors generated from that library.
 */
#include "mum_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <limits.h>

/* ------------------------------------------------------------------ suffix array (prefix doubling) */
static const int64_t* g_rank;
static int64_t g_h, g_n;
static int cmp_sa(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    if (g_rank[x] != g_rank[y]) return g_rank[x] < g_rank[y] ? -1 : 1;
    int64_t rx = x + g_h < g_n ? g_rank[x + g_h] : -1, ry = y + g_h < g_n ? g_rank[y + g_h] : -1;
    return rx < ry ? -1 : (rx > ry ? 1 : 0);
}
/* sa[] and rank[] (inverse) of text[0..n); end of string sorts before every symbol. */
static int build_sa(const uint8_t* t, int64_t n, int64_t* sa, int64_t* rank) {
    int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    if (!tmp) return -1;
    for (int64_t i = 0; i < n; i++) { sa[i] = i; rank[i] = t[i]; }
    for (int64_t h = 1;; h <<= 1) {
        g_rank = rank; g_h = h; g_n = n;
        qsort(sa, (size_t)n, sizeof(int64_t), cmp_sa);
        tmp[sa[0]] = 0;
        for (int64_t i = 1; i < n; i++) tmp[sa[i]] = tmp[sa[i - 1]] + (cmp_sa(&sa[i - 1], &sa[i]) < 0);
        memcpy(rank, tmp, sizeof(int64_t) * (size_t)n);
        if (n == 0 || rank[sa[n - 1]] == n - 1 || h > n) break;
    }
    free(tmp);
    return 0;
}

int oracle_rep(const uint8_t* ref, int64_t n, int32_t* rep) {
    if (n <= 0) return 0;
    int64_t* sa = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* rank = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int32_t* lcp = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t)); /* lcp[r] = LCP(sa[r-1], sa[r]) */
    if (!sa || !rank || !lcp || build_sa(ref, n, sa, rank)) return -1;
    int64_t h = 0; /* Kasai */
    for (int64_t i = 0; i < n; i++) {
        if (rank[i] > 0) {
            int64_t j = sa[rank[i] - 1];
            while (i + h < n && j + h < n && ref[i + h] == ref[j + h]) h++;
            lcp[rank[i]] = (int32_t)h;
            if (h > 0) h--;
        } else h = 0;
    }
    for (int64_t i = 0; i < n; i++) {
        int32_t a = lcp[rank[i]], b = rank[i] + 1 < n ? lcp[rank[i] + 1] : 0;
        rep[i] = a > b ? a : b;
    }
    free(sa); free(rank); free(lcp);
    return 0;
}

/* ------------------------------------------------------------------ events (Find_UM restated) */
typedef struct { const uint8_t* r; int64_t n; int64_t* sa; int32_t* rep; } Index;

static int index_build(Index* ix, const uint8_t* ref, int64_t n) {
    ix->r = ref; ix->n = n;
    ix->sa = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    ix->rep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t* rank = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!ix->sa || !ix->rep || !rank) return -1;
    if (n > 0 && build_sa(ref, n, ix->sa, rank)) return -1;
    free(rank);
    return oracle_rep(ref, n, ix->rep);
}
static void index_free(Index* ix) { free(ix->sa); free(ix->rep); }

static int64_t lce(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb) {
    int64_t k = 0, lim = na < nb ? na : nb;
    while (k < lim && a[k] == b[k]) k++;
    return k;
}
/* Longest match of q[0..mq) against any suffix of R: returns length, *pos = a start of it in R. */
static int64_t longest_match(const Index* ix, const uint8_t* q, int64_t mq, int64_t* pos) {
    int64_t lo = 0, hi = ix->n; /* first suffix >= q */
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2, s = ix->sa[mid];
        int64_t k = lce(ix->r + s, ix->n - s, q, mq);
        int less; /* suffix < q ? (end of suffix sorts first) */
        if (k == mq) less = 0;
        else if (s + k == ix->n) less = 1;
        else less = ix->r[s + k] < q[k];
        if (less) lo = mid + 1; else hi = mid;
    }
    int64_t best = 0; *pos = -1;
    for (int64_t c = lo - 1; c <= lo; c++) {
        if (c < 0 || c >= ix->n) continue;
        int64_t s = ix->sa[c], k = lce(ix->r + s, ix->n - s, q, mq);
        if (k > best) { best = k; *pos = s; }
    }
    return best;
}

typedef void (*event_fn)(void* ctx, int64_t j, int64_t l, int64_t len, int32_t rep);

/* All (j,l,len): q[j..j+len) == R[l..l+len), left- and right-maximal, R[l..l+len) unique in R (len > rep[l]),
 * len >= min_len; reported in increasing j (the order Find_UM meets them, mum.c:177-250). */
static void for_each_event(const Index* ix, const uint8_t* q, int64_t m, int min_len, event_fn fn, void* ctx) {
    for (int64_t j = 0; j < m; j++) {
        int64_t l, len = longest_match(ix, q + j, m - j, &l);
        if (len < 1 || len < min_len) continue;
        if (len <= ix->rep[l]) continue;                       /* not unique in R */
        if (j > 0 && l > 0 && q[j - 1] == ix->r[l - 1]) continue; /* not left-maximal: suffix of a longer event */
        fn(ctx, j, l, len, ix->rep[l]);
    }
}

typedef struct { int32_t* UP; int32_t* EP; int64_t* SP; } Pair;
/* Test_UM (mum.c:27-45): MMP = l+len, pos_label = l+rep[l] (mum.c:219-224), SP = query start. */
static void test_um(void* ctx, int64_t j, int64_t l, int64_t len, int32_t rep) {
    Pair* p = (Pair*)ctx;
    int32_t mmp = (int32_t)(l + len), label = (int32_t)(l + rep);
    if (p->EP[l] == 0) { p->EP[l] = mmp; p->UP[l] = label; p->SP[l] = j; return; }
    if (mmp > p->UP[l]) {
        if (mmp > p->EP[l]) { p->UP[l] = p->EP[l]; p->EP[l] = mmp; p->SP[l] = j; }
        else p->UP[l] = mmp;
    }
}

static int find_um_ix(const Index* ix, const uint8_t* q, int64_t m, int min_len, int32_t* UP, int32_t* EP, int64_t* SP) {
    for (int64_t i = 0; i < ix->n; i++) { UP[i] = 0; EP[i] = 0; SP[i] = 0; }
    Pair p = {UP, EP, SP};
    for_each_event(ix, q, m, min_len, test_um, &p);
    return 0;
}

int oracle_find_um(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_len,
                   int32_t* UP, int32_t* EP, int64_t* SP) {
    Index ix;
    if (index_build(&ix, ref, n)) return -1;
    find_um_ix(&ix, query, m, min_len, UP, EP, SP);
    index_free(&ix);
    return 0;
}

typedef struct { int64_t cap, cnt; int64_t* j; int64_t* l; int32_t* len; int32_t* rep; } EvList;
static void push_event(void* ctx, int64_t j, int64_t l, int64_t len, int32_t rep) {
    EvList* e = (EvList*)ctx;
    if (e->cnt < e->cap) { e->j[e->cnt] = j; e->l[e->cnt] = l; e->len[e->cnt] = (int32_t)len; e->rep[e->cnt] = rep; }
    e->cnt++;
}
int64_t oracle_events(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_len,
                      int64_t cap, int64_t* ev_j, int64_t* ev_l, int32_t* ev_len, int32_t* ev_rep) {
    Index ix;
    if (index_build(&ix, ref, n)) return -1;
    EvList e = {cap, 0, ev_j, ev_l, ev_len, ev_rep};
    for_each_event(&ix, query, m, min_len, push_event, &e);
    index_free(&ix);
    return e.cnt;
}

/* ------------------------------------------------------------------ carry (Intersect_UM restated)
 * Two-value carry along the reference: with (U,E) the propagated values at k-1 and (u,e) the raw values at k,
 *   e > E  : the match starting at k reaches furthest -> (max(u,E), e), SP = own
 *   else   : the earlier match still wins            -> (max(U,e), E), SP = SP[k-1]+1
 * (U,E) == (UP,EP) of mum.c:125-175 everywhere; SP agrees wherever UP < EP (elsewhere it is never read). */
int oracle_propagate(int64_t n, int32_t* UP, int32_t* EP, int64_t* SP) {
    for (int64_t k = 1; k < n; k++) {
        if (EP[k] > EP[k - 1]) { if (EP[k - 1] > UP[k]) UP[k] = EP[k - 1]; }
        else { UP[k] = UP[k - 1] > EP[k] ? UP[k - 1] : EP[k]; EP[k] = EP[k - 1]; SP[k] = SP[k - 1] + 1; }
    }
    return 0;
}

/* ------------------------------------------------------------------ one region of setMums1 */
static uint8_t comp(uint8_t c) { /* Aligner::reversec, src/parsnp.cpp:1294-1393, on ingested symbols ACGTN */
    switch (toupper(c)) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                          case 'U': return 'T'; default: return 'N'; }
}

int oracle_multi_mum(int cnt, const uint8_t* const* seqs, const int64_t* lens, int minsize, int min_event_len,
                     int64_t* out_c, int64_t** out_k, int32_t** out_lon, int64_t** out_sp, uint8_t** out_fwd,
                     int32_t* masterUP, int32_t* masterEP) {
    int64_t n = lens[0];
    int q = cnt - 1;
    Index ix;
    if (index_build(&ix, seqs[0], n)) return -1;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int32_t *mUP = calloc(nn, 4), *mEP = malloc(nn * 4), *fUP = malloc(nn * 4), *fEP = malloc(nn * 4),
            *rUP = malloc(nn * 4), *rEP = malloc(nn * 4);
    int64_t *fSP = malloc(nn * 8), *rSP = malloc(nn * 8);
    int64_t* SPg = malloc(nn * 8 * (size_t)(q > 0 ? q : 1));   /* per query: chosen SP */
    uint8_t* FWg = malloc(nn * (size_t)(q > 0 ? q : 1));       /* per query: chosen strand */
    for (int64_t k = 0; k < n; k++) mEP[k] = (int32_t)n;       /* src/parsnp.cpp:1591-1597 */
    for (int g = 0; g < q; g++) {                              /* src/parsnp.cpp:1600-1619, strictly in ini order */
        int64_t m = lens[g + 1];
        uint8_t* rc = malloc((size_t)(m > 0 ? m : 1));
        for (int64_t i = 0; i < m; i++) rc[i] = comp(seqs[g + 1][m - 1 - i]);
        find_um_ix(&ix, seqs[g + 1], m, min_event_len, fUP, fEP, fSP);
        find_um_ix(&ix, rc, m, min_event_len, rUP, rEP, rSP);
        oracle_propagate(n, fUP, fEP, fSP);
        oracle_propagate(n, rUP, rEP, rSP);
        for (int64_t k = 0; k < n; k++) {
            /* Intersect_UM's fold (mum.c:162-163) on each strand, then Merge_Master (mum.c:92-123): ties -> reverse */
            int32_t fE = mEP[k] < fEP[k] ? mEP[k] : fEP[k], fU = mUP[k] > fUP[k] ? mUP[k] : fUP[k];
            int32_t rE = mEP[k] < rEP[k] ? mEP[k] : rEP[k], rU = mUP[k] > rUP[k] ? mUP[k] : rUP[k];
            if (fE > rE) { mEP[k] = fE; mUP[k] = fU; SPg[(size_t)g * nn + k] = fSP[k]; FWg[(size_t)g * nn + k] = 1; }
            else         { mEP[k] = rE; mUP[k] = rU; SPg[(size_t)g * nn + k] = rSP[k]; FWg[(size_t)g * nn + k] = 0; }
        }
        free(rc);
    }
    /* extraction, src/parsnp.cpp:1633-1695 */
    int64_t c = 0, capc = 16;
    int64_t* K = malloc(sizeof(int64_t) * capc); int32_t* L = malloc(sizeof(int32_t) * capc);
    int32_t prevEP = 0;
    for (int64_t k = 0; k < n; k++) {
        if (mEP[k] > prevEP && mUP[k] < mEP[k] && mEP[k] - k >= minsize) {
            if (c == capc) { capc *= 2; K = realloc(K, sizeof(int64_t) * capc); L = realloc(L, sizeof(int32_t) * capc); }
            K[c] = k; L[c] = (int32_t)(mEP[k] - k); c++;
        }
        prevEP = mEP[k];
    }
    int64_t* S = malloc(sizeof(int64_t) * (size_t)(c * q + 1)); uint8_t* F = malloc((size_t)(c * q + 1));
    for (int64_t i = 0; i < c; i++)
        for (int g = 0; g < q; g++) { S[i * q + g] = SPg[(size_t)g * nn + K[i]]; F[i * q + g] = FWg[(size_t)g * nn + K[i]]; }
    if (masterUP) memcpy(masterUP, mUP, (size_t)n * 4);
    if (masterEP) memcpy(masterEP, mEP, (size_t)n * 4);
    *out_c = c; *out_k = K; *out_lon = L; *out_sp = S; *out_fwd = F;
    free(mUP); free(mEP); free(fUP); free(fEP); free(rUP); free(rEP); free(fSP); free(rSP); free(SPg); free(FWg);
    index_free(&ix);
    return 0;
}
void oracle_free(void* p) { free(p); }

/* ------------------------------------------------------------------ calcmumi, one query genome (src/parsnp.cpp:1991-2069) */
int64_t oracle_mumi_coverage(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_event_len) {
    Index ix;
    if (index_build(&ix, ref, n)) return -1;
    size_t nn = (size_t)(n > 0 ? n : 1);
    int32_t *fUP = malloc(nn * 4), *fEP = malloc(nn * 4), *rUP = malloc(nn * 4), *rEP = malloc(nn * 4);
    int64_t *fSP = malloc(nn * 8), *rSP = malloc(nn * 8);
    uint8_t* mark = calloc(nn, 1);
    uint8_t* rc = malloc((size_t)(m > 0 ? m : 1));
    for (int64_t i = 0; i < m; i++) rc[i] = comp(query[m - 1 - i]);
    find_um_ix(&ix, query, m, min_event_len, fUP, fEP, fSP);
    find_um_ix(&ix, rc, m, min_event_len, rUP, rEP, rSP);
    oracle_propagate(n, fUP, fEP, fSP);
    oracle_propagate(n, rUP, rEP, rSP);
    int32_t last = 0;                                   /* M_EP1: EP of the last ACCEPTED position (:2044-2047) */
    for (int64_t k = 0; k < n; k++) {
        /* Master starts at (UP 0, EP n): the fold of :2017-2019 leaves the strand with the larger EP, ties to reverse */
        int32_t fE = fEP[k] < n ? fEP[k] : (int32_t)n, rE = rEP[k] < n ? rEP[k] : (int32_t)n;
        int32_t EP = fE > rE ? fE : rE, UP = fE > rE ? fUP[k] : rUP[k];
        if (EP > last && UP < EP && EP - k < n) {
            last = EP;
            if (EP - k >= 15) for (int64_t x = k; x < EP; x++) mark[x] = 1;
        }
    }
    int64_t cov = 0;
    for (int64_t k = 0; k < n; k++) cov += mark[k];
    free(fUP); free(fEP); free(rUP); free(rEP); free(fSP); free(rSP); free(mark); free(rc);
    index_free(&ix);
    return cov;
}

/* ------------------------------------------------------------------ minimum MUM length (Converter/Calculator restated)
 * float32 arithmetic throughout, Log(x) = float( double(logf(x)) / log(2.0) ) (src/Converter.cpp:268-270),
 * result ceil()ed in float (Converter.cpp:283-284) and again by the caller (src/parsnp.cpp:1506,1513). */
typedef struct { const char* s; int err; float S; } Px;
static void skipws(Px* p) { while (*p->s == ' ' || *p->s == '\t') p->s++; }
static float p_expr(Px* p);
static float p_base(Px* p) {
    skipws(p);
    if (*p->s == '(') { p->s++; float v = p_expr(p); skipws(p); if (*p->s == ')') p->s++; else p->err = 1; return v; }
    if (isdigit((unsigned char)*p->s)) {
        const char* b = p->s;
        while (isdigit((unsigned char)*p->s)) { p->s++; if (*p->s == '.') p->s++; }
        char buf[64]; size_t len = (size_t)(p->s - b); if (len > 63) len = 63;
        memcpy(buf, b, len); buf[len] = 0;
        return (float)atof(buf);
    }
    if (*p->s == 'S' || *p->s == 's') { p->s++; return p->S; }
    if (p->s[0] == 'L' && p->s[1] == 'o' && p->s[2] == 'g') {
        p->s += 3; skipws(p);
        if (*p->s != '(') { p->err = 1; return 0; }
        float x = p_base(p);
        return (float)((double)logf(x) / log(2.0));
    }
    p->err = 1; return 0;
}
static float p_pow(Px* p) { float v = p_base(p); for (skipws(p); *p->s == '^'; skipws(p)) { p->s++; float x = p_base(p); v = powf(v, x); } return v; }
static float p_term(Px* p) {
    float v = p_pow(p);
    for (skipws(p); *p->s == '*' || *p->s == '/'; skipws(p)) {
        char op = *p->s++; float x = p_pow(p);
        if (op == '*') v = v * x; else { if (x == 0) { p->err = 1; return 0; } v = v / x; }
    }
    return v;
}
static float p_expr(Px* p) {
    float v = p_term(p);
    for (skipws(p); *p->s == '+' || *p->s == '-'; skipws(p)) { char op = *p->s++; float x = p_term(p); v = op == '+' ? v + x : v - x; }
    return v;
}
int32_t oracle_min_length(const char* expr, int64_t S) {
    Px p = {expr, 0, (float)S};
    float v = p_expr(&p);
    skipws(&p);
    if (p.err || *p.s) return INT32_MIN;
    return (int32_t)ceil((double)ceilf(v));
}
