// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C API around the REFERENCE's own csgmum code.  The two reference files are textually
// #included from /root/reference (via -I$(REF)/src) exactly the way src/parsnp.cpp:73-78 does;
// nothing of them is copied into this repository, and the resulting library lives in
// oracle/_ref/ (git-ignored).  The orchestration below restates the call sequence of
// Aligner::setMums1 (src/parsnp.cpp:1570-1695) so that tests can ask the real reference for
//   (a) the raw Find_UM output of one strand            -> ref_find_um
//   (b) the propagated (Intersect_UM) arrays             -> ref_find_um_propagated
//   (c) the multi-MUM candidate list of one region       -> ref_multi_mum
// Used by tests/ to pin oracle/mum_oracle.c and by tests/golden/make_golden.py.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

extern "C" {
#include "csgmum/csg.c"
#include "csgmum/mum.c"
}

namespace {
struct Graph {
    CSG* csg = nullptr;
    std::string text;  // ref + '\x05' (kept alive: csg->seq points into it)
    Graph(const char* ref, long n, int factor) : text(ref, n) {
        text.push_back((char)5);                                   // src/parsnp.cpp:1542
        csg = new_CSG(csg, (ulong)factor * (ulong)n, text.c_str(), n, 0);   // :1570
        build_CSG(csg, text.c_str(), n, 0);                        // :1571
        find_leaves(csg);                                          // :1572
    }
    ~Graph() { free_CSG(csg); }
};
std::string term(const char* s, long m) { std::string t(s, m); t.push_back((char)5); return t; }
}  // namespace

extern "C" {

// Raw Find_UM of one query strand against ref (src/csgmum/mum.c:177-250). Arrays sized n, zeroed here.
int ref_find_um(const char* ref, long n, const char* query, long m, int factor,
                int* UP, int* EP, unsigned long* SP) {
    Graph g(ref, n, factor);
    std::vector<UM> pair(n);
    for (long i = 0; i < n; i++) { pair[i].UP = pair[i].EP = 0; SP[i] = 0; }
    std::string q = term(query, m);
    Find_UM(g.csg, q.c_str(), SP, pair.data());
    for (long i = 0; i < n; i++) { UP[i] = pair[i].UP; EP[i] = pair[i].EP; }
    return 0;
}

// Find_UM followed by Intersect_UM into a fresh Master (UP=0, EP=n) (src/parsnp.cpp:1591-1597,1614-1616).
int ref_find_um_propagated(const char* ref, long n, const char* query, long m, int factor,
                           int* UP, int* EP, unsigned long* SP) {
    Graph g(ref, n, factor);
    std::vector<UM> pair(n), master(n);
    for (long i = 0; i < n; i++) { pair[i].UP = pair[i].EP = 0; master[i].UP = 0; master[i].EP = (int)n; SP[i] = 0; }
    std::string q = term(query, m);
    Find_UM(g.csg, q.c_str(), SP, pair.data());
    Intersect_UM(g.csg, master.data(), pair.data(), (int)n, SP);
    for (long i = 0; i < n; i++) { UP[i] = master[i].UP; EP[i] = master[i].EP; }
    return 0;
}

// One region of setMums1: seqs[0] = reference substring, seqs[1..cnt-1] = query substrings,
// rcs[i] = reverse complement of seqs[i] (Aligner::reversec).  Outputs (malloc'ed, free with ref_free):
//   k[c], lon[c], sp[c*(cnt-1)], fwd[c*(cnt-1)]  for the c candidates of src/parsnp.cpp:1633-1695,
//   and optionally the final Master arrays (mUP/mEP may be NULL).
int ref_multi_mum(int cnt, const char** seqs, const long* lens, const char** rcs, int minsize, int factor,
                  long* out_c, long** out_k, int** out_lon, unsigned long** out_sp, char** out_fwd,
                  int* mUP, int* mEP) {
    long n = lens[0];
    Graph g(seqs[0], n, factor);
    std::vector<UM> Master(n), MasterRC(n), Pair(n), PairRC(n);
    for (long i = 0; i < n; i++) {                                  // :1591-1597 (+ MasterRC.UP := 0, see zero_new.cpp)
        Pair[i].UP = Pair[i].EP = Master[i].UP = 0;
        PairRC[i].UP = PairRC[i].EP = 0;
        Master[i].EP = (int)n; MasterRC[i].EP = (int)n; MasterRC[i].UP = 0;
    }
    std::vector<SP> SPF(cnt - 1), SPR(cnt - 1);
    std::vector<std::vector<ulong>> msp(cnt - 1, std::vector<ulong>(n, 0));
    std::vector<std::vector<char>> ff(cnt - 1, std::vector<char>(n, 1)), fr(cnt - 1, std::vector<char>(n, 0));
    std::vector<ulong> tempMSP(n, 0);
    ulong two[2] = {0, 0};
    for (int i = 0; i < cnt - 1; i++) {                             // :1600-1619
        SPF[i].MSP = msp[i].data(); SPF[i].forward = ff[i].data();
        SPR[i].MSP = two;           SPR[i].forward = fr[i].data();
        for (long j = 0; j < n; j++) tempMSP[j] = 0;
        std::string q = term(seqs[i + 1], lens[i + 1]), qr = term(rcs[i + 1], lens[i + 1]);
        Find_UM(g.csg, q.c_str(), SPF[i].MSP, Pair.data());
        Find_UM(g.csg, qr.c_str(), tempMSP.data(), PairRC.data());
        Intersect_UM(g.csg, Master.data(), Pair.data(), (int)n, SPF[i].MSP);
        Intersect_UM(g.csg, MasterRC.data(), PairRC.data(), (int)n, tempMSP.data());
        Merge_Master(Master.data(), MasterRC.data(), (int)n, (int)lens[i + 1], SPF.data(), SPR.data(), tempMSP.data(), i);
    }
    std::vector<long> K; std::vector<int> L; std::vector<ulong> S; std::vector<char> F;
    int M_EP = 0;
    for (long k = 0; k < n; k++) {                                   // :1633-1695
        if (Master[k].EP > M_EP && Master[k].UP < Master[k].EP) {
            M_EP = Master[k].EP;
            int lon = M_EP - (int)k;
            if (lon >= minsize) {
                K.push_back(k); L.push_back(lon);
                for (int j = 1; j < cnt; j++) { F.push_back(SPF[j - 1].forward[k]); S.push_back(SPF[j - 1].MSP[k]); }
            }
        }
        M_EP = Master[k].EP;
    }
    if (mUP) for (long i = 0; i < n; i++) mUP[i] = Master[i].UP;
    if (mEP) for (long i = 0; i < n; i++) mEP[i] = Master[i].EP;
    *out_c = (long)K.size();
    *out_k = (long*)malloc(sizeof(long) * (K.size() + 1));
    *out_lon = (int*)malloc(sizeof(int) * (K.size() + 1));
    *out_sp = (unsigned long*)malloc(sizeof(unsigned long) * (S.size() + 1));
    *out_fwd = (char*)malloc(F.size() + 1);
    memcpy(*out_k, K.data(), sizeof(long) * K.size());
    memcpy(*out_lon, L.data(), sizeof(int) * L.size());
    memcpy(*out_sp, S.data(), sizeof(unsigned long) * S.size());
    memcpy(*out_fwd, F.data(), F.size());
    return 0;
}

// calcmumi for one query genome: the body of the OpenMP loop of Aligner::setMumi (src/parsnp.cpp:1991-2069).
long ref_mumi_coverage(const char* ref, long n, const char* query, const char* query_rc, long m, int factor) {
    Graph g(ref, n, factor);
    std::vector<UM> Master(n), MasterRC(n), Pair(n), PairRC(n);
    std::vector<ulong> msp(n, 0), tempMSP(n, 0);
    std::vector<char> ff(n, 1), fr(n, 0);
    SP SPF[1], SPR[1];
    ulong two[2] = {0, 0};
    SPF[0].MSP = msp.data(); SPF[0].forward = ff.data(); SPR[0].MSP = two; SPR[0].forward = fr.data();
    for (long i = 0; i < n; i++) {
        Pair[i].UP = Pair[i].EP = Master[i].UP = 0; PairRC[i].UP = PairRC[i].EP = 0;
        Master[i].EP = (int)n; MasterRC[i].EP = (int)n; MasterRC[i].UP = 0;
    }
    std::string q = term(query, m), qr = term(query_rc, m);
    Find_UM(g.csg, q.c_str(), SPF[0].MSP, Pair.data());
    Find_UM(g.csg, qr.c_str(), tempMSP.data(), PairRC.data());
    Intersect_UM(g.csg, Master.data(), Pair.data(), (int)n, SPF[0].MSP);
    Intersect_UM(g.csg, MasterRC.data(), PairRC.data(), (int)n, tempMSP.data());
    Merge_Master(Master.data(), MasterRC.data(), (int)n, (int)m, SPF, SPR, tempMSP.data(), 0);
    std::vector<char> amums(n, 0);
    int M_EP1 = 0;
    for (long k = 0; k < n; k++) {
        if (Master[k].EP > M_EP1 && Master[k].UP < Master[k].EP && Master[k].EP - k < n) {
            M_EP1 = Master[k].EP;
            int lon = M_EP1 - (int)k;
            if (lon >= 15) for (int ii = 0; ii < lon; ii++) amums[k + ii] = 1;
        }
    }
    long total = 0;
    for (long k = 0; k < n; k++) total += amums[k];
    return total;
}

void ref_free(void* p) { free(p); }

}  // extern "C"
