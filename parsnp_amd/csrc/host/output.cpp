// output.cpp -- parsnpAligner.xmfa + parsnpAligner.log, the output contract of the reference's
// Aligner::writeOutput (src/parsnp.cpp:505-1191).  MUM columns are lower case, inter-MUM gap columns upper case.
// Gaps in which every genome has >= 1 base and some genome has >= 2 go to libMUSCLE in the reference
// (:790-865, src/MuscleInterface.cpp:37-78); here they go to gapalign.cpp, the restatement of that aligner.  Should it
// ever decline an input, the gap is emitted left-justified and '-'-padded and the log carries a note.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <future>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <sys/mman.h>
#include <unistd.h>

#include "aligner.h"
#include "hooks.h"
#include "gapalign.h"

namespace parsnp {

namespace {
std::string upper(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::toupper); return s; }
std::string sub(const std::string& g, long pos, long len) {   // std::string::substr semantics incl. size_t wrap of len
    if (pos < 0 || (size_t)pos > g.size()) { std::cerr << "parsnp_core: substr out of range" << std::endl; exit(1); }
    return g.substr((size_t)pos, (size_t)len);
}

// rows of one LCB: MUMs interleaved with the gaps between consecutive MUMs (:663-916), in three steps so that the gap
// alignments of ALL LCBs can be spread over the threads (one long LCB holds hundreds of them):
//   gaps_to_align()  the gaps the reference would hand to MUSCLE (the others are padded on the fly)
//   gap_align()      per gap, by the caller, in one flat parallel loop
//   build_rows()     concatenation
struct Gap {
    std::vector<std::string> seq;       // per genome
    std::vector<std::string> aligned;   // filled when `align`
    unsigned max_len = 0;
    bool align = false, failed = false;
};

// the gap between MUM t and MUM t+1 of an LCB (gp is reused by the caller: no allocation in the common one-column case)
void gap_between(const Aligner& a, const Lcb& ct, size_t t, Gap* gp) {
    const size_t n = a.n;
    const Mum& first = a.pool[(size_t)ct.mums[0]];
    const Mum& m = a.pool[(size_t)ct.mums[t]];
    const Mum& nx = a.pool[(size_t)ct.mums[t + 1]];
    gp->seq.resize(n);
    unsigned max_len = 0, min_len = 1000000;
    for (size_t i = 0; i < n; i++) {
        const std::string& g = a.genomes[i].seq;
        if (!first.fwd[i]) {
            if (m.start[i] - nx.end(i) >= 1) gp->seq[i] = upper(reverse_complement(sub(g, nx.end(i), m.start[i] - nx.end(i))));
            else gp->seq[i].clear();
        } else {
            gp->seq[i] = upper(sub(g, m.end(i), nx.start[i] - m.end(i)));
        }
        if (gp->seq[i].size() > max_len) max_len = (unsigned)gp->seq[i].size();
        if (gp->seq[i].size() < min_len) min_len = (unsigned)gp->seq[i].size();
    }
    gp->max_len = max_len;
    // reference: MUSCLE(maxiters=1) over the n gap strings (:848-861); a single genome is never aligned (:809-826)
    gp->align = max_len > 1 && min_len > 0 && n > 1;
    gp->failed = false;
}

// the gaps of an LCB that go to the aligner, in order, each with its position t
void gaps_to_align(const Aligner& a, const Lcb& ct, std::vector<std::pair<size_t, Gap>>* out) {
    Gap gp;
    for (size_t t = 0; t + 1 < ct.mums.size(); t++) {
        gap_between(a, ct, t, &gp);
        if (gp.align) out->emplace_back(t, gp);
    }
}

void build_rows(const Aligner& a, const Lcb& ct, const std::vector<std::pair<size_t, Gap>>& aligned, std::vector<std::string>* rows, bool* gap_note) {
    const size_t n = a.n;
    rows->assign(n, "");
    {   // final row length, roughly: reference span plus a little for gap columns (avoids repeated regrowth)
        const long span = ct.end[0] - ct.start[0];
        const size_t guess = span > 0 ? (size_t)span + (size_t)span / 16 + 64 : 64;
        for (size_t i = 0; i < n; i++) (*rows)[i].reserve(guess);
    }
    const Mum& first = a.pool[(size_t)ct.mums[0]];
    // a MUM's text, lower case, the reverse complement for a reverse member (= lower(reverse_complement(sub(...))), written
    // straight into the row: 14 million of these per run at 200 x 5 Mb)
    static const struct Tables { char low[256], rc[256]; Tables() {
        for (int c = 0; c < 256; c++) { low[c] = (char)tolower(c); rc[c] = 'n'; }
        rc[(unsigned char)'A'] = rc[(unsigned char)'a'] = 't'; rc[(unsigned char)'C'] = rc[(unsigned char)'c'] = 'g';
        rc[(unsigned char)'G'] = rc[(unsigned char)'g'] = 'c'; rc[(unsigned char)'T'] = rc[(unsigned char)'t'] = 'a';
        rc[(unsigned char)'U'] = rc[(unsigned char)'u'] = 't';
        for (char c : {'\r', '\n', '\t', ' ', '>', '.'}) rc[(unsigned char)c] = 0;       // dropped by reversec (parsnp.cpp:1369-1380)
    } } T;
    auto append_mum = [&](std::string& dst, const Mum& m, size_t i) {
        const std::string& g = a.genomes[i].seq;
        const long pos = m.start[i];
        if (pos < 0 || (size_t)pos > g.size()) { std::cerr << "parsnp_core: substr out of range" << std::endl; exit(1); }
        const size_t len = std::min<size_t>((size_t)m.length, g.size() - (size_t)pos);     // substr clamps
        const size_t at = dst.size();
        if (first.fwd[i]) {
            dst.resize(at + len);
            const char* src = g.data() + pos; char* out = &dst[at];
            for (size_t x = 0; x < len; x++) out[x] = T.low[(unsigned char)src[x]];
        } else {
            dst.reserve(at + len);
            const char* src = g.data() + pos;
            for (size_t x = len; x-- > 0;) { const char c = T.rc[(unsigned char)src[x]]; if (c) dst.push_back(c); }
        }
    };
    Gap gp;
    size_t next_aligned = 0;
    for (size_t t = 0; t < ct.mums.size(); t++) {
        const Mum& m = a.pool[(size_t)ct.mums[t]];
        for (size_t i = 0; i < n; i++) append_mum((*rows)[i], m, i);
        if (t + 1 == ct.mums.size()) break;
        if (next_aligned < aligned.size() && aligned[next_aligned].first == t) {
            const Gap& ag = aligned[next_aligned++].second;
            if (!ag.failed) { for (size_t i = 0; i < n; i++) (*rows)[i] += ag.aligned[i]; continue; }
            *gap_note = true;
        }
        gap_between(a, ct, t, &gp);
        if (gp.max_len > 0)
            for (size_t i = 0; i < n; i++) (*rows)[i] += gp.seq[i] + std::string(gp.max_len - gp.seq[i].size(), '-');
    }
}
}  // namespace

void write_output(Aligner& a, const std::string& stem, bool* gap_note) {
    using namespace std;
    const size_t n = a.n;
    const Params& prm = a.prm;
    if (prm.do_align) cerr << "Writing output files & aligning LCBs..." << endl;
    const string dir = prm.outdir + "/";
    {
        ofstream probe((dir + "parsnpAligner.log").c_str());
        if (!probe.good()) {
            if (system(("mkdir " + prm.outdir).c_str())) {
                cerr << "ParSNP:: error creating output directory, exiting.." << endl;
                exit(1);
            }
        }
    }
    if (prm.recomb_filter) { int rc = system(("mkdir " + dir + "blocks/").c_str()); (void)rc; }
    ofstream xmfa((dir + stem + ".xmfa").c_str());
    ofstream log((dir + stem + ".log").c_str());
    { ofstream allmums("allmums.out"); }   // the reference leaves an empty file in the cwd (:600-601)

    int printable = 0;
    for (const Lcb& c : a.lcbs) if (c.type == 1) printable++;
    xmfa << "#FormatVersion Mauve" << endl;
    xmfa << "#SequenceCount " << (int)n << endl;
    for (size_t i = 0; i < n; i++) {
        xmfa << "##SequenceIndex " << i + 1 << endl;
        xmfa << "##SequenceFile " << a.genomes[i].fname << endl;
        xmfa << "##SequenceHeader " << a.genomes[i].header << endl;
        xmfa << "##SequenceLength " << a.genomes[i].size_nopad << "bp" << endl;
    }
    xmfa << "#IntervalCount " << printable << endl;

    // Every printable LCB is laid out first (which gaps are aligned, how many columns each contributes), the aligned
    // gaps of ALL LCBs go to the device in one batch, and then every record is generated straight into its place in the
    // file: with the column counts known, the size of every record -- and so the file offset of every row -- is known
    // before a single base is formatted, and the threads stream the rows (MUM text from the genomes, gap rows from the
    // aligner's output) through small buffers and positioned writes.  Nothing of the 1 GB text (200 x 5 Mb) is held in
    // memory.  An LCB the stream cannot print as it stands -- one that overlaps the previous printed LCB on the reference
    // (the trim of :928-952 works on the finished rows), or whose coordinates would make std::string::substr clamp or
    // wrap -- is built as strings, the way the reference does it ("slow" below); so is everything with recombfilter
    // (blocks/b<k>/seq.fna wants the records twice) or PARSNP_PLAIN_OUTPUT=1 (test hook: both ways give the same file).
    const long nl = (long)a.lcbs.size();
    const int threads = prm.cores > 0 ? prm.cores : 1;
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    auto clock_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = clock_s();
    auto lap = [&](const char* what) { if (dbg) { double t = clock_s(); fprintf(stderr, "[output] %-14s %.4f s\n", what, t - tl); tl = t; } };
    auto printable_lcb = [&](const Lcb& ct) { return ct.type == 1 && !ct.mums.empty() && prm.do_align != 0; };
    static const bool plain_output = test_hook("PARSNP_PLAIN_OUTPUT") != nullptr;
    const bool all_slow = plain_output || prm.recomb_filter;

    // ---- layout of every LCB
    struct Plan {
        bool regular = false;              // the stream can print it (unless it turns out to need the overlap trim)
        std::vector<int32_t> gmax;         // per gap: longest gap string
        std::vector<int32_t> gjob;         // per gap: index into `jobs` when the gap is aligned, else -1
        std::vector<int32_t> gcols;        // per gap: columns in the rows
        long cols = 0;                     // row length
    };
    vector<Plan> plan((size_t)nl);
    auto gap_length = [&](const Mum& first, const Mum& m, const Mum& nx, size_t i, bool* clean) -> long {
        const long gs = (long)a.genomes[i].seq.size();
        if (!first.fwd[i]) {
            const long pos = nx.end(i), len = m.start[i] - nx.end(i);
            if (pos < 0 || pos > gs) { *clean = false; return 0; }
            if (len < 1) return 0;
            if (pos + len > gs) *clean = false;
            return len;
        }
        const long pos = m.end(i), len = nx.start[i] - m.end(i);
        if (pos < 0 || pos > gs || len < 0 || pos + len > gs) { *clean = false; return 0; }
        return len;
    };
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (long z = 0; z < nl; z++) {
        const Lcb& ct = a.lcbs[(size_t)z];
        if (!printable_lcb(ct)) continue;
        Plan& pl = plan[(size_t)z];
        const size_t T = ct.mums.size();
        pl.gmax.assign(T - 1, 0); pl.gjob.assign(T - 1, -1); pl.gcols.assign(T - 1, 0);
        const Mum& first = a.pool[(size_t)ct.mums[0]];
        bool clean = true;
        for (size_t t = 0; t < T; t++) {
            const Mum& m = a.pool[(size_t)ct.mums[t]];
            for (size_t i = 0; i < n; i++) {
                const long pos = m.start[i];
                if (pos < 0 || m.length < 0 || pos + m.length > (long)a.genomes[i].seq.size()) clean = false;
            }
            if (t + 1 == T) break;
            const Mum& nx = a.pool[(size_t)ct.mums[t + 1]];
            long mx = 0, mn = 1000000;
            for (size_t i = 0; i < n; i++) { const long l = gap_length(first, m, nx, i, &clean); mx = std::max(mx, l); mn = std::min(mn, l); }
            pl.gmax[t] = (int32_t)mx;
            // reference: MUSCLE(maxiters=1) over the n gap strings (:848-861); a single genome is never aligned (:809-826)
            if (mx > 1 && mn > 0 && n > 1) pl.gjob[t] = 0;      // (numbered below)
        }
        pl.regular = clean;
    }
    // ---- the gaps that are aligned
    struct Job { size_t z, t; unsigned max_len; long dev = -1; int grp = 0; Gap host; bool on_host = false, failed = false; };
    vector<Job> jobs;
    for (long z = 0; z < nl; z++) {
        Plan& pl = plan[(size_t)z];
        if (!pl.regular) continue;                               // (an irregular LCB aligns its own gaps, below)
        for (size_t t = 0; t < pl.gjob.size(); t++)
            if (pl.gjob[t] == 0) { Job j; j.z = (size_t)z; j.t = t; j.max_len = (unsigned)pl.gmax[t]; jobs.push_back(std::move(j)); }
            else pl.gjob[t] = -1;
    }
    // longest first: the cost of one alignment grows with the square of the gap length, and a long one started last
    // would leave every other thread idle
    std::stable_sort(jobs.begin(), jobs.end(), [](const Job& x, const Job& y) { return x.max_len > y.max_len; });
    const long nj = (long)jobs.size();
    for (long x = 0; x < nj; x++) plan[jobs[(size_t)x].z].gjob[jobs[(size_t)x].t] = (int32_t)x;
    static const struct Up { char up[256], rcu[256]; Up() {
        for (int c = 0; c < 256; c++) {
            up[c] = (char)toupper(c);
            switch (toupper(c)) {                                 // = upper(reverse_complement()), ingest.cpp / parsnp.cpp:1294-1393
                case 'A': rcu[c] = 'T'; break; case 'C': rcu[c] = 'G'; break; case 'G': rcu[c] = 'C'; break;
                case 'T': rcu[c] = 'A'; break; case 'U': rcu[c] = 'T'; break;
                case '\r': case '\n': case '\t': case ' ': case '>': case '.': rcu[c] = 0; break;
                default: rcu[c] = 'N'; break;
            }
        }
    } } U;
    // the gap string of genome i between MUM t and t+1 of a regular LCB (upper case; reverse complement on a reverse LCB row)
    auto gap_text = [&](const Lcb& ct, size_t t, size_t i, char* dst) -> size_t {
        const Mum& first = a.pool[(size_t)ct.mums[0]];
        const Mum& m = a.pool[(size_t)ct.mums[t]];
        const Mum& nx = a.pool[(size_t)ct.mums[t + 1]];
        const char* g = a.genomes[i].seq.data();
        if (first.fwd[i]) {
            const long pos = m.end(i), len = nx.start[i] - pos;
            for (long x = 0; x < len; x++) dst[x] = U.up[(unsigned char)g[pos + x]];
            return (size_t)len;
        }
        const long pos = nx.end(i), len = m.start[i] - pos;
        size_t w = 0;
        for (long x = len; x-- > 0;) { const char c = U.rcu[(unsigned char)g[pos + x]]; if (c) dst[w++] = c; }
        return w;
    };
    // The gaps go to the device in ONE batch (pm_gap_align_batch: one wavefront per gap, include/parsnp_mum.h); the few the
    // device does not take -- wider than its 96-column limit, or declined -- are aligned here by the host threads, the
    // widest ones while the device works on the rest.  PARSNP_HOST_GAPS=1: everything on the host (measurement / tests).
    constexpr unsigned kDeviceCols = 96;
    static const bool host_gaps = test_hook("PARSNP_HOST_GAPS") != nullptr;
    // The LCBs are cut into a few groups of consecutive LCBs with about the same alignment work, one device batch each:
    // the file offsets of a group's records only depend on the groups before it, so its records are written while the
    // device aligns the gaps of the next group.  PARSNP_GAP_GROUPS (test hook) sets the number.
    struct Group { size_t z0 = 0, z1 = 0, y0 = 0, y1 = 0; };      // its LCBs [z0, z1) and its device jobs [y0, y1)
    struct Batch { vector<int32_t> nseq, maxcols, cols; vector<int64_t> seqoff, rowoff; vector<uint8_t> chars, out; vector<long> job; } B;
    static const long want_groups = test_hook("PARSNP_GAP_GROUPS") ? atol(test_hook("PARSNP_GAP_GROUPS")) : 0;
    const size_t ngroups = (size_t)std::max<long>(1, std::min<long>(16, want_groups > 0 ? want_groups : (nj >= 6000 && !all_slow ? 3 : 1)));
    vector<Group> batch(ngroups);
    {
        vector<double> work((size_t)nl + 1, 0.0);            // alignment work before LCB z (a gap costs about its width squared)
        for (long x = 0; x < nj; x++) work[jobs[(size_t)x].z + 1] += (double)(jobs[(size_t)x].max_len + 4) * (jobs[(size_t)x].max_len + 4);
        for (long z = 0; z < nl; z++) work[(size_t)z + 1] += work[(size_t)z] + 1e-9;
        size_t z = 0;
        for (size_t g = 0; g < ngroups; g++) {
            batch[g].z0 = z;
            const double upto = work[(size_t)nl] * (double)(g + 1) / (double)ngroups;
            while (z < (size_t)nl && (g + 1 == ngroups || work[z + 1] <= upto)) z++;
            batch[g].z1 = z;
        }
        batch[ngroups - 1].z1 = (size_t)nl;
    }
    vector<int> group_of((size_t)nl, 0);
    for (size_t g = 0; g < ngroups; g++) for (size_t z = batch[g].z0; z < batch[g].z1; z++) group_of[z] = (int)g;
    for (long x = 0; x < nj; x++) jobs[(size_t)x].grp = group_of[jobs[(size_t)x].z];
    if (!host_gaps && n <= 512) {
        int64_t out_bytes = 0;
        for (size_t g = 0; g < ngroups; g++) {               // group after group; the sorted order carries over: every group is longest first
            batch[g].y0 = B.job.size();
            for (long x = 0; x < nj; x++) {
                Job& j = jobs[(size_t)x];
                if (j.grp != (int)g || j.max_len > kDeviceCols) continue;
                j.dev = (long)B.job.size(); B.job.push_back(x);
                const int32_t cap = (int32_t)std::min<unsigned>(kDeviceCols, j.max_len + j.max_len / 2 + 16);
                B.nseq.push_back((int32_t)n); B.maxcols.push_back(cap); B.rowoff.push_back(out_bytes);
                out_bytes += (int64_t)n * cap;
            }
            batch[g].y1 = B.job.size();
        }
        int short_rows = 0;
        const long nd = (long)B.job.size();
        // the strings, job after job: lengths first (one offset per sequence), then the text by all threads
        B.seqoff.assign((size_t)nd * n + 1, 0);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
        for (long y = 0; y < nd; y++) {
            const Job& j = jobs[(size_t)B.job[(size_t)y]];
            const Lcb& ct = a.lcbs[j.z];
            const Mum& first = a.pool[(size_t)ct.mums[0]];
            const Mum& m = a.pool[(size_t)ct.mums[j.t]];
            const Mum& nx = a.pool[(size_t)ct.mums[j.t + 1]];
            bool clean = true;
            for (size_t i = 0; i < n; i++) B.seqoff[(size_t)y * n + i + 1] = gap_length(first, m, nx, i, &clean);
        }
        for (size_t k = 1; k < B.seqoff.size(); k++) B.seqoff[k] += B.seqoff[k - 1];
        B.chars.resize((size_t)B.seqoff.back() + 1);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(| : short_rows)
        for (long y = 0; y < nd; y++) {
            const Job& j = jobs[(size_t)B.job[(size_t)y]];
            for (size_t i = 0; i < n; i++) {
                const size_t at = (size_t)B.seqoff[(size_t)y * n + i], want = (size_t)(B.seqoff[(size_t)y * n + i + 1] - B.seqoff[(size_t)y * n + i]);
                if (gap_text(a.lcbs[j.z], j.t, i, (char*)B.chars.data() + at) != want) short_rows = 1;
            }
        }
        B.out.resize((size_t)out_bytes); B.cols.assign((size_t)nd, -1);
        if (short_rows) { cerr << "parsnp_core: a genome holds a character its reverse complement drops" << endl; exit(1); }
    }
    lap("gap strings");
    // The file's blocks are reserved while the gaps are being aligned (an estimate of its size: aligned gaps at their row
    // capacity), and the records are later stored through a shared mapping: positioned writes to ONE file queue up behind
    // its inode lock whatever the number of threads (2 GB/s here), page faults on a mapping do not.  A file system that
    // cannot reserve (or PARSNP_OUTPUT_PWRITE=1, test hook) gets the positioned writes.
    xmfa.flush();
    const long long text_at = (long long)xmfa.tellp();
    xmfa.close();
    const string xmfa_path = dir + stem + ".xmfa";
    const int fd = open(xmfa_path.c_str(), O_RDWR);
    if (fd < 0) { cerr << "parsnp_core: cannot write " << xmfa_path << endl; exit(1); }
    static const bool force_pwrite = test_hook("PARSNP_OUTPUT_PWRITE") != nullptr;
    long long reserved = 0;
    std::future<bool> reserve_done;
    if (!force_pwrite && !all_slow) {
        long long est = text_at;
        for (long z = 0; z < nl; z++) {
            const Plan& pl = plan[(size_t)z];
            if (!pl.regular) continue;
            const Lcb& ct = a.lcbs[(size_t)z];
            long long cols = 0;
            for (size_t t = 0; t < ct.mums.size(); t++) {
                cols += a.pool[(size_t)ct.mums[t]].length;
                if (t + 1 < ct.mums.size()) cols += pl.gjob[t] >= 0 && jobs[(size_t)pl.gjob[t]].dev >= 0 ? B.maxcols[(size_t)jobs[(size_t)pl.gjob[t]].dev] : pl.gmax[t];
            }
            est += (long long)n * (cols + cols / 80 + 2 + 64) + 2;
        }
        if (test_hook("PARSNP_RESERVE_TINY")) est = text_at + 4096;      // test hook: the mapping has to grow with every group
        reserved = est;
        reserve_done = std::async(std::launch::async, [fd, est, dbg, clock_s] {
            const double t0 = clock_s();
            const bool ok = fallocate(fd, 0, 0, (off_t)est) == 0;
            if (dbg) fprintf(stderr, "[output] reserve %lld MB: %s, %.4f s\n", est >> 20, ok ? "ok" : "not supported", clock_s() - t0);
            return ok;
        });
    }
    // one side thread makes the ONE device call; the library reports every group as its rows arrive (pm_gap_align_groups),
    // and the main thread takes them in that order
    vector<std::promise<int>> batch_done(ngroups);
    vector<std::future<int>> batch_ready;
    for (auto& pr : batch_done) batch_ready.push_back(pr.get_future());
    const size_t on_device = B.job.size();
    struct Reported { vector<std::promise<int>>* done; size_t n = 0; } reported{&batch_done, 0};
    std::future<void> device_side = std::async(std::launch::async, [&] {
        int rc = PM_OK;
        if (!B.job.empty()) {
            vector<int64_t> group_end(ngroups);
            for (size_t g = 0; g < ngroups; g++) group_end[g] = (int64_t)batch[g].y1;
            const double t0 = clock_s();
            rc = pm_gap_align_groups(-1, (int64_t)B.job.size(), B.nseq.data(), B.seqoff.data(), B.chars.data(), B.maxcols.data(), B.rowoff.data(),
                                     B.out.data(), (int64_t)B.out.size(), B.cols.data(), (int)ngroups, group_end.data(),
                                     [](void* ctx, int) { Reported* r = (Reported*)ctx; (*r->done)[r->n++].set_value(PM_OK); }, &reported);
            if (dbg) fprintf(stderr, "[output] gaps: device   %.4f s (%zu gaps in %zu groups)\n", clock_s() - t0, B.job.size(), ngroups);
        }
        while (reported.n < ngroups) batch_done[reported.n++].set_value(rc);      // (no device jobs, or a failure: the rest hears of it)
    });
    vector<double> jt(dbg ? (size_t)nj : 0);
    auto host_align = [&](const vector<long>& which) {
        const long nw = (long)which.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (long y = 0; y < nw; y++) {
            const long x = which[(size_t)y];
            Job& j = jobs[(size_t)x];
            const double t0 = dbg ? clock_s() : 0;
            gap_between(a, a.lcbs[j.z], j.t, &j.host);
            j.on_host = true;
            j.failed = !gap_align(j.host.seq, &j.host.aligned);
            if (!j.failed)                       // (rows of one alignment have one length; anything else is not printable here)
                for (const string& r : j.host.aligned) if (r.size() != j.host.aligned[0].size()) { cerr << "parsnp_core: ragged gap alignment" << endl; exit(1); }
            if (dbg) jt[(size_t)x] = clock_s() - t0;
        }
    };
    vector<long> rest;
    for (long x = 0; x < nj; x++) if (jobs[(size_t)x].dev < 0) rest.push_back(x);
    host_align(rest);
    long declined = 0;
    bool device_gaps_failed = false;
    lap("wide gaps: host");
    // row i of an aligned gap: pointer + length (nullptr: the alignment failed, the gap is padded instead)
    auto aligned_row = [&](const Job& j, size_t i, size_t* len) -> const char* {
        if (j.on_host) { if (j.failed) return nullptr; *len = j.host.aligned[i].size(); return j.host.aligned[i].data(); }
        const size_t y = (size_t)j.dev;
        *len = (size_t)B.cols[y];
        return (const char*)B.out.data() + B.rowoff[y] + (int64_t)i * B.maxcols[y];
    };
    vector<char> notes((size_t)nl, 0);
    auto stage_columns = [&](size_t gz0, size_t gz1) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (long z = (long)gz0; z < (long)gz1; z++) {
        Plan& pl = plan[(size_t)z];
        if (!pl.regular) continue;
        const Lcb& ct = a.lcbs[(size_t)z];
        long cols = 0;
        for (size_t t = 0; t < ct.mums.size(); t++) {
            cols += a.pool[(size_t)ct.mums[t]].length;
            if (t + 1 == ct.mums.size()) break;
            int32_t gc = pl.gmax[t];
            if (pl.gjob[t] >= 0) {
                size_t len = 0;
                if (aligned_row(jobs[(size_t)pl.gjob[t]], 0, &len)) gc = (int32_t)len; else notes[(size_t)z] = 1;
            }
            pl.gcols[t] = gc; cols += gc;
        }
        pl.cols = cols;
    }
    };
    // ---- the slow way for one LCB: its rows as strings (build_rows), from the same gap alignments
    vector<vector<string>> rows((size_t)nl);
    auto slow_rows = [&](size_t z) {
        const Lcb& ct = a.lcbs[z];
        const Plan& pl = plan[z];
        vector<pair<size_t, Gap>> aligned;
        if (!pl.regular) {                                   // its gaps were not in the batch
            gaps_to_align(a, ct, &aligned);
            for (auto& tg : aligned) tg.second.failed = !gap_align(tg.second.seq, &tg.second.aligned);
        } else {
            for (size_t t = 0; t < pl.gjob.size(); t++) {
                if (pl.gjob[t] < 0) continue;
                const Job& j = jobs[(size_t)pl.gjob[t]];
                aligned.emplace_back(t, Gap());
                Gap& gp = aligned.back().second;
                gap_between(a, ct, t, &gp);
                gp.aligned.resize(n);
                for (size_t i = 0; i < n; i++) { size_t len = 0; const char* r = aligned_row(j, i, &len); if (!r) { gp.failed = true; break; } gp.aligned[i].assign(r, len); }
            }
        }
        bool note = false;
        build_rows(a, ct, aligned, &rows[z], &note);
        if (note) notes[z] = 1;
    };
    vector<char> slow((size_t)nl, 0);
    for (long z = 0; z < nl; z++) if (printable_lcb(a.lcbs[(size_t)z]) && (all_slow || !plan[(size_t)z].regular)) slow[(size_t)z] = 1;
    auto stage_slow_rows = [&](size_t gz0, size_t gz1) {
        vector<long> which;
        for (long z = (long)gz0; z < (long)gz1; z++) if (slow[(size_t)z]) which.push_back(z);
        const long nw = (long)which.size();
#pragma omp parallel for schedule(dynamic) num_threads(threads)
        for (long y = 0; y < nw; y++) slow_rows((size_t)which[(size_t)y]);
    };

    // Pass 1, in LCB order: the overlap trim against the previous printed LCB (it shortens rows and shifts the starts).
    // Pass 2, all threads: headers and sizes of the records of every printed LCB, hence their places in the file.
    // Pass 3, all threads: the records, wrapped at 80 columns, written where they belong (1 GB at 200 x 5 Mb).
    int prev_end = 0;
    vector<Lcb> trimmed((size_t)nl);
    vector<char> printed((size_t)nl, 0);
    auto stage_in_order = [&](size_t gz0, size_t gz1) {
    for (size_t z = gz0; z < gz1; z++) {
        const Lcb& c0 = a.lcbs[z];
        if (!printable_lcb(c0)) continue;
        const int lcb_start = (int)c0.start[0] + 1, lcb_end = (int)c0.end[0];
        if (!slow[z]) {
            if (!(plan[z].cols > (long)(prm.c * 1))) continue;
            static const long mix = test_hook("PARSNP_OUTPUT_MIX") ? atol(test_hook("PARSNP_OUTPUT_MIX")) : 0;   // test hook: every mix-th LCB takes the late string route
            if (std::max(0, prev_end - lcb_start) == 0 && !(mix > 0 && z % (size_t)mix == 0)) { prev_end = lcb_end; trimmed[z] = c0; printed[z] = 1; continue; }
            slow_rows(z); slow[z] = 1;                       // it has to be trimmed: as strings
        }
        Lcb ct = c0;
        vector<string>& row = rows[z];
        if (!(row[0].size() > (size_t)(prm.c * 1))) continue;
        // trim the overlap with the previous printed LCB on the reference (:928-952, quirks kept: the column scan
        // never advances, and the start shift counts the non-gap columns of the whole reference row -- for genomes
        // after the first through the already shortened row 0, whose buffer still holds the old tail)
        int overlap = std::max(0, prev_end - lcb_start);
        if (overlap > 0) {
            if (row[0].empty() || row[0][0] == '-') { cerr << "parsnp_core: overlap trim would not terminate in the reference" << endl; exit(1); }
            const int cols = overlap;
            const string orig0 = row[0];
            const size_t newsize = orig0.size() > (size_t)cols ? orig0.size() - (size_t)cols : 0;
            auto stale0 = [&](size_t pos) -> char {   // row 0 as the reference sees it after its erase(0, cols)
                if (pos < newsize) return orig0[pos + (size_t)cols];
                if (pos == newsize) return '\0';
                return pos < orig0.size() ? orig0[pos] : '\0';
            };
            for (size_t i = 0; i < n; i++) {
                int count = 0;
                for (size_t pos = 0; pos < row[i].size(); pos++) {
                    char ch = i == 0 ? (pos < orig0.size() ? orig0[pos] : '\0') : stale0(pos);
                    if (ch != '-') count++;
                }
                ct.start[i] += count;
                row[i].erase(0, (size_t)cols);
            }
        }
        prev_end = lcb_end;
        if (!(row[0].size() > (size_t)(prm.c * 1))) continue;
        trimmed[z] = ct; printed[z] = 1;
    }
    };
    // headers of LCB z, one per genome, each ending in '\n' (:980-1049)
    auto headers = [&](size_t z, string* out, vector<uint32_t>* ends) {
        const Lcb& ct = trimmed[z];
        const Mum& first = a.pool[(size_t)ct.mums.front()];
        const Mum& lastm = a.pool[(size_t)ct.mums.back()];
        out->clear(); ends->clear();
        char b[160];
        for (size_t i = 0; i < n; i++) {
            // contig label and offset: last pos2hdr entry at or before the LCB start (:994-1037)
            // (the reference scans the map in key order; the entry it ends on is the last key <= start)
            const string* hdr = nullptr;
            int seqstart = 0;
            {
                const auto& p2h = a.genomes[i].pos2hdr;
                auto it = p2h.upper_bound((int)ct.start[i]);
                if (it != p2h.begin()) { --it; hdr = &it->second; seqstart = it->first; }
            }
            static const string s1 = "s1";
            int offset = 0;
            if (!hdr || *hdr == "") { hdr = &s1; offset = -1; }
            else if (*hdr != "s1") offset = -1;
            int w;
            if (first.fwd[i]) w = snprintf(b, sizeof b, "> %zu:%ld-%ld + cluster%d ", i + 1, (long)ct.start[i] + 1, (long)ct.end[i], (int)z + 1);
            else w = snprintf(b, sizeof b, "> %zu:%ld-%ld - cluster%d ", i + 1, (long)lastm.start[i] + 1, (long)first.end(i), (int)z + 1);
            out->append(b, (size_t)w);
            out->append(*hdr);
            if (first.fwd[i]) w = snprintf(b, sizeof b, ":p%ld\n", (long)(ct.start[i] - seqstart) + 1 + offset);
            else w = snprintf(b, sizeof b, ":p%ld\n", (long)(ct.start[i] - seqstart) + 1 + first.length + offset);
            out->append(b, (size_t)w);
            ends->push_back((uint32_t)out->size());
        }
    };
    auto wrapped = [](long long s) -> long long { return s + (s == 0 ? 1 : (s + 79) / 80); };     // s columns + their line ends
    vector<string> text((size_t)nl);                 // slow LCBs: the whole record text
    vector<string> heads((size_t)nl);                // streamed LCBs: the headers
    vector<vector<uint32_t>> head_end((size_t)nl);
    vector<long long> bytes((size_t)nl, 0);
    auto stage_sizes = [&](size_t gz0, size_t gz1) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long zz = (long)gz0; zz < (long)gz1; zz++) {
        const size_t z = (size_t)zz;
        if (!printed[z]) continue;
        headers(z, &heads[z], &head_end[z]);
        if (!slow[z]) { bytes[z] = (long long)heads[z].size() + (long long)n * wrapped(plan[z].cols) + 2; continue; }
        vector<string>& row = rows[z];
        string& out = text[z];
        size_t total = 4 + heads[z].size();
        for (size_t i = 0; i < n; i++) total += row[i].size() + row[i].size() / 80 + 2;
        out.reserve(total);
        for (size_t i = 0; i < n; i++) {
            const string& s = row[i];
            out.append(heads[z], i ? head_end[z][i - 1] : 0, head_end[z][i] - (i ? head_end[z][i - 1] : 0));
            size_t k = 0;
            const size_t width = 80;
            for (; k + width < s.size(); k += width) { out.append(s, k, width); out += '\n'; }
            out.append(s, k, string::npos); out += '\n';
        }
        if (prm.recomb_filter) {        // blocks/b<z+1>/seq.fna: the same records, without the terminator
            string bdir = dir + "blocks/b" + std::to_string((int)z + 1);
            int rc = system(("mkdir -p " + bdir).c_str()); (void)rc;
            ofstream block((bdir + "/seq.fna").c_str());
            block.write(out.data(), (std::streamsize)out.size());
        }
        out += "=\n";
        bytes[z] = (long long)out.size();
        vector<string>().swap(row);
    }
    };
    // the file: `at` = where the next group's records start; the mapping covers the reserved size and grows if a group
    // needs more (only when gaps were aligned wider than the estimate allows for: host-aligned ones)
    long long at = text_at;
    vector<long long> where((size_t)nl, 0);
    const string& path = xmfa_path;
    char* map = nullptr; long long map_len = 0;
    bool map_decided = false;
    int bad = 0;
    auto stage_write = [&](size_t gz0, size_t gz1) {
        for (size_t z = gz0; z < gz1; z++) { where[z] = at; at += bytes[z]; }
        if (!map_decided) {
            map_decided = true;
            if (reserve_done.valid() && reserve_done.get()) {
                map_len = std::max(reserved, at);
                if (map_len > reserved && fallocate(fd, 0, (off_t)reserved, (off_t)(map_len - reserved)) != 0) { cerr << "parsnp_core: cannot write " << path << endl; exit(1); }
                if (map_len > 0) {
                    void* m = mmap(nullptr, (size_t)map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    if (m != MAP_FAILED) map = (char*)m;
                }
            }
            if (dbg) fprintf(stderr, "[output] records through %s\n", map ? "a shared mapping" : "positioned writes");
        }
        if (map && at > map_len) {
            const long long grown = at + (at - map_len);
            if (fallocate(fd, 0, (off_t)map_len, (off_t)(grown - map_len)) != 0) { cerr << "parsnp_core: cannot write " << path << endl; exit(1); }
            void* m = mremap(map, (size_t)map_len, (size_t)grown, MREMAP_MAYMOVE);
            if (m == MAP_FAILED) { cerr << "parsnp_core: cannot write " << path << endl; exit(1); }
            map = (char*)m; map_len = grown;
        }
        // work items: a slow LCB's text, or a run of rows of a streamed LCB (about 1 MB of file each; the rows of one LCB
        // have one size, so every row's place follows from its number)
        struct Item { size_t z; size_t i0, i1; };
        vector<Item> items;
        for (size_t z = gz0; z < gz1; z++) {
            if (!printed[z]) continue;
            if (slow[z]) { items.push_back(Item{z, 0, 0}); continue; }
            const long long rowb = wrapped(plan[z].cols) + 48;
            const size_t per = (size_t)std::max<long long>(1, ((long long)1 << 20) / rowb);
            for (size_t i0 = 0; i0 < n; i0 += per) items.push_back(Item{z, i0, std::min(n, i0 + per)});
        }
        static const struct Low { char low[256], rc[256]; Low() {
            for (int c = 0; c < 256; c++) { low[c] = (char)tolower(c); rc[c] = (char)tolower((unsigned char)U.rcu[c]); }
        } } Lw;
        const long ni = (long)items.size();
#pragma omp parallel num_threads(threads) reduction(| : bad)
        {
            // a thread's buffer: filled with wrapped text, written out with positioned writes whenever it is nearly full
            constexpr size_t kBuf = (size_t)1 << 18, kPiece = 4096;
            vector<char> buf(kBuf + 2 * kPiece);
            size_t w = 0; long long off = 0; int col = 0; long long produced = 0;
            auto flush = [&] {
                size_t done = 0;
                if (map) { memcpy(map + off, buf.data(), w); done = w; }
                while (done < w) {
                    const ssize_t k = pwrite(fd, buf.data() + done, w - done, (off_t)(off + (long long)done));
                    if (k <= 0) { bad = 1; break; }
                    done += (size_t)k;
                }
                off += (long long)w; w = 0;
            };
            // `len` sequence characters src[0..len) through `table`, forwards or backwards, with a line end before every 81st
            auto put = [&](const char* src, size_t len, const char* table, bool backwards) {
                produced += (long long)len;
                size_t x = 0;
                while (x < len) {
                    if (w >= kBuf) flush();
                    size_t piece = std::min(len - x, kPiece);
                    while (piece) {
                        if (col == 80) { buf[w++] = '\n'; col = 0; }
                        const size_t k = std::min(piece, (size_t)(80 - col));
                        if (!table) memcpy(&buf[w], src + x, k);
                        else if (!backwards) for (size_t y = 0; y < k; y++) buf[w + y] = table[(unsigned char)src[x + y]];
                        else for (size_t y = 0; y < k; y++) { const char c = table[(unsigned char)src[len - 1 - (x + y)]]; buf[w + y] = c; if (!c) bad = 2; }
                        w += k; x += k; col += (int)k; piece -= k;
                    }
                }
            };
            auto fill = [&](char c, size_t len) {
                produced += (long long)len;
                while (len) {
                    if (w >= kBuf) flush();
                    if (col == 80) { buf[w++] = '\n'; col = 0; }
                    const size_t k = std::min(std::min(len, (size_t)(80 - col)), kPiece);
                    memset(&buf[w], c, k); w += k; col += (int)k; len -= k;
                }
            };
            vector<char> gbuf;
#pragma omp for schedule(dynamic, 1)
            for (long it = 0; it < ni; it++) {
                const Item& item = items[(size_t)it];
                const size_t z = item.z;
                if (slow[z]) {
                    const string& t = text[z];
                    size_t done = 0;
                    if (map) { memcpy(map + where[z], t.data(), t.size()); done = t.size(); }
                    while (done < t.size()) {
                        const ssize_t k = pwrite(fd, t.data() + done, t.size() - done, (off_t)(where[z] + (long long)done));
                        if (k <= 0) { bad = 1; break; }
                        done += (size_t)k;
                    }
                    continue;
                }
                const Lcb& ct = a.lcbs[z];
                const Plan& pl = plan[z];
                const Mum& first = a.pool[(size_t)ct.mums[0]];
                const long long rowb = wrapped(pl.cols);
                const size_t T = ct.mums.size();
                w = 0; col = 0;
                off = where[z] + (long long)(item.i0 ? head_end[z][item.i0 - 1] : 0) + (long long)item.i0 * rowb;
                for (size_t i = item.i0; i < item.i1; i++) {
                    const uint32_t h0 = i ? head_end[z][i - 1] : 0, h1 = head_end[z][i];
                    if (w + (h1 - h0) >= kBuf) flush();
                    memcpy(&buf[w], heads[z].data() + h0, h1 - h0); w += h1 - h0;
                    col = 0; produced = 0;
                    const char* g = a.genomes[i].seq.data();
                    const bool fw = first.fwd[i] != 0;
                    for (size_t t = 0; t < T; t++) {
                        const Mum& m = a.pool[(size_t)ct.mums[t]];
                        // a MUM's text, lower case, the reverse complement for a reverse member
                        put(g + m.start[i], (size_t)m.length, fw ? Lw.low : Lw.rc, !fw);
                        if (t + 1 == T) break;
                        const long gc = pl.gcols[t];
                        size_t len = 0;
                        const char* r = pl.gjob[t] >= 0 ? aligned_row(jobs[(size_t)pl.gjob[t]], i, &len) : nullptr;
                        if (r) { put(r, len, nullptr, false); continue; }
                        if (gc == 0) continue;
                        if (gbuf.size() < (size_t)gc + 8) gbuf.resize((size_t)gc + 8);
                        len = gap_text(ct, t, i, gbuf.data());          // left-justified, '-' padded
                        put(gbuf.data(), len, nullptr, false);
                        fill('-', (size_t)gc - len);
                    }
                    buf[w++] = '\n';
                    if (produced != pl.cols) bad = 2;                     // (a dropped character: cannot happen with ingest()'s alphabet)
                }
                if (item.i1 == n) { buf[w++] = '='; buf[w++] = '\n'; }
                flush();
                const long long end = where[z] + (long long)head_end[z][item.i1 - 1] + (long long)item.i1 * rowb + (item.i1 == n ? 2 : 0);
                if (off != end) bad = 2;
            }
        }
        if (bad) { cerr << "parsnp_core: error writing " << path << (bad == 2 ? " (record sizes)" : "") << endl; exit(1); }
    };
    // ---- group after group: wait for its alignments, lay it out, write it
    for (size_t g = 0; g < ngroups; g++) {
        const Group& G = batch[g];
        if (batch_ready[g].get() != PM_OK) {
            // the alignment itself is complete by now; a device that cannot take the gaps (out of memory for the workspace, a
            // part that refuses the LDS request, a stream error) only costs time: every job of the group counts as declined
            // and goes through the host aligner, which produces the same rows
            if (!device_gaps_failed) cerr << "parsnp_core: gap alignment on the device failed (" << pm_gap_last_error() << "): aligning the gaps on the host" << endl;
            device_gaps_failed = true;
            for (size_t y = G.y0; y < G.y1; y++) B.cols[y] = -1;
        }
        rest.clear();
        for (size_t y = G.y0; y < G.y1; y++) if (B.cols[y] < 0) rest.push_back(B.job[y]);
        declined += (long)rest.size();
        host_align(rest);
        stage_columns(G.z0, G.z1);
        stage_slow_rows(G.z0, G.z1);
        stage_in_order(G.z0, G.z1);
        stage_sizes(G.z0, G.z1);
        stage_write(G.z0, G.z1);
        if (dbg) { char what[48]; snprintf(what, sizeof what, "group %zu written", g + 1); lap(what); }
    }
    device_side.get();
    if (map && munmap(map, (size_t)map_len) != 0) bad = 1;
    if (ftruncate(fd, (off_t)at) != 0 || close(fd) != 0 || bad) { cerr << "parsnp_core: error writing " << path << endl; exit(1); }
    for (char c : notes) if (c) *gap_note = true;
    if (dbg && nj) {
        double sum = 0, mx = 0; long arg = 0;
        for (long x = 0; x < nj; x++) { sum += jt[(size_t)x]; if (jt[(size_t)x] > mx) { mx = jt[(size_t)x]; arg = x; } }
        fprintf(stderr, "[output] %ld gap alignments: %zu on the device (%ld declined), %.3f s of host work, longest %.3f s (gap of %u columns)\n",
                nj, on_device, declined, sum, mx, jobs[(size_t)arg].max_len);
    }
    if (dbg) {
        long ns = 0, np = 0, nt = 0;
        for (size_t z = 0; z < (size_t)nl; z++) { np += printed[z]; ns += printed[z] && slow[z]; nt += printed[z] && trimmed[z].start != a.lcbs[z].start; }
        fprintf(stderr, "[output] %ld LCBs printed: %ld streamed, %ld as strings (%ld trimmed against their predecessor); %lld MB in %zu group(s)\n", np, np - ns, ns, nt, at >> 20, ngroups);
    }

    lap("xmfa records");
    // ---- log (:1082-1190); stream flags are sticky exactly as in the reference
    log << "Number of sequences analyzed:" << setiosflags(ios::fixed) << setprecision(1) << setw(10) << n << endl << endl;
    for (size_t i = 0; i < n; i++) {
        log << "Sequence " << i + 1 << " : " << a.genomes[i].path << endl;
        log << a.genomes[i].fname << endl;
        log << "Length:" << setw(10) << a.genomes[i].size_nopad << " bps" << endl;
        log << " GC:" << setw(10) << setiosflags(ios::fixed) << setprecision(1) << a.genomes[i].gc << endl;
        log << " AT:" << setw(10) << setiosflags(ios::fixed) << setprecision(1) << a.genomes[i].at << endl;
    }
    log << setw(2) << setiosflags(ios::left) << "d value:   " << setw(2) << prm.d << endl;
    log << setw(2) << "q value:   " << setw(2) << prm.q << endl << endl;
    log << setw(2) << "Mum anchor size:   " << setw(2) << a.l << endl;
    log << setw(2) << "Number of MUM anchors found:   " << setw(2) << a.m0 << endl;
    const long total = (long)a.mums.size() + a.filtered;
    log << setw(2) << "Number of MUMs found:   " << setw(2) << (total >= a.m0 ? total - a.m0 : 0) << endl;
    log << setw(2) << "Total MUMs found((Anchors+MUMs)-filtered):   " << setw(2) << a.mums.size() << endl << endl;
    log << setw(2) << "Random MUM length:   " << setw(2) << a.random << endl;
    log << setw(2) << "Minimum Cluster length:   " << setw(2) << prm.c << endl;
    log << setw(2) << "Number of MUMs filtered:   " << setw(2) << a.filtered << endl;
    log << setw(2) << "Number of Clusters filtered:   " << setw(2) << a.filtered_lcbs << endl << endl;
    long ccount = 0;
    for (const Lcb& c : a.lcbs) if (c.type && !c.mums.empty()) ccount++;
    log << setw(2) << "Number of clusters created:   " << setw(2) << ccount << endl;
    if (a.lcbs.empty()) log << setw(2) << "Number of clusters created:   " << setw(2) << "NONE" << endl;
    if (ccount == 0) { cerr << "parsnp_core: no clusters (the reference divides by zero here)" << endl; exit(1); }
    log << setw(2) << "Average number of MUMs per cluster:   " << setw(2) << a.mums.size() / (size_t)ccount << endl;
    vector<long> coverage(n, 0);
    long avg = 0, totcoverage = 0, totsize = 0;
    for (size_t i = 0; i < n; i++) {
        for (const Lcb& c : a.lcbs) {
            if (!c.type || c.mums.empty()) continue;
            const Mum& f = a.pool[(size_t)c.mums.front()];
            const Mum& bk = a.pool[(size_t)c.mums.back()];
            long span = f.fwd[i] ? labs((bk.start[i] + bk.length) - f.start[i]) : labs((f.start[i] + f.length) - bk.start[i]);
            coverage[i] += span;
            if (i == 0) avg += span;
        }
    }
    log << setw(2) << "Average cluster length:   " << avg / ccount << " bps" << endl;
    float percent;
    for (size_t i = 0; i < n; i++) {
        percent = (float)coverage[i] / ((float)a.genomes[i].gc + (float)a.genomes[i].at);
        log << setw(2) << "Cluster coverage in sequence " << i + 1 << ":   " << setiosflags(ios::fixed) << setprecision(1)
            << 100.00 * percent << "%" << endl;
        totcoverage += coverage[i];
        totsize += a.genomes[i].size_nopad;
    }
    percent = (float)totcoverage / (float)totsize;
    log << setw(2) << "Total coverage among all sequences:   " << setiosflags(ios::fixed) << setprecision(1) << 100.00 * percent
        << "%" << endl << endl;
    log << setw(2) << " MUM anchor search elapsed time:   " << a.anchor_time << "s " << endl;
    log << setw(2) << " MUM coarsening elapsed time:   " << a.coarsen_time << "s " << endl;
    if (prm.random) log << setw(2) << " MUM filtering elapsed time:   " << a.random_time << "s " << endl;
    log << setw(2) << " MUM clustering elapsed time:   " << a.clusters_time << "s " << endl;
    log << setw(2) << " Inter-clustering elapsed time:   " << a.iclusters_time << "s " << endl;
    log << setw(2) << " Total running time:   "
        << a.anchor_time + a.coarsen_time + a.random_time + a.clusters_time + a.iclusters_time << "s " << endl;
    if (*gap_note)
        log << "NOTE: multi-column inter-MUM gaps were emitted unaligned ('-' padded); MUM and LCB coordinates are unaffected." << endl;
    log.close();
}

// parsnp.unalign (Aligner::setUnalignableRegions, src/parsnp.cpp:2310-2381): round-robin over the genomes, each round
// emitting the next run of bases no MUM covers.  Quirks kept: the record holds end-start bases (the last base of the run is
// dropped), runs of one base are skipped, a trailing single character of the 80-column wrap is not printed, and the
// coordinates are 0-based positions of the padded in-memory genome.
void write_unaligned(Aligner& a) {
    using namespace std;
    a.wait_layout();
    const size_t n = a.n;
    ofstream out((a.prm.outdir + "/parsnp.unalign").c_str());
    vector<long> lastpos(n, 0);
    vector<char> exhausted(n, 0);   // no unmarked base left at or after lastpos: the reference rescans to the same answer
    string rec;
    bool stop = false;
    while (!stop) {
        for (size_t k = 0; k < n; k++) {
            Bitmap& bm = a.layout[k];
            const long size = (long)bm.bits();
            long startpos = -1, endpos = -1;
            if (!exhausted[k]) {
                long m = lastpos[k];
                while (m < size && bm.get(m)) m++;
                if (m >= size) exhausted[k] = 1;
                else {
                    startpos = m;
                    while (m < size && !bm.get(m)) m++;
                    endpos = m - 1;
                    bm.set_range(startpos, m);
                    if (m < size) lastpos[k] = endpos + 1;
                }
            }
            if (startpos != endpos) {
                rec.clear();
                rec += ">" + to_string(k + 1) + ":" + to_string(startpos) + "-" + to_string(endpos) + " + " + a.genomes[k].fname + "\n";
                const string& g = a.genomes[k].seq;
                const size_t len = (size_t)(endpos - startpos);
                const size_t from = (size_t)startpos;
                const size_t avail = from <= g.size() ? min(len, g.size() - from) : 0;
                size_t pos = 0;
                while (pos + 80 < avail) { rec.append(g, from + pos, 80); rec.push_back('\n'); pos += 80; }
                if (pos + 1 < avail) { rec.append(g, from + pos, avail - pos); rec.push_back('\n'); }
                if (avail == 0) rec += "-\n";
                rec += "=\n";
                out.write(rec.data(), (streamsize)rec.size());
            } else if (startpos == -1 && k == n - 1) {
                stop = true;
            }
        }
    }
    out.close();
}

}  // namespace parsnp
