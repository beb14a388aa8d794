// output.cpp -- parsnpAligner.xmfa + parsnpAligner.log, the output contract of the reference's
// Aligner::writeOutput (src/parsnp.cpp:505-1191).  MUM columns are lower case, inter-MUM gap columns upper case.
// Gaps in which every genome has >= 1 base and some genome has >= 2 go to libMUSCLE in the reference
// (:790-865, src/MuscleInterface.cpp:37-78); here they go to gapalign.cpp, the restatement of that aligner.  Should it
// ever decline an input, the gap is emitted left-justified and '-'-padded and the log carries a note.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <fstream>
#include <future>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <unistd.h>

#include "aligner.h"
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

    // rows of every printable LCB (the reference does this under OpenMP over the LCBs; rows are independent)
    vector<vector<string>> rows(a.lcbs.size());
    vector<char> notes(a.lcbs.size(), 0);
    const long nl = (long)a.lcbs.size();
    const int threads = prm.cores > 0 ? prm.cores : 1;
    vector<vector<pair<size_t, Gap>>> gaps(a.lcbs.size());
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    auto clock_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = clock_s();
    auto lap = [&](const char* what) { if (dbg) { double t = clock_s(); fprintf(stderr, "[output] %-14s %.4f s\n", what, t - tl); tl = t; } };
    auto printable_lcb = [&](const Lcb& ct) { return ct.type == 1 && !ct.mums.empty() && prm.do_align != 0; };
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (long z = 0; z < nl; z++)
        if (printable_lcb(a.lcbs[(size_t)z])) gaps_to_align(a, a.lcbs[(size_t)z], &gaps[(size_t)z]);
    lap("gap strings");
    vector<Gap*> jobs;
    for (auto& g : gaps) for (auto& tg : g) jobs.push_back(&tg.second);
    // longest first: the cost of one alignment grows with the square of the gap length, and a long one started last
    // would leave every other thread idle
    std::sort(jobs.begin(), jobs.end(), [](const Gap* x, const Gap* y) { return x->max_len > y->max_len; });
    const long nj = (long)jobs.size();
    // The gaps go to the device in ONE batch (pm_gap_align_batch: one wavefront per gap, include/parsnp_mum.h); the few the
    // device does not take -- wider than its 96-column limit, or declined -- are aligned here by the host threads, the
    // widest ones while the device works on the rest.  PARSNP_HOST_GAPS=1: everything on the host (measurement / tests).
    constexpr unsigned kDeviceCols = 96;
    static const bool host_gaps = getenv("PARSNP_HOST_GAPS") != nullptr;
    vector<char> on_device((size_t)nj, 0);
    vector<int32_t> d_nseq, d_maxcols, d_cols; vector<int64_t> d_seqoff{0}, d_rowoff; vector<uint8_t> d_chars, d_out;
    vector<long> d_job;
    if (!host_gaps) {
        int64_t out_bytes = 0;
        for (long x = 0; x < nj; x++) {
            const Gap& gp = *jobs[(size_t)x];
            if (gp.max_len > kDeviceCols || gp.seq.size() > 512) continue;
            on_device[(size_t)x] = 1; d_job.push_back(x);
            const int32_t cap = (int32_t)std::min<unsigned>(kDeviceCols, gp.max_len + gp.max_len / 2 + 16);
            d_nseq.push_back((int32_t)gp.seq.size()); d_maxcols.push_back(cap); d_rowoff.push_back(out_bytes);
            out_bytes += (int64_t)gp.seq.size() * cap;
            for (const string& q : gp.seq) { d_chars.insert(d_chars.end(), q.begin(), q.end()); d_seqoff.push_back((int64_t)d_chars.size()); }
        }
        d_out.resize((size_t)out_bytes); d_cols.assign(d_job.size(), -1);
    }
    std::future<int> device_done;
    if (!d_job.empty())
        device_done = std::async(std::launch::async, [&] {
            return pm_gap_align_batch(-1, (int64_t)d_job.size(), d_nseq.data(), d_seqoff.data(), d_chars.data(), d_maxcols.data(), d_rowoff.data(),
                                      d_out.data(), (int64_t)d_out.size(), d_cols.data());
        });
    vector<double> jt(dbg ? (size_t)nj : 0);
    auto host_align = [&](const vector<long>& which) {
        const long nw = (long)which.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (long y = 0; y < nw; y++) {
            const long x = which[(size_t)y];
            Gap& gp = *jobs[(size_t)x];
            const double t0 = dbg ? clock_s() : 0;
            gp.failed = !gap_align(gp.seq, &gp.aligned);
            if (dbg) jt[(size_t)x] = clock_s() - t0;
        }
    };
    vector<long> rest;
    for (long x = 0; x < nj; x++) if (!on_device[(size_t)x]) rest.push_back(x);
    host_align(rest);
    long declined = 0;
    if (!d_job.empty()) {
        if (device_done.get() != PM_OK) { cerr << "parsnp_core: gap alignment on the device failed: " << pm_gap_last_error() << endl; exit(1); }
        lap("gaps: device");
        rest.clear();
        const long nd = (long)d_job.size();
#pragma omp parallel for schedule(static) num_threads(threads)
        for (long y = 0; y < nd; y++) {
            if (d_cols[(size_t)y] < 0) continue;
            Gap& gp = *jobs[(size_t)d_job[(size_t)y]];
            gp.aligned.resize(gp.seq.size());
            const char* base = (const char*)d_out.data() + d_rowoff[(size_t)y];
            for (size_t i = 0; i < gp.seq.size(); i++) gp.aligned[i].assign(base + i * (size_t)d_maxcols[(size_t)y], (size_t)d_cols[(size_t)y]);
            gp.failed = false;
        }
        for (long y = 0; y < nd; y++) if (d_cols[(size_t)y] < 0) rest.push_back(d_job[(size_t)y]);
        declined = (long)rest.size();
        host_align(rest);
    }
    if (dbg && nj) {
        double sum = 0, mx = 0; long arg = 0;
        for (long x = 0; x < nj; x++) { sum += jt[(size_t)x]; if (jt[(size_t)x] > mx) { mx = jt[(size_t)x]; arg = x; } }
        fprintf(stderr, "[output] %ld gap alignments: %zu on the device (%ld declined), %.3f s of host work, longest %.3f s (gap of %u columns)\n",
                nj, d_job.size(), declined, sum, mx, jobs[(size_t)arg]->max_len);
    }
    lap("gap alignment");
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (long z = 0; z < nl; z++) {
        const Lcb& ct = a.lcbs[(size_t)z];
        rows[(size_t)z].assign(n, "");
        if (printable_lcb(ct)) {
            bool note = false;
            build_rows(a, ct, gaps[(size_t)z], &rows[(size_t)z], &note);
            notes[(size_t)z] = note;
            vector<pair<size_t, Gap>>().swap(gaps[(size_t)z]);
        }
    }
    for (char c : notes) if (c) *gap_note = true;
    lap("rows");

    // Pass 1, in LCB order: the overlap trim against the previous printed LCB (it shortens rows and shifts the starts).
    // Pass 2, all threads: the records of every printed LCB, wrapped at 80 columns, into one buffer per LCB.
    // Pass 3: the buffers go to their places in the file with positioned writes (1 GB at 200 x 5 Mb).
    int prev_end = 0;
    vector<Lcb> trimmed(a.lcbs.size());
    vector<char> printed(a.lcbs.size(), 0);
    for (size_t z = 0; z < a.lcbs.size(); z++) {
        Lcb ct = a.lcbs[z];
        vector<string>& row = rows[z];
        if (!(ct.type == 1 && !ct.mums.empty() && prm.do_align != 0 && row[0].size() > (size_t)(prm.c * 1))) continue;
        // trim the overlap with the previous printed LCB on the reference (:928-952, quirks kept: the column scan
        // never advances, and the start shift counts the non-gap columns of the whole reference row -- for genomes
        // after the first through the already shortened row 0, whose buffer still holds the old tail)
        int lcb_start = (int)ct.start[0] + 1, lcb_end = (int)ct.end[0];
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
    vector<string> text(a.lcbs.size());
    const long nz = (long)a.lcbs.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long zz = 0; zz < nz; zz++) {
        const size_t z = (size_t)zz;
        if (!printed[z]) continue;
        const Lcb& ct = trimmed[z];
        vector<string>& row = rows[z];
        string& out = text[z];
        size_t total = 4;
        for (size_t i = 0; i < n; i++) total += row[i].size() + row[i].size() / 80 + 96;
        out.reserve(total);
        char b[16];
        snprintf(b, sizeof b, "%d", (int)z + 1);
        const Mum& first = a.pool[(size_t)ct.mums.front()];
        const Mum& lastm = a.pool[(size_t)ct.mums.back()];
        for (size_t i = 0; i < n; i++) {
            std::ostringstream hd;
            if (first.fwd[i]) hd << "> " << i + 1 << ":" << ct.start[i] + 1 << "-" << ct.end[i] << " ";
            else hd << "> " << i + 1 << ":" << lastm.start[i] + 1 << "-" << first.end(i) << " ";
            // contig label and offset: last pos2hdr entry at or before the LCB start (:994-1037)
            // (the reference scans the map in key order; the entry it ends on is the last key <= start)
            string hdr;
            int seqstart = 0;
            {
                const auto& p2h = a.genomes[i].pos2hdr;
                auto it = p2h.upper_bound((int)ct.start[i]);
                if (it != p2h.begin()) { --it; hdr = it->second; seqstart = it->first; }
            }
            int offset = 0;
            if (hdr == "") { hdr = "s1"; offset = -1; }
            else if (hdr != "s1") offset = -1;
            if (!first.fwd[i]) hd << "- cluster" << b << " " << hdr << ":p" << (ct.start[i] - seqstart) + 1 + first.length + offset;
            else hd << "+ cluster" << b << " " << hdr << ":p" << (ct.start[i] - seqstart) + 1 + offset;
            const string& s = row[i];
            out += hd.str(); out += '\n';
            size_t k = 0;
            const size_t width = 80;
            for (; k + width < s.size(); k += width) { out.append(s, k, width); out += '\n'; }
            out.append(s, k, string::npos); out += '\n';
        }
        if (prm.recomb_filter) {        // blocks/b<z+1>/seq.fna: the same records, without the terminator
            string bdir = dir + "blocks/b" + b;
            int rc = system(("mkdir -p " + bdir).c_str()); (void)rc;
            ofstream block((bdir + "/seq.fna").c_str());
            block.write(out.data(), (std::streamsize)out.size());
        }
        out += "=\n";
        vector<string>().swap(row);
    }
    {
        xmfa.flush();
        long long at = (long long)xmfa.tellp();
        xmfa.close();
        vector<long long> where(a.lcbs.size(), 0);
        for (size_t z = 0; z < a.lcbs.size(); z++) { where[z] = at; at += (long long)text[z].size(); }
        const string path = dir + stem + ".xmfa";
        const int fd = open(path.c_str(), O_WRONLY);
        if (fd < 0 || ftruncate(fd, (off_t)at) != 0) { cerr << "parsnp_core: cannot write " << path << endl; exit(1); }
        int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(| : bad)
        for (long zz = 0; zz < nz; zz++) {
            const string& t = text[(size_t)zz];
            size_t done = 0;
            while (done < t.size()) {
                const ssize_t w = pwrite(fd, t.data() + done, t.size() - done, (off_t)(where[(size_t)zz] + (long long)done));
                if (w <= 0) { bad = 1; break; }
                done += (size_t)w;
            }
        }
        if (close(fd) != 0 || bad) { cerr << "parsnp_core: error writing " << path << endl; exit(1); }
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
