// partition_merge.cpp -- partition mode's merge of the per-partition alignments into one parsnp.xmfa.
//
// Restates what the reference driver does after its partitions have run (parsnp:1601-1615), i.e. partition.py:
//   read_xmfa                Bio.AlignIO "mauve" as partition.py sees it: start = printed start - 1, end = printed end,
//                            strand +1/-1, name = sequence index, id = "clusterN sC:pP"
//   chunk_intervals          get_interval + get_chunked_intervals      partition.py:64-83, :507-536  (cut_overlaps :86-96)
//   intersect                interval_intersection, get_intersected_intervals   :35-61, :539-583 (pieces shorter than 10 dropped)
//   trim_lcb                 trim                                       :99-216  (prefix / suffix base counts, bisect_left)
//   combined header          combine_header_info, write_combined_header :245-318
//   merge_cluster            merge_blocks                               :320-433
//   parsnp_partition_merge   trim_xmfas + merge_xmfas                   :586-736
// The partitions' files are 1.2 GB each at 250 x 5 Mb, so nothing is copied: the files are mapped, a row is a view into
// its file (80 columns per line), the trimmed blocks exist only as column ranges of those views (`.trimmed` files are
// written on request), and the clusters are merged by all threads, run of reference-anchored columns by run.
//
// One deliberate difference from the reference: columns that are insertions relative to the reference are
// re-aligned there with spoa.poa (:386), a third-party library that is not part of the reference tree; here they go
// through this project's gap aligner (gapalign.h, the libMUSCLE restatement behind the XMFA writer).  Reference-anchored
// columns, coordinates, headers, block order and the set of bases in every row do not depend on it.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>
#include <thread>

#include "gapalign.h"

namespace parsnp {
namespace {

typedef std::pair<long, long> Interval;

// n independent items over `threads` plain threads, handed out one at a time (no OpenMP here: this code also runs inside
// Python processes, where spinning OpenMP workers of several libraries fight over a container's CPU quota)
template <class F> void parallel_items(long n, int threads, F fn) {
    if (n <= 0) return;
    const int nt = (int)std::max<long>(1, std::min<long>(threads, n));
    if (nt == 1) { for (long i = 0; i < n; i++) fn(i); return; }
    std::atomic<long> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back([&] { for (long i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& th : pool) th.join();
}
double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Mapped {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    void open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd, &st)) throw std::runtime_error("cannot stat " + path);
        n = (size_t)st.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) throw std::runtime_error("cannot map " + path);
            p = (const char*)m;
        }
    }
    ~Mapped() { if (p) munmap((void*)p, n); if (fd >= 0) close(fd); }
};

// a row of an alignment block: `ncol` columns, stored either in its file (lines of `width` columns, each followed by a
// newline) or -- a row whose lines are not all of one width -- unwrapped in `own`
struct Row {
    const char* p = nullptr; size_t ncol = 0, width = 80;
    std::string own;
    char at(size_t c) const { return p[c + c / width]; }
    size_t bytes_to(size_t c) const { return c + c / width; }          // offset of column c
    size_t gaps(size_t a, size_t b) const {                            // '-' among columns [a, b)
        if (a >= b) return 0;
        return (size_t)std::count(p + bytes_to(a), p + bytes_to(b - 1) + 1, '-');
    }
    size_t bases(size_t a, size_t b) const { return a >= b ? 0 : (b - a) - gaps(a, b); }
    size_t next_gap(size_t from) const {                               // first column >= from holding '-', or ncol
        if (from >= ncol) return ncol;
        const char* s = p + bytes_to(from);
        const char* e = p + bytes_to(ncol - 1) + 1;
        const char* g = (const char*)memchr(s, '-', (size_t)(e - s));
        if (!g) return ncol;
        const size_t off = (size_t)(g - p);
        return off - off / (width + 1);
    }
    void append(size_t a, size_t b, std::string* out) const {          // columns [a, b)
        while (a < b) {
            const size_t line_end = (a / width + 1) * width;
            const size_t stop = std::min(b, line_end);
            out->append(p + bytes_to(a), stop - a);
            a = stop;
        }
    }
};

struct Rec {
    int name = 0;             // sequence index of the file
    long start = 0, end = 0;  // AlignIO's view: printed start - 1 (unless the record is absent: "0-0"), printed end
    int strand = 1;
    long contig = 0, pos = 0; // "clusterN sC:pP"
    Row row;
};
typedef std::vector<Rec> Block;

struct SeqEntry { int index; std::string file, header; long length; };

struct XFile {
    std::string path;
    Mapped map;
    std::string header_text;        // the '#' lines, verbatim (copy_header, :218-229)
    std::vector<SeqEntry> seqs;
    std::vector<Block> blocks;
};

long to_long(const char*& s, const char* e) {
    long v = 0; bool any = false;
    while (s < e && *s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); s++; any = true; }
    if (!any) throw std::runtime_error("malformed XMFA record header");
    return v;
}

// "> 3:101-250 + cluster7 s1:p101"
void parse_record_header(const char* s, const char* e, Rec* r) {
    auto expect = [&](char c) { if (s >= e || *s != c) throw std::runtime_error("malformed XMFA record header"); s++; };
    expect('>'); expect(' ');
    r->name = (int)to_long(s, e); expect(':');
    long a = to_long(s, e); expect('-');
    const long b = to_long(s, e); expect(' ');
    if (s >= e || (*s != '+' && *s != '-')) throw std::runtime_error("malformed XMFA record header");
    r->strand = *s == '+' ? 1 : -1; s++; expect(' ');
    if (b != 0) a -= 1;               // Mauve's "0-0" marks a sequence that is absent from the block
    r->start = a; r->end = b;
    static const char kCluster[] = "cluster";
    if ((size_t)(e - s) < sizeof(kCluster) - 1 || memcmp(s, kCluster, sizeof(kCluster) - 1)) throw std::runtime_error("record id is not 'clusterN sC:pP'");
    s += sizeof(kCluster) - 1;
    (void)to_long(s, e); expect(' '); expect('s');
    r->contig = to_long(s, e); expect(':'); expect('p');
    r->pos = to_long(s, e);
}

void read_xmfa(const std::string& path, XFile* x) {
    x->path = path;
    x->map.open(path);
    const char* p = x->map.p; const char* const end = p + x->map.n;
    SeqEntry cur{0, "", "", 0};
    auto field = [](const char* s, const char* e) {      // text after the first space, up to the second (line.split(" ")[1])
        const char* a = (const char*)memchr(s, ' ', (size_t)(e - s));
        if (!a) return std::string();
        a++;
        const char* b = (const char*)memchr(a, ' ', (size_t)(e - a));
        return std::string(a, b ? b : e);
    };
    while (p < end && *p == '#') {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        x->header_text.append(p, (size_t)(le - p)); x->header_text.push_back('\n');
        const std::string line(p, le);
        if (!line.compare(0, 15, "##SequenceIndex")) cur.index = atoi(field(p, le).c_str());
        else if (!line.compare(0, 14, "##SequenceFile")) cur.file = field(p, le);
        else if (!line.compare(0, 16, "##SequenceHeader")) cur.header = field(p, le);
        else if (!line.compare(0, 16, "##SequenceLength")) { std::string v = field(p, le); cur.length = atol(v.substr(0, v.size() >= 2 ? v.size() - 2 : 0).c_str()); x->seqs.push_back(cur); }
        p = nl ? nl + 1 : end;
    }
    Block block;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (*p == '>') {
            Rec r;
            parse_record_header(p, le, &r);
            const char* s = nl ? nl + 1 : end;
            // the sequence lines: up to the next line that starts with '>' or '='
            r.row.p = s;
            size_t ncol = 0, width = 0; bool regular = true, last_short = false;
            const char* q = s;
            while (q < end && *q != '>' && *q != '=' && *q != '#') {
                const char* e2 = (const char*)memchr(q, '\n', (size_t)(end - q));
                const size_t len = (size_t)((e2 ? e2 : end) - q);
                if (last_short) regular = false;                     // a short line followed by another
                if (!width) width = len;
                if (len != width) { if (len < width) last_short = true; else regular = false; }
                ncol += len;
                q = e2 ? e2 + 1 : end;
            }
            r.row.ncol = ncol; r.row.width = width ? width : 80;
            if (!regular) {
                r.row.own.reserve(ncol);
                for (const char* t = s; t < q;) {
                    const char* e2 = (const char*)memchr(t, '\n', (size_t)(q - t));
                    r.row.own.append(t, (size_t)((e2 ? e2 : q) - t));
                    t = e2 ? e2 + 1 : q;
                }
                r.row.width = (size_t)1 << 60;
            }
            block.push_back(std::move(r));
            p = q;
            continue;
        }
        if (*p == '=') { if (!block.empty()) { x->blocks.push_back(std::move(block)); block.clear(); } }
        p = nl ? nl + 1 : end;
    }
    if (!block.empty()) x->blocks.push_back(std::move(block));
    for (Block& b : x->blocks) for (Rec& r : b) if (!r.row.own.empty()) r.row.p = r.row.own.data();     // (after the last move of the record)
}

// partition.py:35-61 (the test is lo < hi: touching intervals do not intersect)
std::vector<Interval> interval_intersection(const std::vector<Interval>& A, const std::vector<Interval>& B) {
    std::vector<Interval> out;
    size_t i = 0, j = 0;
    while (i < A.size() && j < B.size()) {
        const long lo = std::max(A[i].first, B[j].first), hi = std::min(A[i].second, B[j].second);
        if (lo < hi) out.emplace_back(lo, hi);
        if (A[i].second < B[j].second) i++; else j++;
    }
    return out;
}

// get_interval, :64-83: (contig, interval) of the block's FIRST record
void lcb_interval(const Block& b, long* contig, Interval* iv) {
    const Rec& r = b[0];
    const long len = r.end - r.start;
    *contig = r.contig;
    *iv = r.strand == -1 ? Interval(r.pos - len, r.pos) : Interval(r.pos, r.pos + len);
}

typedef std::map<long, std::vector<Interval>> IntervalMap;

IntervalMap chunk_intervals(const XFile& x) {       // get_chunked_intervals, :507-536
    IntervalMap d;
    for (const Block& b : x.blocks) { long c; Interval iv; lcb_interval(b, &c, &iv); d[c].push_back(iv); }
    for (auto& kv : d) {
        std::vector<Interval>& v = kv.second;
        std::sort(v.begin(), v.end());
        for (size_t i = 0; i + 1 < v.size(); i++)       // cut_overlaps, :86-96
            if (v[i].second > v[i + 1].first) v[i + 1] = Interval(v[i].second + 1, v[i + 1].second);
    }
    return d;
}

IntervalMap intersected_intervals(const std::vector<IntervalMap>& per_chunk, long min_size) {      // :539-583
    IntervalMap cur = per_chunk[0];
    for (const IntervalMap& d : per_chunk) {
        std::set<long> keys;
        for (auto& kv : cur) keys.insert(kv.first);
        for (auto& kv : d) keys.insert(kv.first);
        for (long k : keys) {
            auto it = d.find(k);
            cur[k] = interval_intersection(cur[k], it == d.end() ? std::vector<Interval>() : it->second);
        }
    }
    for (auto& kv : cur) {
        std::vector<Interval> keep;
        for (const Interval& iv : kv.second) if (iv.second - iv.first >= min_size) keep.push_back(iv);
        kv.second.swap(keep);
    }
    return cur;
}

// one record of a trimmed block: columns [c0, c1) of a record of the partition's file, with its new coordinates
struct TRec { const Rec* src; size_t c0, c1; long start, end, pos; };
struct TBlock { std::vector<TRec> recs; };

// trim, :99-216: the pieces of one block whose reference coordinates are the given intervals
void trim_lcb(const Block& lcb, const IntervalMap& intervals, int seqidx, std::vector<TBlock>* out) {
    const Rec* ref = nullptr;
    for (const Rec& r : lcb) if (r.name == seqidx) { ref = &r; break; }
    if (!ref) throw std::runtime_error("Reference alignment not found!");
    const long aln_len = ref->end - ref->start;
    long super_start = ref->pos, super_end;
    if (ref->strand == -1) { super_end = super_start; super_start -= aln_len; } else super_end = super_start + aln_len;
    auto it = intervals.find(ref->contig);
    if (it == intervals.end()) throw std::runtime_error("reference contig without intersected intervals");      // KeyError in partition.py
    const std::vector<Interval> pieces = interval_intersection(it->second, std::vector<Interval>{Interval(super_start, super_end)});
    const size_t ncol = ref->row.ncol;
    for (const Rec& r : lcb) if (r.row.ncol < ncol) throw std::runtime_error("rows of one block differ in length");
    for (const Interval& piece : pieces) {
        const long left_bases = piece.first - super_start, right_bases = super_end - piece.second;
        // bisect_left over the prefix sums: the fewest leading (trailing) columns that hold that many reference bases
        size_t left_cols = 0, right_cols = 0;
        for (long seen = 0; seen < left_bases && left_cols < ncol; left_cols++) seen += ref->row.at(left_cols) != '-';
        for (long seen = 0; seen < right_bases && right_cols < ncol; right_cols++) seen += ref->row.at(ncol - 1 - right_cols) != '-';
        TBlock tb;
        for (const Rec& r : lcb) {
            const long lb = (long)r.row.bases(0, left_cols);
            // the last right_cols columns of the row (rec.seq[-i]); the row may be longer than the reference row
            const long rb = (long)r.row.bases(r.row.ncol - right_cols, r.row.ncol);
            TRec t;
            t.src = &r;
            t.c0 = left_cols;
            t.c1 = right_cols > 0 ? (r.row.ncol > right_cols ? r.row.ncol - right_cols : 0) : r.row.ncol;      // seq[l:-r] / seq[l:]
            if (t.c1 < t.c0) t.c1 = t.c0;
            t.start = r.start; t.end = r.end; t.pos = r.pos;
            if (r.strand == -1) { t.start += rb; t.end -= lb; t.pos -= lb; }
            else { t.start += lb; t.end -= rb; t.pos += lb; }
            tb.recs.push_back(t);
        }
        out->push_back(std::move(tb));
    }
}

void wrap80(const std::string& seq, std::string* out) {       // write_aln_to_fna, :231-248
    for (size_t i = 0; i < seq.size(); i += 80) { out->append(seq, i, 80); out->push_back('\n'); }
}
void record_header(int name, long start, long end, int strand, long cluster, long contig, long pos, std::string* out) {
    char buf[160];
    snprintf(buf, sizeof buf, "> %d:%ld-%ld %c cluster%ld s%ld:p%ld\n", name, start + 1, end, strand == 1 ? '+' : '-', cluster, contig, pos);
    out->append(buf);
}

struct Part { XFile x; std::vector<TBlock> trimmed; std::map<int, int> new_index; };

// Where the merged alignment can depend on the insertion aligner (the reference re-aligns insertion runs with
// spoa.poa, partition.py:386; here they go through the gap aligner of the XMFA writer): counted per merge.
//   runs            runs of merged columns in which some partition's reference row holds a gap
//   shared          ... that collected bases from MORE THAN ONE sequence (one sequence is its own alignment either way)
//   shared_diverse  ... whose sequences are not all the same string (identical strings align column by column in any aligner)
//   shared_columns  merged columns of the shared runs
// (one set per merge, handed to the workers of that merge: two merges may run side by side in one process)
struct InsertionCounts { std::atomic<long> runs{0}, shared{0}, shared_diverse{0}, shared_columns{0}; };

// merge_blocks, :320-433: the same trimmed cluster of every partition -> one block.  Columns in which every partition's
// reference row holds a base are concatenated partition after partition (the reference row from the first); a run of
// columns in which some reference row holds a gap is an insertion: its bases are collected per sequence and aligned
// among themselves.
void merge_cluster(const std::vector<Part>& parts, size_t cluster, std::string* text, InsertionCounts* ins) {
    struct Out { int name; const TRec* t; int strand; std::string seq; };
    std::vector<Out> rows;
    std::vector<std::vector<size_t>> row_of(parts.size());      // per partition and record: index into rows, or npos (skipped reference)
    const size_t npos = (size_t)-1;
    size_t total_cols = 0;
    for (size_t p = 0; p < parts.size(); p++) {
        const TBlock& tb = parts[p].trimmed[cluster];
        row_of[p].assign(tb.recs.size(), npos);
        for (size_t k = (p == 0 ? 0 : 1); k < tb.recs.size(); k++) {
            const TRec& t = tb.recs[k];
            auto it = parts[p].new_index.find(t.src->name);
            if (it == parts[p].new_index.end()) throw std::runtime_error("sequence index without an entry in the combined header");      // KeyError in partition.py
            row_of[p][k] = rows.size();
            rows.push_back(Out{it->second, &t, t.src->strand, std::string()});
        }
        if (!tb.recs.empty()) total_cols = std::max(total_cols, tb.recs[0].c1 - tb.recs[0].c0);
    }
    for (Out& o : rows) o.seq.reserve(total_cols + total_cols / 16 + 16);
    std::vector<size_t> col(parts.size()), len(parts.size());
    for (size_t p = 0; p < parts.size(); p++) {
        const TBlock& tb = parts[p].trimmed[cluster];
        if (tb.recs.empty()) throw std::runtime_error("empty block");
        col[p] = tb.recs[0].c0; len[p] = tb.recs[0].c1;
    }
    // insertion bases per row, in the order the rows first received one (the dict order partition.py hands to the aligner)
    std::vector<size_t> gap_order;
    std::vector<std::string> gap_seq(rows.size());
    std::vector<char> in_gap_set(rows.size(), 0);
    auto flush_gap = [&]() {
        std::vector<std::string> seqs;
        for (size_t r : gap_order) seqs.push_back(gap_seq[r]);
        std::vector<std::string> aligned;
        bool ok = seqs.size() > 1 && gap_align(seqs, &aligned);
        size_t width = 0;
        if (!ok) { aligned = seqs; }
        for (const std::string& s : aligned) width = std::max(width, s.size());
        ins->runs++;
        if (seqs.size() > 1) {
            ins->shared++; ins->shared_columns += (long)width;
            bool same = true;
            for (const std::string& s : seqs) same = same && s == seqs[0];
            if (!same) ins->shared_diverse++;
        }
        if (!ok) for (std::string& s : aligned) s.append(width - s.size(), '-');
        for (size_t k = 0; k < gap_order.size(); k++) rows[gap_order[k]].seq += aligned[k];
        for (size_t r = 0; r < rows.size(); r++) if (!in_gap_set[r]) rows[r].seq.append(width, '-');
        for (size_t r : gap_order) { gap_seq[r].clear(); in_gap_set[r] = 0; }
        gap_order.clear();
    };
    for (;;) {
        bool in_gap = false, all_done = true;
        for (size_t p = 0; p < parts.size(); p++) {
            const Row& ref = parts[p].trimmed[cluster].recs[0].src->row;
            if (col[p] >= len[p] || ref.at(col[p]) == '-') in_gap = true;
            if (col[p] < len[p]) all_done = false;
        }
        if ((!in_gap || all_done) && !gap_order.empty()) flush_gap();
        else if (!in_gap && !all_done) {
            // a run of columns that are reference bases in every partition: as many as the shortest such run
            size_t run = (size_t)-1;
            for (size_t p = 0; p < parts.size(); p++) {
                const Row& ref = parts[p].trimmed[cluster].recs[0].src->row;
                run = std::min(run, std::min(ref.next_gap(col[p]), len[p]) - col[p]);
            }
            for (size_t p = 0; p < parts.size(); p++) {
                const TBlock& tb = parts[p].trimmed[cluster];
                for (size_t k = (p == 0 ? 0 : 1); k < tb.recs.size(); k++) tb.recs[k].src->row.append(col[p], col[p] + run, &rows[row_of[p][k]].seq);
                col[p] += run;
            }
        } else if (!all_done) {
            bool moved = false;
            for (size_t p = 0; p < parts.size(); p++) {
                const TBlock& tb = parts[p].trimmed[cluster];
                const Row& ref = tb.recs[0].src->row;
                size_t c = col[p];
                while (c < len[p] && ref.at(c) == '-') {
                    for (size_t k = (p == 0 ? 0 : 1); k < tb.recs.size(); k++) {
                        const Row& row = tb.recs[k].src->row;
                        if (c < row.ncol && row.at(c) != '-') {
                            const size_t r = row_of[p][k];
                            if (!in_gap_set[r]) { in_gap_set[r] = 1; gap_order.push_back(r); }
                            gap_seq[r].push_back(row.at(c));
                        }
                    }
                    c++; moved = true;
                }
                col[p] = c;
            }
            if (!moved) throw std::runtime_error("trimmed blocks of one cluster hold different numbers of reference bases");      // partition.py would not terminate
        }
        if (all_done) break;
    }
    for (const Out& o : rows) {
        record_header(o.name, o.t->start, o.t->end, o.strand, (long)cluster + 1, o.t->src->contig, o.t->pos, text);
        wrap80(o.seq, text);
    }
    text->append("=\n");
}

void write_all(int fd, const std::string& s) {
    const char* p = s.data(); size_t n = s.size();
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) throw std::runtime_error("write failed");
        p += w; n -= (size_t)w;
    }
}

}  // namespace

struct MergeStats { long clusters = 0, sequences = 0, intervals = 0, ref_bases = 0, ins_runs = 0, ins_shared = 0, ins_shared_diverse = 0, ins_shared_columns = 0; };

// xmfas: the partitions' alignments (parsnp_core's XMFA of every good partition, in chunk-label order).  Writes the
// merged alignment to out_path and, with keep_trimmed, <xmfa>.trimmed next to every input (trim_single_xmfa, :586-618).
MergeStats partition_merge(const std::vector<std::string>& xmfas, const std::string& out_path, long min_interval_size, int threads, bool keep_trimmed) {
    if (xmfas.empty()) throw std::runtime_error("no partition to merge");
    if (threads < 1) threads = 1;
    InsertionCounts ins;
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double tl = wall_s();
    auto lap = [&](const char* what) { if (dbg) { const double t = wall_s(); fprintf(stderr, "[merge] %-12s %.3f s\n", what, t - tl); tl = t; } };
    std::vector<Part> parts(xmfas.size());
    std::vector<std::string> errors(xmfas.size());
    const long np = (long)parts.size();
    parallel_items(np, threads, [&](long p) {
        try { read_xmfa(xmfas[(size_t)p], &parts[(size_t)p].x); } catch (const std::exception& e) { errors[(size_t)p] = e.what(); }
    });
    for (long p = 0; p < np; p++) if (!errors[(size_t)p].empty()) throw std::runtime_error(xmfas[(size_t)p] + ": " + errors[(size_t)p]);
    lap("read");
    std::vector<IntervalMap> per_chunk;
    for (const Part& pt : parts) per_chunk.push_back(chunk_intervals(pt.x));
    const IntervalMap inter = intersected_intervals(per_chunk, min_interval_size);
    MergeStats st;
    for (auto& kv : inter) { st.intervals += (long)kv.second.size(); for (const Interval& iv : kv.second) st.ref_bases += iv.second - iv.first; }
    lap("intervals");
    // trim every partition (blocks in file order; a block may give several pieces)
    for (long p = 0; p < np; p++) {
        Part& pt = parts[(size_t)p];
        const long nb = (long)pt.x.blocks.size();
        std::vector<std::vector<TBlock>> per_block((size_t)nb);
        std::vector<std::string> errs((size_t)nb);      // one slot per block: chunks run side by side
        const long chunk = 16, nchunks = (nb + chunk - 1) / chunk;
        parallel_items(nchunks, threads, [&](long c) {
            for (long b = c * chunk; b < std::min(nb, (c + 1) * chunk); b++) {
                try { trim_lcb(pt.x.blocks[(size_t)b], inter, 1, &per_block[(size_t)b]); }
                catch (const std::exception& e) { errs[(size_t)b] = e.what(); }
            }
        });
        for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(pt.x.path + ": " + e);
        for (auto& v : per_block) for (TBlock& tb : v) pt.trimmed.push_back(std::move(tb));
    }
    for (const Part& pt : parts)
        if (pt.trimmed.size() != parts[0].trimmed.size()) throw std::runtime_error("One of the partitions has a different number of clusters after trimming...");   // partition.py:644-646
    const size_t nclusters = parts[0].trimmed.size();
    lap("trim");
    if (keep_trimmed) {
        std::vector<std::string> werr((size_t)np);
        parallel_items(np, threads, [&](long p) {
            const Part& pt = parts[(size_t)p];
            FILE* f = fopen((pt.x.path + ".trimmed").c_str(), "w");
            if (!f) { werr[(size_t)p] = "cannot write " + pt.x.path + ".trimmed"; return; }
            fputs(pt.x.header_text.c_str(), f);
            std::string text, seq;
            for (size_t c = 0; c < pt.trimmed.size(); c++) {
                text.clear();
                for (const TRec& t : pt.trimmed[c].recs) {
                    record_header(t.src->name, t.start, t.end, t.src->strand, (long)c + 1, t.src->contig, t.pos, &text);
                    seq.clear();
                    t.src->row.append(t.c0, t.c1, &seq);
                    wrap80(seq, &text);
                }
                text.append("=\n");
                if (fwrite(text.data(), 1, text.size(), f) != text.size()) { werr[(size_t)p] = "short write to " + pt.x.path + ".trimmed"; break; }
            }
            if (fclose(f) != 0 && werr[(size_t)p].empty()) werr[(size_t)p] = "cannot finish " + pt.x.path + ".trimmed";
        });
        for (const std::string& e : werr) if (!e.empty()) throw std::runtime_error(e);
    }
    if (keep_trimmed) lap("write trimmed");
    // combined header (combine_header_info, :245-292): a (file, header) pair seen before -- the reference, present in
    // every partition -- is not added again; such a sequence of a later partition has no new index
    std::vector<SeqEntry> order;
    {
        std::set<std::pair<std::string, std::string>> seen;
        for (Part& pt : parts)
            for (const SeqEntry& e : pt.x.seqs)
                if (seen.insert(std::make_pair(e.file, e.header)).second) {
                    SeqEntry ne = e; ne.index = (int)order.size() + 1;
                    order.push_back(ne);
                    pt.new_index[e.index] = ne.index;
                }
    }
    st.sequences = (long)order.size(); st.clusters = (long)nclusters;
    const int fd = ::open(out_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + out_path);
    {
        std::string h = "#FormatVersion Mauve\n#SequenceCount " + std::to_string(order.size()) + "\n";       // write_combined_header, :295-318
        for (const SeqEntry& e : order)
            h += "##SequenceIndex " + std::to_string(e.index) + "\n##SequenceFile " + e.file + "\n##SequenceHeader " + e.header + "\n##SequenceLength " + std::to_string(e.length) + "bp\n";
        h += "#IntervalCount " + std::to_string(nclusters) + "\n";
        write_all(fd, h);
    }
    // clusters: merged by all threads a batch at a time, written in order
    const size_t batch = (size_t)threads * 4;
    for (size_t c0 = 0; c0 < nclusters; c0 += batch) {
        const size_t c1 = std::min(nclusters, c0 + batch);
        std::vector<std::string> text(c1 - c0), errs(c1 - c0);
        parallel_items((long)(c1 - c0), threads, [&](long k) {
            try { merge_cluster(parts, c0 + (size_t)k, &text[(size_t)k], &ins); } catch (const std::exception& e) { errs[(size_t)k] = e.what(); }
        });
        for (const std::string& e : errs) if (!e.empty()) { close(fd); throw std::runtime_error(e); }
        for (const std::string& t : text) write_all(fd, t);
    }
    close(fd);
    lap("merge + write");
    st.ins_runs = ins.runs; st.ins_shared = ins.shared; st.ins_shared_diverse = ins.shared_diverse; st.ins_shared_columns = ins.shared_columns;
    return st;
}

}  // namespace parsnp

// C entry (include/parsnp_merge.h): what the reference driver does between "Computing intersection of all partition
// LCBs..." and the end of merge_xmfas (parsnp:1601-1615).  Returns 0, or 1 with a message in err.
static thread_local long g_last_ins[4] = {0, 0, 0, 0};      // of the calling thread's last merge
extern "C" void parsnp_partition_merge_insertions(long* runs, long* shared, long* shared_diverse, long* shared_columns) {
    if (runs) *runs = g_last_ins[0];
    if (shared) *shared = g_last_ins[1];
    if (shared_diverse) *shared_diverse = g_last_ins[2];
    if (shared_columns) *shared_columns = g_last_ins[3];
}
extern "C" int parsnp_partition_merge(int n_xmfas, const char* const* xmfa_paths, const char* out_path, long min_interval_size, int threads,
                                      int keep_trimmed, long* clusters, long* sequences, long* ref_bases, char* err, long err_cap) {
    try {
        std::vector<std::string> xs;
        for (int i = 0; i < n_xmfas; i++) xs.push_back(xmfa_paths[i]);
        const parsnp::MergeStats st = parsnp::partition_merge(xs, out_path, min_interval_size, threads, keep_trimmed != 0);
        if (clusters) *clusters = st.clusters;
        if (sequences) *sequences = st.sequences;
        if (ref_bases) *ref_bases = st.ref_bases;
        g_last_ins[0] = st.ins_runs; g_last_ins[1] = st.ins_shared; g_last_ins[2] = st.ins_shared_diverse; g_last_ins[3] = st.ins_shared_columns;
        return 0;
    } catch (const std::exception& e) {
        if (err && err_cap > 0) { strncpy(err, e.what(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
        return 1;
    }
}
