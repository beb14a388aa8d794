// aligner.cpp -- MUM validation, recursive extension and LCB formation on the host.
// See aligner.h for the map onto the reference.  The match finding itself (csgmum) is NOT here: it is
// requested through include/parsnp_mum.h in batches and runs on the GPU.
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "aligner.h"
#include "hooks.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>
#include <omp.h>

#include "minlen.h"

namespace parsnp {

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
[[noreturn]] void fatal(const std::string& msg) {
    std::cerr << "parsnp_core: " << msg << std::endl;
    exit(1);
}
// Order of a std::sort that only looks at the reference start (TMum/TRegion/Cluster operator<, TMum.cpp:151-156,
// LCR.cpp:42-46, LCB.cpp:53-57).  Sorting (key, index) handles with the same algorithm and comparator performs the
// same comparisons and moves as sorting the objects, so the (unstable) permutation is the reference's.
struct Handle { long key; int idx; };
inline bool operator<(const Handle& a, const Handle& b) { return a.key < b.key; }
}  // namespace

// ---------------------------------------------------------------------------------------------- Bitmap
void Bitmap::init(size_t nbits) {
    nbits_ = nbits;
    const size_t words = (nbits + 63) / 64 + 1;
    if (words_ == words) std::fill(w_, w_ + words_, 0);           // a recycled bitmap keeps its pages (attached words: cleared in place)
    else fresh(words);                                            // fresh zero pages, untouched until used
    logging_ = false; log_.clear();
    if (nbits) w_[(nbits - 1) >> 6] |= 1ull << ((nbits - 1) & 63);
}
void Bitmap::set_range_slow(long a, long b) {
    if (a < 0) a = 0;
    if (b > (long)nbits_) b = (long)nbits_;
    while (a < b) {
        size_t wi = (size_t)a >> 6;
        int lo = (int)(a & 63);
        long span = std::min<long>(64 - lo, b - a);
        uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
        store(wi, w_[wi] | mask);
        a += span;
    }
}
void Bitmap::set_range_atomic(long a, long b) {
    if (a < 0) a = 0;
    if (b > (long)nbits_) b = (long)nbits_;
    while (a < b) {
        const size_t wi = (size_t)a >> 6;
        const int lo = (int)(a & 63);
        const long span = std::min<long>(64 - lo, b - a);
        const uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
        __atomic_fetch_or(&w_[wi], mask, __ATOMIC_RELAXED);
        a += span;
    }
}
void Bitmap::clear_range_atomic(long a, long b) {
    if (a < 0) a = 0;
    if (b > (long)nbits_) b = (long)nbits_;
    while (a < b) {
        const size_t wi = (size_t)a >> 6;
        const int lo = (int)(a & 63);
        const long span = std::min<long>(64 - lo, b - a);
        const uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
        __atomic_fetch_and(&w_[wi], ~mask, __ATOMIC_RELAXED);
        a += span;
    }
}
void Bitmap::clear_range(long a, long b) {
    if (a < 0) a = 0;
    if (b > (long)nbits_) b = (long)nbits_;
    while (a < b) {
        size_t wi = (size_t)a >> 6;
        int lo = (int)(a & 63);
        long span = std::min<long>(64 - lo, b - a);
        uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
        store(wi, w_[wi] & ~mask);
        a += span;
    }
}
long Bitmap::next_set(long from) const {
    if (from < 0) from = 0;
    if ((size_t)from >= nbits_) return (long)nbits_;
    size_t wi = (size_t)from >> 6;
    uint64_t w = w_[wi] & (~0ull << (from & 63));
    const size_t nw = (nbits_ + 63) / 64;
    while (!w) { if (++wi >= nw) return (long)nbits_; w = w_[wi]; }
    return (long)(wi * 64 + (size_t)__builtin_ctzll(w));
}
long Bitmap::prev_set(long from) const {
    if (from < 0) return -1;
    if ((size_t)from >= nbits_) from = (long)nbits_ - 1;
    size_t wi = (size_t)from >> 6;
    int hi = (int)(from & 63);
    uint64_t w = w_[wi] & (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1));
    while (!w) { if (wi == 0) return -1; w = w_[--wi]; }
    return (long)(wi * 64 + 63 - (size_t)__builtin_clzll(w));
}
void Bitmap::rollback() {
    for (size_t i = log_.size(); i-- > 0;) w_[log_[i].first] = log_[i].second;
    log_.clear();
}

// ---------------------------------------------------------------------------------------------- Region
static void finish_region(Region& r, size_t n) {   // TRegion ctor, LCR.cpp:16-37
    long s = 500000000, l = 0;
    for (size_t i = 0; i < n; i++) {
        r.length[i] = r.end[i] - r.start[i];
        if (r.length[i] < s) s = r.length[i];
        if (r.length[i] > l) l = r.length[i];
    }
    r.slength = s; r.llength = l;
}
bool Region::same_as(const Region& o, size_t n) const {
    for (size_t i = 0; i < n; i++)
        if (start[i] != o.start[i] || end[i] != o.end[i]) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- Aligner
Aligner::Aligner(std::vector<Genome>& g, const Params& p, pm_session* session, AlignerMemory* memory)
    : own_memory_(memory ? nullptr : new AlignerMemory), memory_(memory ? memory : own_memory_.get()),
      n(g.size()), prm(p), genomes(g), layout(memory_->layout), session_(session), rows_(memory_->rows), irows_(memory_->irows), brows_(memory_->brows), cache_rows_(memory_->cache_rows),
      req_rows_(memory_->req_rows) {
    layout.resize(n);
    pool.swap(memory_->pool_store);      // the previous run's MUM records: capacity (and mapped pages) kept, see ~Aligner
    pool.clear();
    gsize_.resize(n);
    for (size_t i = 0; i < n; i++) {
        if (genomes[i].seq.size() > (size_t)INT32_MAX - 64) fatal("genome longer than 2^31 bases: " + genomes[i].path);   // Mum rows are int32
        gsize_[i] = (long)genomes[i].seq.size();
    }
    // 125 MB of bitmaps at 200 x 5 Mb.  Their storage stays mapped in the run's AlignerMemory (fresh pages cost a fault
    // each when they are first marked, and giving them back costs as much again: 78 vs 61 ms per step), so a repeated
    // run only has to clear them -- beside the anchor call, while the host has nothing else to do; the first reader of
    // the layout (validate, or the end of find_anchors) awaits it.
    // (a few threads, not one: one core clears ~10 GB/s and would still be at it when the anchor call returns;
    // not all: pages first touched by the worker threads can end up away from the thread that does most of the walking)
    // ... unless the previous run never wrote to them (the resident route keeps the layout on the device) ...
    if (memory_->layout_clean && layout.size() == n) {
        bool fits = true;
        for (size_t i = 0; i < n && fits; i++) fits = layout[i].bits() == genomes[i].seq.size() + 1 && !layout[i].attached();
        memory_->layout_clean = false;
        if (fits && test_hook("PARSNP_CHECK_ZERO"))
            for (size_t i = 0; i < n; i++)
                if (layout[i].count_set() != 1 || !layout[i].get((long)genomes[i].seq.size())) fatal("the layout of a resident run is not empty");
        if (fits) return;
    }
    memory_->layout_clean = false;
    const size_t parts = std::min<size_t>(4, std::max<size_t>(1, n / 8));
    for (size_t k = 0; k < parts; k++)
        layout_ready_.push_back(std::async(std::launch::async, [this, k, parts] {
            for (size_t i = n * k / parts; i < n * (k + 1) / parts; i++) layout[i].init(genomes[i].seq.size() + 1);
        }));
}
void Aligner::wait_layout() {
    if (layout_ready_.empty()) return;
    const double t = now_s();
    for (auto& f : layout_ready_) f.get();
    layout_ready_.clear();
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[setup] waited %.4f s for the layout to be cleared\n", now_s() - t);
}

Aligner::~Aligner() {
    wait_layout();
    // the resident route never writes to the host's bitmaps: the next run starts on them as they are
    memory_->layout_clean = res_.active && !layout.empty() && !layout[0].attached();
    // bitmaps attached to this run's layout image (materialize(), parsnp.unalign) outlive the image in the shared AlignerMemory:
    // released, so that the next run's init() takes fresh storage instead of clearing words that are gone
    for (Bitmap& b : layout) if (b.attached()) b.release();
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double t = now_s();
    auto lap = [&](const char* what) { if (dbg) { double u = now_s(); fprintf(stderr, "[release] %-10s %.4f s\n", what, u - t); t = u; } };
    cache_.clear(); lap("cache");
    pool.clear(); pool.swap(memory_->pool_store); lap("pool");
    std::vector<Lcb>().swap(lcbs); lap("lcbs");
    own_memory_.reset(); lap("arenas");
}

Region Aligner::new_region() {
    Region r;
    r.start = rows_.alloc(n); r.end = rows_.alloc(n); r.length = rows_.alloc(n);
    return r;
}
void Aligner::neighbour_into(const Mum& m, bool left, Region* out) const {
    long* start = out->start; long* end = out->end;
    for (size_t i = 0; i < n; i++) {
        if (left) {   // walk left to the previous marked base; at the genome start the region begins at 1 (:1216-1231)
            long p = layout[i].prev_set((long)m.start[i] - 1);
            if (p < 0) p = 0;
            start[i] = p + 1;
            end[i] = (long)m.start[i] - 1;
        } else {      // walk right to the next marked base or the genome end (:1254-1268)
            long nxt = m.end(i) + 1, size = (long)genomes[i].seq.size();
            long p = nxt >= size ? nxt : layout[i].next_set(nxt);
            start[i] = nxt;
            end[i] = p - 1;
        }
    }
    finish_region(*out, n);
}
// neighbour_into for a caller that only wants regions longer than q in every genome (:2158-2170, :233-238 drop the
// others unseen): gives up at the first genome that is not, returning false with the genome, the length there and the
// position at which the walk stopped.  Between anchors of a population sample most gaps are a single base: one genome
// looked at instead of all of them.
bool Aligner::neighbour_if_longer(const Mum& m, bool left, long q, Region* out, long* short_j, long* short_len, long* short_stop) const {
    long* start = out->start; long* end = out->end; long* length = out->length;
    long s = 500000000, l = 0;
    // most regions are dropped in the first genome or two; one that survives those is usually kept, and then every genome's
    // walk starts with a cache miss in a different bitmap: ask for all of those words at once
    constexpr size_t kProbe = 3;
    for (size_t i = 0; i < n; i++) {
        if (i == kProbe)
            for (size_t x = kProbe; x < n; x++) layout[x].prefetch_read(left ? (long)m.start[x] - 1 : m.end(x) + 1);
        long a, b, stop;
        if (left) {
            long p = layout[i].prev_set((long)m.start[i] - 1);
            if (p < 0) p = 0;
            a = p + 1; b = (long)m.start[i] - 1; stop = p;
        } else {
            const long nxt = m.end(i) + 1, size = (long)genomes[i].seq.size();
            const long p = nxt >= size ? nxt : layout[i].next_set(nxt);
            a = nxt; b = p - 1; stop = p;
        }
        const long len = b - a;
        if (len <= q) { *short_j = (long)i; *short_len = len; *short_stop = stop; return false; }
        start[i] = a; end[i] = b; length[i] = len;
        if (len < s) s = len;
        if (len > l) l = len;
    }
    out->slength = s; out->llength = l;
    return true;
}
Region Aligner::neighbour_region(const Mum& m, bool left) {
    struct Tm { double t0; double* acc; ~Tm() { *acc += now_s() - t0; } } tm{now_s(), &stats.t_neighbour};
    Region r = new_region();
    neighbour_into(m, left, &r);
    return r;
}

int Aligner::min_length(bool anchors, long slength) {
    // a pure function of the ini's expression and the length: short lengths (the recursion's 8 000 regions per step) through a
    // flat table that the runs of one process share
    if (slength >= 0 && slength < (1 << 16)) {
        std::vector<int>& flat = memory_->minlen_flat[anchors ? 1 : 0];
        std::string& owner = memory_->minlen_expr[anchors ? 1 : 0];
        const std::string& e = anchors ? prm.anchors : prm.mums;
        if (flat.empty() || owner != e) { flat.assign(1 << 16, INT32_MIN); owner = e; }
        int& v = flat[(size_t)slength];
        if (v == INT32_MIN) {
            int w = 0;
            if (!min_mum_length(e, slength, &w)) fatal("cannot evaluate minimum MUM length expression '" + e + "'");
            v = w;
        }
        return v;
    }
    auto& memo = minlen_memo_[anchors ? 1 : 0];
    auto it = memo.find(slength);
    if (it != memo.end()) return it->second;
    int v = 0;
    const std::string& e = anchors ? prm.anchors : prm.mums;
    if (!min_mum_length(e, slength, &v)) fatal("cannot evaluate minimum MUM length expression '" + e + "'");
    memo.emplace(slength, v);
    return v;
}

static uint64_t hash_rows(const long* a, const long* b, size_t n, int32_t minsize) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)(uint32_t)minsize;
    for (size_t i = 0; i < n; i++) {
        h = (h ^ (uint64_t)a[i]) * 0xff51afd7ed558ccdull; h ^= h >> 32;
        h = (h ^ (uint64_t)b[i]) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
    }
    return h;
}

Aligner::CacheEntry* Aligner::cache_find(const Request& q) {
    auto range = cache_.equal_range(q.hash);
    for (auto it = range.first; it != range.second; ++it) {
        CacheEntry& e = it->second;
        if (e.minsize == q.minsize && !memcmp(e.start, q.start, n * sizeof(long)) && !memcmp(e.len, q.len, n * sizeof(long))) return &e;
    }
    return nullptr;
}
Aligner::CacheEntry* Aligner::cache_put(const Request& q, bool pending) {
    CacheEntry e;
    long* st = cache_rows_.alloc(n); long* ln = cache_rows_.alloc(n);
    memcpy(st, q.start, n * sizeof(long)); memcpy(ln, q.len, n * sizeof(long));
    e.start = st; e.len = ln; e.minsize = q.minsize; e.pending = pending;
    return &cache_.emplace(q.hash, std::move(e))->second;
}

// The reference cuts the reference side of a region into chunks of p bases and re-streams every query against each
// chunk (parsnp.cpp:1519-1547); one finder request per chunk.
void Aligner::chunk_requests(const Region& r, int minsize, std::vector<Request>* out) {
    out->clear();
    long len0 = r.length[0];
    long p = prm.p > len0 ? len0 : prm.p;
    if (p <= 0 && len0 > 0) fatal("LCB p must be positive");
    if (p == len0 && len0 > 0) {
        // the common case: one chunk = the region itself; its rows are the request unless substr() would clamp them
        bool plain = true;
        for (size_t g = 0; g < n && plain; g++) {
            if (r.start[g] < 0 || r.start[g] > gsize_[g]) fatal("region start outside genome");   // std::string::substr would throw
            plain = r.length[g] >= 0 && r.start[g] + r.length[g] <= gsize_[g];
        }
        if (plain) {
            out->push_back(Request{r.start, r.length, minsize, r.start[0], hash_rows(r.start, r.length, n, minsize), true});
            return;
        }
    }
    long partpos = 0;
    while (partpos < len0) {
        if (partpos + p > len0) {
            p = len0 - partpos;
            if (p < 50) { p = 50 + p; partpos = partpos - 50; }
            if (partpos < 0) fatal("reference chunk underflow (p < 50)");
        }
        long* st = req_rows_.alloc(n); long* ln = req_rows_.alloc(n);
        for (size_t g = 0; g < n; g++) {
            long s0 = g == 0 ? r.start[0] + partpos : r.start[g];
            long l0 = g == 0 ? p : r.length[g];
            if (s0 < 0 || s0 > gsize_[g]) fatal("region start outside genome");   // std::string::substr would throw
            if (l0 < 0) l0 = 0;
            if (s0 + l0 > gsize_[g]) l0 = gsize_[g] - s0;                         // substr clamps
            st[g] = s0; ln[g] = l0;
        }
        out->push_back(Request{st, ln, minsize, r.start[0] + partpos, hash_rows(st, ln, n, minsize)});
        partpos += p;
    }
}

void Aligner::collect_engine_timing() {
    int cnt = 64; const char* names[64]; float ms[64];
    if (pm_last_timing(session_, &cnt, names, ms) != PM_OK) return;
    if (timing_first_call_) stats.anchor_ms.clear();
    if (getenv("PARSNP_DEBUG_TIMERS")) {      // one line per engine call: its wall time and its device phases
        double wall = 0, dev = 0;
        for (int i = 0; i < cnt; i++) {
            if (!strcmp(names[i], "call_wall")) wall = ms[i];
            else if (strncmp(names[i], "alg_", 4) && strncmp(names[i], "n_", 2) && strcmp(names[i], "events") && strcmp(names[i], "rest_samples") && strcmp(names[i], "budget_retries") &&
                     strcmp(names[i], "exact_cluster_tests") && strcmp(names[i], "deferred_regions") && strcmp(names[i], "tail_repeats") && strcmp(names[i], "outside_writes")) dev += ms[i];
        }
        fprintf(stderr, "[engine call] wall %.3f ms, device phases %.3f ms\n", wall, dev);
    }
    for (int i = 0; i < cnt; i++) {
        // (counts that travel in the timing list: the algorithmic bytes of a search whose rows only the engine held)
        if (!strcmp(names[i], "alg_survey")) { stats.alg_bytes += ms[i]; continue; }
        if (!strcmp(names[i], "alg_kernel")) { stats.alg_bytes_kernel += ms[i]; continue; }
        if (!strcmp(names[i], "alg_query")) { stats.alg_bytes_query += ms[i]; continue; }
        if (timing_first_call_) stats.anchor_ms.emplace_back(names[i], ms[i]);
        bool merged = false;
        for (auto& kv : stats.engine_ms) if (kv.first == names[i]) { kv.second += ms[i]; merged = true; }
        if (!merged) stats.engine_ms.emplace_back(names[i], ms[i]);
    }
}

// the per-region views into one engine result (nothing is copied; the views keep the result alive)
void Aligner::unpack_result(pm_result* res, size_t nregions, bool rows, std::vector<Raw>* out) {
    std::shared_ptr<pm_result> own(res, pm_result_free);
    out->clear();
    out->resize(nregions);
    const int64_t* off = pm_result_offsets(res);
    const int32_t* k = pm_result_k(res);
    const int32_t* lon = pm_result_lon(res);
    const int32_t* sp = pm_result_sp(res);
    const uint8_t* fw = pm_result_fwd(res);
    int32_t* rstart = rows ? pm_result_start(res) : nullptr;
    uint8_t* rstrand = rows ? pm_result_strand(res) : nullptr;
    const uint32_t* rflags = rows ? pm_result_flags(res) : nullptr;
    const bool dirty_known = rows && pm_result_dirty_known(res) != 0;
    const size_t q = n - 1;
    for (size_t i = 0; i < nregions; i++) {
        Raw& r = (*out)[i];
        size_t a = (size_t)off[i], b = (size_t)off[i + 1];
        r.k = k + a; r.lon = lon + a;
        if (rows) { r.start = rstart + a * n; r.strand = rstrand + a * n; r.flags = rflags + a; r.dirty_known = dirty_known; r.row0 = a; }
        else { r.sp = sp + a * q; r.fwd = fw + a * q; }
        r.count = b - a;
        r.owner = own;
    }
}

void Aligner::run_batch(const std::vector<Request>& reqs, std::vector<Raw>* out, bool rows) {
    out->clear();
    out->resize(reqs.size());
    if (reqs.empty()) return;
    double t0 = now_s();
    // results as MUM rows built on the device where the provider can (the HIP engine) and every request is its region
    static const bool no_rows = test_hook("PARSNP_NO_DEVICE_ROWS") != nullptr;      // test hook: the host builds the rows from sp / fwd
    rows = rows && rows_supported_ && !no_rows;
    {
        int mode = rows ? (resident_try_ ? 2 : 1) : 0;
        if (mode != rows_mode_) {
            if (mode == 2 && pm_session_rows(session_, 2) != PM_OK) mode = 1;      // (a provider without a MUM store: the result carries its rows)
            if (mode == 2 || pm_session_rows(session_, mode) == PM_OK) rows_mode_ = mode;
            else { rows_supported_ = false; rows = false; rows_mode_ = 0; }
        }
    }
    // the flat arrays of the C ABI live in the run's memory: 2 x 13 MB for the recursion batch, not faulted in per call
    std::vector<int64_t>& starts = memory_->batch_starts; std::vector<int64_t>& lens = memory_->batch_lens;
    if (starts.size() < reqs.size() * n) { starts.resize(reqs.size() * n); lens.resize(reqs.size() * n); }
    std::vector<int32_t> mins(reqs.size());
    double alg = 0, algk = 0, algq = 0;
    const long nreq = (long)reqs.size();
#pragma omp parallel for schedule(dynamic, 64) num_threads(prm.cores > 0 ? prm.cores : 1) reduction(+ : alg, algk, algq) if (nreq > 256)
    for (long i = 0; i < nreq; i++) {
        const Request& q = reqs[(size_t)i];
        memcpy(&starts[(size_t)i * n], q.start, n * 8); memcpy(&lens[(size_t)i * n], q.len, n * 8);
        mins[(size_t)i] = q.minsize;
        // SURVEY 8d's model (one 8-byte index probe and 16 bytes of state per query suffix), and what the event search of THIS
        // engine has to move: per (region, query genome) the query piece once (16 B per 32 bases; the reverse strand is not
        // streamed) and 8 B per sampled K-mer -- one 64-byte index request per LEADER, one sample in eight -- unless both sides
        // fit 128 bases (no index), and per region the reference window once.  The samples SeedRest probes on their own (64 B
        // each) and the events (16 B each) are added by the caller from the engine's counts.  (Until round 4 the model charged
        // an index request per sample and the reference window per genome: 1.9 x what the fabric counters saw.)
        const long minlen = q.minsize < 1 ? 1 : q.minsize, K = minlen < 16 ? minlen : 16, stride = minlen - K + 1;
        double a = 0, k = 0, qb = 0;
        for (size_t g = 1; g < n; g++) {
            const double m = (double)q.len[g], nr = (double)q.len[0];
            a += m * 16.25 + 16.0 * nr;
            k += 0.5 * m;
            qb += 0.5 * m;
            if (!(q.len[g] <= 128 && q.len[0] <= 128) && q.len[g] >= K && q.len[0] >= K) k += 8.0 * (double)((q.len[g] - K) / stride + 1);
        }
        k += 0.5 * (double)q.len[0];
        alg += a; algk += k; algq += qb;
    }
    stats.alg_bytes += alg;
    stats.alg_bytes_kernel += algk;
    stats.alg_bytes_query += algq;
    stats.t_pack += now_s() - t0;
    const bool dbg_b = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    if (dbg_b) fprintf(stderr, "[run_batch] %zu requests packed %.4f s\n", reqs.size(), now_s() - t0);
    const double tcall = now_s();
    pm_result* res = nullptr;
    const int rc = pm_multi_mum_batch(session_, (int64_t)reqs.size(), starts.data(), lens.data(), mins.data(), &res);
    if (rc == PM_ELIMIT) {     // a size limit of the engine (include/parsnp_mum.h), not a malfunction: its own exit code
        std::cerr << "parsnp_core: input exceeds a limit of the multi-MUM engine: " << pm_last_error() << std::endl;
        exit(5);
    }
    if (rc != PM_OK) fatal(std::string("multi-MUM engine failed: ") + pm_last_error());
    if (dbg_b) fprintf(stderr, "[run_batch] engine call %.4f s\n", now_s() - tcall);
    double tu = now_s();
    unpack_result(res, reqs.size(), rows, out);
    if (const char* dump = test_hook("PARSNP_DUMP_BATCH")) {      // test hook: every request's reference column and its candidates (k, length), for a diff of two providers
        if (FILE* f = fopen(dump, "a")) {
            for (size_t i = 0; i < reqs.size(); i++) {
                fprintf(f, "R %ld %ld min %d n %zu:", reqs[i].start[0], reqs[i].len[0], (int)reqs[i].minsize, (*out)[i].count);
                for (size_t c = 0; c < (*out)[i].count; c++) fprintf(f, " %d+%d", (*out)[i].k[c], (*out)[i].lon[c]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    stats.t_unpack += now_s() - tu;
    timing_first_call_ = stats.finder_calls == 0;
    collect_engine_timing();      // the device phase times of this call
    stats.finder_calls++;
    stats.finder_regions += (long)reqs.size();
    stats.finder_s += now_s() - t0;
}

// setMums1 for one region: minimum length, finder request(s), validation.
void Aligner::region_mums(const Region& r, bool anchors, std::vector<int>* accepted, bool speculative) {
    int minsize = min_length(anchors, r.slength);
    if (anchors) l = (float)minsize;
    std::vector<Request> reqs;
    const Arena<long>::Mark qmark = req_rows_.mark();
    chunk_requests(r, minsize, &reqs);
    for (Request& q : reqs) {
        double tk = now_s();
        CacheEntry* e = cache_find(q);
        stats.t_key += now_s() - tk;
        if (!e || e->pending) {
            if (speculative) continue;     // the sweep only consumes what its batch computed
            if (speculation_ && remaining_ && (sweeps_ == 0 || misses_since_sweep_ >= 64)) {
                std::vector<Region> rest;
                rest.push_back(r);
                remaining_(&rest);
                sweeps_++; misses_since_sweep_ = 0;
                speculate(std::move(rest));
                e = cache_find(q);
                if (e && !e->pending) { validate(r, q, e->raw, accepted); continue; }
            }
            stats.cache_misses++; misses_since_sweep_++;
            std::vector<Request> one{q};
            std::vector<Raw> raw;
            run_batch(one, &raw, q.plain);
            if (!e) e = cache_put(q, false);
            e->raw = std::move(raw[0]); e->pending = false;
        } else if (!speculative) {
            stats.cache_hits++;
        }
        { double tv = now_s(); validate(r, q, e->raw, accepted); stats.t_validate += now_s() - tv; }
    }
    req_rows_.rewind(qmark);
    if (!speculative) stats.regions_processed++;
}

// rows of candidate c exactly as the reference derives them: DSP (:1671,:1681) with its bound test (:1723), then the
// TMum constructor (TMum.cpp:25-60): forward = DSP-1, reverse = flipped against the WHOLE genome length even inside a
// sub-region.  Returns false when the reference skips the candidate before constructing the TMum.
bool Aligner::candidate_rows(const Region& r, const Request& q, const Raw& raw, size_t c, Mum& m, bool* ok, bool* any_reverse) const {
    if (raw.start) {      // built on the device (CompactCandidates): copy the row, read the verdicts
        memcpy(m.start, raw.start + c * n, n * sizeof(int32_t));
        memcpy(m.fwd, raw.strand + c * n, n);
        m.length = raw.lon[c];
        const uint32_t f = raw.flags[c];
        *ok = !(f & PM_ROW_OUTSIDE); *any_reverse = (f & PM_ROW_REVERSE) != 0;
        return !(f & PM_ROW_BAD);
    }
    const size_t nq = n - 1;
    const long lon = raw.lon[c];
    const int32_t* __restrict sp = &raw.sp[c * nq];
    const uint8_t* __restrict fw = &raw.fwd[c * nq];
    const long* __restrict rstart = r.start;
    const long* __restrict rlen = r.length;
    const long* __restrict gs = gsize_.data();
    int32_t* __restrict ms = m.start; uint8_t* __restrict mf = m.fwd;
    // genome 0: DSP = k + 1 + ini of the reference chunk, always forward
    const unsigned long dsp0 = (unsigned long)raw.k[c] + 1 + (unsigned long)q.ref_ini;
    unsigned long bad = dsp0 - (unsigned long)rstart[0] > (unsigned long)(unsigned int)rlen[0];
    long st0 = (long)(dsp0 - 1);
    ms[0] = (int32_t)st0; mf[0] = 1;   // out-of-range values wrap here, but such a candidate is refused below (bad / notgood)
    unsigned long notgood = (st0 + lon > gs[0]) | (st0 < 0), rev = 0;
    for (size_t j = 1; j < n; j++) {
        // dsp - r.start[j] == sp + 1 in unsigned arithmetic (:1723); startpos = dsp - 1
        const unsigned long spj = (unsigned long)(long)sp[j - 1];
        bad |= spj + 1 > (unsigned long)(unsigned int)rlen[j];
        const long startpos = (long)(spj + (unsigned long)rstart[j]);
        const long f = fw[j - 1] != 0;
        const long flipped = gs[j] - (startpos + lon);
        const long st = f ? startpos : flipped;
        ms[j] = (int32_t)st; mf[j] = fw[j - 1];
        rev |= (unsigned long)!f;
        notgood |= (unsigned long)(st + lon > gs[j]) | (unsigned long)(st < 0);          // never for in-range candidates
    }
    m.length = lon;
    *ok = !notgood; *any_reverse = rev != 0;
    return !bad;
}

// the rest of the per-candidate block of setMums1 (:1781-1833): trim against the layout, length tests, reverse-strand
// members must spell the reverse complement of the reference member (:1791-1825).  Does NOT mark the layout.
bool Aligner::settle(Mum& m, bool touches, bool any_reverse) const {
    if (m.length < 5) return false;
    if (touches) trim(m);   // trim() only acts when the first or last base of some genome is already marked
    if (m.length < 2 || n <= 1) return false;
    if (!m.fwd[0]) return false;
    return !any_reverse || reverse_members_spell(m);
}
bool Aligner::reverse_members_spell(const Mum& m) const {
    const std::string& g0 = genomes[0].seq;
    for (size_t j = 0; j < n; j++) {
        if (m.fwd[j]) continue;
        const std::string& gj = genomes[j].seq;
        long l1 = m.start[j], l2 = m.length;
        if (l1 > (long)gj.size() || m.start[0] > (long)g0.size()) fatal("MUM outside genome");
        long have = std::min<long>(l2, (long)gj.size() - l1), have0 = std::min<long>(l2, (long)g0.size() - m.start[0]);
        if (have != have0) return false;
        for (long x = 0; x < have; x++) {
            char cj = gj[(size_t)(l1 + have - 1 - x)], want;
            switch (cj) { case 'A': want = 'T'; break; case 'C': want = 'G'; break; case 'G': want = 'C'; break;
                          case 'T': want = 'A'; break; default: want = 'N'; }
            if (g0[(size_t)(m.start[0] + x)] != want) return false;
        }
    }
    return true;
}

// Candidate -> MUM, in candidate order (parsnp.cpp:1717-1841).
void Aligner::validate(const Region& r, const Request& q, const Raw& raw, std::vector<int>* accepted) {
    wait_layout();
    anchors_ordered_ = false;      // only validate_parallel, for a list accepted into an empty layout, can say otherwise
    const size_t ncand = raw.count;
    const int threads = prm.cores > 1 ? prm.cores : 1;
    static const size_t par_min = test_hook("PARSNP_PARALLEL_MIN") ? (size_t)atol(test_hook("PARSNP_PARALLEL_MIN")) : 4096;   // test hook
    if (ncand >= par_min && threads > 1 && !layout[0].logging()) { validate_parallel(r, q, raw, accepted, threads); return; }
    const double tser = now_s();
    struct Rep { double t0; size_t n; ~Rep() { if (n > 1000 && getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[validate serial] %zu candidates %.4f s\n", n, now_s() - t0); } } rep_{tser, ncand};
    for (size_t c = 0; c < ncand; c++) {
        Mum m;
        const Arena<int32_t>::Mark imark = irows_.mark();
        const Arena<uint8_t>::Mark bmark = brows_.mark();
        m.start = irows_.alloc(n); m.fwd = brows_.alloc(n);
        bool ok, any_reverse;
        if (!candidate_rows(r, q, raw, c, m, &ok, &any_reverse)) { irows_.rewind(imark); brows_.rewind(bmark); continue; }
        m.id = next_id_++;
        bool touches = false;
        if (ok && m.length > 0)
            for (size_t j = 0; j < n; j++) touches |= layout[j].get(m.start[j]) | layout[j].get(m.end(j) - 1);
        if (const char* dump = test_hook("PARSNP_DUMP_VALIDATION"))      // test hook: every candidate's rows as the host validates them
            if (FILE* f = fopen(dump, "a")) {
                fprintf(f, "host candidate of region %ld+%ld: len %ld ok %d touches %d rev %d rows", r.start[0], r.length[0], m.length, (int)ok, (int)touches, (int)any_reverse);
                for (size_t j = 0; j < n; j++) fprintf(f, " %d%c", m.start[j], m.fwd[j] ? '+' : '-');
                fprintf(f, "\n"); fclose(f);
            }
        if (!ok || !settle(m, touches, any_reverse)) { irows_.rewind(imark); brows_.rewind(bmark); continue; }
        if (const char* dump = test_hook("PARSNP_DUMP_VALIDATION"))
            if (FILE* f = fopen(dump, "a")) { fprintf(f, "   accepted: start0 %d len %ld\n", m.start[0], m.length); fclose(f); }
        for (size_t j = 0; j < n; j++) layout[j].set_range(m.start[j], m.end(j));
        m.slength = r.slength;
        pool.push_back(m);
        accepted->push_back((int)pool.size() - 1);
    }
}

// The same result for a long candidate list (the anchor call), with the per-genome work spread over threads.
// A candidate whose ranges touch nothing marked before it -- neither the layout nor an EARLIER candidate of this
// list -- cannot be trimmed, and nothing it marks can be seen by such a candidate: these "clean" candidates are
// settled and marked in parallel.  The others see exactly what the sequential loop would show them once the clean
// ones are marked (a later clean candidate never overlaps them, or it would not be clean), and run through the
// sequential path in their original order.
void Aligner::validate_parallel(const Region& r, const Request& q, const Raw& raw, std::vector<int>* accepted, int threads) {
    const size_t ncand = raw.count;
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double tp = now_s();
    auto lap = [&](const char* what) { if (dbg) { double t = now_s(); fprintf(stderr, "[validate_parallel] %-10s %.4f s (cpu %.4f)\n", what, t - tp, cpu_lap_s()); tp = t; } };
    // rows: built on the device where the engine delivers them (the candidates' rows ARE the result blocks then: nothing
    // is copied, trim() works on them in place and the result is kept alive), else from (sp, fwd) here
    const bool device_rows = raw.start != nullptr;
    int32_t* srow = device_rows ? raw.start : irows_.alloc(ncand * n);
    uint8_t* frow = device_rows ? raw.strand : brows_.alloc(ncand * n);
    if (device_rows) kept_results_.push_back(raw.owner);
    std::vector<Mum>& cand = memory_->candidates;      // (the run's memory: 3 MB of records per anchor call, not faulted in again)
    cand.clear(); cand.resize(ncand);
    std::vector<uint8_t> state(ncand, 0);   // bit0 constructed, bit1 ok, bit2 any_reverse, bit3 dirty, bit4 accepted
    const long nc = (long)ncand;
    const bool layout_empty = pool.empty();      // nothing accepted yet: the layout holds no mark (the anchor call)
    static const bool force_exact = test_hook("PARSNP_EXACT_OVERLAP") != nullptr;   // test hook: always the bitmap test
    static const bool host_overlap = test_hook("PARSNP_HOST_OVERLAP") != nullptr;   // test hook: the cheap test on the host although the device ran it
    const bool device_dirty = device_rows && raw.dirty_known && layout_empty && !host_overlap;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (long c = 0; c < nc; c++) {
        Mum& m = cand[(size_t)c];
        m.start = srow + (size_t)c * n; m.fwd = frow + (size_t)c * n;
        if (device_rows) {
            const uint32_t f = raw.flags[c];
            m.length = raw.lon[c];
            if (!(f & PM_ROW_BAD)) state[(size_t)c] = (uint8_t)(1 | ((f & PM_ROW_OUTSIDE) ? 0 : 2) | ((f & PM_ROW_REVERSE) ? 4 : 0) | ((device_dirty && (f & PM_ROW_DIRTY)) ? 8 : 0));
            continue;
        }
        bool ok, rev;
        if (candidate_rows(r, q, raw, (size_t)c, m, &ok, &rev)) state[(size_t)c] = 1 | (ok ? 2 : 0) | (rev ? 4 : 0);
    }
    lap("rows");
    // dirty = overlaps the layout or an earlier candidate in some genome.  Each thread owns a stripe of genomes and
    // walks the candidates in order (rows are candidate-major: a stripe reads contiguous entries of every row).
    // First a cheap sufficient test: a candidate that lies entirely after, or entirely before, EVERYTHING earlier in a
    // genome overlaps nothing earlier there (running max of ends / min of starts).  It can only err towards "dirty", and
    // treating a clean candidate as dirty is harmless (it takes the ordered path and sees the same marks); where genomes
    // are rearranged enough for it to flag too many, the exact test with scratch bitmaps decides instead.
    const int nstripes = threads;
    if (!device_dirty) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int t = 0; t < nstripes; t++) {
            const size_t j0 = n * (size_t)t / (size_t)nstripes, j1 = n * (size_t)(t + 1) / (size_t)nstripes;
            std::vector<long> maxend_l(j1 - j0 + 16, -1), minstart_l(j1 - j0 + 16, (long)1 << 62);   // per stripe: no cache line shared with a neighbour
            long* maxend = maxend_l.data() + 8 - j0; long* minstart = minstart_l.data() + 8 - j0;
            for (size_t c = 0; c < ncand; c++) {
                // a stripe reads a few entries of every row, 4 n bytes apart: no hardware prefetcher follows that
                __builtin_prefetch(srow + (c + 24) * n + j0); __builtin_prefetch(srow + (c + 24) * n + j1 - 1);
                if ((state[c] & 3) != 3 || cand[c].length < 5) continue;        // never marks anything
                const int32_t* st = cand[c].start; const long lon = cand[c].length;
                bool hit = false;
                for (size_t j = j0; j < j1; j++) {
                    const long a = st[j], b = a + lon;
                    hit |= !(a >= maxend[j] || b <= minstart[j]);
                    if (!layout_empty) hit |= layout[j].any_set(a, b);
                    if (b > maxend[j]) maxend[j] = b;
                    if (a < minstart[j]) minstart[j] = a;
                }
                if (hit) __atomic_fetch_or(&state[c], (uint8_t)8, __ATOMIC_RELAXED);
            }
        }
    }
    size_t flagged = 0;
    for (size_t c = 0; c < ncand; c++) flagged += (state[c] >> 3) & 1;
    if (dbg) fprintf(stderr, "[validate_parallel] cheap overlap test flags %zu of %zu\n", flagged, ncand);
    if (flagged * 8 > ncand || force_exact) {
        for (size_t c = 0; c < ncand; c++) state[c] &= (uint8_t)~8;
        std::vector<Bitmap>& scratch = memory_->scratch;
        scratch.resize(n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int t = 0; t < nstripes; t++) {
            const size_t j0 = n * (size_t)t / (size_t)nstripes, j1 = n * (size_t)(t + 1) / (size_t)nstripes;
            for (size_t j = j0; j < j1; j++) scratch[j].init_zero_lazy((size_t)gsize_[j] + 1);
            for (size_t c = 0; c < ncand; c++) {
                if ((state[c] & 3) != 3 || cand[c].length < 5) continue;
                const int32_t* st = cand[c].start; const long lon = cand[c].length;
                bool hit = false;
                for (size_t j = j0; j < j1; j++) hit |= scratch[j].test_and_set(st[j], (long)st[j] + lon) | layout[j].any_set(st[j], (long)st[j] + lon);
                if (hit) __atomic_fetch_or(&state[c], (uint8_t)8, __ATOMIC_RELAXED);
            }
        }
    }
    lap("overlap");
    // clean candidates: settle in parallel (no trimming possible), then mark genome by genome
    {
        const long kRun = 1024, nruns = (nc + kRun - 1) / kRun;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (long rr = 0; rr < nruns; rr++) {
            const long c0 = rr * kRun, c1 = std::min(nc, c0 + kRun);
            for (long c = c0; c < c1; c++) {
                const uint8_t st = state[(size_t)c];
                if ((st & 3) != 3 || (st & 8)) continue;
                if (settle(cand[(size_t)c], false, (st & 4) != 0)) state[(size_t)c] |= 16;
            }
        }
    }
    lap("settle");
    // the marks of the clean candidates, genome by genome; the same pass notes whether the accepted clean candidates of a
    // genome come one after the other (anchors_ordered_)
    int disorder = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(| : disorder)
    for (int t = 0; t < nstripes; t++) {
        const size_t j0 = n * (size_t)t / (size_t)nstripes, j1 = n * (size_t)(t + 1) / (size_t)nstripes;
        std::vector<long> last_l(j1 - j0 + 16, 0);      // end of the previous accepted candidate, per genome of the stripe
        long* last = last_l.data() + 8 - j0;
        long bad = 0;
        for (size_t c = 0; c < ncand; c++) {
            __builtin_prefetch(srow + (c + 24) * n + j0); __builtin_prefetch(srow + (c + 24) * n + j1 - 1);
            if ((state[c] & 24) == 16)
                { const int32_t* st = cand[c].start; const long lon = cand[c].length;     // accepted: inside the genome, length >= 5
                  for (size_t j = j0; j < j1; j++) {
                      const long a = st[j], b = a + lon;
                      bad |= a - last[j];               // negative (sign bit) when this one starts before the previous one ends
                      last[j] = b;
                      layout[j].set_range_inside(a, b);
                  } }
        }
        disorder |= bad < 0 ? 1 : 0;
    }
    lap("mark");
    // the rest: ids and pool order as the sequential loop assigns them
    std::vector<uint32_t> ordered;        // the flagged candidates, in candidate order
    for (size_t c = 0; c < ncand; c++) if ((state[c] & 11) == 11) ordered.push_back((uint32_t)c);
    // Two flagged candidates whose ranges meet in no genome read and write different bits: they commute.  A flagged
    // candidate that meets NO other flagged candidate ("free") is settled by any thread, marks atomic; only the tangled
    // rest -- runs of candidates overlapping each other -- keeps the candidate order.  Per genome the test is exact:
    // intervals sorted by start, one overlaps an earlier one iff it starts before the running maximum end; it and the
    // holder of that maximum are tangled (every member of an overlapping pair is caught as one or the other).
    const size_t nord = ordered.size();
    std::vector<uint8_t> tangled(nord, 1), settled(nord, 0);
    static const bool no_free = test_hook("PARSNP_ORDERED_FLAGGED") != nullptr;     // test hook: everything flagged stays ordered
    static const size_t free_min = test_hook("PARSNP_FREE_MIN") ? (size_t)atol(test_hook("PARSNP_FREE_MIN")) : 32;   // test hook
    if (nord >= free_min && !no_free) {
        std::fill(tangled.begin(), tangled.end(), 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int t = 0; t < nstripes; t++) {
            const size_t j0 = n * (size_t)t / (size_t)nstripes, j1 = n * (size_t)(t + 1) / (size_t)nstripes;
            std::vector<std::pair<long, uint32_t>> iv(nord);
            for (size_t j = j0; j < j1; j++) {
                for (size_t o = 0; o < nord; o++) iv[o] = std::make_pair((long)cand[ordered[o]].start[j], (uint32_t)o);
                std::sort(iv.begin(), iv.end());
                long maxend = -1; uint32_t holder = 0;
                for (size_t x = 0; x < nord; x++) {
                    const uint32_t o = iv[x].second;
                    const long a = iv[x].first, b = a + cand[ordered[o]].length;
                    if (a < maxend) { __atomic_store_n(&tangled[o], (uint8_t)1, __ATOMIC_RELAXED); __atomic_store_n(&tangled[holder], (uint8_t)1, __ATOMIC_RELAXED); }
                    if (b > maxend) { maxend = b; holder = o; }
                }
            }
        }
        const long no = (long)nord;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
        for (long o = 0; o < no; o++) {
            if (tangled[(size_t)o]) continue;
            Mum& m = cand[ordered[(size_t)o]];
            bool touches = false;
            if (m.length > 0)
                for (size_t j = 0; j < n; j++) touches |= layout[j].get(m.start[j]) | layout[j].get(m.end(j) - 1);
            if (settle(m, touches, (state[ordered[(size_t)o]] & 4) != 0)) {
                for (size_t j = 0; j < n; j++) layout[j].set_range_atomic(m.start[j], m.end(j));
                settled[(size_t)o] = 1;
            }
        }
    }
    lap("free");
    // what is left reads two words per genome that another core has just written: requested two candidates ahead
    std::vector<uint32_t> left;
    for (size_t o = 0; o < nord; o++) if (tangled[o]) left.push_back(ordered[o]);
    if (dbg) fprintf(stderr, "[validate_parallel] %zu flagged, %zu of them tangled\n", nord, left.size());
    auto warm = [&](size_t o) {
        if (o >= left.size()) return;
        const Mum& w = cand[left[o]];
        if (w.length <= 0) return;
        for (size_t j = 0; j < n; j++) { layout[j].prefetch(w.start[j]); layout[j].prefetch(w.end(j) - 1); }
    };
    warm(0); warm(1);
    // (1) the tangled ones, in candidate order: the only part in which the order shows (everything else is marked already)
    {
        size_t onext = 0;
        for (size_t o = 0; o < nord; o++) {
            if (!tangled[o]) continue;
            const size_t c = ordered[o];
            Mum& m = cand[c];
            warm(++onext + 1);
            bool touches = false;
            if (m.length > 0)
                for (size_t j = 0; j < n; j++) touches |= layout[j].get(m.start[j]) | layout[j].get(m.end(j) - 1);
            const bool acc = settle(m, touches, (state[c] & 4) != 0);
            if (acc) for (size_t j = 0; j < n; j++) layout[j].set_range(m.start[j], m.end(j));
            settled[o] = acc ? 1 : 0;
            stats.parallel_tangled++;
        }
    }
    lap("tangled");
    // (2) ids and pool places as the sequential loop would assign them: every constructed candidate takes the next id, every
    // accepted one the next place -- one pass over the state bytes -- then (3) the MUM records are written by all threads
    constexpr uint32_t kNoPlace = 0xffffffffu;
    std::vector<uint32_t> place(ncand, kNoPlace), idrank(ncand, 0);
    size_t nacc = 0; long nid = 0, ndirty = 0;
    {
        size_t oi = 0;
        for (size_t c = 0; c < ncand; c++) {
            const uint8_t st = state[c];
            if (!(st & 1)) continue;
            idrank[c] = (uint32_t)nid++;
            bool acc = (st & 16) != 0;
            if ((st & 2) && (st & 8)) acc = settled[oi++] != 0;
            if (!acc) continue;
            place[c] = (uint32_t)nacc++;
            ndirty += (st & 8) ? 1 : 0;
        }
    }
    lap("places");
    const size_t pool0 = pool.size(), acc0 = accepted->size();
    const long id0 = next_id_;
    pool.resize(pool0 + nacc);
    accepted->resize(acc0 + nacc);
    lap("resize");
    {
        Mum* const pout = pool.data() + pool0;
        int* const aout = accepted->data() + acc0;
        const long slen = r.slength;
#pragma omp parallel for schedule(dynamic, 2048) num_threads(threads)
        for (long c = 0; c < nc; c++) {
            const uint32_t at = place[(size_t)c];
            if (at == kNoPlace) continue;
            Mum m = cand[(size_t)c];
            m.id = id0 + (long)idrank[(size_t)c];
            m.slength = slen;
            m.dirty = (state[(size_t)c] & 8) != 0;
            m.touched = m.dirty && (!device_rows || m.length != (long)raw.lon[(size_t)c]);      // (every trim shortens the MUM)
            pout[at] = m;
            aout[at] = (int)(pool0 + at);
        }
    }
    lap("records");
    next_id_ += nid;
    stats.parallel_dirty += ndirty;
    stats.parallel_candidates += (long)ncand;
    // order of the whole accepted list per genome = order of the clean ones (above) + every accepted flagged candidate
    // between its list neighbours (it may have been trimmed: its row holds the final coordinates)
    if (layout_empty && !disorder) {
        int unordered = 0;
        const long na = (long)accepted->size();
#pragma omp parallel for schedule(dynamic, 2048) num_threads(threads) reduction(| : unordered)
        for (long x = 0; x < na; x++) {
            const Mum& m = pool[(size_t)(*accepted)[(size_t)x]];
            if (!m.dirty || unordered) continue;
            for (int side = 0; side < 2; side++) {
                if ((side == 0 && x == 0) || (side == 1 && x + 1 == na)) continue;
                const Mum& a = side == 0 ? pool[(size_t)(*accepted)[(size_t)x - 1]] : m;
                const Mum& b = side == 0 ? m : pool[(size_t)(*accepted)[(size_t)x + 1]];
                for (size_t j = 0; j < n; j++) if ((long)b.start[j] < a.end(j)) { unordered = 1; break; }
            }
        }
        anchors_ordered_ = !unordered;
    } else anchors_ordered_ = false;
    lap("sequential");
}

// Overlap trimming against already marked bases: from the left, then from the right, genome by genome; every trim
// shortens the MUM in ALL genomes (Aligner::trim :1399-1477, TMum::trimleft/right TMum.cpp:104-148).
void Aligner::trim(Mum& m) const {
    for (size_t j = 0; j < n; j++) {
        for (long x = m.start[j]; x < m.end(j); x++) {      // start moves right in every genome, end stays
            if (!layout[j].get(x)) break;
            for (size_t i = 0; i < n; i++) m.start[i] += 1;
            m.length -= 1; m.slength -= 1;
        }
        for (long x = m.end(j) - 1; x >= m.start[j]; x--) { // end moves left in every genome: end = start + length
            if (!layout[j].get(x)) break;
            m.length -= 1; m.slength -= 1;
        }
    }
}

bool Aligner::find_anchors() {
    double t0 = now_s();
    Region whole = new_region();
    for (size_t i = 0; i < n; i++) { whole.start[i] = 0; whole.end[i] = (long)genomes[i].seq.size(); }
    finish_region(whole, n);
    std::vector<int> found;
    if (announce_) {
        std::cerr << std::endl << "        Constructing device index of the reference...\n";
        std::cerr << "        Performing initial search for exact matches in the sequences...\n";
    }
    const bool dbg_a = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double ta = now_s();
    auto lap_a = [&](const char* what) { if (dbg_a) { const double t = now_s(); fprintf(stderr, "[anchors] %-18s %.4f s (cpu %.4f)\n", what, t - ta, cpu_lap_s()); ta = t; } };
    lap_a("set-up");
    if (resident_anchors(whole, &found)) {      // the resident route (resident.cpp): validated on the device, the seed regions stay there
        lap_a("resident anchors");
        mums = found;
        m0 = (long)found.size();
        stats.resident = 1;
        stats.anchor_s = now_s() - t0;
        return m0 != 0;
    }
    region_mums(whole, true, &found, false);
    lap_a("search + validation");
    wait_layout();
    mums = found;
    m0 = (long)found.size();
    // seed regions: left and right neighbour of every anchor, longer than q in every genome (:2150-2172).  The layout
    // is final here, so the 2*m0 bitmap walks are independent: computed in parallel, consumed in order.
    double tn = now_s();
    // Only a region longer than q in every genome is ever looked at again (one in fifteen at 200 x 5 Mb), and a region
    // that is dropped cannot equal one that is kept (equal rows, equal slength): rows are worked out in per-thread
    // scratch and kept -- copied into the arena -- only for those.
    std::vector<Region> lRs(found.size()), rRs(found.size());   // start == nullptr: dropped
    const long nf = (long)found.size();
    static const bool check_derived = test_hook("PARSNP_CHECK_NEIGHBOURS") != nullptr;
    // left neighbour: where the walk to the right of the previous anchor ended exactly at this anchor in every genome,
    // the walk back from this anchor crosses the same unmarked bases and stops at the previous anchor's last base
    // (prev_set :1216-1231), or one base later when the base after it is marked: no second walk over the bitmap.
    // Each thread takes a contiguous run of anchors, left then right of each, so the bitmap words stay in its cache.
    const int nthreads = prm.cores > 0 ? prm.cores : 1;
    const long q = prm.q;
    while ((int)memory_->per_thread.size() < nthreads) memory_->per_thread.emplace_back(new AlignerMemory::PerThread);
    // runs of 1024 anchors handed out dynamically: with more threads than the container's CPU quota some thread regularly
    // starts milliseconds late, and a fixed share per thread makes everybody wait for it
    const long kRun = 1024;
    const int nruns = (int)((nf + kRun - 1) / kRun);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int t = 0; t < nruns; t++) {
        const long i0 = (long)t * kRun, i1 = std::min(nf, i0 + kRun);
        std::vector<long> buf(9 * n);
        auto scratch = [&](int k) { Region r; r.start = &buf[(size_t)k * 3 * n]; r.end = r.start + n; r.length = r.end + n; return r; };
        Region lS = scratch(0), rS[2] = {scratch(1), scratch(2)};
        auto keep = [&](const Region& s, Region* out) {
            if (s.slength <= q) return;
            // rows from this thread's own arena: 8 000 kept regions through ONE arena behind a critical section cost 3 ms of
            // lock hand-overs between 48 threads (measured), the copies themselves nothing
            AlignerMemory::PerThread& tl = *memory_->per_thread[(size_t)omp_get_thread_num()];
            Region r;
            r.start = tl.rows.alloc(n); r.end = tl.rows.alloc(n); r.length = tl.rows.alloc(n);
            memcpy(r.start, s.start, n * sizeof(long)); memcpy(r.end, s.end, n * sizeof(long)); memcpy(r.length, s.length, n * sizeof(long));
            r.slength = s.slength; r.llength = s.llength;
            *out = r;
        };
        // what is known about the right neighbour of the previous anchor of this run: its rows (kept), or a genome in
        // which it was shorter than q and where its walk stopped.  If it stopped at THIS anchor there, the left neighbour
        // of this anchor spans the same gap plus at most the base before it: not longer than q either.
        bool have_pr = false;
        long drop_j = -1, drop_stop = 0;
        long sj = 0, sl = 0, sp = 0;
        auto checked = [&](const Mum& m, bool left, bool kept, const Region& got) {      // test hook: the full walk must agree
            std::vector<long> b(3 * n);
            Region f; f.start = b.data(); f.end = f.start + n; f.length = f.end + n;
            neighbour_into(m, left, &f);
            if ((f.slength > q) != kept) fatal("seed region kept/dropped differently from the full bitmap walk");
            if (kept && (!std::equal(f.start, f.start + n, got.start) || !std::equal(f.end, f.end + n, got.end) || f.slength != got.slength || f.llength != got.llength))
                fatal("seed region differs from the full bitmap walk");
        };
        // Anchors in list order in every genome (validate_parallel checked it): the marked base next to an anchor is its list
        // neighbour's first / last base, so both regions follow from three rows -- no bitmap is read.  The walks they
        // replace: determineRegion :1216-1231 (left: previous marked base, region starts one after it, at 1 at the genome
        // start) and :1254-1268 (right: from one base after the MUM's end to the base before the next marked one or the
        // sentinel at the genome end).
        auto from_rows = [&](long i, bool left, Region* out) -> bool {
            const Mum& m = pool[(size_t)found[(size_t)i]];
            const Mum* other = left ? (i > 0 ? &pool[(size_t)found[(size_t)i - 1]] : nullptr) : (i + 1 < nf ? &pool[(size_t)found[(size_t)i + 1]] : nullptr);
            long s = 500000000, l = 0;
            for (size_t j = 0; j < n; j++) {
                long a, b;
                if (left) {
                    a = other ? other->end(j) : 1;            // prev_set = the neighbour's last base e-1 -> start e; none: p = 0 -> start 1
                    b = (long)m.start[j] - 1;
                } else {
                    const long nxt = m.end(j) + 1, size = gsize_[j];
                    long p = nxt;                              // nxt >= size: the walk does not start
                    if (nxt < size) p = other ? std::max<long>((long)other->start[j], nxt) : size;
                    a = nxt; b = p - 1;
                }
                const long len = b - a;
                if (len <= q) return false;
                out->start[j] = a; out->end[j] = b; out->length[j] = len;
                if (len < s) s = len;
                if (len > l) l = len;
            }
            out->slength = s; out->llength = l;
            return true;
        };
        static const bool no_rows_path = test_hook("PARSNP_WALK_NEIGHBOURS") != nullptr;      // test hook: always the bitmap walks
        if (anchors_ordered_ && !no_rows_path) {
            for (long i = i0; i < i1; i++) {
                const Mum& m = pool[(size_t)found[(size_t)i]];
                // fourteen regions in fifteen are dropped at the first genome (one SNP between two anchors): the pass waits
                // for the first line of a row it has not seen, so that line is asked for a few anchors ahead
                if (i + 6 < nf) __builtin_prefetch(pool[(size_t)found[(size_t)i + 6]].start);
                Region& rR = rS[0];
                const bool l_kept = from_rows(i, true, &lS);
                if (check_derived) checked(m, true, l_kept, lS);
                if (l_kept) keep(lS, &lRs[(size_t)i]);
                const bool r_kept = from_rows(i, false, &rR);
                if (check_derived) checked(m, false, r_kept, rR);
                if (r_kept) keep(rR, &rRs[(size_t)i]);
            }
            continue;
        }
        for (long i = i0; i < i1; i++) {
            const Mum& m = pool[(size_t)found[(size_t)i]];
            Region& pr = rS[(i - i0 + 1) & 1];       // right neighbour of the previous anchor of this run (valid when have_pr)
            Region& rR = rS[(i - i0) & 1];
            bool l_kept = false, l_done = false;
            if (i > i0 && drop_j >= 0 && drop_stop == (long)m.start[(size_t)drop_j]) l_done = true;
            else if (i > i0 && have_pr) {
                bool derived = true;
                for (size_t j = 0; j < n; j++) if (pr.end[j] + 1 != (long)m.start[j]) { derived = false; break; }
                if (derived) {
                    const Mum& pm = pool[(size_t)found[(size_t)i - 1]];
                    for (size_t j = 0; j < n; j++) {
                        const long e = pm.end(j);
                        lS.start[j] = layout[j].get(e) ? e + 1 : e;
                        lS.end[j] = (long)m.start[j] - 1;
                    }
                    finish_region(lS, n);
                    l_kept = lS.slength > q; l_done = true;
                }
            }
            if (!l_done) l_kept = neighbour_if_longer(m, true, q, &lS, &sj, &sl, &sp);
            if (check_derived) checked(m, true, l_kept, lS);
            if (l_kept) keep(lS, &lRs[(size_t)i]);
            have_pr = neighbour_if_longer(m, false, q, &rR, &sj, &sl, &sp);
            drop_j = (!have_pr && sl <= q - 1) ? sj : -1; drop_stop = sp;
            if (check_derived) checked(m, false, have_pr, rR);
            if (have_pr) keep(rR, &rRs[(size_t)i]);
        }
    }
    stats.t_neighbour += now_s() - tn;
    lap_a(anchors_ordered_ ? "seeds from rows" : "seeds by walks");
    for (size_t i = 0; i < found.size(); i++) {
        const Region& lR = lRs[i];
        if (lR.start && (i == 0 || !rRs[i - 1].start || !lR.same_as(rRs[i - 1], n))) regions.push_back(lR);
        const Region& rR = rRs[i];
        if (rR.start && (!lR.start || !rR.same_as(lR, n))) regions.push_back(rR);
    }
    lap_a("region list");
    stats.anchor_s = now_s() - t0;
    return m0 != 0;
}

// One pass of the reference's work-list loop (doWork :192-308): pop the front region, find its MUMs, push the left /
// right neighbour regions of every new MUM, std::sort the list by reference start, drop a region equal to its successor.
//
// The list is kept in a std::map while its keys are unique: then the sorted order is unique too, so sort+dedup equals
// "insert unless an identical region is already there", and nothing has to be re-sorted per iteration (the reference
// spends a third of its recursion time there).  The moment two DIFFERENT regions share a reference start the unstable
// std::sort's tie order becomes observable; from then on the literal vector + std::sort + adjacent-dedup of the
// reference runs on exactly the array the reference would hold, until the keys are unique again.
bool Aligner::extend_pass(bool speculative, bool sorted_start) {
    wait_layout();
    std::vector<Region> rpool = std::move(regions);
    regions.clear();
    std::map<long, int> uniq;            // key -> rpool index       (fast mode)
    std::vector<Handle> work;            // literal mode
    bool literal = false;
    {   // the seed list is processed in push order until the first sort (:194-195 precede :291-292)
        work.reserve(rpool.size());
        for (size_t i = 0; i < rpool.size(); i++) work.push_back(Handle{rpool[i].start[0], (int)i});
        literal = true;
    }
    size_t head = 0;
    std::vector<int> found;
    std::vector<Handle> kids;
    const bool force_literal = test_hook("PARSNP_FORCE_LITERAL_WORKLIST") != nullptr;   // debug: always the reference's vector + std::sort
    remaining_ = [&](std::vector<Region>* out) {   // what is still on the work list, in order
        if (literal) for (size_t x = head; x < work.size(); x++) out->push_back(rpool[(size_t)work[x].idx]);
        else for (const auto& kv : uniq) out->push_back(rpool[(size_t)kv.second]);
    };
    auto keys_unique = [&]() {
        for (size_t x = head; x + 1 < work.size(); x++) if (!(work[x].key < work[x + 1].key)) return false;
        return true;
    };
    // literal mode: sort the waiting part, drop a region equal to its successor (adjacent duplicates only, :294-306), and go
    // back to the map when the keys are unique again
    auto normalize = [&]() {
        if (head < work.size()) std::sort(work.begin() + (long)head, work.end());
        long rsize = (long)(work.size() - head);
        if (rsize) {
            for (long x = 0; x < rsize - 1; x++) {
                if (rpool[(size_t)work[head + (size_t)x].idx].same_as(rpool[(size_t)work[head + (size_t)x + 1].idx], n)) {
                    work.erase(work.begin() + (long)head + x);
                    x -= 1; rsize -= 1;
                }
            }
        }
        if (!force_literal && keys_unique()) {
            for (size_t x = head; x < work.size(); x++) uniq.emplace_hint(uniq.end(), work[x].key, work[x].idx);
            work.clear(); head = 0;
            literal = false;
        }
    };
    if (sorted_start) normalize();     // continuation of a run whose first region has been processed already: the list is sorted
    for (;;) {
        int cur;
        if (literal) { if (head >= work.size()) break; cur = work[head++].idx; }
        else { if (uniq.empty()) break; cur = uniq.begin()->second; uniq.erase(uniq.begin()); }
        found.clear();
        {
            Region curRegion = rpool[(size_t)cur];   // rows live in the arena; the pool vector may grow below
            region_mums(curRegion, false, &found, speculative);
        }
        kids.clear();
        Region lR, rR;
        for (size_t i = 0; i < found.size(); i++) {
            if (i == 0) lR = neighbour_region(pool[(size_t)found[0]], true);
            rR = neighbour_region(pool[(size_t)found[i]], false);
            if (lR.slength > prm.q) { rpool.push_back(lR); kids.push_back(Handle{lR.start[0], (int)rpool.size() - 1}); }
            if (rR.slength > prm.q) { rpool.push_back(rR); kids.push_back(Handle{rR.start[0], (int)rpool.size() - 1}); }
            if (i + 1 < found.size()) lR = neighbour_region(pool[(size_t)found[i + 1]], true);
            mums.push_back(found[i]);
        }
        double ts = now_s();
        if (!literal) {
            // can the children be merged without a tie between different regions?
            bool clean = true;
            for (size_t a = 0; a < kids.size() && clean; a++) {
                auto it = uniq.find(kids[a].key);
                if (it != uniq.end() && !rpool[(size_t)it->second].same_as(rpool[(size_t)kids[a].idx], n)) clean = false;
                for (size_t b = 0; b < a && clean; b++)
                    if (kids[b].key == kids[a].key && !rpool[(size_t)kids[b].idx].same_as(rpool[(size_t)kids[a].idx], n)) clean = false;
            }
            if (clean) {
                for (const Handle& k : kids) uniq.emplace(k.key, k.idx);   // an identical region is already there: dropped
            } else {
                work.clear(); head = 0;
                for (const auto& kv : uniq) work.push_back(Handle{kv.first, kv.second});
                uniq.clear();
                literal = true;
                if (!speculative) stats.tie_fallbacks++;
            }
        }
        if (literal) {
            if (!speculative) stats.literal_iterations++;
            for (const Handle& k : kids) work.push_back(k);
            normalize();
        }
        stats.t_sort += now_s() - ts;
    }
    return !mums.empty();
}

// Batch-compute the engine requests of `gen` that are not cached yet (one engine call).
void Aligner::prefetch(const std::vector<Region>& gen) {
    wanted_.clear(); wanted_entries_.clear();
    const Arena<long>::Mark qmark = req_rows_.mark();
    std::vector<Request> reqs;
    for (const Region& r : gen) {
        int minsize = min_length(false, r.slength);
        chunk_requests(r, minsize, &reqs);
        for (Request& q : reqs) {
            if (cache_find(q)) continue;                       // known, or already wanted in this round
            wanted_entries_.push_back(cache_put(q, true));
            Request w = q;                                      // the cache entry owns a stable copy of the rows
            w.start = wanted_entries_.back()->start; w.len = wanted_entries_.back()->len;
            wanted_.push_back(w);
        }
    }
    req_rows_.rewind(qmark);
    std::vector<Raw> raws;
    bool all_plain = true;
    for (const Request& w : wanted_) all_plain = all_plain && w.plain;
    run_batch(wanted_, &raws, all_plain);
    for (size_t i = 0; i < wanted_.size(); i++) { wanted_entries_[i]->raw = std::move(raws[i]); wanted_entries_[i]->pending = false; }
    wanted_.clear(); wanted_entries_.clear();
}

// Speculative breadth-first sweep from the regions in `gen`: discover the regions the exact in-order replay will ask
// for, generation by generation, and compute each generation in ONE batched engine call.  Everything it does to the
// layout, the MUM pool and the arenas is undone afterwards; only the cache of raw engine results (a pure function of
// the request coordinates) survives.  The reference's processing order is observable, so the replay stays
// authoritative; a request the sweep did not predict is computed on demand there.
void Aligner::speculate(std::vector<Region> gen) {
    double t0 = now_s();
    for (size_t i = 0; i < n; i++) layout[i].begin_log();
    const std::vector<int> saved_mums = mums;
    const size_t saved_pool = pool.size();
    const long saved_id = next_id_;
    const Arena<long>::Mark rmark = rows_.mark();
    const Arena<int32_t>::Mark imark = irows_.mark();
    const Arena<uint8_t>::Mark bmark = brows_.mark();
    while (!gen.empty()) {
        stats.spec_rounds++;
        prefetch(gen);
        std::vector<Region> next;
        std::vector<int> found;
        for (const Region& r : gen) {
            found.clear();
            region_mums(r, false, &found, true);
            for (size_t i = 0; i < found.size(); i++) {
                Region a = neighbour_region(pool[(size_t)found[i]], true), b = neighbour_region(pool[(size_t)found[i]], false);
                if (a.slength > prm.q) next.push_back(a);
                if (b.slength > prm.q) next.push_back(b);
            }
        }
        // same clean-up the reference applies to its work list: order by reference start, drop adjacent duplicates
        std::stable_sort(next.begin(), next.end(), [](const Region& x, const Region& y) { return x.start[0] < y.start[0]; });
        gen.clear();
        for (Region& r : next)
            if (gen.empty() || !gen.back().same_as(r, n)) gen.push_back(r);
    }
    for (size_t i = 0; i < n; i++) { layout[i].rollback(); layout[i].end_log(); }
    pool.resize(saved_pool);
    rows_.rewind(rmark); irows_.rewind(imark); brows_.rewind(bmark);
    next_id_ = saved_id;
    mums = saved_mums;
    stats.t_sweep += now_s() - t0;
}

// The regions of `w` (sorted by reference start) fall into clusters: maximal runs that overlap or touch on the reference
// (the left region of a MUM and the right region of its predecessor are the same gap, one base apart at the start, so
// pairs are the rule).  Are the clusters pairwise disjoint in EVERY genome?  Then no MUM found in one cluster (nor in
// anything derived from it: children lie inside their parent) can touch, trim or bound anything found in another, and the
// order in which the reference would interleave the clusters is not observable.  first[c] = index of cluster c's first
// region, first[nclusters] = w.size().
bool Aligner::disjoint_clusters(const std::vector<Region>& w, std::vector<size_t>* first) const {
    first->clear();
    const size_t m = w.size();
    long reach = -1;
    for (size_t i = 0; i < m; i++) {
        if (i == 0 || w[i].start[0] > reach + 1) first->push_back(i);
        if (w[i].end[0] > reach) reach = w[i].end[0];
    }
    first->push_back(m);
    const long nc = (long)first->size() - 1;
    if (nc < 2) return true;
    // Collinear genomes hold the clusters in reference order: then "pairwise disjoint" is "every cluster starts after its
    // predecessor ends" -- a pass over the regions' rows, cluster by cluster (contiguous reads).  A genome in which that
    // fails for some pair may still hold them disjoint in another order: those genomes take the sort below.
    std::vector<uint8_t> unsure(n, 0);
    {
        const int threads = prm.cores > 0 ? prm.cores : 1;
        const long kBlock = 128, nblocks = (nc + kBlock - 1) / kBlock;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (long bl = 0; bl < nblocks; bl++) {
            const long c0 = bl * kBlock, c1 = std::min(nc, c0 + kBlock);
            std::vector<long> lo(n), hi(n), prev(n);
            auto extent = [&](long c, long* plo, long* phi) {
                const size_t f = (*first)[(size_t)c], l = (*first)[(size_t)c + 1];
                memcpy(plo, w[f].start, n * sizeof(long)); memcpy(phi, w[f].end, n * sizeof(long));
                for (size_t i = f + 1; i < l; i++) {
                    const long* st = w[i].start; const long* en = w[i].end;
                    for (size_t g = 0; g < n; g++) { if (st[g] < plo[g]) plo[g] = st[g]; if (en[g] > phi[g]) phi[g] = en[g]; }
                }
            };
            if (c0 > 0) extent(c0 - 1, lo.data(), prev.data());
            for (long c = c0; c < c1; c++) {
                extent(c, lo.data(), hi.data());
                if (c > 0)
                    for (size_t g = 1; g < n; g++)
                        if (lo[g] <= prev[g] + 1) __atomic_store_n(&unsure[g], (uint8_t)1, __ATOMIC_RELAXED);      // touching counts too
                prev.swap(hi);
            }
        }
    }
    std::vector<long> todo;
    for (size_t g = 1; g < n; g++) if (unsure[g]) todo.push_back((long)g);
    if (todo.empty()) return true;
    int bad = 0;
    const long ntodo = (long)todo.size();
#pragma omp parallel for schedule(dynamic, 4) num_threads(prm.cores > 0 ? prm.cores : 1) reduction(| : bad)
    for (long t = 0; t < ntodo; t++) {
        if (bad) continue;
        const long g = todo[(size_t)t];
        std::vector<std::pair<long, long>> iv((size_t)nc);
        for (long c = 0; c < nc; c++) {
            long lo = w[(*first)[(size_t)c]].start[g], hi = w[(*first)[(size_t)c]].end[g];
            for (size_t i = (*first)[(size_t)c] + 1; i < (*first)[(size_t)c + 1]; i++) {
                if (w[i].start[g] < lo) lo = w[i].start[g];
                if (w[i].end[g] > hi) hi = w[i].end[g];
            }
            iv[(size_t)c] = std::make_pair(lo, hi);
        }
        if (!std::is_sorted(iv.begin(), iv.end())) std::sort(iv.begin(), iv.end());      // a collinear genome holds them in reference order already
        for (long c = 0; c + 1 < nc; c++)
            if (iv[(size_t)c + 1].first <= iv[(size_t)c].second + 1) { bad = 1; break; }   // touching counts too
    }
    return !bad;
}

// The recursive extension, generation by generation.  The reference pops the region with the smallest reference start,
// processes it and merges its children into the sorted list (doWork :173-317).  Whenever the regions waiting on the list
// are pairwise disjoint in every genome that order cannot be seen in the result (the set of accepted MUMs; MUM ids are
// never read), so all of them -- one generation -- are validated by all threads at once, their engine results coming
// from one batched call, and their children form the next generation.  The first seed is processed alone, as the
// reference does before its first sort (:194-195 precede :291-292).  The moment a generation is not disjoint (rearranged
// genomes, repeats) the in-order replay takes over on exactly the list the reference would hold.
bool Aligner::extend_generations() {
    // what a restart of the in-order replay needs (taken when a LATER generation turns out not to be disjoint: by then the
    // reference's own list -- and with it the tie order of its unstable sort -- can no longer be reconstructed)
    const std::vector<Region> seeds = regions;
    const size_t pool0 = pool.size();
    const std::vector<int> mums0 = mums;
    std::vector<int> seeds_raw;                   // engine results of the seeds, for the restart
    std::function<void()> before_restart = [] {};
    auto restart_in_order = [&]() {
        wait_layout();
        before_restart();
        pool.resize(pool0);
        mums = mums0;
        const long nl = (long)n;
#pragma omp parallel for schedule(dynamic, 4) num_threads(prm.cores > 0 ? prm.cores : 1)
        for (long g = 0; g < nl; g++) {        // the layout before the extension = the marks of the MUMs accepted so far
            layout[(size_t)g].init(genomes[(size_t)g].seq.size() + 1);
            for (size_t x = 0; x < pool0; x++) layout[(size_t)g].set_range(pool[x].start[g], pool[x].end((size_t)g));
        }
        regions = seeds;
        stats.generation_restarts++;
        if (speculation_) prefetch(regions);
        return extend_pass(false);
    };
    std::vector<Region> gen = std::move(regions);
    regions.clear();
    // Engine results are held per region here (no cache, no hashing of 2 x n coordinates per lookup): raw_of[i] indexes
    // raws for gen[i], -1 = not fetched yet.  A hand-over to the in-order replay files them into its cache first.
    std::vector<Raw> raws;
    std::vector<int> gen_raw(gen.size(), -1);
    auto plain_shape = [&](const Region& r) {                    // the region is ONE engine request (rows only: any thread)
        const long len0 = r.length[0];
        if (len0 <= 0 || prm.p < len0) return false;              // chunked reference (p): chunk_requests' business
        for (size_t g = 0; g < n; g++)
            if (r.start[g] < 0 || r.length[g] < 0 || r.start[g] + r.length[g] > gsize_[g]) return false;
        return true;
    };
    auto plain_request = [&](const Region& r, Request* q) {
        if (!plain_shape(r)) return false;
        *q = Request{r.start, r.length, min_length(false, r.slength), r.start[0], 0, true};
        return true;
    };
    // the same for a list: the row checks (8 000 regions x 201 genomes per generation) by all threads, the minimum
    // lengths (memoised per slength) by this one
    auto plain_requests = [&](const std::vector<Region>& rs, const std::vector<int>* skip, std::vector<Request>* out) {
        const long m = (long)rs.size();
        std::vector<uint8_t> ok((size_t)m, 1);
#pragma omp parallel for schedule(dynamic, 64) num_threads(prm.cores > 0 ? prm.cores : 1) if (m > 256)
        for (long i = 0; i < m; i++) if (!skip || (*skip)[(size_t)i] < 0) ok[(size_t)i] = plain_shape(rs[(size_t)i]) ? 1 : 0;
        out->assign((size_t)m, Request{nullptr, nullptr, 0, 0, 0});
        for (long i = 0; i < m; i++) {
            if (skip && (*skip)[(size_t)i] >= 0) continue;
            if (!ok[(size_t)i]) return false;
            const Region& r = rs[(size_t)i];
            (*out)[(size_t)i] = Request{r.start, r.length, min_length(false, r.slength), r.start[0], 0, true};
        }
        return true;
    };
    auto fetch = [&](const std::vector<Region>& rs, std::vector<int>* raw_of) {      // one engine call for the regions without a result
        std::vector<Request> want, all; std::vector<size_t> who;
        const double tf = now_s();
        if (!plain_requests(rs, raw_of, &all)) return false;
        for (size_t i = 0; i < rs.size(); i++) {
            if ((*raw_of)[i] >= 0) continue;
            want.push_back(all[i]); who.push_back(i);
        }
        if (want.empty()) return true;
        const double tr = now_s();
        std::vector<Raw> got;
        run_batch(want, &got, true);
        const double tb = now_s();
        for (size_t k = 0; k < who.size(); k++) { (*raw_of)[who[k]] = (int)raws.size(); raws.push_back(std::move(got[k])); }
        if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[fetch] requests %.4f s, batch %.4f s, filing %.4f s\n", tr - tf, tb - tr, now_s() - tb);
        return true;
    };
    auto file_into_cache = [&](const std::vector<Region>& rs, const std::vector<int>& raw_of) {
        for (size_t i = 0; i < rs.size(); i++) {
            if (raw_of[i] < 0) continue;
            Request q;
            if (!plain_request(rs[i], &q)) continue;
            q.hash = hash_rows(q.start, q.len, n, q.minsize);
            if (cache_find(q)) continue;
            cache_put(q, false)->raw = raws[(size_t)raw_of[i]];
        }
    };
    before_restart = [&] { if (seeds_raw.size() == seeds.size()) file_into_cache(seeds, seeds_raw); };
    const int threads = prm.cores > 0 ? prm.cores : 1;
    while ((int)memory_->per_thread.size() < threads) memory_->per_thread.emplace_back(new AlignerMemory::PerThread);
    struct Out { std::vector<Mum> accepted; std::vector<Region> kids; };
    // candidates with a reverse-strand member outside their region (see the validation loop), of every generation so far, and
    // for every MUM of the recursion (pool[pool0 + i]) the order key of its region, the region and its place among the region's MUMs
    std::vector<AlignerMemory::ForeignCase> foreign_cases;
    std::vector<long> mum_key; std::vector<int> mum_uid, mum_rank;
    std::vector<std::vector<uint8_t>> rec_words(n);      // one byte per 64 bases of every genome: the recursion has marked there
    for (size_t g = 0; g < n; g++) rec_words[g].assign(genomes[g].seq.size() / 64 + 2, 0);
    int uid_base = 0;
    for (auto& t : memory_->per_thread) t->foreign.clear();
    int gi = 0;                                  // 0: the first seed alone; 1: the other seeds + its children; 2..: children
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double tl = now_s();
    auto lap = [&](const char* what) { if (dbg) { double t = now_s(); fprintf(stderr, "[generation %d] %-12s %.4f s (cpu %.4f)\n", gi, what, t - tl, cpu_lap_s()); tl = t; } };
    while (!gen.empty()) {
        std::vector<Region> now;
        std::vector<int> now_raw;
        std::vector<size_t> first;               // clusters of `now`
        bool trouble = false;
        if (gi == 0) {                           // the first pushed seed, before anything is sorted
            trouble = !fetch(gen, &gen_raw);     // every seed's engine result in ONE call
            seeds_raw = gen_raw;
            now.push_back(gen.front()); now_raw.push_back(gen_raw.front());
            first = {0, 1};
        } else {                                 // sort by reference start, drop a region equal to its successor (:291-306)
            std::vector<Handle> h(gen.size());
            for (size_t i = 0; i < gen.size(); i++) h[i] = Handle{gen[i].start[0], (int)i};
            std::sort(h.begin(), h.end());
            for (size_t i = 0; i < h.size(); i++) {
                const Region& r = gen[(size_t)h[i].idx];
                if (!now.empty() && now.back().start[0] == r.start[0]) {
                    if (now.back().same_as(r, n)) continue;
                    trouble = true;              // two different regions share a reference start: the unstable sort decides
                }
                now.push_back(r); now_raw.push_back(gen_raw[(size_t)h[i].idx]);
            }
            lap("sort");
            if (!trouble) trouble = !disjoint_clusters(now, &first);
            lap("clusters");
            if (!trouble) trouble = !fetch(now, &now_raw);       // children: one more call (usually nothing to fetch)
        }
        lap("sort+fetch");
        std::vector<Request> req;
        if (!trouble && !plain_requests(now, nullptr, &req)) trouble = true;
        lap("requests");
        if (trouble) {
            stats.generation_handover = gi;
            file_into_cache(gen, gen_raw);
            if (gi == 0) { regions = std::move(gen); if (speculation_) prefetch(regions); return extend_pass(false); }
            // after the first seed the reference's list is [other seeds in push order, children of the first in push order],
            // sorted: exactly `gen` as it stands
            if (gi == 1) { regions = std::move(gen); if (speculation_) prefetch(regions); return extend_pass(false, true); }
            return restart_in_order();
        }
        if (gi == 0) { gen.erase(gen.begin()); gen_raw.erase(gen_raw.begin()); } else { gen.clear(); gen_raw.clear(); }
        const long m = (long)now.size();
        const long nclusters = (long)first.size() - 1;
        std::vector<Out> out((size_t)m);
        int cluster_trouble = 0;
        wait_layout();
        double tv = now_s();
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(| : cluster_trouble)
        for (long cl = 0; cl < nclusters; cl++) {
            AlignerMemory::PerThread& tl = *memory_->per_thread[(size_t)omp_get_thread_num()];
            long pending_min = -1;                // smallest reference start among the children produced so far in this cluster
            for (size_t x = first[(size_t)cl]; x < first[(size_t)cl + 1]; x++) {      // the cluster's regions, in the reference's order
                const Region& r = now[x];
                // a child that sorts before (or with) this region would have been processed first, and its engine result
                // is not there yet: give up on the whole generation
                if (pending_min >= 0 && pending_min <= r.start[0]) { cluster_trouble = 1; break; }
                const Raw& raw = raws[(size_t)now_raw[x]];
                Out& o = out[x];
                // the words the touch test of a candidate reads (first and last base in every genome) are requested one
                // candidate ahead: 2 n cache misses in n different bitmaps otherwise
                auto warm = [&](const Raw& w, size_t c) {
                    if (!w.start || c >= w.count) return;
                    const int32_t* st = w.start + c * n; const long lon = w.lon[c];
                    for (size_t j = 0; j < n; j++) { layout[j].prefetch((long)st[j]); layout[j].prefetch((long)st[j] + lon - 1); }
                };
                if (x == first[(size_t)cl]) warm(raw, 0);
                if (x + 1 < first[(size_t)cl + 1]) warm(raws[(size_t)now_raw[x + 1]], 0);
                for (size_t c = 0; c < raw.count; c++) {      // = validate(), with per-thread rows and atomic marks
                    warm(raw, c + 1);
                    Mum mm;
                    const Arena<int32_t>::Mark imark = tl.irows.mark();
                    const Arena<uint8_t>::Mark bmark = tl.brows.mark();
                    mm.start = tl.irows.alloc(n); mm.fwd = tl.brows.alloc(n);
                    bool ok, any_reverse;
                    if (!candidate_rows(r, req[x], raw, c, mm, &ok, &any_reverse)) { tl.irows.rewind(imark); tl.brows.rewind(bmark); continue; }
                    bool touches = false;
                    if (ok && mm.length > 0)
                        for (size_t j = 0; j < n; j++) touches |= layout[j].get(mm.start[j]) | layout[j].get(mm.end(j) - 1);
                    // a reverse-strand member flipped far outside this region (TMum.cpp:33-35) makes the touch test and the trimming
                    // READ layout bits in another cluster's territory, and what is marked there when the reference looks depends on
                    // its order.  The candidate is noted with what it saw and decided again, in the reference's order, when the
                    // recursion is over (engine/store_kernels.h: ForeignRead / ForeignBound; fuzz seed 7059 of round 5)
                    AlignerMemory::ForeignCase* noted = nullptr;
                    if (ok && any_reverse && mm.length >= 5) {
                        bool outside = false;
                        for (size_t j = 0; j < n; j++) outside |= !mm.fwd[j] && ((long)mm.start[j] < r.start[j] - 1 || mm.end(j) > r.end[j] + 1);
                        if (outside && mm.length > 64) cluster_trouble = 1;
                        else if (outside) {
                            tl.foreign.emplace_back();
                            noted = &tl.foreign.back();
                            noted->start.assign(mm.start, mm.start + n); noted->fwd.assign(mm.fwd, mm.fwd + n); noted->mask.resize(n);
                            for (size_t j = 0; j < n; j++) noted->mask[j] = layout[j].bits64(mm.start[j], mm.length);
                            noted->length = mm.length; noted->key = gi == 0 ? -1 : r.start[0] * 4096 + std::min(gi, 4095); noted->uid = uid_base + (int)x;
                            noted->own_before = o.accepted.size(); noted->rstart = r.start; noted->rend = r.end;
                            noted->accepted = false; noted->acc_start0 = 0; noted->acc_length = 0;
                        }
                    }
                    if (!ok || !settle(mm, touches, any_reverse)) { tl.irows.rewind(imark); tl.brows.rewind(bmark); continue; }
                    if (noted) { noted->accepted = true; noted->acc_start0 = mm.start[0]; noted->acc_length = mm.length; }
                    // a reverse-strand member is flipped against the WHOLE genome length (TMum.cpp:33-35), so it can pass the
                    // sequence check while lying outside this region's interval (inverted repeats): its marks could then meet
                    // another cluster's and the order would show.  The in-order replay decides such a generation.
                    if (any_reverse)
                        for (size_t j = 0; j < n; j++)
                            if (!mm.fwd[j] && ((long)mm.start[j] < r.start[j] - 1 || mm.end(j) > r.end[j] + 1)) cluster_trouble = 1;
                    for (size_t j = 0; j < n; j++) {
                        layout[j].set_range_atomic(mm.start[j], mm.end(j));
                        for (long w = (long)mm.start[j] >> 6; w <= (mm.end(j) - 1) >> 6; w++) __atomic_store_n(&rec_words[j][(size_t)w], (uint8_t)1, __ATOMIC_RELAXED);      // (every thread of the generation stores 1)
                    }
                    mm.slength = r.slength;
                    o.accepted.push_back(mm);
                }
                // children: left of the first new MUM, right of each, left of the next (:215-254); one that equals a region
                // still waiting in this cluster is dropped, as the work list drops a region equal to one it holds
                // (worked out in two scratch rows of the thread; only a child that is kept gets rows in the arena)
                if (tl.scratch.size() < 6 * n) tl.scratch.resize(6 * n);
                auto slot = [&](int k) { Region r; r.start = &tl.scratch[(size_t)k * 3 * n]; r.end = r.start + n; r.length = r.end + n; return r; };
                Region lR = slot(0), rR = slot(1);
                auto keep = [&](const Region& k) {
                    if (k.slength <= prm.q) return;
                    for (size_t y = x + 1; y < first[(size_t)cl + 1]; y++) if (now[y].same_as(k, n)) return;
                    Region c;
                    c.start = tl.rows.alloc(n); c.end = tl.rows.alloc(n); c.length = tl.rows.alloc(n);
                    memcpy(c.start, k.start, n * sizeof(long)); memcpy(c.end, k.end, n * sizeof(long)); memcpy(c.length, k.length, n * sizeof(long));
                    c.slength = k.slength; c.llength = k.llength;
                    o.kids.push_back(c);
                    if (pending_min < 0 || c.start[0] < pending_min) pending_min = c.start[0];
                };
                long sj, sl, sp;
                bool lok = false;
                for (size_t i = 0; i < o.accepted.size(); i++) {
                    if (i == 0) lok = neighbour_if_longer(o.accepted[0], true, prm.q, &lR, &sj, &sl, &sp);
                    const bool rok = neighbour_if_longer(o.accepted[i], false, prm.q, &rR, &sj, &sl, &sp);
                    if (lok) keep(lR);
                    if (rok) keep(rR);
                    if (i + 1 < o.accepted.size()) lok = neighbour_if_longer(o.accepted[i + 1], true, prm.q, &lR, &sj, &sl, &sp);
                }
            }
        }
        for (auto& t : memory_->per_thread) { for (auto& f : t->foreign) foreign_cases.push_back(std::move(f)); t->foreign.clear(); }
        if (cluster_trouble) { stats.generation_handover = gi; return restart_in_order(); }
        stats.t_validate += now_s() - tv;
        lap("validate");
        stats.generations++; stats.generation_regions += m;
        for (long x = 0; x < m; x++) {           // commit in list order
            int rank = 0;
            for (Mum& mm : out[(size_t)x].accepted) {
                mm.id = next_id_++; pool.push_back(mm); mums.push_back((int)pool.size() - 1);
                mum_key.push_back(gi == 0 ? -1 : now[(size_t)x].start[0] * 4096 + std::min(gi, 4095)); mum_uid.push_back(uid_base + (int)x); mum_rank.push_back(rank++);
            }
            for (Region& k : out[(size_t)x].kids) { gen.push_back(k); gen_raw.push_back(-1); }
            stats.regions_processed++; stats.cache_hits++;
        }
        lap("commit");
        uid_base += (int)m;
        gi++;
    }
    // The noted candidates again, with the marks the reference's order (the first seed, then always the waiting region with the
    // smallest reference start) had in place in the genomes of their outside members: the anchors' and those of the recursion's
    // MUMs from regions with a smaller (reference start, generation), or earlier in the same region.  As on the device
    // (engine/store_kernels.h: ForeignBound): first with a lower and an upper bound of those marks -- the final marks outside /
    // inside the 64-base words the recursion has touched; trimming is monotone, so equal results, or less than two bases left
    // with the lower bound, settle the candidate -- and only for the others the exact marks, from a walk over the recursion's
    // MUMs.  A different verdict, shift or length: the order shows, the in-order replay decides the run.
    const long ncases = (long)foreign_cases.size();
    int differs = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(| : differs) if (ncases > 8)
    for (long ci = 0; ci < ncases; ci++) {
        const AlignerMemory::ForeignCase& e = foreign_cases[(size_t)ci];
        auto outside = [&](size_t j) { return !e.fwd[j] && ((long)e.start[j] < e.rstart[j] - 1 || (long)e.start[j] + e.length > e.rend[j] + 1); };
        auto rec_bits = [&](size_t j) {      // the bases of the member's interval that lie in a word the recursion has marked
            uint64_t m = 0;
            for (long t = 0; t < e.length; ) {
                const long p = (long)e.start[j] + t, span = std::min<long>(64 - (p & 63), e.length - t);
                if (rec_words[j][(size_t)p >> 6]) m |= (span == 64 ? ~0ull : ((1ull << span) - 1)) << t;
                t += span;
            }
            return m;
        };
        auto exact = [&](size_t j) {         // the anchors' marks + the recursion's that the reference's order had in place
            const long a = e.start[j];
            uint64_t owned = 0, present = 0;
            for (size_t k = pool0; k < pool.size(); k++) {
                const long ar = pool[k].start[j], lo = std::max(ar, a), hi = std::min(ar + pool[k].length, a + e.length);
                if (lo >= hi) continue;
                const uint64_t bits = ((hi - lo) == 64 ? ~0ull : ((1ull << (hi - lo)) - 1)) << (lo - a);
                owned |= bits;
                const size_t i = k - pool0;
                if (mum_key[i] < e.key || (mum_uid[i] == e.uid && (size_t)mum_rank[i] < e.own_before)) present |= bits;
            }
            return (layout[j].bits64(a, e.length) & ~owned) | present;
        };
        auto trim_with = [&](int how, long* pdl, long* plen) {      // how: 0 lower bound, 1 upper bound, 2 exact
            long dl = 0, len = e.length;
            for (size_t j = 0; j < n && len > 0; j++) {
                uint64_t mk = e.mask[j];
                if (outside(j)) mk = how == 2 ? exact(j) : how == 1 ? layout[j].bits64(e.start[j], e.length) : (layout[j].bits64(e.start[j], e.length) & ~rec_bits(j));
                const uint64_t x = (mk >> dl) & (len == 64 ? ~0ull : ((1ull << len) - 1));
                if (!x) continue;
                long l = ~x ? __builtin_ctzll(~x) : 64;
                if (l > len) l = len;
                long rr = 0;
                if (l < len) {
                    const uint64_t y = ~(x << (64 - len));
                    rr = y ? __builtin_clzll(y) : 64;
                    if (rr > len - l) rr = len - l;
                }
                dl += l; len -= l + rr;
            }
            *pdl = dl; *plen = len;
        };
        long dl, len;
        trim_with(0, &dl, &len);
        if (len >= 2) {
            long dh, lh;
            trim_with(1, &dh, &lh);
            if (dh != dl || lh != len) trim_with(2, &dl, &len);
        }
        bool acc = len >= 2 && n > 1 && e.fwd[0];
        if (acc) {
            std::vector<int32_t> st(e.start);
            for (size_t j = 0; j < n; j++) st[j] += (int32_t)dl;
            Mum t; t.start = st.data(); t.fwd = const_cast<uint8_t*>(e.fwd.data()); t.length = len;
            acc = reverse_members_spell(t);
        }
        if (acc != e.accepted || (acc && (e.start[0] + dl != e.acc_start0 || len != e.acc_length))) differs = 1;
    }
    if (differs) { stats.generation_handover = gi; return restart_in_order(); }
    return !mums.empty();
}

// Recursive extension.  Every seed region is known before the loop starts, so their engine results come from one
// batched call.  Children only exist once their parent has been processed in order; the first time the replay needs one
// that is not cached, a speculative sweep over everything still on the work list predicts and batches the rest.
bool Aligner::extend() {
    double t0 = now_s();
    if (res_.active) {
        const bool any = resident_extend();
        if (getenv("PARSNP_DEBUG_TIMERS"))
            fprintf(stderr, "[extend] resident route: %ld generations (%ld regions)%s%s\n", stats.generations, stats.generation_regions, res_.failed ? ", left: " : "", res_.failed ? res_.why.c_str() : "");
        return any;
    }
    speculation_ = test_hook("PARSNP_NO_SPECULATION") == nullptr;
    sweeps_ = 0; misses_since_sweep_ = 0;
    double tr = now_s();
    bool any;
    if (test_hook("PARSNP_SEQUENTIAL_REPLAY") == nullptr) any = extend_generations();
    else { if (speculation_) prefetch(regions); tr = now_s(); any = extend_pass(false); }
    stats.t_replay = now_s() - tr - stats.t_sweep;
    remaining_ = nullptr;
    cache_.clear();
    stats.extend_s = now_s() - t0;
    if (getenv("PARSNP_DEBUG_TIMERS"))
        fprintf(stderr, "[extend] %ld parallel generations (%ld regions), in-order replay from generation %ld, %ld restarts, %ld regions in all\n",
                stats.generations, stats.generation_regions, stats.generation_handover, stats.generation_restarts, stats.regions_processed);
    return any;
}

// filterRandom1 (:327-425): a MUM no longer than rvalue survives only if it is collinear (0..5000 bases, no marked
// base in between) with its successor -- and, when it has one, its predecessor -- in every genome.
void Aligner::filter_mums(int rvalue) {
    wait_layout();
    double t0 = now_s();
    {
        std::vector<Handle> h(mums.size());
        const long nh = (long)mums.size();        // one cache miss per MUM (its row): spread over the threads
        // (the resident route holds the keys in one array: no cache miss per MUM to spread over threads, and no worker to wake)
#pragma omp parallel for schedule(dynamic, 1024) num_threads(prm.cores > 0 ? prm.cores : 1) if (nh > 4096 && !res_.active)
        for (long i = 0; i < nh; i++) h[(size_t)i] = Handle{key0(mums[(size_t)i]), mums[(size_t)i]};
        if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[filter_mums] keys %.4f s\n", now_s() - t0);
        // the list is the anchors followed by the MUMs of the recursion, generation by generation, each in reference order:
        // a few increasing runs, one of them (the anchors) much longer than the rest.  With all keys different there is one
        // sorted order, and merging finds it: the short runs into one list, that list into the long run by block copies
        // (a binary search per element of the short list).  With equal keys the order std::sort leaves is the reference's
        // (sort( mums ) :338), so it runs on the list as it stands.
        std::vector<size_t> runs{0};
        for (size_t i = 1; i < h.size(); i++) if (!(h[i - 1].key < h[i].key)) runs.push_back(i);
        runs.push_back(h.size());
        const size_t nruns = runs.size() - 1;
        bool merged = nruns <= 1;
        if (nruns >= 2 && nruns <= 64) {
            size_t big = 0;
            for (size_t r = 1; r < nruns; r++) if (runs[r + 1] - runs[r] > runs[big + 1] - runs[big]) big = r;
            merged = true;
            std::vector<Handle> small, tmp;
            for (size_t r = 0; r < nruns && merged; r++) {
                if (r == big) continue;
                tmp.clear();
                tmp.reserve(small.size() + runs[r + 1] - runs[r]);
                size_t a = 0, b = runs[r];
                while (a < small.size() && b < runs[r + 1]) {
                    if (small[a].key == h[b].key) { merged = false; break; }
                    tmp.push_back(small[a].key < h[b].key ? small[a++] : h[b++]);
                }
                while (a < small.size()) tmp.push_back(small[a++]);
                while (b < runs[r + 1]) tmp.push_back(h[b++]);
                small.swap(tmp);
            }
            if (merged) {
                std::vector<Handle> out(h.size());
                const Handle* B = h.data() + runs[big]; const Handle* const Bend = h.data() + runs[big + 1];
                size_t o = 0;
                for (size_t x = 0; x < small.size() && merged; x++) {
                    const Handle* at = std::lower_bound(B, Bend, small[x]);
                    if (at != Bend && at->key == small[x].key) { merged = false; break; }
                    std::copy(B, at, out.begin() + (long)o);
                    o += (size_t)(at - B);
                    out[o++] = small[x];
                    B = at;
                }
                if (merged) { std::copy(B, Bend, out.begin() + (long)o); h.swap(out); }
            }
        }
        if (!merged) std::sort(h.begin(), h.end());
        for (size_t i = 0; i < mums.size(); i++) mums[i] = h[i].idx;
    }
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[filter_mums] order %.4f s\n", now_s() - t0);
    long numums = (long)mums.size();
    for (long x = 0; x < numums - 1; x++) {
        const Mum& mt = pool[(size_t)mums[(size_t)x]];
        if (mt.length > rvalue) continue;
        if (res_.active) { res_.failed = true; res_.why = "a MUM short enough for filterRandom1"; return; }      // (the rows and the layout are the device's: host route)
        const Mum& nt = pool[(size_t)mums[(size_t)x + 1]];
        const Mum* prev = x > 0 ? &pool[(size_t)mums[(size_t)x - 1]] : nullptr;
        bool adjacent = true;
        for (size_t k = 0; k < n && adjacent; k++) {
            long gap = labs(nt.start[k]) - labs(mt.end(k));
            if (gap < 0 || gap > 5000) { adjacent = false; break; }
            for (int m = (int)mt.end(k) + 1; m < nt.start[k]; m++)
                if (layout[k].get(m)) { adjacent = false; break; }
            if (prev) {
                long pgap = labs(mt.start[k]) - labs(prev->end(k));
                if (pgap < 0 || pgap > 5000) { adjacent = false; break; }
                for (int m = (int)prev->end(k) + 1; m < mt.start[k]; m++)
                    if (layout[k].get(m)) { adjacent = false; break; }
            }
        }
        if (!adjacent) {
            filtered += 1;
            for (size_t k = 0; k < n; k++) layout[k].clear_range(mt.start[k], mt.end(k));
            mums.erase(mums.begin() + x);
            x -= 1; numums -= 1;
        }
    }
    stats.filter_s += now_s() - t0;
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[filter_mums] all %.4f s\n", now_s() - t0);
}

// Greedy collinear chaining of the MUMs in reference order (setFinalClusters :2563-2719).  The float32 / double
// mix of the gap-ratio test is the reference's.
// the test of one MUM against the open chain (:2596-2700).  It reads the chain only through its last MUM: the chain
// end IS that MUM's end, and every member has the strand flags of the first, so "back" decides alone.
uint8_t Aligner::judge_pair(const Mum& nt, const Mum& back) const {
    const int d = prm.d;
    const float diag_diff = prm.diag_diff;
    bool addmum = true;
    float max_gap = 0;
    float min_gap = d + 10;
    const long blen = back.length;
    for (size_t k = 0; k < n; k++) {
        const long fgap = (long)nt.start[k] - ((long)back.start[k] + blen);   // forward: next start - chain end
        const long rgap = (long)back.start[k] - nt.end(k);   // reverse: previous MUM start - next end
        const bool f = nt.fwd[k] != 0;
        if (f && fgap > max_gap) max_gap = fgap;
        else if (!f && rgap > max_gap) max_gap = fgap;       // sic (:2608-2611)
        if (f && fgap < min_gap) min_gap = fgap;
        else if (!f && rgap < min_gap) min_gap = rgap;
        if (nt.fwd[k] != back.fwd[k]) addmum = false;
        else if (f && fgap < 0) addmum = false;
        else if (!f && fgap >= 0) addmum = false;
        else if (f && fgap > d) addmum = false;
        else if (!f && rgap > d) addmum = false;
        if (!addmum) break;
    }
    if (!addmum) return kClose;
    if (min_gap == 0) min_gap = 1;
    if (max_gap == 0) max_gap = 1;
    if (diag_diff > 1.0) return max_gap - min_gap < diag_diff ? kJoin : kPass;   // kPass: neither joined nor closed (:2684-2692)
    return min_gap / max_gap >= 1.0 - diag_diff ? kJoin : kClose;
}

void Aligner::chain() {
    double t0 = now_s();
    lcbs.clear();
    unique_order = true;
    {
        std::vector<Handle> h(mums.size());
        const long nh = (long)mums.size();        // one cache miss per MUM (its row): spread over the threads
#pragma omp parallel for schedule(dynamic, 1024) num_threads(prm.cores > 0 ? prm.cores : 1) if (nh > 4096 && !res_.active)
        for (long i = 0; i < nh; i++) h[(size_t)i] = Handle{key0(mums[(size_t)i]), mums[(size_t)i]};
        // strictly increasing keys (the list as filter_mums left it) have one sorted order: nothing to do.  With ties
        // the order std::sort leaves is the reference's, so it runs.
        bool increasing = true;
        for (size_t i = 1; i < h.size() && increasing; i++) increasing = h[i - 1].key < h[i].key;
        unique_order = increasing;
        if (!increasing) {
            std::sort(h.begin(), h.end());
            for (size_t i = 0; i < mums.size(); i++) mums[i] = h[i].idx;
        }
    }
    if (mums.empty()) return;
    enum : uint8_t { JOIN = kJoin, CLOSE = kClose, PASS = kPass };
    auto judge = [&](int cur, int back) -> uint8_t { return res_.active ? resident_judge_rows(cur, back) : judge_pair(pool[(size_t)cur], pool[(size_t)back]); };
    if (res_.active) resident_verdicts();      // (the pairs whose predecessor changed: one engine call)
    // almost always the chain's last MUM is the previous MUM of the list: those verdicts are independent, computed ahead
    const long m = (long)mums.size();
    // (a verdict depends on the two MUMs alone: the second chaining pass, after a few LCBs were dissolved, reuses the
    // verdicts of every MUM whose predecessor is still the same)
    std::vector<uint8_t> ahead((size_t)m, CLOSE);
    if (judged_pred_.size() < pool.size()) { judged_pred_.resize(pool.size(), -1); judged_verdict_.resize(pool.size(), CLOSE); }   // earlier verdicts (the first pass) stay
    std::vector<long> lens((size_t)m);          // gathered here: the sequential pass below would miss the cache once per MUM
    lens[0] = pool[(size_t)mums[0]].length;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(prm.cores > 0 ? prm.cores : 1) if (m > 4096 && !res_.active)
    for (long x = 1; x < m; x++) {
        const int cur = mums[(size_t)x], prev = mums[(size_t)x - 1];
        lens[(size_t)x] = pool[(size_t)cur].length;
        if (judged_pred_[(size_t)cur] != prev) { judged_verdict_[(size_t)cur] = judge(cur, prev); judged_pred_[(size_t)cur] = prev; }
        ahead[(size_t)x] = judged_verdict_[(size_t)cur];
    }

    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[chain] order + verdicts %.4f s\n", now_s() - t0);
    auto open_chain = [&](int idx) { Lcb c; c.type = 1; c.mums.push_back(idx); c.length = pool[(size_t)idx].length; return c; };
    auto close_chain = [&](Lcb& c) {      // start of the first MUM, end of the last (Cluster(TMum) LCB.cpp:21-28 + the joins)
        // the reference column now; the rows of the other genomes when the list of LCBs is final (complete_lcbs() on the host route,
        // materialize() on the resident one): the LCB filter and the second chaining pass read the reference column only, and on a
        // rearranged set of 500 genomes the rows of 23 000 LCBs are 2 x 23 MB per pass
        const Mum& b = pool[(size_t)c.mums.back()];
        c.start.assign(1, key0(c.mums.front())); c.end.assign(1, key0(c.mums.back()) + b.length);
        lcbs.push_back(c);
    };
    Lcb cluster = open_chain(mums[0]);
    bool addmum = true;
    for (long x = 1; x < m; x++) {
        const long nt_length = lens[(size_t)x];
        if (nt_length < random) { addmum = true; continue; }
        if (!addmum) cluster = open_chain(mums[(size_t)x - 1]);
        addmum = true;
        const int back = cluster.mums.back();
        const uint8_t v = back == mums[(size_t)x - 1] ? ahead[(size_t)x] : judge(mums[(size_t)x], back);
        if (v == PASS) continue;
        if (v == JOIN) {
            cluster.length += nt_length;
            cluster.mums.push_back(mums[(size_t)x]);
        } else {
            addmum = false;
            close_chain(cluster);
        }
    }
    if (!addmum) cluster = open_chain(mums.back());
    close_chain(cluster);
    stats.lcb_s += now_s() - t0;
}

// start / end rows of the LCBs (Cluster(TMum) LCB.cpp:21-28 + the joins: the start of the first MUM, the end of the last), for the
// list as it stands; the host route's rows live with the MUMs
void Aligner::complete_lcbs() {
    if (res_.active) return;      // (resident route: the rows arrive with materialize())
    const long nl = (long)lcbs.size();
#pragma omp parallel for schedule(dynamic, 64) num_threads(prm.cores > 0 ? prm.cores : 1) if (nl > 256)
    for (long x = 0; x < nl; x++) {
        Lcb& c = lcbs[(size_t)x];
        if (c.type != 1 || c.mums.empty() || c.start.size() == n) continue;
        const Mum& f = pool[(size_t)c.mums.front()];
        const Mum& b = pool[(size_t)c.mums.back()];
        c.start.assign(f.start, f.start + n);
        c.end.resize(n);
        for (size_t k = 0; k < n; k++) c.end[k] = b.end(k);
    }
}

static void sort_lcbs(std::vector<Lcb>& v) {
    std::vector<Handle> h(v.size());
    for (size_t i = 0; i < v.size(); i++) h[i] = Handle{v[i].start[0], (int)i};
    std::sort(h.begin(), h.end());
    std::vector<Lcb> out;
    out.reserve(v.size());
    for (auto& x : h) out.push_back(std::move(v[(size_t)x.idx]));
    v.swap(out);
}

// filterRandomClustersSimple1 (:433-497): LCBs whose MUM lengths sum to <= c are dissolved (the last one is never
// examined); their MUMs leave the layout and the MUM list.
void Aligner::filter_lcbs() {
    double t0 = now_s();
    sort_lcbs(lcbs);
    const long count = (long)lcbs.size();
    std::vector<int32_t> unmark;      // (resident route: the layout is the device's)
    // (the reference erases every dissolved MUM from the list and every dissolved LCB from its list one at a time, :460-470 -- quadratic
    // on a rearranged set with 20 000 short LCBs; which ones go does not depend on the order, so: marked, then both lists swept once)
    std::vector<char> dead_mum(pool.size(), 0), dead_lcb((size_t)count, 0);
    std::vector<int> gone;
    bool any = false;
    for (long x = 0; x < count - 1; x++) {      // the last LCB is never examined (:447)
        if (lcbs[(size_t)x].length > prm.c) continue;
        filtered_lcbs += 1;
        dead_lcb[(size_t)x] = 1; any = true;
        for (int idx : lcbs[(size_t)x].mums) {
            filtered += 1;
            if (res_.active) unmark.push_back(pool[(size_t)idx].row);
            else gone.push_back(idx);
            dead_mum[(size_t)idx] = 1;
        }
    }
    if (!gone.empty()) {      // out of the layout (:460-466): a genome per task (its bitmap is its own)
        const long ng = (long)n;
#pragma omp parallel for schedule(dynamic, 4) num_threads(prm.cores > 0 ? prm.cores : 1) if (gone.size() * n > 100000)
        for (long k = 0; k < ng; k++)
            for (int idx : gone) { const Mum& mt = pool[(size_t)idx]; layout[(size_t)k].clear_range(mt.start[k], mt.end((size_t)k)); }
    }
    if (any) {
        size_t w = 0;
        for (size_t i = 0; i < mums.size(); i++) if (!dead_mum[(size_t)mums[i]]) mums[w++] = mums[i];
        mums.resize(w);
        w = 0;
        for (size_t x = 0; x < (size_t)count; x++) if (!dead_lcb[x]) { if (w != x) lcbs[w] = std::move(lcbs[x]); w++; }
        lcbs.resize(w);
    }
    if (!unmark.empty() && pm_store_unmark(session_, unmark.data(), (int64_t)unmark.size()) != PM_OK) fatal(std::string("cannot take dissolved LCBs out of the layout: ") + pm_last_error());
    stats.lcb_s += now_s() - t0;
}

// setInterClusterRegions (:2389-2460): between consecutive LCBs that do not overlap in any genome, a type-0 filler
// from the end of the first to the next marked base is PREPENDED; fillers are never printed but shift LCB numbers.
void Aligner::fill_between() {
    double t0 = now_s();
    sort_lcbs(lcbs);
    if (res_.active) { resident_fill_between(); stats.lcb_s += now_s() - t0; return; }
    complete_lcbs();
    // every pair of consecutive LCBs is looked at on its own (bitmap reads only): all threads, results kept in order
    const long npairs = (long)lcbs.size() - 1;
    std::vector<std::unique_ptr<Lcb>> made(npairs > 0 ? (size_t)npairs : 0);
#pragma omp parallel for schedule(dynamic, 16) num_threads(prm.cores > 0 ? prm.cores : 1)
    for (long x = 0; x < npairs; x++) {
        const Lcb& ct = lcbs[(size_t)x];
        const Lcb& nx = lcbs[(size_t)x + 1];
        bool add = true;
        std::vector<long> start, end;
        int flag = 0;
        for (size_t a = 0; a < n; a++) {
            if (nx.start[a] - ct.end[a] <= 0) { add = false; break; }
            long stop = (long)genomes[a].seq.size();
            start.push_back(ct.end[a]);
            if (ct.end[a] + 1 <= stop) {            // the scan runs: first marked base in (end, stop]; [stop] is the sentinel
                long m = layout[a].next_set(ct.end[a] + 1);
                flag = 1;
                end.push_back(m - 1);
            } else if (!flag) {                     // scan did not run: the flag of the previous genome decides (:2419-2433)
                end.push_back(stop - 1);
            }
        }
        if (!add) continue;
        if (end.size() != n) fatal("inter-cluster region bookkeeping would overrun in the reference");
        std::unique_ptr<Lcb> f(new Lcb);
        f->type = 0;
        f->start = start;
        f->end.resize(n);
        for (size_t a = 0; a < n; a++) f->end[a] = end[a] + 1;   // Cluster(bmum,0).addMum(emum): end = emum.end = end+1
        f->length = 2;
        for (size_t a = 0; a < n; a++)
            if (f->end[a] - f->start[a] < 5) { add = false; break; }
        if (add) made[(size_t)x] = std::move(f);
    }
    std::vector<Lcb> fillers;
    for (auto& f : made) if (f) fillers.push_back(std::move(*f));
    lcbs.insert(lcbs.begin(), fillers.begin(), fillers.end());
    stats.lcb_s += now_s() - t0;
}

}  // namespace parsnp
