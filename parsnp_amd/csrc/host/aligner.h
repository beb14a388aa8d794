// aligner.h -- host side of the parsnp_core replacement: everything of the reference's Aligner that sits
// AROUND the csgmum calls (the csgmum work itself runs on the GPU behind include/parsnp_mum.h).
//
// Reference map (all src/parsnp.cpp unless noted):
//   Genome / ingest()          main() FASTA loop                      :2913-3160
//   Bitmap                     mumlayout (vector<vector<bool>>)       :3181-3186
//   Mum                        TMum                                   src/TMum.cpp:8-162
//   Region                     TRegion                                src/LCR.cpp:12-58
//   Lcb                        Cluster                                src/LCB.cpp:8-57
//   Aligner::find_anchors      setInitialClusters()                   :2121-2174
//   Aligner::validate          setMums1, second half                  :1713-1841
//   Aligner::trim              trim                                   :1399-1477
//   Aligner::neighbour_region  determineRegion                        :1199-1290
//   Aligner::extend            doWork                                 :173-317
//   Aligner::filter_mums       filterRandom1                          :327-425
//   Aligner::chain             setFinalClusters()                     :2563-2719
//   Aligner::filter_lcbs       filterRandomClustersSimple1            :433-497
//   Aligner::fill_between      setInterClusterRegions                 :2389-2460
//   write_output               writeOutput                            :505-1191
#pragma once
#include <ctime>
#include <cstdint>
#include <algorithm>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/parsnp_mum.h"

namespace parsnp {

// CPU seconds of the whole process since the previous call (PARSNP_DEBUG_TIMERS laps: wall time beside the work of all threads)
inline double cpu_lap_s() {
    static double last = 0;
    timespec ts;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    const double t = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, d = t - last;
    last = t;
    return d;
}


struct Params {
    int c = 0, d = 0, q = 0, p = 0, do_align = 0, cores = 2, random = 0;
    bool unaligned = false, recomb_filter = false, anchors_only = false, calc_mumi = false, extend_mums = false;
    float diag_diff = 1.0f, factor = 0.0f;
    std::string anchors, mums, anchorfile, mumfile, prefix, outdir;
};

struct Genome {
    std::string path, fname, header;   // files / fasta / headers of the reference's Aligner
    std::string seq;                   // 'A','C','G','T','N' only; query contigs joined by d+10 'N'
    int size_nopad = 0;                // genome_sizes: length without the contig padding
    float gc = 0, at = 0;              // gcCount / atCount
    std::map<int, std::string> pos2hdr;
};
// returns false when the file cannot be opened; *console receives what the reference prints to stdout for this file
bool ingest(const std::string& path, bool is_ref, bool reverse, int d, Genome* out, std::string* console);

// n+1 bits per genome, bit [n] is the sentinel (parsnp.cpp:3184-3185)
class Bitmap {
public:
    void init(size_t nbits_with_sentinel);
    void init_zero_lazy(size_t nbits) {   // scratch: no sentinel; storage (and its mapped pages) is kept when the size repeats
        const size_t words = (nbits + 63) / 64 + 1;
        nbits_ = nbits;
        if (words_ == words) std::fill(w_, w_ + words_, 0); else fresh(words);
    }
    void release() { fresh(0); nbits_ = 0; }
    // a bitmap whose words live elsewhere (the layout image the engine delivers, pm_layout_image): not cleared, not owned
    void attach(uint64_t* words, size_t nwords, size_t nbits_with_sentinel) {
        decltype(own_)().swap(own_);
        w_ = words; words_ = nwords; nbits_ = nbits_with_sentinel;
        logging_ = false; log_.clear();
    }
    bool attached() const { return w_ && own_.empty(); }
    uint64_t bits64(long a, long len) const {      // marks of [a, a + len), len <= 64: bit t = base a + t (outside the bitmap: unmarked)
        uint64_t m = 0;
        for (long t = a < 0 ? -a : 0; t < len; ) {
            const long p = a + t;
            if (p >= (long)nbits_) break;
            const int lo = (int)(p & 63);
            const long span = std::min<long>(64 - lo, len - t);
            m |= ((__atomic_load_n(&w_[(size_t)p >> 6], __ATOMIC_RELAXED) >> lo) & (span == 64 ? ~0ull : ((1ull << span) - 1))) << t;      // (other threads may be marking: set_range_atomic)
            t += span;
        }
        return m;
    }
    size_t count_set() const { size_t c = 0; for (size_t i = 0; i < words_; i++) c += (size_t)__builtin_popcountll(w_[i]); return c; }
    Bitmap() = default;
    Bitmap(Bitmap&& o) noexcept { *this = std::move(o); }
    Bitmap& operator=(Bitmap&& o) noexcept {
        own_ = std::move(o.own_); w_ = o.w_; words_ = o.words_; nbits_ = o.nbits_; logging_ = o.logging_; log_ = std::move(o.log_);
        o.w_ = nullptr; o.words_ = 0; o.nbits_ = 0;
        return *this;
    }
    Bitmap(const Bitmap&) = delete;
    Bitmap& operator=(const Bitmap&) = delete;
    bool test_and_set(long a, long b) {    // [a,b) := 1; was any of it marked?  (scratch use: no undo log)
        if (a < 0) a = 0;
        if (b > (long)nbits_) b = (long)nbits_;
        bool hit = false;
        while (a < b) {
            const size_t wi = (size_t)a >> 6;
            const int lo = (int)(a & 63);
            const long span = std::min<long>(64 - lo, b - a);
            const uint64_t mask = (span == 64 ? ~0ull : ((1ull << span) - 1)) << lo;
            hit |= (w_[wi] & mask) != 0;
            w_[wi] |= mask;
            a += span;
        }
        return hit;
    }
    bool get(long i) const { return (w_[(size_t)i >> 6] >> (i & 63)) & 1; }
    void prefetch(long i) const { if (i >= 0 && (size_t)i < nbits_) __builtin_prefetch(&w_[(size_t)i >> 6], 1, 1); }
    void prefetch_read(long i) const { if (i >= 0 && (size_t)i < nbits_) __builtin_prefetch(&w_[(size_t)i >> 6], 0, 1); }
    void set_range(long a, long b) {    // [a,b) := 1
        if (a >= 0 && b <= (long)nbits_ && a < b && ((size_t)a >> 6) == ((size_t)(b - 1) >> 6)) {   // inside one word: the common case
            const size_t wi = (size_t)a >> 6;
            const long span = b - a;
            store(wi, w_[wi] | ((span == 64 ? ~0ull : ((1ull << span) - 1)) << (a & 63)));
        } else set_range_slow(a, b);
    }
    void set_range_slow(long a, long b);
    // [a,b) := 1 for a range known to lie inside the bitmap, no undo log active (the bulk marking of validate_parallel)
    void set_range_inside(long a, long b) {
        uint64_t* w = w_;
        const size_t wa = (size_t)a >> 6, wb = (size_t)(b - 1) >> 6;
        const uint64_t first = ~0ull << (a & 63), last = ~0ull >> (63 - ((b - 1) & 63));
        if (wa == wb) { w[wa] |= first & last; return; }
        w[wa] |= first;
        for (size_t i = wa + 1; i < wb; i++) w[i] = ~0ull;
        w[wb] |= last;
    }
    void set_range_atomic(long a, long b);   // [a,b) := 1 with atomic word updates: threads marking neighbouring ranges may share a word
    void clear_range_atomic(long a, long b); // [a,b) := 0, likewise (no undo log)
    void clear_range(long a, long b);   // [a,b) := 0
    long next_set(long from) const;     // smallest i >= from with bit set; the sentinel guarantees one for from <= n
    long prev_set(long from) const;     // largest i <= from with bit set, or -1
    size_t bits() const { return nbits_; }
    // undo log support (speculative replay): words changed since begin_log() are restored by rollback()
    void begin_log() { logging_ = true; log_.clear(); }
    bool logging() const { return logging_; }
    bool any_set(long a, long b) const {   // any marked base in [a,b)?  (touches only the words of the range)
        if (a < 0) a = 0;
        if (b > (long)nbits_) b = (long)nbits_;
        if (a >= b) return false;
        size_t wa = (size_t)a >> 6, wb = (size_t)(b - 1) >> 6;
        uint64_t first = ~0ull << (a & 63), last = ((b & 63) == 0) ? ~0ull : ((1ull << (b & 63)) - 1);
        if (wa == wb) return (w_[wa] & first & last) != 0;
        if (w_[wa] & first) return true;
        for (size_t w = wa + 1; w < wb; w++) if (w_[w]) return true;
        return (w_[wb] & last) != 0;
    }
    void rollback();
    void end_log() { logging_ = false; log_.clear(); }
private:
    // zero-on-demand storage: calloc hands out fresh zero pages for a block this size and resize() does not write to
    // them, so a new bitmap costs nothing until its words are touched (125 MB of bitmaps per run at 200 x 5 Mb)
    template <class T> struct ZeroAlloc {
        typedef T value_type;
        ZeroAlloc() = default;
        template <class U> ZeroAlloc(const ZeroAlloc<U>&) {}
        T* allocate(size_t k) { void* p = calloc(k ? k : 1, sizeof(T)); if (!p) throw std::bad_alloc(); return (T*)p; }
        void deallocate(T* p, size_t) { free(p); }
        template <class U, class... A> void construct(U*, A&&...) {}      // storage is zero already
        template <class U> bool operator==(const ZeroAlloc<U>&) const { return true; }
        template <class U> bool operator!=(const ZeroAlloc<U>&) const { return false; }
    };
    std::vector<uint64_t, ZeroAlloc<uint64_t>> own_;
    uint64_t* w_ = nullptr;       // own_.data(), or the attached words
    size_t words_ = 0;
    void fresh(size_t words) { decltype(own_)().swap(own_); own_.resize(words); w_ = words ? own_.data() : nullptr; words_ = words; }
    size_t nbits_ = 0;
    bool logging_ = false;
    std::vector<std::pair<size_t, uint64_t>> log_;
    inline void store(size_t wi, uint64_t v) { if (logging_ && w_[wi] != v) log_.emplace_back(wi, w_[wi]); w_[wi] = v; }
};

// bump allocator with stable addresses and rewind (per-genome coordinate rows of MUMs and regions live here: at 200
// genomes the reference's three std::vector per object dominate its run time)
template <class T>
class Arena {
public:
    T* alloc(size_t n) {
        if (blocks_.empty() || off_ + n > cap_[cur_]) {
            size_t next = blocks_.empty() ? 0 : cur_ + 1;
            if (next == blocks_.size()) {
                size_t c = std::max(n, kBlock);
                blocks_.emplace_back(new T[c]);
                cap_.push_back(c);
            } else if (cap_[next] < n) {
                blocks_[next].reset(new T[n]); cap_[next] = n;
            }
            cur_ = next; off_ = 0;
        }
        T* p = blocks_[cur_].get() + off_;
        off_ += n;
        return p;
    }
    struct Mark { size_t cur, off; bool empty; };
    Mark mark() const { return Mark{cur_, off_, blocks_.empty()}; }
    void rewind(const Mark& m) { if (m.empty) { cur_ = 0; off_ = 0; } else { cur_ = m.cur; off_ = m.off; } }
    void reset() { cur_ = 0; off_ = 0; }      // start over, keeping the blocks (and their mapped pages)
private:
    static constexpr size_t kBlock = 1 << 20;
    std::vector<std::unique_ptr<T[]>> blocks_;
    std::vector<size_t> cap_;
    size_t cur_ = 0, off_ = 0;
};

// rows of n entries in Aligner's arenas.  TMum keeps start, end and a strand flag per genome as vector<long>/<bool>
// (TMum.h); end == start + length in every genome at all times (constructor TMum.cpp:25-60, trimleft/trimright
// :104-148 move start or end together with length), so only start is stored, as int32 (a genome longer than 2^31 - 1
// is refused at construction): 5 bytes per genome and MUM instead of 20 -- the host phases that walk 60 000 anchors x
// 200 genomes are bound by the bytes of these rows, not by arithmetic.
struct Mum {
    long id = 0;
    long length = 0;
    long slength = 0;
    int32_t* start = nullptr;
    uint8_t* fwd = nullptr;
    bool dirty = false;      // (anchor validation) overlapped an earlier candidate and went through the ordered pass
    bool touched = false;    // ... and was trimmed there
    int32_t row = -1;        // (resident route) its row in the engine's MUM store
    long end(size_t j) const { return (long)start[j] + length; }
};

struct Region {          // rows of n entries in Aligner's arenas; immutable once built
    long* start = nullptr;
    long* end = nullptr;
    long* length = nullptr;
    long slength = 0, llength = 0;
    bool same_as(const Region& o, size_t n) const;   // TRegion operator== (LCR.cpp:48-58)
};

struct Lcb {
    int type = 1;
    std::vector<int> mums;         // indices into Aligner::pool
    std::vector<long> start, end;
    long length = 0;
};

// raw candidate list of one finder request (one reference chunk of one region)
// (a view into the engine's result, which stays alive as long as a cache entry refers to it: nothing is copied)
struct Raw {
    const int32_t* k = nullptr; const int32_t* lon = nullptr;
    const int32_t* sp = nullptr;
    const uint8_t* fwd = nullptr;
    // MUM-row mode of the engine (pm_session_rows): rows of n entries per candidate built on the device, and their flags
    int32_t* start = nullptr; uint8_t* strand = nullptr; const uint32_t* flags = nullptr;
    bool dirty_known = false;
    size_t row0 = 0;             // this list's first candidate in the result
    size_t count = 0;
    std::shared_ptr<pm_result> owner;
};

struct Stats {   // wall-clock split reported next to the reference's own phase timers
    double anchor_s = 0, extend_s = 0, filter_s = 0, lcb_s = 0, finder_s = 0;
    double t_pack = 0;      // request rows -> the flat arrays of the C ABI
    double alg_bytes = 0;   // sum over the regions sent to the engine of (m/4 + 16 m + 16 n) per query genome
    double alg_bytes_query = 0;    // ... of which the query pieces (m/2): the one coalesced stream that reaches the fabric
    double alg_bytes_kernel = 0;   // the same sum of what THIS engine's event search must move: (m + n)/2 + 64 B per sampled K-mer (run_batch)
    long finder_calls = 0, finder_regions = 0, regions_processed = 0, cache_hits = 0, cache_misses = 0, spec_rounds = 0;
    // device-side phase times (HIP events, pm_last_timing): summed over every engine call of the step, and of the
    // anchor call alone (the one launch that sees whole genomes)
    std::vector<std::pair<std::string, double>> engine_ms, anchor_ms;
    long generations = 0, generation_regions = 0, generation_handover = -1;   // parallel generations, regions in them, generation at which the in-order replay took over (-1: never)
    long generation_restarts = 0;   // generation-parallel extension abandoned after its second generation (see extend_generations)
    long regions_deferred = 0;      // (resident route) regions a generation left on the work list: their cluster met an earlier one in some genome, or a child sorted first
    long tie_runs = 0, tie_runs_open = 0;   // (resident route) runs of different regions with one reference start; ... of which more than one region had candidates (the route is left)
    long resident = 0;              // 1: phases A-D ran on the resident route (resident.cpp: rows, layout and regions stayed on the device)
    long device_chain = 0;          // 1: phases C-D (sort, chaining, LCB filter, fillers) came from the device in one call (pm_store_chain_*)
    long resident_retry = 0;        // 1: the resident route was left (the reference's processing order would have shown) and the step ran again on the host route
    long tie_fallbacks = 0, literal_iterations = 0, parallel_candidates = 0, parallel_dirty = 0, parallel_tangled = 0;   // work-list ties between different regions (extend_pass)
    double t_validate = 0, t_neighbour = 0, t_key = 0, t_sweep = 0, t_replay = 0, t_sort = 0, t_unpack = 0;   // host split
};

// The arenas of a run.  A caller that runs the path repeatedly (CoreRun::step) keeps one of these across runs: giving
// hundreds of MB back to the OS and faulting them in again costs more than the reset.
struct AlignerMemory {
    Arena<long> rows, cache_rows, req_rows;
    Arena<int32_t> irows;                    // MUM start rows
    Arena<uint8_t> brows;                    // MUM strand rows
    std::vector<Bitmap> layout;              // the run's mumlayout: storage kept mapped across runs, cleared per run
    bool layout_clean = false;               // `layout` holds the sentinels and nothing else: the run that used it never wrote to it (the resident route)
    std::vector<Bitmap> scratch;             // validate_parallel's scratch bitmaps (each stripe thread clears and fills its own)
    std::vector<int64_t> batch_starts, batch_lens;   // run_batch's flat request arrays
    std::vector<pm_row_info> anchor_info_store;      // resident route: the block the anchors' per-row scalars arrive in, kept between runs (1.3 MB: not zero-filled per step)
    std::vector<Mum> pool_store, candidates;         // storage of Aligner::pool / validate_parallel's candidate records between runs
    std::vector<int> minlen_flat[2]; std::string minlen_expr[2];   // Aligner::min_length for lengths below 2^16, by expression (mums, anchors)
    // extend_generations: a candidate with a reverse-strand member outside its region, with the marks it saw in every genome's
    // interval and what was decided (engine/store_kernels.h: ForeignRead / ForeignBound)
    struct ForeignCase {
        std::vector<int32_t> start; std::vector<uint8_t> fwd; std::vector<uint64_t> mask;
        long length, key; int uid; size_t own_before; const long* rstart; const long* rend;
        bool accepted; long acc_start0, acc_length;
    };
    struct PerThread { Arena<long> rows; Arena<int32_t> irows; Arena<uint8_t> brows; std::vector<long> scratch; std::vector<ForeignCase> foreign; };
    std::vector<std::unique_ptr<PerThread>> per_thread;   // rows written by the threads of the generation-parallel replay
    void reset() { rows.reset(); cache_rows.reset(); req_rows.reset(); irows.reset(); brows.reset(); for (auto& t : per_thread) { t->rows.reset(); t->irows.reset(); t->brows.reset(); } }
};

class Aligner {
    std::unique_ptr<AlignerMemory> own_memory_;   // when the caller did not lend one (declared first: `layout` refers into it)
    AlignerMemory* memory_;
public:
    Aligner(std::vector<Genome>& genomes, const Params& prm, pm_session* session, AlignerMemory* memory = nullptr);
    ~Aligner();
    size_t n;
    Params prm;
    std::vector<Genome>& genomes;
    std::vector<Bitmap>& layout;  // mumlayout; lives in the run's AlignerMemory
    std::vector<Mum> pool;        // every MUM ever accepted; `mums` and Lcb::mums index into it
    std::vector<int> mums;        // this->mums of the reference, in its order
    std::vector<Lcb> lcbs;        // this->clusters
    std::vector<Region> regions;  // this->regions (work list of extend())
    float l = 0;                  // anchor min length ("Mum anchor size")
    long m0 = 0, filtered = 0, filtered_lcbs = 0;
    bool unique_order = false;   // chain(): the MUM list had no two MUMs with the same reference start (one sorted order)
    int random = 0;
    float anchor_time = 0, coarsen_time = 0, random_time = 0, clusters_time = 0, iclusters_time = 0;
    Stats stats;
    bool sharded_ = false;       // (set by the owner) a rank of a sharded run: every rank makes the same engine calls in the same order, no helper thread
    // The resident route (resident.cpp): MUM rows, layout and regions stay on the device; the host keeps the list logic.
    bool resident_allowed_ = true;                       // (set by the owner) false: the run that repeats a step the route was left in
    bool announce_ = true;      // find_anchors prints its two progress lines (false: the step is being run again on the host route, they are out already)
    bool resident_active() const { return res_.active; }
    bool resident_failed() const { return res_.failed; }
    const std::string& resident_why() const { return res_.why; }
    bool resident_chain();                               // phases C-D from the device (resident.cpp); false: the caller runs them
    void materialize();                                  // rows of the LCBs' MUMs (and the layout, for parsnp.unalign) for the writer; no-op on the host route
    long key0(int idx) const { return res_.active ? (long)res_.start0[(size_t)idx] : (long)pool[(size_t)idx].start[0]; }      // reference start of a MUM

    bool find_anchors();       // returns m0 != 0
    bool extend();             // returns !mums.empty()
    void filter_mums(int rvalue);
    void chain();
    void filter_lcbs();
    void fill_between();
    void complete_lcbs();      // the LCBs' start / end rows in every genome (host route; chain() leaves the reference column)

    Region neighbour_region(const Mum& m, bool left);
    void neighbour_into(const Mum& m, bool left, Region* out) const;   // rows of *out already allocated
    bool neighbour_if_longer(const Mum& m, bool left, long q, Region* out, long* short_j, long* short_len, long* short_stop) const;
    Region new_region();

    void wait_layout();           // the layout bitmaps are set up in the background (constructor); the first reader awaits them
    enum : uint8_t { kJoin = 0, kClose = 1, kPass = 2 };
    uint8_t judge_pair(const Mum& nt, const Mum& back) const;     // chain()'s test of a MUM against the open chain's last MUM

private:
    struct Resident {
        bool active = false, failed = false, materialized = false;
        std::string why;
        int64_t table = 0;
        std::vector<int32_t> start0;                                           // reference start of pool[i]
        std::vector<int64_t> found_key;                                        // pool[i] of the recursion: where its REGION stands in the reference's processing order (reference start, generation; -1: the first pushed seed)
        std::vector<pm_region_info> gen_info; std::vector<int32_t> gen_id;     // the seed regions, in push order
        std::vector<int32_t> fallback_start; std::vector<uint8_t> fallback_strand;   // rows fetched for the host route
        std::vector<uint64_t> image;                                           // the layout, fetched for parsnp.unalign
        std::vector<pm_row_info> anchor_info; size_t anchor_rows = 0; const int32_t* anchor_lon = nullptr; long anchor_slength = 0; size_t anchor_accepted = 0;   // what resident_records() writes the anchors' MUM records from
        bool records_done = true;
        std::vector<int32_t> of_row, len_of_row;      // store row -> index of its MUM record in the pool (-1: none), and its length
        bool chain_queued = false;                                             // pm_store_chain_begin is in flight (resident_chain() collects it)
        std::string chain_why;                                                 // why phases C-D fell back to the host's list logic
    } res_;
    bool resident_try_ = false;           // run_batch: ask for resident mode (the anchor call of the route)
    bool layout_written_ = false;         // this run wrote to `layout`
    bool resident_anchors(const Region& whole, std::vector<int>* found);
    bool resident_extend();
    void resident_chain_begin(size_t expect);
    void resident_records();
    void resident_verdicts();
    uint8_t resident_judge_rows(int cur, int back);
    void resident_fill_between();
    std::vector<std::future<void>> layout_ready_;
    // set by validate_parallel for the list it accepted into an EMPTY layout (the anchor call): in every genome the accepted
    // MUMs lie in list order, one after the other without overlap -- then the marked base next to a MUM is its list
    // neighbour's, and find_anchors() derives the seed regions from the rows instead of walking bitmaps
    bool anchors_ordered_ = false;
    std::vector<int> judged_pred_;            // chain(): predecessor against which a MUM was last judged, and the verdict
    std::vector<uint8_t> judged_verdict_;
    pm_session* session_;
    long next_id_ = 1;
    Arena<long>& rows_;      // region coordinate rows
    Arena<int32_t>& irows_;  // MUM start rows
    Arena<uint8_t>& brows_;  // MUM strand rows
    // --- finder plumbing -------------------------------------------------------------------------------------
    // one engine request = one reference chunk of one region; rows of n entries (the region's own rows when the
    // region is a single unclamped chunk, else rows in req_rows_)
    struct Request {
        const long* start; const long* len; int32_t minsize; long ref_ini; uint64_t hash; bool plain = false;   // plain: the rows are the region's own (one unclamped chunk)
    };
    void chunk_requests(const Region& r, int minsize, std::vector<Request>* out);   // the p-chunk loop, :1519-1547
    void run_batch(const std::vector<Request>& reqs, std::vector<Raw>* out, bool rows = false);   // rows: every request is its region (plain)
    void unpack_result(pm_result* res, size_t nregions, bool rows, std::vector<Raw>* out);
    int rows_mode_ = -1; bool rows_supported_ = true;     // pm_session_rows: 0 (sp, fwd), 1 MUM rows, 2 resident; -1: as an earlier run of the session left it
    bool timing_first_call_ = false;
    void collect_engine_timing();
    std::vector<std::shared_ptr<pm_result>> kept_results_;   // results whose row blocks hold the rows of accepted MUMs
    // cache of raw results keyed by request coordinates (results are a pure function of them); entries own a copy of
    // the coordinates and are compared in full on a hash hit
    struct CacheEntry { const long* start; const long* len; int32_t minsize; bool pending; Raw raw; };
    std::unordered_multimap<uint64_t, CacheEntry> cache_;
    Arena<long>& cache_rows_;
    Arena<long>& req_rows_;
    CacheEntry* cache_find(const Request& q);
    CacheEntry* cache_put(const Request& q, bool pending);
    std::unordered_map<long, int> minlen_memo_[2];
    std::vector<long> gsize_;
    // --- setMums1 ---------------------------------------------------------------------------------------------
    int min_length(bool anchors, long slength);
    // finder for one region + validate(); `speculative`: a missing cache entry is recorded in `wanted_` and the
    // region is treated as yielding nothing instead of calling the GPU.
    void region_mums(const Region& r, bool anchors, std::vector<int>* accepted, bool speculative);
    void validate(const Region& r, const Request& q, const Raw& raw, std::vector<int>* accepted);
    void validate_parallel(const Region& r, const Request& q, const Raw& raw, std::vector<int>* accepted, int threads);
    bool candidate_rows(const Region& r, const Request& q, const Raw& raw, size_t c, Mum& m, bool* ok, bool* any_reverse) const;
    bool settle(Mum& m, bool touches, bool any_reverse) const;
    bool reverse_members_spell(const Mum& m) const;      // the sequence check of settle() alone (:1791-1825)
    void trim(Mum& m) const;
    bool extend_pass(bool speculative, bool sorted_start = false);
    bool extend_generations();
    bool disjoint_clusters(const std::vector<Region>& w, std::vector<size_t>* first) const;
    void prefetch(const std::vector<Region>& gen);
    void speculate(std::vector<Region> gen);
    std::function<void(std::vector<Region>*)> remaining_;
    bool speculation_ = true;
    int sweeps_ = 0;
    long misses_since_sweep_ = 0;
    std::vector<Request> wanted_;
    std::vector<CacheEntry*> wanted_entries_;
};

// XMFA + log (writeOutput).  gap_note: set when the gap aligner (gapalign.h) declined an inter-MUM gap and it was written '-'-padded.
void write_output(Aligner& a, const std::string& stem, bool* gap_note);
// parsnp.unalign (setUnalignableRegions); marks every base of the layout bitmaps.
void write_unaligned(Aligner& a);

std::string reverse_complement(const std::string& s);   // Aligner::reversec, :1294-1393

}  // namespace parsnp
