// resident.cpp -- the RESIDENT route of the host side: phases A-D with the MUM rows, the layout and the regions kept on the
// device (include/parsnp_mum.h: pm_store_*).  What the engine does per MUM and genome there -- validation and trimming of the
// candidates (second half of setMums1, src/parsnp.cpp:1713-1841, :1399-1477), the neighbour walks of determineRegion
// (:1199-1290), the pairwise test of setFinalClusters (:2596-2700), setInterClusterRegions (:2389-2460) -- this file does not;
// what stays here is the part of the reference whose ORDER is observable and which needs a few bytes per MUM only: the work
// list of doWork (:173-317: sort by reference start, drop a region equal to its successor, ties), the order in which accepted
// MUMs enter the list, the sort of filterRandom1 (:338), the chain walk (:2563-2719) and filterRandomClustersSimple1 (:433-497).
//
// The route is taken for a long anchor list (the engine keeps its rows: the anchor table) and left for the host route of
// aligner.cpp -- by running the step again on a fresh Aligner (CoreRun::step) -- the moment the reference's processing order
// would show: two different regions with one reference start, clusters of waiting regions that are not disjoint in every
// genome, a child region sorting before a region still waiting in its cluster, a reverse-strand member outside its region
// (Aligner::extend_generations hands over to the in-order replay at the same points).  Rows travel to the host once, after
// phase D, for the XMFA writer (materialize()).
#include "aligner.h"
#include "hooks.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>

namespace parsnp {

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
[[noreturn]] void fatal(const std::string& msg) {
    std::cerr << "parsnp_core: " << msg << std::endl;
    exit(1);
}
struct Handle { long key; int idx; };
inline bool operator<(const Handle& a, const Handle& b) { return a.key < b.key; }
void engine_error(const char* what, int rc) {
    if (rc == PM_ELIMIT) {
        std::cerr << "parsnp_core: input exceeds a limit of the multi-MUM engine: " << pm_last_error() << std::endl;
        exit(5);
    }
    fatal(std::string(what) + ": " + pm_last_error());
}
}  // namespace

// Phase A.  The anchor call in resident mode, the validation of its list on the device (pm_store_settle), the accepted anchors
// as MUM records without rows, the seed regions (pm_store_seeds).  Returns true when the route is taken; false leaves the run
// to find_anchors()'s host route with nothing changed -- the engine result, rows fetched, waits in the request cache.
bool Aligner::resident_anchors(const Region& whole, std::vector<int>* found) {
    static const bool off = test_hook("PARSNP_NO_RESIDENT") != nullptr;      // test hook: always the host route
    if (off || !resident_allowed_ || !session_ || n < 2 || prm.anchors_only || prm.random >= 2 || !pool.empty()) { res_.why = "not applicable (switched off, anchors only or a MUM filter length >= 2)"; return false; }
    const int minsize = min_length(true, whole.slength);
    l = (float)minsize;
    std::vector<Request> reqs;
    chunk_requests(whole, minsize, &reqs);
    if (reqs.size() != 1 || !reqs[0].plain) { res_.why = "chunked reference"; return false; }      // (p): the p-loop is the host route's
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    double tl = now_s();
    auto lap = [&](const char* what) { if (dbg) { const double t = now_s(); fprintf(stderr, "[resident anchors] %-14s %.6f s\n", what, t - tl); tl = t; } };
    resident_try_ = true;
    std::vector<Raw> raw;
    run_batch(reqs, &raw, true);
    resident_try_ = false;
    lap("search");
    Raw& a = raw[0];
    const int64_t table = pm_result_table_id(a.owner.get());
    const bool kept = table != 0 && pm_result_store_base(a.owner.get()) == 0;      // the rows stayed on the device
    std::vector<pm_row_info> info = std::move(memory_->anchor_info_store);      // (its storage is kept between runs)
    memory_->anchor_info_store.clear();
    int rc = PM_EAGAIN;
    static const bool fused = test_hook("PARSNP_SPLIT_SETTLE") == nullptr;
    int64_t nreg = 0;
    if (kept) {
        if (info.size() < a.count) info.resize(a.count);
        const double ts = now_s();
        // validation and seed regions in one call (one round trip); the test hook takes the two calls it replaces
        if (fused) rc = pm_store_settle_seeds(session_, table, (int32_t)prm.q, info.data(), &nreg);
        else rc = pm_store_settle(session_, table, info.data());
        if (rc != PM_OK && rc != PM_EAGAIN) engine_error("validation of the anchors on the device failed", rc);
        timing_first_call_ = false; collect_engine_timing();
        stats.t_validate += now_s() - ts;
        lap("settle");
    }
    if (rc != PM_OK) {
        // not this list (short: no anchor table; or too many rows overlapping earlier ones): the host route, from the same result
        if (kept) {
            res_.fallback_start.resize(a.count * n); res_.fallback_strand.resize(a.count * n);
            if (pm_store_rows(session_, nullptr, 0, (int64_t)a.count, 1, res_.fallback_start.data(), res_.fallback_strand.data()) != PM_OK)
                engine_error("cannot fetch the anchor rows", PM_EHIP);
            a.start = res_.fallback_start.data(); a.strand = res_.fallback_strand.data();
        }
        CacheEntry* e = cache_put(reqs[0], false);
        e->raw = std::move(a);
        res_.why = kept ? "too many anchor candidates overlap earlier ones" : "short anchor list (no anchor table)";
        return false;
    }
    // the accepted anchors, in list order: counted now; their MUM records (ids as the sequential loop assigns them: one per
    // constructed candidate) are written by resident_records() while the device works on phases C-D -- nothing on the way to the
    // recursion's search waits for them (a helper thread wrote them until round 5: starting it cost more than it hid)
    res_.active = true; res_.table = table;
    kept_results_.push_back(a.owner);
    pool.clear(); res_.start0.clear(); res_.found_key.clear();
    std::vector<int32_t> acc;
    size_t nacc = 0;
    if (fused) { for (size_t c = 0; c < a.count; c++) nacc += (info[c].state_flags & PM_ST_ACCEPTED) != 0; }
    else {
        acc.reserve(a.count);
        for (size_t c = 0; c < a.count; c++) if (info[c].state_flags & PM_ST_ACCEPTED) acc.push_back((int32_t)c);
        nacc = acc.size();
    }
    found->resize(nacc);
    for (size_t i = 0; i < nacc; i++) (*found)[i] = (int)i;
    res_.anchor_info = std::move(info); res_.anchor_rows = a.count; res_.anchor_lon = a.lon; res_.anchor_slength = whole.slength; res_.anchor_accepted = nacc; res_.records_done = false;
    stats.parallel_candidates += (long)a.count;
    stats.regions_processed++;
    lap("records");
    // seed regions: both neighbours of every anchor, longer than q in every genome (:2150-2172)
    const double tn = now_s();
    if (!fused) {
        rc = pm_store_seeds(session_, table, acc.data(), (int64_t)acc.size(), (int32_t)prm.q, &nreg);
        if (rc != PM_OK) engine_error("seed regions on the device failed", rc);
        collect_engine_timing();
    }
    const pm_region_info* ri = pm_store_new_regions(session_);
    const int32_t* rid = pm_store_new_region_ids(session_);
    // the reference pushes lR unless it equals the right region of the previous anchor, rR unless it equals lR (:2158-2170): equal
    // regions have equal reference columns -- only then is the device asked
    res_.gen_info.clear(); res_.gen_id.clear();
    for (int64_t i = 0; i < nreg; i++) {
        bool drop = false;
        if (!res_.gen_info.empty()) {
            const pm_region_info& p = res_.gen_info.back();
            const bool neighbours = p.key == ri[i].key - 1;      // rR(i) right after lR(i), lR(i) right after rR(i-1) (key = 2 i + side)
            if (neighbours && p.ref_start == ri[i].ref_start && p.ref_len == ri[i].ref_len && p.slength == ri[i].slength) {
                uint8_t same = 0;
                const int32_t x = res_.gen_id.back(), y = rid[i];
                if (pm_store_regions_equal(session_, &x, &y, 1, &same) != PM_OK) engine_error("region comparison failed", PM_EHIP);
                drop = same != 0;
            }
        }
        if (drop) continue;
        res_.gen_info.push_back(ri[i]); res_.gen_id.push_back(rid[i]);
    }
    stats.t_neighbour += now_s() - tn;
    lap("seeds");
    return true;
}

// the MUM records of the accepted anchors (no rows: those stay on the device), in list order, at the head of the pool
void Aligner::resident_records() {
    if (res_.records_done) return;
    res_.records_done = true;
    const size_t nacc = res_.anchor_accepted, count = res_.anchor_rows;
    pool.resize(nacc); res_.start0.resize(nacc);
    res_.of_row.assign(count, -1); res_.len_of_row.assign(count, 0);      // store row -> MUM record (resident_chain reads the device's list through it)
    size_t at = 0;
    long dirty = 0, tangled = 0;
    for (size_t c = 0; c < count; c++) {
        const pm_row_info& r = res_.anchor_info[c];
        const uint32_t st = r.state_flags & 0xffu;
        if (st & PM_ST_BUILT) next_id_++;
        if (!(st & PM_ST_ACCEPTED)) continue;
        Mum m;
        m.id = next_id_ - 1; m.length = r.len; m.slength = res_.anchor_slength; m.row = (int32_t)c;
        m.dirty = (st & PM_ST_FLAGGED) != 0; m.touched = r.len != res_.anchor_lon[c];
        pool[at] = m; res_.start0[at] = r.start0;
        res_.of_row[c] = (int32_t)at; res_.len_of_row[c] = (int32_t)r.len;
        at++;
        dirty += m.dirty; tangled += (st & PM_ST_TANGLED) != 0;
    }
    stats.parallel_dirty += dirty; stats.parallel_tangled += tangled;
    memory_->anchor_info_store = std::move(res_.anchor_info); res_.anchor_info.clear();      // (the storage goes back for the next run)
}

static const char* const kOrderWhy = "a candidate with a reverse-strand member outside its region is decided differently by the reference's order";
// Phase B: the generations of extend_generations() with the per-genome work on the device.
bool Aligner::resident_extend() {
    const double t0 = now_s();
    struct Records { Aligner* a; ~Records() { a->resident_records(); } } records_on_every_way_out{this};
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    std::vector<pm_region_info> gen = std::move(res_.gen_info);
    std::vector<int32_t> gen_id = std::move(res_.gen_id);
    // where the candidates of a region lie in the MUM store (-1: not searched yet), by region id
    std::vector<int64_t> row0; std::vector<int32_t> cnt;
    auto known = [&](int32_t id) { return (size_t)id < row0.size() && row0[(size_t)id] >= 0; };
    std::vector<pm_row_info> info;
    struct Batch { int64_t first, total; };
    std::vector<Batch> batches;
    auto search = [&](const std::vector<pm_region_info>& rs, const std::vector<int32_t>& ids) {      // one engine call for the regions without a result
        std::vector<int32_t> want, mins;
        for (size_t i = 0; i < ids.size(); i++) {
            if (known(ids[i])) continue;
            want.push_back(ids[i]); mins.push_back((int32_t)min_length(false, rs[i].slength));
        }
        if (want.empty()) return;
        const double ts = now_s();
        std::vector<int64_t> off(want.size() + 1);
        int64_t first = 0;
        const int rc = pm_store_search(session_, want.data(), mins.data(), (int64_t)want.size(), &first, off.data());
        if (rc != PM_OK) engine_error("multi-MUM engine failed", rc);
        for (size_t i = 0; i < want.size(); i++) {
            const size_t id = (size_t)want[i];
            if (row0.size() <= id) { row0.resize(id + 1, -1); cnt.resize(id + 1, 0); }
            row0[id] = first + off[i]; cnt[id] = (int32_t)(off[i + 1] - off[i]);
        }
        batches.push_back(Batch{first, off[want.size()]});
        timing_first_call_ = false;
        collect_engine_timing();
        stats.finder_calls++; stats.finder_regions += (long)want.size();
        stats.finder_s += now_s() - ts;
    };
    int gi = 0;
    double tl = now_s();
    auto lap = [&](const char* what) { if (dbg) { double t = now_s(); fprintf(stderr, "[resident generation %d] %-12s %.6f s\n", gi, what, t - tl); tl = t; } };
    // The work list of a generation (:291-306), in two steps around the search of its new regions.
    // sort_unique: sorted by reference start, a region equal to another one with its reference start dropped.  Regions that share a
    // reference start and DIFFER stay, next to each other -- inside an inverted block of some genome the right neighbour of one anchor
    // and the left neighbour of the next are one gap of the reference and two different gaps of that genome -- and tie_runs says
    // where; a run that also lost a duplicate is noted (the reference erases ADJACENT duplicates only, and its unstable sort decides
    // what is adjacent).
    struct TieRun { size_t first, count; bool lost_duplicate; };
    auto sort_unique = [&](const std::vector<pm_region_info>& in, const std::vector<int32_t>& in_id, std::vector<pm_region_info>* out, std::vector<int32_t>* out_id, std::vector<TieRun>* ties) {
        std::vector<Handle> h(in.size());
        for (size_t i = 0; i < in.size(); i++) h[i] = Handle{(long)in[i].ref_start, (int)i};
        std::sort(h.begin(), h.end());
        size_t run0 = out->size();
        bool lost = false;
        auto close_run = [&]() { if (out->size() - run0 > 1) ties->push_back(TieRun{run0, out->size() - run0, lost}); };
        for (size_t i = 0; i < h.size(); i++) {
            const pm_region_info& r = in[(size_t)h[i].idx];
            if (out->size() > run0 && (*out)[run0].ref_start != r.ref_start) { close_run(); run0 = out->size(); lost = false; }
            bool dup = false;
            for (size_t y = run0; y < out->size() && !dup; y++) {
                if ((*out)[y].ref_len != r.ref_len || (*out)[y].slength != r.slength) continue;
                uint8_t same = 0;
                const int32_t x = (*out_id)[y], z = in_id[(size_t)h[i].idx];
                if (pm_store_regions_equal(session_, &x, &z, 1, &same) != PM_OK) engine_error("region comparison failed", PM_EHIP);
                dup = same != 0;
            }
            if (dup) { lost = true; continue; }
            out->push_back(r); out_id->push_back(in_id[(size_t)h[i].idx]);
        }
        close_run();
    };
    // settle_ties (after the search): regions with one reference start are processed in the order the reference's unstable sort
    // leaves them in, which nothing here can know -- but a region WITHOUT candidates changes nothing when it is processed (no MUM,
    // no mark, no child), so the order only shows where two regions of a run have candidates (or the run lost a duplicate of a region
    // that has some: the reference may process that one twice).  Then the route is left; else the run is put in an order of ours.
    auto settle_ties = [&](std::vector<pm_region_info>* list, std::vector<int32_t>* ids, const std::vector<TieRun>& ties) {
        for (const TieRun& t : ties) {
            size_t with = 0, at = t.first;
            for (size_t y = t.first; y < t.first + t.count; y++) if (cnt[(size_t)(*ids)[y]] > 0) { with++; at = y; }
            stats.tie_runs++;
            if (with > 1 || (with == 1 && t.lost_duplicate)) {
                stats.tie_runs_open++;
                res_.failed = true; res_.why = "two different regions with candidates share a reference start";      // the unstable sort decides: host route
                return false;
            }
            if (with == 1 && at != t.first + t.count - 1) { std::swap((*list)[at], (*list)[t.first + t.count - 1]); std::swap((*ids)[at], (*ids)[t.first + t.count - 1]); }
        }
        return true;
    };
    // clusters: maximal runs that overlap or touch on the reference (what the other genomes do to them: the device)
    auto cluster = [&](const std::vector<pm_region_info>& list, size_t base, std::vector<int64_t>* first) {
        long reach = -1;
        for (size_t i = base; i < list.size(); i++) {
            if (i == base || list[i].ref_start > reach + 1) first->push_back((int64_t)i);
            const long end = (long)(list[i].ref_start + list[i].ref_len);
            if (end > reach) reach = end;
        }
        first->push_back((int64_t)list.size());
    };
    static const bool two_stages = test_hook("PARSNP_ONE_STAGE") == nullptr;      // test hook: every generation its own call
    struct Commit { std::vector<pm_region_info> now; std::vector<int32_t> now_id; std::vector<uint8_t> ran; std::vector<int64_t> r0; std::vector<int32_t> rc; int gi; size_t second_stage_from; };
    std::vector<Commit> commits;
    size_t accepted_ahead = 0;      // MUMs of the generations not committed yet
    while (!gen.empty()) {
        std::vector<pm_region_info> now; std::vector<int32_t> now_id;
        std::vector<int64_t> first;
        std::vector<TieRun> ties;
        int64_t stage_first = 0;                      // > 0: the call holds two generations (pm_store_validate)
        std::vector<pm_region_info> rest; std::vector<int32_t> rest_id;
        if (gi == 0) {                  // the first pushed seed, before anything is sorted (:194-195 precede :291-292); every seed's search in ONE call
            search(gen, gen_id);
            lap("search call");
            now.push_back(gen.front()); now_id.push_back(gen_id.front());
            first = {0};
            gen.erase(gen.begin()); gen_id.erase(gen_id.begin());
            // ... and, in the same call, the generation that follows if the first seed pushes no child region (it almost never
            // does): the remaining seeds, sorted.  The device leaves them alone if it does
            if (two_stages && !gen.empty()) {
                rest = gen; rest_id = gen_id;
                sort_unique(gen, gen_id, &now, &now_id, &ties);
                if (!settle_ties(&now, &now_id, ties)) return false;
                cluster(now, 1, &first);
                stage_first = 1;
                gen.clear(); gen_id.clear();
            } else first.push_back(1);
        } else {
            sort_unique(gen, gen_id, &now, &now_id, &ties);
            lap("sort");
            search(now, now_id);
            if (!settle_ties(&now, &now_id, ties)) return false;
            cluster(now, 0, &first);
            gen.clear(); gen_id.clear();
        }
        lap(gi == 0 ? "lists" : "search");
        std::vector<int64_t> r0(now.size()); std::vector<int32_t> rc_(now.size());
        for (size_t i = 0; i < now.size(); i++) { r0[i] = row0[(size_t)now_id[i]]; rc_[i] = cnt[(size_t)now_id[i]]; }
        uint32_t trouble = 0; int64_t nkids = 0;
        const double tv = now_s();
        // the scalars of the candidates about to be decided travel back with the call: the ranges of the searches they came from
        int64_t lo = INT64_MAX, hi = -1;
        for (size_t i = 0; i < now.size(); i++) if (rc_[i] > 0) { lo = std::min(lo, r0[i]); hi = std::max(hi, r0[i] + rc_[i]); }
        if (hi > lo && info.size() < (size_t)hi) info.resize((size_t)hi);
        int32_t second_ran = 0;
        const int64_t ncl = (int64_t)first.size() - 1;
        std::vector<int32_t> done((size_t)ncl, 0);
        int rc = pm_store_validate(session_, now_id.data(), r0.data(), rc_.data(), (int64_t)now.size(), first.data(), ncl, (int32_t)prm.q, &trouble, &nkids,
                                   hi > lo ? lo : 0, hi > lo ? hi - lo : 0, hi > lo ? info.data() + lo : nullptr, stage_first, &second_ran, (int32_t)gi, done.data());
        if (rc != PM_OK) engine_error("validation of a generation on the device failed", rc);
        lap("validate call");
        collect_engine_timing();
        lap("timing");
        if (trouble) {
            res_.failed = true;
            res_.why = (trouble & 2) ? "a reverse-strand member was accepted outside its region where the reference's order shows"
                     : "a region with too many candidates, or more candidates with a member outside their region than the engine notes";
            stats.generation_handover = gi;
            return false;
        }
        // children (the engine lists them parent by parent in push order) -> the next generation, after what is still waiting
        const pm_region_info* ki = pm_store_new_regions(session_);
        const int32_t* kid = pm_store_new_region_ids(session_);
        if (stage_first > 0 && !second_ran) {
            // the first seed pushed children: only it was validated.  What waits is the other seeds, then the children (:215-254)
            first.resize(2); first[1] = 1; done.resize(1);
            gen = std::move(rest); gen_id = std::move(rest_id);
            stage_first = 0;
        }
        // which regions were processed: all of a cluster, none of it (it meets an earlier cluster in some genome and waits for that
        // one), or its first few (a child sorts before the next one).  What was not stays on the work list
        std::vector<uint8_t> ran(now.size(), 0);
        long waiting = 0;
        for (size_t cl = 0; cl + 1 < first.size(); cl++)
            for (int64_t x = first[cl]; x < first[cl + 1]; x++) {
                if (x - first[cl] < done[cl]) ran[(size_t)x] = 1;
                else if (x < (int64_t)now.size()) { gen.push_back(now[(size_t)x]); gen_id.push_back(now_id[(size_t)x]); waiting++; }
            }
        if (!ran[0]) fatal("a generation on the device processed nothing");
        stats.regions_deferred += waiting;
        for (int64_t i = 0; i < nkids; i++) { gen.push_back(ki[i]); gen_id.push_back(kid[i]); }
        stats.t_validate += now_s() - tv;
        lap("validate");
        // The MUM records of this generation (commit) are host work nobody on the device waits for: they are written when the last
        // generation has queued phases C-D, beside them (until round 6 every generation committed at once -- the anchors' 80 000
        // records with the first one, 0.2 ms with the device idle)
        size_t more = 0;
        for (size_t i = 0; i < now.size(); i++)
            for (int64_t c = r0[i]; ran[i] && c < r0[i] + rc_[i]; c++) more += (info[(size_t)c].state_flags & PM_ST_ACCEPTED) != 0;
        accepted_ahead += more;
        commits.push_back(Commit{std::move(now), std::move(now_id), std::move(ran), std::move(r0), std::move(rc_), gi, stage_first > 0 ? (size_t)first[(size_t)stage_first] : (size_t)-1});
        if (gen.empty()) {
            // the last generation: the accepted rows of the store are the run's MUM list.  Phases C-D are queued on the device now
            // (resident_chain() collects them) and run beside the commits below
            resident_chain_begin(mums.size() + accepted_ahead);
            lap("chain queued");
        }
        gi += stage_first > 0 ? 2 : 1;
    }
    resident_records();      // the anchors' records first: the recursion's MUMs follow them in the pool
    for (Commit& cm : commits) {
        // commit in list order (:215-254 push the MUMs of a region in candidate order)
        long processed = 0;
        for (size_t i = 0; i < cm.now.size(); i++) {
            if (!cm.ran[i]) continue;
            processed++;
            if (const char* dump = test_hook("PARSNP_DUMP_VALIDATION"))      // test hook: what the device decided for every candidate of every region
                if (FILE* f = fopen(dump, "a")) {
                    fprintf(f, "region %ld+%ld (generation %d):", (long)cm.now[i].ref_start, (long)cm.now[i].ref_len, cm.gi);
                    for (int64_t c = cm.r0[i]; c < cm.r0[i] + cm.rc[i]; c++) fprintf(f, " [%d len %d shift %d state %02x flags %x]", info[(size_t)c].start0, info[(size_t)c].len, info[(size_t)c].shift, info[(size_t)c].state_flags & 0xffu, info[(size_t)c].state_flags >> 8);
                    fprintf(f, "\n");
                    for (int64_t c = cm.r0[i]; c < cm.r0[i] + cm.rc[i]; c++) {
                        std::vector<int32_t> st(n); std::vector<uint8_t> fw(n);
                        const int32_t row = (int32_t)c;
                        if (pm_store_rows(session_, &row, 0, 1, 1, st.data(), fw.data()) == PM_OK) {
                            fprintf(f, "   device candidate rows (raw)");
                            for (size_t j = 0; j < n; j++) fprintf(f, " %d%c", st[j], fw[j] ? '+' : '-');
                            fprintf(f, "\n");
                        }
                    }
                    fclose(f);
                }
            for (int64_t c = cm.r0[i]; c < cm.r0[i] + cm.rc[i]; c++) {
                const uint32_t st = info[(size_t)c].state_flags & 0xffu;
                if (st & PM_ST_BUILT) next_id_++;
                if (!(st & PM_ST_ACCEPTED)) continue;
                Mum m;
                m.id = next_id_ - 1; m.length = info[(size_t)c].len; m.slength = cm.now[i].slength; m.row = (int32_t)c;
                pool.push_back(m); res_.start0.push_back(info[(size_t)c].start0);
                if (res_.of_row.size() <= (size_t)c) { res_.of_row.resize((size_t)c + 1 + (size_t)c / 8, -1); res_.len_of_row.resize(res_.of_row.size(), 0); }
                res_.of_row[(size_t)c] = (int32_t)pool.size() - 1; res_.len_of_row[(size_t)c] = (int32_t)info[(size_t)c].len;
                // (generation of the region: the second stage of a two-stage call is one later; 0 = the first pushed seed, processed before anything is sorted)
                const int g = cm.gi + (i >= cm.second_stage_from ? 1 : 0);
                res_.found_key.resize(pool.size(), -1);
                res_.found_key.back() = g <= 0 ? -1 : (int64_t)cm.now[i].ref_start * 4096 + (g < 4095 ? g : 4095);
                mums.push_back((int)pool.size() - 1);
            }
            stats.regions_processed++; stats.cache_hits++;
        }
        stats.generations += cm.second_stage_from != (size_t)-1 ? 2 : 1; stats.generation_regions += processed;
    }
    lap("commit");
    if (!res_.chain_queued && !mums.empty()) resident_chain_begin(mums.size());      // (no seed region at all: the anchors are the list)
    if (!res_.chain_queued && gi > 0) {
        // candidates that read marks outside their region, decided again in the reference's order (a queued chain does this itself)
        uint32_t trouble = 0;
        if (pm_store_order_check(session_, &trouble) != PM_OK) engine_error("the order check on the device failed", PM_EHIP);
        collect_engine_timing();
        if (trouble) { res_.failed = true; res_.why = kOrderWhy; stats.generation_handover = gi; return false; }
    }
    stats.extend_s = now_s() - t0;
    stats.t_replay = stats.extend_s;
    return !mums.empty();
}

// Phases C-D on the device (pm_store_chain_begin / _end: the sort of filterRandom1 :338, setFinalClusters :2563-2719,
// filterRandomClustersSimple1 :433-497, the second chaining pass :3261-3268, setInterClusterRegions :2389-2460) where their list
// logic is order-free: diag_diff <= 1 (the ratio test joins or closes: the chain's last MUM is the list predecessor), a filter
// length below every MUM's (accepted MUMs have >= 2 bases), and -- the device's own finding -- no two MUMs with one reference start.
void Aligner::resident_chain_begin(size_t expect) {
    static const bool off = test_hook("PARSNP_NO_DEVICE_CHAIN") != nullptr;      // test hook: the host's list logic over pm_store_judge / _unmark / _fill
    res_.chain_queued = false;
    if (off || expect == 0 || prm.random > 1 || !(prm.diag_diff <= 1.0f)) return;
    const int rc = pm_store_chain_begin(session_, (int64_t)expect, (int32_t)prm.d, prm.diag_diff, (int64_t)prm.c);
    if (rc != PM_OK) engine_error("phases C-D on the device failed", rc);
    collect_engine_timing();
    res_.chain_queued = true;
}
// returns true when mums, lcbs and the counters of the log are those of phases C-D; false: the caller runs filter_mums(), chain(),
// filter_lcbs(), chain(), fill_between() (nothing on the device has changed)
bool Aligner::resident_chain() {
    if (!res_.active || !res_.chain_queued) return false;
    res_.chain_queued = false;
    const double t0 = now_s();
    pm_chain_info ci; const int32_t* rows = nullptr; const uint8_t* heads = nullptr;
    const int rc = pm_store_chain_end(session_, &ci, &rows, &heads);
    if (rc != PM_OK) engine_error("phases C-D on the device failed", rc);
    const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
    if (dbg) fprintf(stderr, "[resident chain] wait         %.6f s\n", now_s() - t0);
    collect_engine_timing();
    if (ci.n_in != (int64_t)mums.size()) fatal("the device's MUM list and the host's differ");
    if (ci.trouble & 4) { res_.failed = true; res_.why = kOrderWhy; return false; }
    if (ci.trouble & 1) {
        // sort( mums ) (:338, :2571) is unstable: what it does with two MUMs of one reference start depends on the list it is handed.
        // The reference's list is the anchors followed by the recursion's MUMs as doWork found them -- region after region in ITS
        // order (the first pushed seed, then always the waiting region with the smallest reference start: a parent before its
        // children), candidate after candidate -- while this route found them generation by generation.  Put them in that order
        // (a region's rows follow one another in the store), then the host's list logic does what the reference's does.
        res_.found_key.resize(pool.size(), -1);
        std::stable_sort(mums.begin() + (long)m0, mums.end(), [&](int a, int b) {
            if (res_.found_key[(size_t)a] != res_.found_key[(size_t)b]) return res_.found_key[(size_t)a] < res_.found_key[(size_t)b];
            return pool[(size_t)a].row < pool[(size_t)b].row;
        });
        res_.chain_why = "two MUMs share a reference start";
        if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[resident] phases C-D by the host's list logic: %s\n", res_.chain_why.c_str());
        return false;
    }
    if (ci.trouble & 2) fatal("inter-cluster region bookkeeping would overrun in the reference");
    // store row -> MUM record (res_.of_row: written with the records, beside phases C-D on the device)
    const std::vector<int32_t>& of = res_.of_row;
    const std::vector<int32_t>& len_of = res_.len_of_row;
    const int64_t top = (int64_t)of.size() - 1;
    mums.resize((size_t)ci.n_mums);
    lcbs.clear();
    lcbs.reserve((size_t)(ci.n_fillers + ci.n_lcbs));
    for (int64_t f = 0; f < ci.n_fillers; f++) { Lcb c; c.type = 0; c.length = 2; lcbs.push_back(std::move(c)); }      // (their rows: nothing reads them)
    if (ci.n_mums > ci.n_in || (ci.n_mums > 0 && !heads[0])) fatal("the device's MUM list does not begin with an LCB head, or is longer than the list it came from");
    for (int64_t x = 0; x < ci.n_mums;) {      // LCB by LCB: from a head to the MUM before the next one
        int64_t y = x + 1;
        while (y < ci.n_mums && !heads[y]) y++;
        lcbs.emplace_back();
        Lcb& c = lcbs.back(); c.type = 1;
        c.mums.resize((size_t)(y - x));
        int* out = c.mums.data(); int* all = mums.data() + x;
        long total = 0;
        for (int64_t z = x; z < y; z++) {
            const int32_t row = rows[z];
            if (row < 0 || row > top || of[(size_t)row] < 0) fatal("the device's MUM list names a row the host does not hold");
            const int idx = of[(size_t)row];
            out[z - x] = idx; all[z - x] = idx;
            total += len_of[(size_t)row];
        }
        c.length = total;
        c.start.assign(1, key0(c.mums.front()));
        c.end.assign(1, key0(c.mums.back()) + pool[(size_t)c.mums.back()].length);
        x = y;
    }
    if ((int64_t)lcbs.size() != ci.n_fillers + ci.n_lcbs) fatal("the device's LCB count and its head flags differ");
    filtered += ci.mums_dissolved; filtered_lcbs += ci.lcbs_dissolved;
    unique_order = true;
    stats.lcb_s += now_s() - t0;
    if (dbg) fprintf(stderr, "[resident chain] with lists   %.6f s\n", now_s() - t0);
    stats.device_chain = 1;
    return true;
}

// setFinalClusters' test of MUM cur against the chain's last MUM, from rows fetched for the pair (a reverse-strand member,
// or a chain whose last MUM is not the previous MUM of the list: rare)
uint8_t Aligner::resident_judge_rows(int cur, int back) {
    const int32_t rows[2] = {pool[(size_t)cur].row, pool[(size_t)back].row};
    std::vector<int32_t> st(2 * n); std::vector<uint8_t> fw(2 * n);
    if (pm_store_rows(session_, rows, 0, 2, 0, st.data(), fw.data()) != PM_OK) engine_error("cannot fetch MUM rows", PM_EHIP);
    Mum a = pool[(size_t)cur], b = pool[(size_t)back];
    a.start = st.data(); a.fwd = fw.data(); b.start = st.data() + n; b.fwd = fw.data() + n;
    return judge_pair(a, b);
}
// verdicts of the consecutive pairs of the list whose predecessor changed since they were last judged
void Aligner::resident_verdicts() {
    const long m = (long)mums.size();
    if (judged_pred_.size() < pool.size()) { judged_pred_.resize(pool.size(), -1); judged_verdict_.resize(pool.size(), kClose); }
    std::vector<int32_t> cur, back; std::vector<int> who;
    for (long x = 1; x < m; x++) {
        const int c = mums[(size_t)x], p = mums[(size_t)x - 1];
        if (judged_pred_[(size_t)c] == p) continue;
        cur.push_back(pool[(size_t)c].row); back.push_back(pool[(size_t)p].row); who.push_back(c);
        judged_pred_[(size_t)c] = p;
    }
    if (cur.empty()) return;
    std::vector<int32_t> mn(cur.size()), mx(cur.size()); std::vector<uint8_t> v(cur.size());
    const int rc = pm_store_judge(session_, cur.data(), back.data(), (int64_t)cur.size(), (int32_t)prm.d, mn.data(), mx.data(), v.data());
    if (rc != PM_OK) engine_error("chaining verdicts on the device failed", rc);
    collect_engine_timing();
    const float diag_diff = prm.diag_diff;
    for (size_t i = 0; i < who.size(); i++) {
        uint8_t out;
        if (v[i] == 2) out = resident_judge_rows(who[i], judged_pred_[(size_t)who[i]]);
        else if (v[i] == 1) out = kClose;
        else {
            // every member forward and every gap in [0, d]: the loop of :2596-2700 leaves max_gap = the largest gap (from 0) and
            // min_gap = the smallest (from d + 10); the ratio test in the reference's float / double mix
            float max_gap = 0, min_gap = (float)(prm.d + 10);
            if ((float)mx[i] > max_gap) max_gap = (float)mx[i];
            if ((float)mn[i] < min_gap) min_gap = (float)mn[i];
            if (min_gap == 0) min_gap = 1;
            if (max_gap == 0) max_gap = 1;
            if (diag_diff > 1.0) out = max_gap - min_gap < diag_diff ? kJoin : kPass;
            else out = min_gap / max_gap >= 1.0 - diag_diff ? kJoin : kClose;
        }
        judged_verdict_[(size_t)who[i]] = out;
    }
}

// setInterClusterRegions (:2389-2460) on the device; the fillers come back as rows
void Aligner::resident_fill_between() {
    const long npairs = (long)lcbs.size() - 1;
    if (npairs <= 0) return;
    std::vector<int32_t> last_of((size_t)npairs), first_next((size_t)npairs);
    for (long x = 0; x < npairs; x++) {
        last_of[(size_t)x] = pool[(size_t)lcbs[(size_t)x].mums.back()].row;
        first_next[(size_t)x] = pool[(size_t)lcbs[(size_t)x + 1].mums.front()].row;
    }
    std::vector<uint8_t> add((size_t)npairs);
    const int rc = pm_store_fill(session_, last_of.data(), first_next.data(), npairs, add.data());
    if (rc != PM_OK) engine_error("inter-LCB regions on the device failed", rc);
    collect_engine_timing();
    const int64_t* fs = pm_store_fill_starts(session_); const int64_t* fe = pm_store_fill_ends(session_);
    std::vector<Lcb> fillers;
    size_t at = 0;
    for (long x = 0; x < npairs; x++) {
        if (add[(size_t)x] == 2) fatal("inter-cluster region bookkeeping would overrun in the reference");
        if (add[(size_t)x] != 1) continue;
        Lcb f;
        f.type = 0; f.length = 2;
        f.start.assign(fs + at, fs + at + n); f.end.assign(fe + at, fe + at + n);
        at += n;
        fillers.push_back(std::move(f));
    }
    lcbs.insert(lcbs.begin(), fillers.begin(), fillers.end());
}

// The rows the XMFA writer reads -- every MUM of an LCB, the LCBs' own start / end rows -- and, for parsnp.unalign, the
// layout: fetched once, after phase D.
void Aligner::materialize() {
    if (!res_.active || res_.materialized) return;
    res_.materialized = true;
    const double t0 = now_s();
    std::vector<int32_t> rows; std::vector<int> who;
    std::vector<uint8_t> seen(pool.size(), 0);
    for (const Lcb& c : lcbs)
        for (int idx : c.mums) if (!seen[(size_t)idx]) { seen[(size_t)idx] = 1; rows.push_back(pool[(size_t)idx].row); who.push_back(idx); }
    int32_t* st = irows_.alloc(rows.size() * n + 1); uint8_t* fw = brows_.alloc(rows.size() * n + 1);
    if (!rows.empty() && pm_store_rows(session_, rows.data(), 0, (int64_t)rows.size(), 0, st, fw) != PM_OK) engine_error("cannot fetch the MUM rows", PM_EHIP);
    for (size_t i = 0; i < who.size(); i++) { pool[(size_t)who[i]].start = st + i * n; pool[(size_t)who[i]].fwd = fw + i * n; }
    for (Lcb& c : lcbs) {
        if (c.type != 1 || c.mums.empty()) continue;
        const Mum& f = pool[(size_t)c.mums.front()]; const Mum& b = pool[(size_t)c.mums.back()];
        c.start.assign(f.start, f.start + n);
        c.end.resize(n);
        for (size_t k = 0; k < n; k++) c.end[k] = b.end(k);
    }
    if (prm.unaligned) {      // write_unaligned walks (and marks) the layout
        wait_layout();
        std::vector<int64_t> off(n + 1);
        const int64_t words = pm_store_layout_words(session_, off.data());
        res_.image.resize((size_t)words);
        if (pm_store_layout(session_, res_.image.data(), words) != PM_OK) engine_error("cannot fetch the layout", PM_EHIP);
        for (size_t j = 0; j < n; j++) layout[j].attach(res_.image.data() + off[j], (size_t)(off[j + 1] - off[j]), genomes[j].seq.size() + 1);
    }
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[resident] rows of %zu MUMs fetched for the writer %.4f s\n", rows.size(), now_s() - t0);
}

}  // namespace parsnp
