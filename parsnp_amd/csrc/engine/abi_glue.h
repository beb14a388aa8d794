// abi_glue.h -- the extern "C" surface of include/parsnp_mum.h on top of Engine<Backend>.
// Included exactly once by a translation unit that has defined `PmBackend` (the backend type),
// `pm_backend_name` and `pm_backend_open(int device, std::string* err)`.
#pragma once
#include <chrono>
#include <memory>
#include <new>

#include "../../../include/parsnp_mum.h"
#include "engine_core.h"

struct pm_session {
    std::unique_ptr<PmBackend> backend;
    std::unique_ptr<pm::Engine<PmBackend>> engine;
    std::vector<pm::PhaseTime> timing;
    float call_wall_ms = 0;
    std::vector<pm::RegInfo> new_regions; std::vector<int32_t> new_region_ids;      // pm_store_seeds / pm_store_validate
    std::vector<int64_t> fill_starts, fill_ends;                                    // pm_store_fill
};
struct pm_result { pm::BatchResult r; };

namespace {
thread_local std::string g_pm_error;
int fail(int code, const std::string& msg) { g_pm_error = msg; return code; }
}  // namespace

extern "C" {

const char* pm_last_error(void) { return g_pm_error.c_str(); }
const char* pm_provider(void) { return pm_backend_name; }

static int session_create(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens, const pm::Collectives* coll) {
    if (!out || n_genomes < 1 || !seqs || !lens) return fail(PM_EINVAL, "bad argument");
    try {
        std::unique_ptr<pm_session> s(new pm_session);
        std::string err;
        const auto t0 = std::chrono::steady_clock::now();
        s->backend.reset(pm_backend_open(device, &err));
        if (!s->backend) return fail(PM_ENODEV, err);
        if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[upload] %-14s %.4f s\n", "runtime start", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        s->engine.reset(new pm::Engine<PmBackend>(*s->backend));
        if (coll) s->engine->set_shard(*coll);
        int rc = s->engine->load_genomes(n_genomes, seqs, lens);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        *out = s.release();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_session_create(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens) {
    return session_create(out, device, n_genomes, seqs, lens, nullptr);
}
int pm_session_create_sharded(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens, int rank,
                              int world, pm_allreduce_min_i32_fn allreduce_min, pm_allgather_fn allgather, void* ctx) {
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && (!allreduce_min || !allgather))) return fail(PM_EINVAL, "bad shard description");
    pm::Collectives c;
    c.rank = rank; c.world = world; c.allreduce_min_i32 = allreduce_min; c.allgather = allgather; c.ctx = ctx;
    return session_create(out, device, n_genomes, seqs, lens, &c);
}
#if defined(PM_HAVE_RCCL)
int pm_rccl_unique_id(uint8_t* id128) {
    if (!id128) return fail(PM_EINVAL, "bad argument");
    Rccl& R = Rccl::get();
    if (!R.ok()) return fail(PM_ENODEV, R.err);
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(PM_EHIP, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return PM_OK;
}
int pm_session_create_rccl(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens, int rank, int world,
                           const uint8_t* id128) {
    if (!out || n_genomes < 1 || !seqs || !lens || !id128 || world < 1 || rank < 0 || rank >= world) return fail(PM_EINVAL, "bad argument");
    try {
        std::unique_ptr<pm_session> s(new pm_session);
        std::string err;
        s->backend.reset(pm_backend_open(device, &err));
        if (!s->backend) return fail(PM_ENODEV, err);
        if (!s->backend->comm_init(rank, world, id128)) return fail(PM_EHIP, s->backend->error());
        s->engine.reset(new pm::Engine<PmBackend>(*s->backend));
        pm::Collectives c;
        c.rank = rank; c.world = world; c.device = true;
        s->engine->set_shard(c);
        int rc = s->engine->load_genomes(n_genomes, seqs, lens);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        *out = s.release();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_session_rccl_ranks(const pm_session* s) { return s ? s->backend->comm_ranks() : 0; }
#endif
void pm_session_destroy(pm_session* s) { delete s; }
int pm_session_genomes(const pm_session* s) { return s ? s->engine->ngen : 0; }

int pm_multi_mum_batch(pm_session* s, int64_t n_regions, const int64_t* starts, const int64_t* lens, const int32_t* minsize, pm_result** out) {
    if (!s || !out || n_regions < 0 || (n_regions > 0 && (!starts || !lens || !minsize))) return fail(PM_EINVAL, "bad argument");
    try {
        std::unique_ptr<pm_result> r(new pm_result);
        const auto w0 = std::chrono::steady_clock::now();
        s->backend->drop_landings();      // (downloads a failed call queued and never waited for must not land in its dead blocks)
        int rc = s->engine->run(n_regions, starts, lens, minsize, &r->r);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        // wall clock of the whole call on the host (uploads, launches, waits, result assembly) next to the device phases
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        *out = r.release();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int64_t pm_result_table_id(const pm_result* r) { return r ? r->r.table_id : 0; }
int64_t pm_result_regions(const pm_result* r) { return r->r.nregions; }
int64_t pm_result_total(const pm_result* r) { return r->r.total; }
const int64_t* pm_result_offsets(const pm_result* r) { return r->r.off.data(); }
const int32_t* pm_result_k(const pm_result* r) { return r->r.k(); }
const int32_t* pm_result_lon(const pm_result* r) { return r->r.lon(); }
const int32_t* pm_result_sp(const pm_result* r) { return r->r.sp(); }
const uint8_t* pm_result_fwd(const pm_result* r) { return r->r.fwd(); }
void pm_result_free(pm_result* r) { delete r; }
int pm_session_tune(pm_session* s, const char* key, int64_t value) {
    if (!s || !key) return fail(PM_EINVAL, "bad argument");
    return s->engine->tune(key, value) ? PM_OK : fail(PM_EINVAL, std::string("unknown tunable or bad value: ") + key);
}
int pm_session_rows(pm_session* s, int enable) {
    if (!s || enable < 0 || enable > 2) return fail(PM_EINVAL, "bad argument");
    s->engine->want_rows = enable != 0; s->engine->resident = enable == 2;
    return PM_OK;
}
int64_t pm_result_store_base(const pm_result* r) { return r ? r->r.store_base : -1; }

// the resident route: one engine method per entry point (engine_core.h)
#define PM_STORE_CALL(expr)                                                                                                             \
    try {                                                                                                                               \
        const auto w0 = std::chrono::steady_clock::now();                                                                               \
        s->backend->drop_landings();                                                                                                    \
        const int rc = (expr);                                                                                                          \
        if (rc) return fail(rc, s->engine->error);                                                                                      \
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());                                                               \
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();                      \
        return PM_OK;                                                                                                                   \
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");                                                 \
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
static_assert(sizeof(pm_row_info) == sizeof(pm::RowInfo) && sizeof(pm_region_info) == sizeof(pm::RegInfo), "resident-route records");
static_assert(PM_EAGAIN == pm::Engine<PmBackend>::kAgain, "PM_EAGAIN");
int pm_store_settle(pm_session* s, int64_t table_id, pm_row_info* rows) {
    if (!s || !rows) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_settle(table_id, (pm::RowInfo*)rows))
}
int pm_store_info(pm_session* s, int64_t first, int64_t count, pm_row_info* out) {
    if (!s || (count > 0 && !out)) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_info(first, count, (pm::RowInfo*)out))
}
int pm_store_seeds(pm_session* s, int64_t table_id, const int32_t* anchors, int64_t n_anchors, int32_t q, int64_t* n_regions) {
    if (!s || n_anchors < 0 || (n_anchors > 0 && !anchors) || !n_regions) return fail(PM_EINVAL, "bad argument");
    *n_regions = 0;
    try {
        const int rc = s->engine->store_seeds(table_id, anchors, n_anchors, q, &s->new_regions, &s->new_region_ids);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        *n_regions = (int64_t)s->new_regions.size();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_store_settle_seeds(pm_session* s, int64_t table_id, int32_t q, pm_row_info* rows, int64_t* n_regions) {
    if (!s || !rows || !n_regions) return fail(PM_EINVAL, "bad argument");
    *n_regions = 0;
    try {
        const auto w0 = std::chrono::steady_clock::now();
        s->backend->drop_landings();      // (downloads a failed call queued and never waited for must not land in its dead blocks)
        const int rc = s->engine->store_settle_seeds(table_id, q, (pm::RowInfo*)rows, &s->new_regions, &s->new_region_ids);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        *n_regions = (int64_t)s->new_regions.size();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
const pm_region_info* pm_store_new_regions(const pm_session* s) { return s ? (const pm_region_info*)s->new_regions.data() : nullptr; }
const int32_t* pm_store_new_region_ids(const pm_session* s) { return s ? s->new_region_ids.data() : nullptr; }
int pm_store_regions_equal(pm_session* s, const int32_t* a, const int32_t* b, int64_t n, uint8_t* same) {
    if (!s || n < 0 || (n > 0 && (!a || !b || !same))) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_regions_equal(a, b, n, same))
}
int pm_store_search(pm_session* s, const int32_t* regions, const int32_t* minsize, int64_t n, int64_t* first_row, int64_t* offsets) {
    if (!s || n < 0 || (n > 0 && (!regions || !minsize)) || !first_row || !offsets) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_search(regions, minsize, n, first_row, offsets))
}
int pm_store_validate(pm_session* s, const int32_t* regions, const int64_t* row_first, const int32_t* row_count, int64_t n_regions,
                      const int64_t* cluster_first, int64_t n_clusters, int32_t q, uint32_t* trouble, int64_t* n_children,
                      int64_t info_first, int64_t info_count, pm_row_info* info, int64_t stage_first, int32_t* second_stage_ran, int32_t generation, int32_t* done) {
    if (!s || generation < 0 || n_regions < 0 || n_clusters < 0 || (n_regions > 0 && (!regions || !row_first || !row_count || !cluster_first)) || !trouble || !n_children || info_count < 0 || (info_count > 0 && !info)) return fail(PM_EINVAL, "bad argument");
    *n_children = 0;
    try {
        const auto w0 = std::chrono::steady_clock::now();
        s->backend->drop_landings();      // (downloads a failed call queued and never waited for must not land in its dead blocks)
        const int rc = s->engine->store_validate(regions, row_first, row_count, n_regions, cluster_first, n_clusters, q, trouble, &s->new_regions, &s->new_region_ids, info_first, info_count, (pm::RowInfo*)info, stage_first, second_stage_ran, generation, done);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        *n_children = (int64_t)s->new_regions.size();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_store_judge(pm_session* s, const int32_t* cur, const int32_t* back, int64_t n, int32_t d, int32_t* min_gap, int32_t* max_gap, uint8_t* verdict) {
    if (!s || n < 0 || (n > 0 && (!cur || !back || !min_gap || !max_gap || !verdict))) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_judge(cur, back, n, d, min_gap, max_gap, verdict))
}
int pm_store_unmark(pm_session* s, const int32_t* rows, int64_t n) {
    if (!s || n < 0 || (n > 0 && !rows)) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_unmark(rows, n))
}
int pm_store_fill(pm_session* s, const int32_t* last_of, const int32_t* first_of_next, int64_t n, uint8_t* add) {
    if (!s || n < 0 || (n > 0 && (!last_of || !first_of_next || !add))) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_fill(last_of, first_of_next, n, add, &s->fill_starts, &s->fill_ends))
}
int pm_store_order_check(pm_session* s, uint32_t* trouble) {
    if (!s || !trouble) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_order_check(trouble))
}
int pm_store_chain_begin(pm_session* s, int64_t n_expected, int32_t d, float diag_diff, int64_t c) {
    if (!s) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_chain_begin(n_expected, d, diag_diff, c))
}
int pm_store_chain_end(pm_session* s, pm_chain_info* info, const int32_t** rows, const uint8_t** heads) {
    if (!s || !info || !rows || !heads) return fail(PM_EINVAL, "bad argument");
    static_assert(sizeof(pm_chain_info) == sizeof(pm::Engine<PmBackend>::ChainInfo), "pm_chain_info");
    PM_STORE_CALL(s->engine->store_chain_end((pm::Engine<PmBackend>::ChainInfo*)info, rows, heads))
}
const int64_t* pm_store_fill_starts(const pm_session* s) { return s ? s->fill_starts.data() : nullptr; }
const int64_t* pm_store_fill_ends(const pm_session* s) { return s ? s->fill_ends.data() : nullptr; }
int pm_store_rows(pm_session* s, const int32_t* rows, int64_t first, int64_t n, int raw, int32_t* start, uint8_t* strand) {
    if (!s || n < 0 || (n > 0 && (!start || !strand))) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_rows(rows, first, n, raw != 0, start, strand))
}
int64_t pm_store_layout_words(pm_session* s, int64_t* word_off) {
    if (!s) return 0;
    try {
        const int64_t w = (int64_t)s->engine->layout_geometry();
        if (word_off) for (size_t j = 0; j < s->engine->lay_off_h.size(); j++) word_off[j] = s->engine->lay_off_h[j];
        return w;
    } catch (...) { return 0; }
}
int pm_store_layout(pm_session* s, uint64_t* out, int64_t words) {
    if (!s || !out) return fail(PM_EINVAL, "bad argument");
    PM_STORE_CALL(s->engine->store_layout(out, words))
}
int pm_session_traffic(const pm_session* s, uint64_t* h2d_bytes, uint64_t* d2h_bytes) {
    if (!s) return PM_EINVAL;
    if (h2d_bytes) *h2d_bytes = s->backend->bytes_h2d;
    if (d2h_bytes) *d2h_bytes = s->backend->bytes_d2h;
    return PM_OK;
}
int32_t* pm_result_start(pm_result* r) { return r->r.start(); }
uint8_t* pm_result_strand(pm_result* r) { return r->r.strand(); }
const uint32_t* pm_result_flags(const pm_result* r) { return r->r.flags(); }
int pm_result_dirty_known(const pm_result* r) { return r->r.dirty_known ? 1 : 0; }

int pm_find_events(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int32_t min_len, int strand,
                   int64_t cap, int64_t* count, int64_t* ev_j, int64_t* ev_l, int32_t* ev_len, int32_t* ev_rep) {
    if (!ref || !query || !count || n < 0 || m < 0) return fail(PM_EINVAL, "bad argument");
    const uint8_t* seqs[2] = {ref, query};
    int64_t lens[2] = {n, m};
    pm_session* s = nullptr;
    int rc = pm_session_create(&s, -1, 2, seqs, lens);
    if (rc) return rc;
    std::unique_ptr<pm_session> guard(s);
    int64_t starts[2] = {0, 0};
    pm::BatchResult br;
    try {
        rc = s->engine->run(1, starts, lens, &min_len, &br, true);
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
    if (rc) return fail(rc, s->engine->error);
    const auto& K = s->engine->ev_key_h; const auto& V = s->engine->ev_val_h;
    const uint64_t lmask = (1ull << s->engine->ev_lbits) - 1;
    int64_t c = 0;
    for (size_t i = 0; i < K.size(); i++) {
        if ((int)(K[i] & 1) != (strand ? 1 : 0)) continue;
        int64_t l = (int64_t)((K[i] >> 1) & lmask);
        if (c < cap) { ev_j[c] = (int64_t)(V[i] >> 32); ev_l[c] = l; ev_len[c] = (int32_t)(V[i] & 0xffffffffu); ev_rep[c] = s->engine->rep_h[(size_t)l]; }
        c++;
    }
    *count = c;
    return PM_OK;
}

int pm_mumi_coverage(pm_session* s, const int64_t* starts, const int64_t* lens, int64_t* covered) {
    if (!s || !starts || !lens || !covered) return fail(PM_EINVAL, "bad argument");
    try {
        pm::BatchResult br;
        int32_t fifteen = 15;
        int rc = s->engine->run(1, starts, lens, &fifteen, &br, false, true);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        s->timing = s->engine->timing;
        for (size_t g = 0; g < s->engine->mumi_covered.size(); g++) covered[g] = s->engine->mumi_covered[g];
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}

int pm_last_timing(const pm_session* cs, int* count, const char** names, float* ms) {
    if (!cs || !count) return PM_EINVAL;
    pm_session* s = const_cast<pm_session*>(cs);
    s->timing = s->engine->timing;
    if (s->call_wall_ms > 0) s->timing.push_back(pm::PhaseTime{"call_wall", s->call_wall_ms});
    if (s->engine->budget_retries) s->timing.push_back(pm::PhaseTime{"budget_retries", (float)s->engine->budget_retries});   // a count, not a time
    s->timing.push_back(pm::PhaseTime{"rest_samples", (float)s->engine->last_rest});      // samples SeedExtend handed to SeedRest
    if (s->engine->last_alg[0] > 0) {      // a search of store regions: its algorithmic bytes (counts as well; the caller never held the rows)
        s->timing.push_back(pm::PhaseTime{"alg_survey", (float)s->engine->last_alg[0]});
        s->timing.push_back(pm::PhaseTime{"alg_kernel", (float)s->engine->last_alg[1]});
        s->timing.push_back(pm::PhaseTime{"alg_query", (float)s->engine->last_alg[2]});
    }
    s->timing.push_back(pm::PhaseTime{"n_positions", (float)s->engine->last_positions});      // counts as well: reference positions of the batch, candidates Master.EP selected, candidates the fold accepted
    s->timing.push_back(pm::PhaseTime{"n_candidates", (float)s->engine->last_candidates});
    s->timing.push_back(pm::PhaseTime{"n_accepted", (float)s->engine->last_accepted});
    s->timing.push_back(pm::PhaseTime{"n_grouped", (float)s->engine->last_grouped});
    if (s->engine->outside_writes) { s->timing.push_back(pm::PhaseTime{"outside_writes", (float)s->engine->outside_writes}); s->engine->outside_writes = 0; }      // accepted members outside their region, checked against the order (OutsideWriteCheck)
    if (s->engine->exact_cluster_tests) { s->timing.push_back(pm::PhaseTime{"exact_cluster_tests", (float)s->engine->exact_cluster_tests}); s->engine->exact_cluster_tests = 0; }      // generations validated with the exact test of their clusters (inversions)
    if (s->engine->tail_repeats) { s->timing.push_back(pm::PhaseTime{"tail_repeats", (float)s->engine->tail_repeats}); s->engine->tail_repeats = 0; }      // searches that repeated a part because a capacity from the last step was too small
    if (s->engine->deferred_regions) { s->timing.push_back(pm::PhaseTime{"deferred_regions", (float)s->engine->deferred_regions}); s->engine->deferred_regions = 0; }      // regions a generation left waiting (their cluster met an earlier one, or a child sorted first)
    s->timing.push_back(pm::PhaseTime{"events", (float)s->engine->last_events});      // a count too: R-unique maximal matches the event search appended (16 B each)
    int capn = *count, n = 0;
    for (const auto& t : s->timing) { if (n < capn) { names[n] = t.name; ms[n] = t.ms; } n++; }
    *count = n < capn ? n : capn;
    return PM_OK;
}

}  // extern "C"
