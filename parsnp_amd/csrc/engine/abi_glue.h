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
int pm_multi_mum_batch_gaps(pm_session* s, int64_t table_id, int64_t n_regions, const pm_gap_ref* gaps, const int64_t* ref_start, const int64_t* ref_len,
                            const int32_t* minsize, int64_t n_explicit, const int64_t* ex_starts, const int64_t* ex_lens, pm_result** out) {
    if (!s || !out || n_regions < 1 || !gaps || !ref_start || !ref_len || !minsize || n_explicit < 0) return fail(PM_EINVAL, "bad argument");
    static_assert(sizeof(pm_gap_ref) == sizeof(pm::GapRef), "pm_gap_ref layout");
    try {
        std::unique_ptr<pm_result> r(new pm_result);
        const auto w0 = std::chrono::steady_clock::now();
        pm::Engine<PmBackend>::GapBatch gb;
        gb.table_id = table_id; gb.gaps = (const pm::GapRef*)gaps; gb.ref_start = ref_start; gb.ref_len = ref_len;
        gb.n_explicit = n_explicit; gb.ex_starts = ex_starts; gb.ex_lens = ex_lens;
        int rc = s->engine->run(n_regions, nullptr, nullptr, minsize, &r->r, false, false, &gb);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        *out = r.release();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_multi_mum_batch_spec(pm_session* s, int64_t table_id, int32_t q, int64_t ref_len_limit, const int32_t* minsize_by_length, int64_t table_len, pm_result** out) {
    if (!s || !out || !minsize_by_length || table_len < 1) return fail(PM_EINVAL, "bad argument");
    try {
        std::unique_ptr<pm_result> r(new pm_result);
        const auto w0 = std::chrono::steady_clock::now();
        s->backend->bind_thread();          // (the call may come from a helper thread: the device is a per-thread setting)
        int rc = s->engine->run_spec(table_id, q, ref_len_limit, minsize_by_length, table_len, &r->r);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        s->call_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
        *out = r.release();
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
const pm_gap_ref* pm_result_spec_refs(const pm_result* r) { return r ? (const pm_gap_ref*)r->r.spec_refs.data() : nullptr; }
const int32_t* pm_result_spec_minsize(const pm_result* r) { return r ? r->r.spec_minsize.data() : nullptr; }
int64_t pm_result_table_id(const pm_result* r) { return r ? r->r.table_id : 0; }
int64_t pm_result_regions(const pm_result* r) { return r->r.nregions; }
int64_t pm_result_total(const pm_result* r) { return r->r.total; }
const int64_t* pm_result_offsets(const pm_result* r) { return r->r.off.data(); }
const int32_t* pm_result_k(const pm_result* r) { return r->r.k(); }
const int32_t* pm_result_lon(const pm_result* r) { return r->r.lon(); }
const int32_t* pm_result_sp(const pm_result* r) { return r->r.sp(); }
const uint8_t* pm_result_fwd(const pm_result* r) { return r->r.fwd(); }
void pm_result_free(pm_result* r) { delete r; }
int pm_layout_image(pm_session* s, int64_t table_id, const int64_t* nbits, const uint8_t* accept, int64_t n_rows,
                    const int32_t* extra_start, const int32_t* extra_len, int64_t n_extra, uint64_t** image) {
    if (!s || !nbits || !image || n_rows < 0 || n_extra < 0) return fail(PM_EINVAL, "bad argument");
    try {
        int rc = s->engine->layout_image(table_id, nbits, accept, n_rows, extra_start, extra_len, n_extra, image);
        if (rc) return fail(rc, s->engine->error);
        if (!s->backend->ok()) return fail(PM_EHIP, s->backend->error());
        return PM_OK;
    } catch (const std::bad_alloc&) { return fail(PM_ENOMEM, "host allocation failed");
    } catch (const pm::Engine<PmBackend>::DeviceOutOfMemory& e) { return fail(PM_ENOMEM, "device allocation of " + std::to_string(e.bytes) + " bytes failed"); }
}
int pm_layout_wait(pm_session* s) {
    if (!s) return fail(PM_EINVAL, "bad argument");
    s->engine->layout_wait();
    return s->backend->ok() ? PM_OK : fail(PM_EHIP, s->backend->error());
}
int pm_session_tune(pm_session* s, const char* key, int64_t value) {
    if (!s || !key) return fail(PM_EINVAL, "bad argument");
    return s->engine->tune(key, value) ? PM_OK : fail(PM_EINVAL, std::string("unknown tunable or bad value: ") + key);
}
int pm_session_rows(pm_session* s, int enable) { if (!s) return fail(PM_EINVAL, "bad argument"); s->engine->want_rows = enable != 0; return PM_OK; }
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
    s->timing.push_back(pm::PhaseTime{"events", (float)s->engine->last_events});      // a count too: R-unique maximal matches the event search appended (16 B each)
    int capn = *count, n = 0;
    for (const auto& t : s->timing) { if (n < capn) { names[n] = t.name; ms[n] = t.ms; } n++; }
    *count = n < capn ? n : capn;
    return PM_OK;
}

}  // extern "C"
