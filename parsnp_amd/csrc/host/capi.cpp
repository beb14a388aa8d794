// capi.cpp -- in-process C API over CoreRun for bench.py and tests (libparsnp_core.so): open once (ingest +
// upload), then step the timed path (phases A-D) any number of times on the resident genomes.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sstream>

#include "core.h"

using namespace parsnp;

extern "C" {

struct pc_run { CoreRun run; StepReport last; std::string json; };

// returns 0 or the process exit code parsnp_core would have used
int pc_open(const char* ini_path, pc_run** out) {
    pc_run* r = new pc_run;
    int rc = r->run.open(ini_path);
    if (rc) { delete r; return rc; }
    *out = r;
    return 0;
}
// the same for rank `rank` of a sharded run (every rank opens the same ini; see pm_session_create_sharded)
int pc_open_sharded(const char* ini_path, int rank, int world, pm_allreduce_min_i32_fn allreduce_min, pm_allgather_fn allgather, void* ctx,
                    pc_run** out) {
    pc_run* r = new pc_run;
    r->run.shard.rank = rank; r->run.shard.world = world;
    r->run.shard.allreduce_min = allreduce_min; r->run.shard.allgather = allgather; r->run.shard.ctx = ctx;
    int rc = r->run.open(ini_path);
    if (rc) { delete r; return rc; }
    *out = r;
    return 0;
}
// the same with the engine's own RCCL communicator (device-resident exchanges): id = the 128 bytes rank 0 got from pc_rccl_id
int pc_rccl_id(uint8_t* id) { return pm_rccl_unique_id(id); }
int pc_open_rccl(const char* ini_path, int rank, int world, const uint8_t* id, pc_run** out) {
    pc_run* r = new pc_run;
    r->run.shard.rank = rank; r->run.shard.world = world; r->run.shard.rccl = true;
    memcpy(r->run.shard.rccl_id, id, PM_RCCL_ID_BYTES);
    int rc = r->run.open(ini_path);
    if (rc) { delete r; return rc; }
    *out = r;
    return 0;
}
int pc_rccl_ranks(pc_run* r) { return pm_session_rccl_ranks(r->run.session); }
// a tunable of the run's engine session (pm_session_tune, include/parsnp_mum.h): bench.py --tune, for before/after measurements
int pc_tune(pc_run* r, const char* key, long long value) { return pm_session_tune(r->run.session, key, (int64_t)value); }
// calcmumi on an opened run: writes <outdir>/all.mumi (rank 0 of a sharded run should be the only one to call pc_write)
int pc_mumi(pc_run* r) { return r->run.mumi(); }
// one pass of phases A-D; returns a JSON report (valid until the next call on this handle)
// (pc_step_brief: the same without the LCBs' reference intervals -- 800 pairs of numbers that only partition mode's exchange step
// reads; a caller that times steps should not pay for printing and parsing them)
static const char* step_report(pc_run* r, bool intervals);
const char* pc_step(pc_run* r) { return step_report(r, true); }
const char* pc_step_brief(pc_run* r) { return step_report(r, false); }
static const char* step_report(pc_run* r, bool intervals) {
    r->last = r->run.step();
    const StepReport& s = r->last;
    std::ostringstream o;
    o.precision(9);
    o << "{\"provider\": \"" << pm_provider() << "\", \"queries\": " << r->run.qfiles << ", \"setup_s\": " << s.setup_s << ", \"path_s\": " << s.path_s << ", \"anchor_s\": " << s.anchor_s
      << ", \"extend_s\": " << s.extend_s << ", \"filter_s\": " << s.filter_s << ", \"lcb_s\": " << s.lcb_s << ", \"finder_s\": " << s.finder_s
      << ", \"ingest_s\": " << r->run.ingest_s << ", \"upload_s\": " << r->run.upload_s << ", \"anchors\": " << s.anchors << ", \"anchor_minsize\": " << (long)r->run.align->l << ", \"mums\": " << s.mums
      << ", \"lcbs\": " << s.lcbs << ", \"core_bp\": " << s.core_bp << ", \"alg_bytes\": " << (long long)s.alg_bytes << ", \"alg_bytes_kernel\": " << (long long)s.alg_bytes_kernel << ", \"alg_bytes_query\": " << (long long)s.alg_bytes_query << ", \"finder_calls\": " << s.finder_calls << ", \"finder_regions\": "
      << s.finder_regions << ", \"regions_processed\": " << s.regions_processed << ", \"cache_hits\": " << s.cache_hits << ", \"cache_misses\": "
      << s.cache_misses << ", \"spec_rounds\": " << s.spec_rounds << ", \"mums_found\": " << (s.mums_found ? "true" : "false")
      << ", \"host_split_s\": {\"validate\": " << s.host.t_validate << ", \"neighbour\": " << s.host.t_neighbour << ", \"key\": " << s.host.t_key
      << ", \"sweep\": " << s.host.t_sweep << ", \"replay\": " << s.host.t_replay << ", \"sort\": " << s.host.t_sort << ", \"unpack\": " << s.host.t_unpack << ", \"pack\": " << s.host.t_pack << "}"
      << ", \"h2d_bytes\": " << (long long)s.h2d_bytes << ", \"d2h_bytes\": " << (long long)s.d2h_bytes << ", \"resident\": " << s.host.resident << ", \"resident_retry\": " << s.host.resident_retry << ", \"device_chain\": " << s.host.device_chain << ", \"tie_fallbacks\": " << s.host.tie_fallbacks << ", \"literal_iterations\": " << s.host.literal_iterations
      << ", \"generations\": " << s.host.generations << ", \"regions_deferred\": " << s.host.regions_deferred << ", \"tie_runs\": " << s.host.tie_runs << ", \"tie_runs_open\": " << s.host.tie_runs_open;
    {
        std::string why = s.resident_why;      // (plain words; quotes and backslashes would break the JSON)
        for (char& ch : why) if (ch == '"' || ch == '\\' || (unsigned char)ch < 32) ch = ' ';
        o << ", \"resident_why\": \"" << why << "\"";
    }
    for (int which = 0; which < 2; which++) {
        o << ", \"" << (which ? "anchor_ms" : "engine_ms") << "\": {";
        const auto& v = which ? s.anchor_ms : s.engine_ms;
        for (size_t i = 0; i < v.size(); i++) o << (i ? ", " : "") << "\"" << v[i].first << "\": " << v[i].second;
        o << "}";
    }
    // reference interval of every real LCB, 1-based inclusive as the XMFA prints it ('> 1:a-b'): the unit that
    // partition mode intersects across partitions (partition.py:35-61)
    o << ", \"lcb_ref_intervals\": [";
    bool first = true;
    for (const Lcb& c : r->run.align->lcbs) {
        if (!intervals) break;
        if (c.type != 1 || c.mums.empty()) continue;
        o << (first ? "" : ", ") << "[" << c.start[0] + 1 << ", " << c.end[0] << "]";
        first = false;
    }
    o << "]}";
    r->json = o.str();
    return r->json.c_str();
}
// the rows of the final MUM list reach the host (resident route: they wait on the device until the writer asks, CoreRun::write);
// returns the milliseconds it took, 0 when they were there already.  bench.py reports it beside ms_per_step (`materialize_ms`)
double pc_materialize(pc_run* r) {
    if (!r->last.mums_found || !r->run.align) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    r->run.align->materialize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int pc_write(pc_run* r) {
    if (!r->last.mums_found) return 1;
    bool note = false;
    r->run.write(&note);
    return 0;
}
void pc_close(pc_run* r) { delete r; }

}  // extern "C"
