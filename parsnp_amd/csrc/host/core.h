// core.h -- one parsnp_core run as an object: configuration, ingest, upload, the timed path (phases A-D of the
// reference's main(), src/parsnp.cpp:3187-3270) and output.  Used by main.cpp (the drop-in binary) and by capi.cpp
// (in-process stepping for bench.py and tests).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "aligner.h"

namespace parsnp {

struct StepReport {
    double setup_s = 0;   // per-run host state: layout bitmaps, arenas (and release of the previous step's)
    double path_s = 0, anchor_s = 0, extend_s = 0, filter_s = 0, lcb_s = 0, finder_s = 0;
    double alg_bytes = 0, alg_bytes_kernel = 0, alg_bytes_query = 0;
    long finder_calls = 0, finder_regions = 0, regions_processed = 0, cache_hits = 0, cache_misses = 0, spec_rounds = 0;
    long anchors = 0, mums = 0, lcbs = 0, core_bp = 0;
    bool mums_found = false;
    double h2d_bytes = 0, d2h_bytes = 0;   // what the engine moved over the host link during the step (pm_session_traffic)
    std::vector<std::pair<std::string, double>> engine_ms, anchor_ms;
    std::string resident_why;   // why the resident route was not taken or was left (empty: it was taken)
    Stats host;
};

// sharded run: one process per GPU, query genomes split into contiguous blocks (include/parsnp_mum.h)
struct ShardSpec {
    int rank = 0, world = 1;
    pm_allreduce_min_i32_fn allreduce_min = nullptr;
    pm_allgather_fn allgather = nullptr;
    void* ctx = nullptr;
    // device collectives: the engine's own RCCL communicator (pm_session_create_rccl) instead of the two callbacks
    bool rccl = false;
    uint8_t rccl_id[PM_RCCL_ID_BYTES] = {0};
};

class CoreRun {
public:
    ~CoreRun();
    ShardSpec shard;   // set before open()
    // reads the ini, ingests every genome (printing what the reference prints) and uploads them to the engine.
    // returns 0, or the exit code the reference would use (1) / 3 when the engine cannot start.
    int open(const std::string& ini_path);
    // calcmumi=1: pairwise MUMi distance of every query to the reference -> <outdir>/all.mumi (Aligner::setMumi,
    // src/parsnp.cpp:1869-2115); returns 0 on success
    int mumi();
    // phases A-D on a fresh Aligner over the resident genomes
    StepReport step();
    StepReport step_once(bool resident);     // (resident: the resident route may be taken, resident.cpp)
    // XMFA + log of the last step (phase E)
    void write(bool* gap_note);
    Params prm;
    std::vector<Genome> genomes;
    int qfiles = 0;
    double ingest_s = 0, upload_s = 0;
    AlignerMemory memory;                 // arenas kept across step() calls (declared before align: destroyed after it)
    std::unique_ptr<Aligner> align;
    pm_session* session = nullptr;
private:
    std::string left_why_;                // why a step of this session left the resident route (later steps take the host route at once)
    int said_ = 0;                        // the last progress line of the running step that was printed
    bool say(int stage);
};

}  // namespace parsnp
