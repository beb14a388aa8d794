// main.cpp -- parsnp_core: same command line, .ini surface, phase order, outputs and exit codes as the
// reference's main() (src/parsnp.cpp:2792-3299), with the MUM search running on the MI355X through
// include/parsnp_mum.h.  Extra, off by default: PARSNP_TIMING=<file> writes a JSON line with the wall-clock split.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>

#include "core.h"
#include "minlen.h"

using namespace parsnp;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char* argv[]) {
    const std::string version = "v1.0.1";
    bool help = false, show_version = false;
    for (int i = 0; i < argc; i++)
        if (argv[i][0] == '-') {
            if (argv[i][1] == 'h') help = true;
            else if (argv[i][1] == 'v') show_version = true;
        }
    if (help) {
        std::cout << "parsnp options:" << std::endl;
        std::cout << "   -h <display this message>" << std::endl;
        std::cout << "   -v <display the version>" << std::endl;
        std::cout << "   <parameter file with options>" << std::endl;
        exit(0);
    }
    if (show_version) { std::cout << "Parsnp " << version << std::endl; exit(0); }
    if (argc < 2) { std::cout << "ERROR: No parameter file specified!" << std::endl; exit(1); }
    if (argc >= 4 && !strcmp(argv[1], "--min-length")) {   // diagnostic: minimum MUM length of an expression for S values
        for (int i = 3; i < argc; i++) {
            int v = 0;
            if (!min_mum_length(argv[2], atol(argv[i]), &v)) { std::cout << "ERROR: cannot evaluate " << argv[2] << std::endl; exit(1); }
            std::cout << argv[i] << " " << v << std::endl;
        }
        exit(0);
    }

    const double t_begin = now_s();
    time_t tstart, start, end;
    time(&tstart);
    CoreRun run;
    // Sharded --no-partition run, one parsnp_core process per GPU: PARSNP_SHARD_WORLD ranks are started by
    // any launcher with PARSNP_SHARD_RANK = 0 .. world-1 (and PARSNP_DEVICE, default = the rank); the engines exchange over
    // their own RCCL communicator, whose id rank 0 publishes in PARSNP_RCCL_ID_FILE.  Rank 0 writes the outputs.
    if (const char* w = getenv("PARSNP_SHARD_WORLD")) {
        const int world = atoi(w), rank = getenv("PARSNP_SHARD_RANK") ? atoi(getenv("PARSNP_SHARD_RANK")) : 0;
        if (world < 1 || rank < 0 || rank >= world) { std::cerr << "parsnp_core: bad PARSNP_SHARD_RANK / PARSNP_SHARD_WORLD" << std::endl; exit(1); }
        if (!getenv("PARSNP_DEVICE")) setenv("PARSNP_DEVICE", std::to_string(rank).c_str(), 1);
        const char* idf = getenv("PARSNP_RCCL_ID_FILE");
        if (!idf || !*idf) { std::cerr << "parsnp_core: a sharded run needs PARSNP_RCCL_ID_FILE (a path every rank can read)" << std::endl; exit(1); }
        run.shard.rank = rank; run.shard.world = world; run.shard.rccl = true;
        // The id file is PARSNP_RCCL_ID_BYTES of communicator id followed by the launch's nonce (PARSNP_SHARD_NONCE: any
        // string the launcher gives to all ranks of ONE launch).  Rank 0 removes what a previous run may have left before it
        // creates its id, publishes id + nonce by rename, and removes the file again at exit; the other ranks accept a file
        // only if it carries their nonce -- or, without a nonce, if it is not older than their own start (a stale id would
        // leave the ranks waiting for each other inside ncclCommInitRank until the engine's watchdog ends them).
        const std::string nonce = getenv("PARSNP_SHARD_NONCE") ? getenv("PARSNP_SHARD_NONCE") : "";
        if (rank == 0) {
            unlink(idf);
            if (pm_rccl_unique_id(run.shard.rccl_id) != PM_OK) { std::cerr << "parsnp_core: " << pm_last_error() << std::endl; exit(3); }
            const std::string tmp = std::string(idf) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(run.shard.rccl_id, 1, PM_RCCL_ID_BYTES, f) != PM_RCCL_ID_BYTES || fwrite(nonce.data(), 1, nonce.size(), f) != nonce.size() || fclose(f) ||
                rename(tmp.c_str(), idf)) {
                std::cerr << "parsnp_core: cannot write " << idf << std::endl; exit(1);
            }
            static std::string id_file_to_remove;
            id_file_to_remove = idf;
            atexit([] { unlink(id_file_to_remove.c_str()); });
        } else {
            bool got = false;
            const time_t started = time(nullptr);
            for (int tries = 0; tries < 1200 && !got; tries++) {       // up to two minutes: rank 0 may still be starting
                FILE* f = fopen(idf, "rb");
                if (f) {
                    char extra[256];
                    struct stat st;
                    const bool whole = fread(run.shard.rccl_id, 1, PM_RCCL_ID_BYTES, f) == PM_RCCL_ID_BYTES;
                    const size_t ne = whole ? fread(extra, 1, sizeof extra, f) : 0;
                    const bool fresh = !nonce.empty() ? std::string(extra, ne) == nonce : (fstat(fileno(f), &st) == 0 && st.st_mtime + 60 >= started);
                    got = whole && fresh;
                    fclose(f);
                }
                if (!got) { struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); }
            }
            if (!got) { std::cerr << "parsnp_core: no RCCL id of this launch in " << idf << " (PARSNP_SHARD_NONCE must be the same for all ranks)" << std::endl; exit(1); }
        }
    }
    const bool writer = run.shard.rank == 0;
    int rc = run.open(argv[1]);
    if (rc) exit(rc);
    if (run.prm.calc_mumi) {          // main() :3188-3209: distances only, no alignment
        std::ofstream touch((run.prm.outdir + "/parsnpAligner.log").c_str());
        std::cerr << "Calculating mumi distances.." << std::endl;
        exit(run.mumi());
    }
    std::ofstream mfile((writer ? run.prm.outdir + "/parsnpAligner.log" : std::string("/dev/null")).c_str());
    StepReport rep = run.step();
    if (!writer) exit(0);          // every rank computed the same alignment; rank 0 writes it
    if (!rep.mums_found) {
        mfile << "NO MUMS FOUND" << std::endl;
        mfile.close();
        return 0;
    }
    mfile << "MUMS FOUND" << std::endl;
    mfile.close();

    if (!run.prm.do_align) std::cerr << "Writing output files..." << std::endl;
    time(&start);
    const double t_out0 = now_s();
    bool gap_note = false;
    run.write(&gap_note);
    const double output_s = now_s() - t_out0;
    time(&end);
    printf("        Output files updated, elapsed time: %.0lf seconds\n\n", difftime(end, start));
    time_t tend;
    time(&tend);
    std::cerr << "Parsnp: Finished core genome alignment" << std::endl;
    printf("        See log file for further details. Total processing time: %.0lf seconds \n\n", difftime(tend, tstart));

    if (const char* tf = getenv("PARSNP_TIMING")) {
        FILE* f = fopen(tf, "w");
        if (f) {
            double outside_writes = 0;      // accepted reverse-strand members outside their region that stayed on the resident route
            for (const auto& kv : rep.host.engine_ms) if (kv.first == "outside_writes") outside_writes = kv.second;
            std::string why = rep.resident_why;      // (plain words; quotes and backslashes would break the JSON)
            for (char& ch : why) if (ch == '"' || ch == '\\') ch = ' ';
            fprintf(f,
                    "{\"provider\": \"%s\", \"genomes\": %zu, \"queries\": %d, \"ingest_s\": %.6f, \"upload_s\": %.6f, \"path_s\": %.6f, "
                    "\"anchor_s\": %.6f, \"extend_s\": %.6f, \"filter_s\": %.6f, \"lcb_s\": %.6f, \"output_s\": %.6f, \"total_s\": %.6f, "
                    "\"finder_s\": %.6f, \"finder_calls\": %ld, \"finder_regions\": %ld, \"regions_processed\": %ld, \"cache_hits\": %ld, "
                    "\"cache_misses\": %ld, \"spec_rounds\": %ld, \"anchors\": %ld, \"mums\": %ld, \"lcbs\": %ld, \"core_bp\": %ld, "
                    "\"gap_note\": %s, \"tie_fallbacks\": %ld, \"literal_iterations\": %ld, \"parallel_candidates\": %ld, \"parallel_dirty\": %ld, \"t_validate\": %.6f, \"t_neighbour\": %.6f, \"t_sweep\": %.6f, \"t_replay\": %.6f, \"t_key\": %.6f, \"resident\": %ld, \"resident_retry\": %ld, \"device_chain\": %ld, \"h2d_bytes\": %.0f, \"d2h_bytes\": %.0f, \"outside_writes\": %.0f, \"resident_why\": \"%s\"}\n",
                    pm_provider(), run.genomes.size(), run.qfiles, run.ingest_s, run.upload_s, rep.path_s, rep.anchor_s, rep.extend_s, rep.filter_s,
                    rep.lcb_s, output_s, now_s() - t_begin, rep.finder_s, rep.finder_calls, rep.finder_regions, rep.regions_processed,
                    rep.cache_hits, rep.cache_misses, rep.spec_rounds, rep.anchors, rep.mums, rep.lcbs, rep.core_bp, gap_note ? "true" : "false",
                    rep.host.tie_fallbacks, rep.host.literal_iterations, rep.host.parallel_candidates, rep.host.parallel_dirty, rep.host.t_validate, rep.host.t_neighbour, rep.host.t_sweep, rep.host.t_replay, rep.host.t_key,
                    rep.host.resident, rep.host.resident_retry, rep.host.device_chain, rep.h2d_bytes, rep.d2h_bytes, outside_writes, why.c_str());
            fclose(f);
        }
    }
    exit(0);
}
