#include "core.h"
#include "hooks.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <future>
#include <iostream>

#include "ini.h"

namespace parsnp {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

CoreRun::~CoreRun() {
    align.reset();
    if (session) pm_session_destroy(session);
}

int CoreRun::open(const std::string& ini_path) {
    IniFile ini;
    ini.read(ini_path);
    prm.c = ini.get_int("LCB", "c");
    prm.d = ini.get_int("LCB", "d");
    prm.diag_diff = (float)ini.get_double("LCB", "diagdiff");
    if (prm.diag_diff < 0.0 || prm.diag_diff > 10000000) prm.diag_diff = 1.0;
    prm.q = ini.get_int("LCB", "q");
    prm.p = ini.get_int("LCB", "p");
    prm.do_align = ini.get_int("LCB", "doalign");
    prm.unaligned = ini.get_bool("LCB", "unaligned");
    std::cout << prm.unaligned << std::endl;
    prm.cores = ini.get_int("LCB", "cores");
    prm.recomb_filter = ini.get_bool("LCB", "recombfilter");
    prm.anchors = ini.get("MUM", "anchors");
    prm.anchorfile = ini.get("MUM", "anchorfile");
    prm.anchors_only = ini.get_bool("MUM", "anchorsonly");
    prm.calc_mumi = ini.get_bool("MUM", "calcmumi");
    prm.extend_mums = ini.get_bool("MUM", "extendmums");
    prm.mums = ini.get("MUM", "mums");
    prm.mumfile = ini.get("MUM", "mumfile");
    prm.random = ini.get_int("MUM", "filter");
    prm.factor = (float)ini.get_double("MUM", "factor");
    prm.prefix = ini.get("Output", "prefix", "parsnp");
    prm.outdir = ini.get("Output", "outdir", "output");
    const bool reverse_ref = ini.get_bool("Reference", "reverse");
    qfiles = (int)ini.count("Query") / 2;

    if (!prm.anchorfile.empty() || !prm.mumfile.empty()) {
        // replay of a stored anchor / MUM list is outside the accelerated path; the driver never sets these keys
        std::cerr << "parsnp_core (MI355X build): anchorfile / mumfile are not supported by this build" << std::endl;
        return 1;
    }

    time_t start, end;
    time(&start);
    const double t0 = now_s();
    // the GPU runtime of a fresh process takes ~0.15 s to start: it does so beside the FASTA parsing
    std::future<int> gpu_ready = std::async(std::launch::async, [] { return pm_warmup(-1); });
    genomes.assign((size_t)qfiles + 1, Genome());
    {   // files are independent: parsed in parallel, reported in file order (the reference reads them one by one)
        std::vector<std::string> paths((size_t)qfiles + 1), console((size_t)qfiles + 1);
        std::vector<char> rev((size_t)qfiles + 1), good((size_t)qfiles + 1, 0);
        for (int i = 0; i <= qfiles; i++) {
            if (i == 0) { paths[0] = ini.get("Reference", "file"); rev[0] = reverse_ref; }
            else {
                char buf[64];
                snprintf(buf, sizeof buf, "file%d", i);
                paths[(size_t)i] = ini.get("Query", buf);
                snprintf(buf, sizeof buf, "reverse%d", i);
                rev[(size_t)i] = ini.get_bool("Query", buf);
            }
        }
#pragma omp parallel for schedule(dynamic) num_threads(prm.cores > 0 ? prm.cores : 1)
        for (int i = 0; i <= qfiles; i++)
            good[(size_t)i] = ingest(paths[(size_t)i], i == 0, rev[(size_t)i] != 0, prm.d, &genomes[(size_t)i], &console[(size_t)i]);
        for (int i = 0; i <= qfiles; i++) {
            std::cout << console[(size_t)i];
            if (!good[(size_t)i]) return 1;
        }
        std::cout.flush();
    }
    ingest_s = now_s() - t0;

    std::cerr << "\n*****************************************************" << std::endl;
    std::cerr << "\nparsnpAligner:: rapid whole genome SNP typing" << std::endl;
    std::cerr << "\n*****************************************************\n" << std::endl;
    time(&end);
    std::cerr << "ParSNP: Preparing to construct global multiple alignment framework" << std::endl;
    std::cerr << "\nPreparing to verify and process input sequences..." << std::endl;
    printf("        Finished processing input sequences, elapsed time: %.0lf seconds\n\n", difftime(end, start));

    // genomes -> HBM (2-bit + N mask, both strands); the engine addresses regions by coordinates from here on
    const double t1 = now_s();
    (void)gpu_ready.get();      // a failure shows up again, with its message, in the session call below
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[upload] waited %.4f s for the GPU runtime (started beside the ingest, %.4f s ago)\n", now_s() - t1, now_s() - t0);
    std::vector<const uint8_t*> ptr(genomes.size());
    std::vector<int64_t> len(genomes.size());
    for (size_t i = 0; i < genomes.size(); i++) { ptr[i] = (const uint8_t*)genomes[i].seq.data(); len[i] = (int64_t)genomes[i].seq.size(); }
    int rc = shard.rccl
                 ? pm_session_create_rccl(&session, -1, (int)genomes.size(), ptr.data(), len.data(), shard.rank, shard.world, shard.rccl_id)
             : shard.world > 1
                 ? pm_session_create_sharded(&session, -1, (int)genomes.size(), ptr.data(), len.data(), shard.rank, shard.world,
                                             shard.allreduce_min, shard.allgather, shard.ctx)
                 : pm_session_create(&session, -1, (int)genomes.size(), ptr.data(), len.data());
    if (rc != PM_OK) {
        std::cerr << "parsnp_core: cannot start the multi-MUM engine (" << pm_provider() << "): " << pm_last_error() << std::endl;
        return 3;
    }
    // (test builds: the engine's thresholds of the long-list routes, lowered so that small inputs take them)
    if (const char* v = test_hook("PM_DIRTY_MIN")) (void)pm_session_tune(session, "dirty_min", atol(v));
    if (const char* v = test_hook("PM_WORK_BUDGET")) (void)pm_session_tune(session, "work_budget", atol(v));
    if (const char* v = test_hook("PM_FLAGGED_DIV")) (void)pm_session_tune(session, "flagged_div", atol(v));
    if (const char* v = test_hook("PM_TANGLED_MAX")) (void)pm_session_tune(session, "tangled_max", atol(v));
    if (const char* v = test_hook("PM_ATOMIC_MARKS")) (void)pm_session_tune(session, "atomic_marks", atol(v));
    if (const char* v = test_hook("PM_GROUP_SMALL")) (void)pm_session_tune(session, "group_small", atol(v));
    if (const char* v = test_hook("PM_MASTER_SEG")) (void)pm_session_tune(session, "master_seg", atol(v));
    if (const char* v = test_hook("PM_STAGE_GATE")) (void)pm_session_tune(session, "stage_gate", atol(v));
    if (const char* v = test_hook("PM_CLUSTER_UNSURE")) (void)pm_session_tune(session, "cluster_unsure", atol(v));
    if (const char* v = test_hook("PM_CHAIN_TIE")) (void)pm_session_tune(session, "chain_tie", atol(v));
    if (const char* v = test_hook("PM_FAST_TAIL")) (void)pm_session_tune(session, "fast_tail", atol(v));
    if (const char* v = test_hook("PM_TANGLE_ROUNDS")) (void)pm_session_tune(session, "tangle_rounds", atol(v));
    upload_s = now_s() - t1;
    return 0;
}

int CoreRun::mumi() {
    const size_t n = genomes.size();
    // setMumi only ever looks at the first p bases of the reference (its chunk loop stops after one pass, :1909)
    long len0 = (long)genomes[0].seq.size();
    long p = prm.p > len0 ? len0 : prm.p;
    std::cerr << std::endl << "        Constructing device index of the reference...\n";
    std::cerr << "        Calculating pairwise MUMi distances...\n";
    std::cout << prm.outdir << std::endl;
    std::string path = prm.outdir + "/all.mumi";
    std::cout << path << std::endl;
    FILE* f = fopen(shard.rank == 0 ? path.c_str() : "/dev/null", "w");   // sharded run: every rank computes, rank 0 writes
    if (!f) { std::cerr << "parsnp_core: cannot write " << path << std::endl; return 1; }
    std::vector<int64_t> starts(n, 0), lens(n), covered(n > 1 ? n - 1 : 1, 0);
    lens[0] = p;
    for (size_t g = 1; g < n; g++) lens[g] = (int64_t)genomes[g].seq.size();
    if (n > 1 && p > 0) {
        int rc = pm_mumi_coverage(session, starts.data(), lens.data(), covered.data());
        if (rc != PM_OK) { std::cerr << "parsnp_core: MUMi engine failed: " << pm_last_error() << std::endl; fclose(f); return 1; }
    }
    for (size_t g = 1; g < n; g++) {
        int total = (int)covered[g - 1];
        int minlen = (int)p;
        float ratio = float(len0) / float(lens[g]);             // r1.length.at(0) / rs[g].len_region (:2074)
        if (ratio > 1.3 || ratio < 0.7) total = 0;
        if (total > minlen) total = minlen;
        fprintf(f, "%d:%f\n", (int)g, 1.0 - (float(total) / float(minlen)));   // (:2080)
    }
    fclose(f);
    return 0;
}

StepReport CoreRun::step() {
    uint64_t h0 = 0, d0 = 0, h1 = 0, d1 = 0;
    (void)pm_session_traffic(session, &h0, &d0);
    said_ = 0;
    // an input that left the resident route once leaves it at the same point every time (the step is a function of the genomes and
    // the parameters): later steps of the session go to the host route at once instead of paying for the attempt again
    StepReport r = step_once(left_why_.empty());
    std::string why = left_why_.empty() ? align->resident_why() : "left in an earlier step of this session (" + left_why_ + ")";
    if (left_why_.empty() && align->resident_failed()) {
        // the resident route was left: the reference's processing order would have shown in the result (resident.cpp).  The
        // step runs again on the host route, from the anchor call.
        if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[resident] route left (%s): the step runs again on the host route\n", align->resident_why().c_str());
        left_why_ = why;
        const double lost = r.path_s;
        r = step_once(false);
        r.path_s += lost; r.host.resident_retry = 1;
    }
    (void)pm_session_traffic(session, &h1, &d1);
    r.h2d_bytes = (double)(h1 - h0); r.d2h_bytes = (double)(d1 - d0);
    r.resident_why = why;
    if (const char* log = test_hook("PARSNP_RESIDENT_LOG"))      // test hook: which route every step took
        if (FILE* f = fopen(log, "a")) {
            double exact = 0, outside = 0;
            for (const auto& kv : r.host.engine_ms) { if (kv.first == "exact_cluster_tests") exact = kv.second; if (kv.first == "outside_writes") outside = kv.second; }
            fprintf(f, "resident=%ld retry=%ld chain=%ld exact=%ld generations=%ld deferred=%ld ties=%ld outside=%ld anchors=%ld mums=%ld why=%s\n", r.host.resident, r.host.resident_retry, r.host.device_chain, (long)exact, r.host.generations, r.host.regions_deferred, r.host.tie_runs, (long)outside, r.anchors, r.mums, why.c_str()); fclose(f);
        }
    return r;
}

// the progress lines of a step (stderr / stdout, as the reference prints them): each ONCE per step, also when the step is run again
// on the host route after the resident route was left
bool CoreRun::say(int stage) {
    if (stage <= said_) return false;
    said_ = stage;
    return true;
}

StepReport CoreRun::step_once(bool resident) {
    StepReport r;
    const double ts = now_s();
    align.reset();
    const double tm = now_s();
    memory.reset();
    align.reset(new Aligner(genomes, prm, session, &memory));
    align->sharded_ = shard.world > 1 || shard.rccl;
    align->resident_allowed_ = resident;
    r.setup_s = now_s() - ts;
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[setup] release of the previous run %.4f s, new state %.4f s\n", tm - ts, now_s() - tm);
    Aligner& a = *align;
    time_t start, end;
    time(&start);
    a.announce_ = say(1);
    if (a.announce_) std::cerr << "Searching for initial MUM anchors..." << std::endl;
    const double t0 = now_s();
    bool found = a.find_anchors();
    time(&end);
    a.anchor_time = (float)difftime(end, start);
    time(&start);
    if (!prm.anchors_only) {
        if (say(2)) std::cerr << "Performing recursive MUM search between MUM anchors..." << std::endl;
        found = a.extend();
    }
    time(&end);
    r.mums_found = found;
    if (const char* dump = test_hook("PARSNP_DUMP_MUMS")) {      // test hook: the MUM list after the recursion, (reference start, length) in list order
        if (FILE* f = fopen(dump, "w")) {
            for (int idx : a.mums) {
                fprintf(f, "%ld %ld", a.key0(idx), a.pool[(size_t)idx].length);
                if (a.pool[(size_t)idx].start) for (size_t j = 0; j < a.n; j++) fprintf(f, " %d%c", a.pool[(size_t)idx].start[j], a.pool[(size_t)idx].fwd[j] ? '+' : '-');
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    if (found && !a.resident_failed()) {
        // (resident route, order-free list logic: phases C-D were queued on the device behind the last generation)
        const bool from_device = a.resident_chain();
        if (a.resident_failed()) { r.path_s = now_s() - t0; return r; }      // (the order check queued ahead of phases C-D)
        if (say(3)) printf("        Finished recursive MUM search, elapsed time: %.0lf seconds\n\n", difftime(end, start));
        a.coarsen_time = (float)difftime(end, start);
        if (prm.random) {
            if (say(4)) std::cerr << "Filtering spurious matches..." << std::endl;
            time(&start);
            a.random = prm.random;
            if (!from_device) a.filter_mums(prm.random);
            time(&end);
            if (say(5)) printf("        Finished filtering spurious matches, elapsed time: %.0lf seconds\n\n", difftime(end, start));
            a.random_time = (float)difftime(end, start);
            if (a.resident_failed()) { r.path_s = now_s() - t0; return r; }
        }
        time(&start);
        if (say(6)) std::cerr << "Creating and verifying final LCBs..." << std::endl;
        const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
        double tl = now_s();
        auto lap = [&](const char* what) { if (dbg) { double t = now_s(); fprintf(stderr, "[lcb] %-12s %.4f s (cpu %.4f)\n", what, t - tl, cpu_lap_s()); tl = t; } };
        if (!from_device) {
        a.chain(); lap("chain");
        const long dissolved = a.filtered_lcbs;
        a.filter_lcbs(); lap("filter_lcbs");
        // the reference chains again (:3261-3268).  With no LCB dissolved and no two MUMs sharing a reference start the
        // MUM list and its sorted order are unchanged, and the second pass would rebuild exactly the list at hand
        // (chain order = order of the first MUMs on the reference = the order sort_lcbs left).
        if (a.resident_failed()) { r.path_s = now_s() - t0; return r; }
        bool same = a.filtered_lcbs == dissolved && a.unique_order && !test_hook("PARSNP_CHAIN_TWICE");
        for (size_t i = 1; i < a.lcbs.size() && same; i++) same = a.lcbs[i - 1].start[0] < a.lcbs[i].start[0];
        if (!same) { a.chain(); lap("chain"); }
        a.fill_between(); lap("fill_between");
        }
        time(&end);
        a.iclusters_time = (float)difftime(end, start);
        if (say(7)) printf("        LCBs created, elapsed time: %.0lf seconds\n\n", difftime(end, start));
    }
    r.path_s = now_s() - t0;
    if (getenv("PARSNP_DEBUG_TIMERS")) fprintf(stderr, "[step] anchors + recursion + lists %.6f s (anchor %.6f, extend %.6f, lcb %.6f; engine calls %.6f)\n", r.path_s, a.stats.anchor_s, a.stats.extend_s, a.stats.lcb_s, a.stats.finder_s);
    const Stats& s = a.stats;
    r.anchor_s = s.anchor_s; r.extend_s = s.extend_s; r.filter_s = s.filter_s; r.lcb_s = s.lcb_s; r.finder_s = s.finder_s;
    r.alg_bytes = s.alg_bytes; r.alg_bytes_kernel = s.alg_bytes_kernel; r.alg_bytes_query = s.alg_bytes_query; r.finder_calls = s.finder_calls; r.finder_regions = s.finder_regions; r.regions_processed = s.regions_processed;
    r.cache_hits = s.cache_hits; r.cache_misses = s.cache_misses; r.spec_rounds = s.spec_rounds;
    r.engine_ms = s.engine_ms; r.anchor_ms = s.anchor_ms; r.host = s;
    r.anchors = a.m0; r.mums = (long)a.mums.size(); r.lcbs = (long)a.lcbs.size();
    for (const Lcb& c : a.lcbs)
        if (c.type == 1 && !c.mums.empty()) r.core_bp += c.end[0] - c.start[0];
    return r;
}

void CoreRun::write(bool* gap_note) {
    align->materialize();      // (resident route: the rows of the LCBs' MUMs come to the host now)
    write_output(*align, "parsnpAligner", gap_note);
    if (prm.unaligned) write_unaligned(*align);   // src/parsnp.cpp:3283-3287
}

}  // namespace parsnp
