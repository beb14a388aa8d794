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

#include "aligner.h"
#include "ini.h"
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
    IniFile ini;
    ini.read(argv[1]);
    Params prm;
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
    const int qfiles = (int)ini.count("Query") / 2;

    if (prm.calc_mumi || !prm.anchorfile.empty() || !prm.mumfile.empty() || prm.unaligned) {
        // calcmumi (setMumi), anchorfile/mumfile replay and parsnp.unalign are outside the accelerated path
        std::cerr << "parsnp_core (MI355X build): calcmumi / anchorfile / mumfile / unaligned are not supported by this build" << std::endl;
        exit(1);
    }

    time(&start);
    const double t_ingest0 = now_s();
    std::vector<Genome> genomes((size_t)qfiles + 1);
    for (int i = 0; i <= qfiles; i++) {
        std::string path;
        bool rev;
        if (i == 0) { path = ini.get("Reference", "file"); rev = reverse_ref; }
        else {
            char buf[64];
            snprintf(buf, sizeof buf, "file%d", i);
            path = ini.get("Query", buf);
            snprintf(buf, sizeof buf, "reverse%d", i);
            rev = ini.get_bool("Query", buf);
        }
        if (!ingest(path, i == 0, rev, prm.d, &genomes[(size_t)i])) exit(1);
    }
    const double ingest_s = now_s() - t_ingest0;

    std::ofstream mfile((prm.outdir + "/parsnpAligner.log").c_str());
    std::cerr << "\n*****************************************************" << std::endl;
    std::cerr << "\nparsnpAligner:: rapid whole genome SNP typing" << std::endl;
    std::cerr << "\n*****************************************************\n" << std::endl;
    time(&end);
    std::cerr << "ParSNP: Preparing to construct global multiple alignment framework" << std::endl;
    std::cerr << "\nPreparing to verify and process input sequences..." << std::endl;
    printf("        Finished processing input sequences, elapsed time: %.0lf seconds\n\n", difftime(end, start));

    // genomes -> HBM (2-bit + N mask, both strands); the engine addresses regions by coordinates from here on
    const double t_up0 = now_s();
    pm_session* session = nullptr;
    {
        std::vector<const uint8_t*> ptr(genomes.size());
        std::vector<int64_t> len(genomes.size());
        for (size_t i = 0; i < genomes.size(); i++) { ptr[i] = (const uint8_t*)genomes[i].seq.data(); len[i] = (int64_t)genomes[i].seq.size(); }
        int rc = pm_session_create(&session, -1, (int)genomes.size(), ptr.data(), len.data());
        if (rc != PM_OK) {
            std::cerr << "parsnp_core: cannot start the multi-MUM engine (" << pm_provider() << "): " << pm_last_error() << std::endl;
            exit(3);
        }
    }
    const double upload_s = now_s() - t_up0;

    Aligner align(genomes, prm, session);
    time(&start);
    std::cerr << "Searching for initial MUM anchors..." << std::endl;
    const double t_path0 = now_s();
    bool mumsfound = align.find_anchors();
    time(&end);
    align.anchor_time = (float)difftime(end, start);
    time(&start);
    if (!prm.anchors_only) {
        std::cerr << "Performing recursive MUM search between MUM anchors..." << std::endl;
        mumsfound = align.extend();
    }
    time(&end);
    if (!mumsfound) {
        mfile << "NO MUMS FOUND" << std::endl;
        mfile.close();
        return 0;
    }
    mfile << "MUMS FOUND" << std::endl;
    mfile.close();
    printf("        Finished recursive MUM search, elapsed time: %.0lf seconds\n\n", difftime(end, start));
    align.coarsen_time = (float)difftime(end, start);

    if (prm.random) {
        std::cerr << "Filtering spurious matches..." << std::endl;
        time(&start);
        align.random = prm.random;
        align.filter_mums(prm.random);
        time(&end);
        printf("        Finished filtering spurious matches, elapsed time: %.0lf seconds\n\n", difftime(end, start));
        align.random_time = (float)difftime(end, start);
    }
    time(&start);
    std::cerr << "Creating and verifying final LCBs..." << std::endl;
    align.chain();
    align.filter_lcbs();
    align.chain();
    align.fill_between();
    time(&end);
    align.iclusters_time = (float)difftime(end, start);
    printf("        LCBs created, elapsed time: %.0lf seconds\n\n", difftime(end, start));
    const double path_s = now_s() - t_path0;

    if (!prm.do_align) std::cerr << "Writing output files..." << std::endl;
    time(&start);
    const double t_out0 = now_s();
    bool gap_note = false;
    write_output(align, "parsnpAligner", &gap_note);
    const double output_s = now_s() - t_out0;
    time(&end);
    printf("        Output files updated, elapsed time: %.0lf seconds\n\n", difftime(end, start));
    time_t tend;
    time(&tend);
    std::cerr << "Parsnp: Finished core genome alignment" << std::endl;
    printf("        See log file for further details. Total processing time: %.0lf seconds \n\n", difftime(tend, tstart));

    if (const char* tf = getenv("PARSNP_TIMING")) {
        const Stats& s = align.stats;
        long core_bp = 0;
        for (const Lcb& c : align.lcbs)
            if (c.type == 1 && !c.mums.empty()) core_bp += c.end[0] - c.start[0];
        FILE* f = fopen(tf, "w");
        if (f) {
            fprintf(f,
                    "{\"provider\": \"%s\", \"genomes\": %zu, \"queries\": %d, \"ingest_s\": %.6f, \"upload_s\": %.6f, \"path_s\": %.6f, "
                    "\"anchor_s\": %.6f, \"extend_s\": %.6f, \"filter_s\": %.6f, \"lcb_s\": %.6f, \"output_s\": %.6f, \"total_s\": %.6f, "
                    "\"finder_s\": %.6f, \"finder_calls\": %ld, \"finder_regions\": %ld, \"regions_processed\": %ld, \"cache_hits\": %ld, "
                    "\"cache_misses\": %ld, \"spec_rounds\": %ld, \"anchors\": %ld, \"mums\": %zu, \"lcbs\": %zu, \"core_bp\": %ld, "
                    "\"gap_note\": %s}\n",
                    pm_provider(), genomes.size(), qfiles, ingest_s, upload_s, path_s, s.anchor_s, s.extend_s, s.filter_s, s.lcb_s, output_s,
                    now_s() - t_begin, s.finder_s, s.finder_calls, s.finder_regions, s.regions_processed, s.cache_hits, s.cache_misses,
                    s.spec_rounds, align.m0, align.mums.size(), align.lcbs.size(), core_bp, gap_note ? "true" : "false");
            fclose(f);
        }
    }
    pm_session_destroy(session);
    exit(0);
}
