// engine_core.h -- orchestration of one pm_multi_mum_batch call, templated on the execution backend.
// HipBackend (engine_hip.hip) = the product: HIP kernels on gfx950, one stream, HIP-event timing.
// tests/emu/ instantiates it with a sequential host backend to check the logic without a GPU (tests only).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "store_kernels.h"

namespace pm {

// Host buffers the results are downloaded into (ordinary memory).  They are recycled: a result gives its
// blocks back when it is freed, the next batch of the same shape takes them, so the steady state allocates nothing and
// touches no fresh pages.  Shared by the session and every result it handed out (a result may outlive the session).
struct HostPool {
    void* (*alloc_fn)(size_t) = nullptr;
    void (*free_fn)(void*) = nullptr;
    struct Block { void* p = nullptr; size_t cap = 0; };
    std::vector<Block> idle;
    std::mutex mu;      // results are freed by whichever thread holds them while another thread's call takes blocks
    Block take(size_t n) {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = idle.size();
        for (size_t i = 0; i < idle.size(); i++)
            if (idle[i].cap >= n && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
        if (best != idle.size() && idle[best].cap <= 2 * n + (1u << 20)) { Block b = idle[best]; idle.erase(idle.begin() + (long)best); return b; }
        Block b; b.cap = n + n / 8 + 256; b.p = alloc_fn(b.cap);
        if (!b.p) throw std::bad_alloc();
        return b;
    }
    void give(Block b) { std::lock_guard<std::mutex> lk(mu); if (b.p) idle.push_back(b); }
    ~HostPool() { for (auto& b : idle) free_fn(b.p); }
};

struct ClearJob { void* p; size_t bytes; int value; };      // backend clear_many: several clears in one launch
struct BatchResult {
    int64_t nregions = 0, total = 0;
    int nq = 0;
    std::vector<int64_t> off;
    HostPool::Block kb, lonb, spb, fwdb;     // int32 k[total], int32 lon[total], int32 sp[total*nq], uint8 fwd[total*nq]
    // a session in MUM-row mode (pm_session_rows) fills these instead of sp / fwd:
    HostPool::Block startb, strandb, flagsb; // int32 start[total*(nq+1)], uint8 strand[total*(nq+1)], uint32 flags[total]
    bool rows = false, dirty_known = false;  // dirty_known: the kRowDirty bits were computed (one-region batch with a long list)
    int64_t table_id = 0;                    // != 0: the rows of this result stay on the device as the session's anchor table = rows [0, A) of the MUM store
    int64_t store_base = -1;                 // resident mode: the result's rows are rows [store_base, store_base + total) of the session's MUM store (no start / strand blocks)
    std::shared_ptr<HostPool> pool;
    int32_t* start() const { return (int32_t*)startb.p; }
    uint8_t* strand() const { return (uint8_t*)strandb.p; }
    const uint32_t* flags() const { return (const uint32_t*)flagsb.p; }
    const int32_t* k() const { return (const int32_t*)kb.p; }
    const int32_t* lon() const { return (const int32_t*)lonb.p; }
    const int32_t* sp() const { return (const int32_t*)spb.p; }
    const uint8_t* fwd() const { return (const uint8_t*)fwdb.p; }
    void release() {
        if (pool) { pool->give(kb); pool->give(lonb); pool->give(spb); pool->give(fwdb); pool->give(startb); pool->give(strandb); pool->give(flagsb); }
        kb = lonb = spb = fwdb = startb = strandb = flagsb = HostPool::Block();
    }
    BatchResult() = default;
    BatchResult(const BatchResult&) = delete;
    BatchResult& operator=(const BatchResult&) = delete;
    ~BatchResult() { release(); }
};

struct PhaseTime { const char* name; float ms; };

// Sharded run: the query genomes are split into contiguous blocks, one per rank/GPU; every rank holds the
// reference.  The two exchange steps of a batch go through these callbacks (host buffers; the embedding process
// implements them with torch.distributed -- RCCL on GPUs, gloo in tests).  Return 0 on success.
struct Collectives {
    int rank = 0, world = 1;
    int (*allreduce_min_i32)(void* ctx, int32_t* buf, int64_t count) = nullptr;
    int (*allgather)(void* ctx, const void* send, int64_t send_bytes, void* recv) = nullptr;   // recv: world * send_bytes
    void* ctx = nullptr;
    // device collectives (pm_session_create_rccl): the backend owns an RCCL communicator and the exchanges run on device
    // buffers on the engine's stream; the callbacks above are not used then
    bool device = false;
};

inline int bits_for(uint64_t v) { int b = 1; while (b < 64 && (v >> b)) b++; return b; }

template <class B>
class Engine {
public:
    explicit Engine(B& backend) : be(backend), pool(std::make_shared<HostPool>()) { pool->alloc_fn = &B::host_alloc; pool->free_fn = &B::host_free; }
    std::shared_ptr<HostPool> pool;
    ~Engine() { release(); }

    std::string error;
    std::vector<PhaseTime> timing;
    int64_t last_events = 0, last_candidates = 0, last_rest = 0, last_positions = 0, last_accepted = 0;
    double last_alg[3] = {0, 0, 0};      // last search of store regions: SURVEY 8d bytes, this engine's bytes, query-stream bytes (AlgBytes)
    uint64_t alg_raw[3] = {0, 0, 0};
    uint64_t alg_sets[8 * kAlgSets] = {};

    int ngen = 0;
    bool want_rows = false;           // pm_session_rows: results as MUM rows (start, strand, flags) instead of (sp, fwd)
    std::vector<int64_t> glen_h;
    Collectives coll;                 // world == 1: not sharded
    int g_first = 1, g_last = 1;      // query genomes [g_first, g_last) are resident on this GPU
    void set_shard(const Collectives& c) { coll = c; }
    // contiguous split of the n-1 query genomes over the ranks
    static void shard_range(int n, int rank, int world, int* first, int* last) {
        int q = n - 1;
        *first = 1 + (int)((int64_t)q * rank / world);
        *last = 1 + (int)((int64_t)q * (rank + 1) / world);
    }

    // ---- genomes -> packed strands in device memory
    int load_genomes(int n, const uint8_t* const* seqs, const int64_t* lens) {
        const bool dbg = getenv("PARSNP_DEBUG_TIMERS") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tl = now();
        auto lap = [&](const char* what) { if (dbg) { be.sync(); const double t = now(); fprintf(stderr, "[upload] %-14s %.4f s\n", what, t - tl); tl = t; } };
        ngen = n;
        glen_h.assign(lens, lens + n);
        shard_range(n, coll.rank, coll.world, &g_first, &g_last);
        // every rank of a sharded run keeps ALL genomes resident (one byte per base and strand: 57 000 5-Mb genomes fit one GPU):
        // the SEARCH is what is sharded -- a rank streams the query genomes of its block only (CountUnits, SmallPairEvents,
        // MasterEP) -- while the resident route's validation checks reverse-strand members of any genome against the sequence
        auto resident = [&](int) { return true; };
        std::vector<int64_t> goff(2 * (size_t)n);
        int64_t words = 2;                       // leading guard (64 bases)
        int64_t maxlen = 0;
        for (int g = 0; g < n; g++) {
            if (!resident(g)) { goff[2 * (size_t)g] = goff[2 * (size_t)g + 1] = 64; continue; }   // never read
            maxlen = std::max(maxlen, lens[g]);
            for (int s = 0; s < 2; s++) {
                goff[2 * (size_t)g + s] = words * 32;
                words += (lens[g] + 31) / 32 + 2;   // + trailing guard
            }
        }
        words += 8;                              // SmallPairEvents reads a 128-base window + 2 blocks from any position of the last strand
        total_words = words;
        blk = (SeqBlock*)be.alloc((size_t)words * sizeof(SeqBlock));
        d_goff = (int64_t*)be.alloc(sizeof(int64_t) * 2 * (size_t)n);
        d_glen = (int64_t*)be.alloc(sizeof(int64_t) * (size_t)n);
        if (!blk || !d_goff || !d_glen) { error = "device allocation failed (genomes)"; return -3; }
        lap("allocate");
        be.memset(blk, 0, (size_t)words * sizeof(SeqBlock));
        be.h2d(d_goff, goff.data(), sizeof(int64_t) * goff.size());
        be.h2d(d_glen, lens, sizeof(int64_t) * (size_t)n);
        // ASCII genomes -> packed strands: a few host threads copy the genomes into page-locked staging slots, each slot's
        // DMA and its two PackStrand launches run on the slot's own stream while the next genome is being staged
        // (the genomes arrive in ordinary memory, from which a direct copy is staged by the runtime at ~6 GB/s)
        std::vector<char> res((size_t)n, 0);
        for (int g = 0; g < n; g++) res[(size_t)g] = resident(g) && lens[g] > 0;
        lap("clear + tables");
        if (!be.stage_genomes(n, seqs, lens, res, goff, blk, maxlen)) { error = "genome upload failed: " + be.error(); return -3; }
        lap("stage + pack");
        P = Packed{blk, d_goff, d_glen};
        return 0;
    }

    // ---- one batch of regions (include/parsnp_mum.h: pm_multi_mum_batch)
    // Host round trips of one call: the event counters (+ error word), the candidate count, the accepted count and the
    // result download.  Everything else the device needs from the host (region table, prefix arrays, request rows) is
    // assembled in ONE page-locked block and sent by asynchronous copies on the engine's stream.
    std::vector<int64_t> mumi_covered;   // result of run(..., mumi = true): per query genome
    // A batch whose repeat structure exhausts the per-thread work budget (a tandem repeat of period > 1 with thousands of
    // copies inside ONE region: every copy's K-mer chain is walked by every sample that hits it) is run again with a
    // budget 256 times larger -- slow, but the reference aligns such input too; only then PM_ELIMIT.  In a sharded run the
    // verdict is common to all ranks (it travels with the first exchange), so every rank repeats the batch together.
    // Requests whose rows are regions of the session's region store (resident route): the host holds their reference columns only.
    struct GapBatch {
        const int64_t* ref_start = nullptr; const int64_t* ref_len = nullptr;   // [nreg]: the reference column of every region (sizes the index)
        const int32_t* store_ids = nullptr;                                      // [nreg]: the regions
    };
    int run(int64_t nreg, const int64_t* starts, const int64_t* lens, const int32_t* minsize, BatchResult* out, bool want_events = false,
            bool mumi = false, const GapBatch* gb = nullptr) {
        budget_exceeded = false;
        int rc = run_once(nreg, starts, lens, minsize, out, want_events, mumi, gb);
        if (rc == -5 && budget_exceeded && work_budget < ((int64_t)1 << 40)) {
            const int64_t keep = work_budget;
            work_budget = keep << 8;
            budget_retries++;
            rc = run_once(nreg, starts, lens, minsize, out, want_events, mumi, gb);
            work_budget = keep;
        }
        return rc;
    }
    int64_t anchor_table_id = 0, anchor_table_rows = 0;      // the anchor table: rows of the last long one-region call in row mode = rows [0, A) of the MUM store
    bool budget_exceeded = false;
    long budget_retries = 0;
    int run_once(int64_t nreg, const int64_t* starts, const int64_t* lens, const int32_t* minsize, BatchResult* out, bool want_events,
                 bool mumi, const GapBatch* gb = nullptr) {
        timing.clear();
        const int nq = ngen - 1;
        out->nregions = nreg; out->nq = nq; out->total = 0;
        out->off.assign((size_t)nreg + 1, 0);
        out->release();
        out->pool = pool;
        if (nreg == 0) return 0;
        if (nq < 1) { error = "need at least one query genome"; return -2; }
        const int no_small = mumi ? 1 : 0;

        // -- host: the page-locked parameter block  [ RegionInfo x nreg | posbase | cbase | starts rows | lens rows ]
        // (regions of the region store: the rows part holds their ids only)
        const size_t nrow = (size_t)nreg * (size_t)ngen;
        const size_t regz = (size_t)nreg;
        const size_t bytes_R = sizeof(RegionInfo) * regz, bytes_pre = 8 * (regz + 1);
        const bool from_store = gb && gb->store_ids;
        if (gb && !from_store) { error = "bad region list"; return -2; }
        const size_t nrow_staged = gb ? 0 : nrow;
        uint8_t* block = (uint8_t*)be.staging(bytes_R + 2 * bytes_pre + 16 * nrow_staged + (gb ? 4 * regz : 0) + 64);
        if (!block) { error = "cannot allocate the request staging block"; return -3; }
        RegionInfo* R = (RegionInfo*)block;
        int64_t* posbase = (int64_t*)(block + bytes_R);
        int64_t* cbase = posbase + regz + 1;
        int64_t* stage = cbase + regz + 1;
        std::vector<size_t> guess_r(regz, 0);
        std::vector<int64_t> units_r(regz, 0);
        if (from_store) {
            for (int64_t r = 0; r < nreg; r++) {
                if (gb->store_ids[r] < 0 || gb->store_ids[r] >= rg_count) { error = "region outside the region store"; return -2; }
                if (gb->ref_start[r] < 0 || gb->ref_len[r] < 0 || gb->ref_start[r] + gb->ref_len[r] > glen_h[0]) { error = "region outside its genome"; return -2; }
                if (gb->ref_len[r] >= (1ll << 31)) { error = "region longer than 2^31"; return -5; }
            }
            memcpy(stage, gb->store_ids, sizeof(int32_t) * regz);
        } else {
            // the request rows (2 x 8 bytes per region and genome: 26 MB for a recursion batch of 8 000 regions x 201) are
            // checked and copied by a few threads; the same pass counts the SeedExtend work units of every region (what
            // CountUnits computes on the device) so that the grid size needs no read-back
            const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, nreg / 512));
            std::vector<int> bad((size_t)nt, 0);
            auto part = [&](int t) {
                const int64_t r0 = nreg * t / nt, r1 = nreg * (t + 1) / nt;
                for (int64_t r = r0; r < r1; r++) {
                    const int64_t* st = starts + r * ngen; const int64_t* ln = lens + r * ngen;
                    const int minlen = minsize[r] < 1 ? 1 : minsize[r];
                    const int div = std::max(8, minlen);
                    const int K = minlen < kMaxK ? minlen : kMaxK, stride = minlen - K + 1;
                    const int64_t nR = ln[0];
                    size_t gs = 0;
                    int64_t units = 0;
                    int b = 0;
                    for (int g = 0; g < ngen; g++) {
                        if (st[g] < 0 || ln[g] < 0 || st[g] + ln[g] > glen_h[(size_t)g]) b |= 1;
                        if (ln[g] >= (1ll << 31)) b |= 2;
                        if (!g) continue;
                        gs += (size_t)(2 * ln[g] / div);
                        int64_t ns = (ln[g] >= K && nR >= K && g >= g_first && g < g_last) ? (ln[g] - K) / stride + 1 : 0;
                        if (small_pair(nR, ln[g]) && !no_small) ns = 0;
                        units += (ns + kUnitSamples - 1) / kUnitSamples;
                    }
                    guess_r[(size_t)r] = gs; units_r[(size_t)r] = units;
                    bad[(size_t)t] |= b;
                    memcpy(stage + r * ngen, st, sizeof(int64_t) * (size_t)ngen);
                    memcpy(stage + nrow + r * ngen, ln, sizeof(int64_t) * (size_t)ngen);
                }
            };
            if (nt == 1) part(0);
            else {
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++) th.emplace_back(part, t);
                part(0);
                for (auto& x : th) x.join();
            }
            int b = 0;
            for (int x : bad) b |= x;
            if (b & 1) { error = "region outside its genome"; return -2; }
            if (b & 2) { error = "region longer than 2^31"; return -5; }
        }
        int64_t npos = 0, tsize = 0, fwords = 0, nunits = 0;
        int32_t max_nr = 1;
        size_t ev_guess = 1 << 16;     // first-call event buffer: a match of length >= minsize every max(8,minsize) bases is generous
        cbase[0] = 0;
        for (int64_t r = 0; r < nreg; r++) {
            RegionInfo& ri = R[(size_t)r];
            ri.ref_pos = gb ? gb->ref_start[r] : starts[r * ngen];
            ri.nR = (int32_t)(gb ? gb->ref_len[r] : lens[r * ngen]);
            ri.minsize = minsize[r];
            ri.minlen = minsize[r] < 1 ? 1 : minsize[r];
            ri.K = ri.minlen < kMaxK ? ri.minlen : kMaxK;
            ri.stride = ri.minlen - ri.K + 1;
            int64_t slots = 16;
            while (slots < slot_factor * (int64_t)ri.nR) slots <<= 1;   // load factor <= 1 / slot_factor (x2: the next power of two)
            ri.tmask = (uint32_t)(slots - 1);
            ri.tbase = tsize; tsize += slots;
            int64_t fbits = 64;
            while (fbits < filter_factor * (int64_t)ri.nR) fbits <<= 1;
            ri.fmask = (uint32_t)(fbits - 1); ri.fbase = fwords; fwords += fbits / 32; ri.pad_ = 0;
            ri.posbase = npos; posbase[(size_t)r] = npos; npos += ri.nR;
            cbase[(size_t)r + 1] = cbase[(size_t)r] + (((int64_t)ri.nR + kChunkPos - 1) >> kCoarseShift) + 1;
            max_nr = std::max(max_nr, ri.nR);
            ev_guess += guess_r[(size_t)r];
            nunits += units_r[(size_t)r];
        }
        posbase[regz] = npos;
        last_positions = npos; last_candidates = 0; last_accepted = 0; last_grouped = 0;
        const int64_t npairs = nreg * nq;
        const int64_t nchunks = cbase[regz] - nreg;                  // 256-position chunks of the batch
        const int64_t centries = cbase[regz] * nq;
        const int lbits = bits_for((uint64_t)max_nr);
        if (bits_for((uint64_t)npairs) + lbits + 1 > 64) { error = "batch too large for 64-bit event keys"; return -5; }
        if (npairs >= (1ll << 31) || nreg >= (1ll << 31)) { error = "too many regions in one batch"; return -5; }
        if (nunits >= (1ll << 31)) { error = "too many work units in one batch"; return -5; }
        if (centries >= (1ll << 31)) { error = "coarse event index too large"; return -5; }

        be.mark("setup");
        ensure(d_R, regz);
        ensure(d_starts, nrow); ensure(d_lens, nrow);
        ensure(d_posbase, regz + 1); ensure(d_cbase, regz + 1);
        be.h2d_staged(d_R.p, R, bytes_R);
        be.h2d_staged(d_posbase.p, posbase, bytes_pre);
        be.h2d_staged(d_cbase.p, cbase, bytes_pre);
        if (!gb) {
            be.h2d_staged(d_starts.p, stage, sizeof(int64_t) * nrow);
            be.h2d_staged(d_lens.p, stage + nrow, sizeof(int64_t) * nrow);
        } else if (from_store) {
            ensure(d_list, regz);
            be.h2d_staged(d_list.p, stage, sizeof(int32_t) * regz);
            be.launch("gather_regions", (int64_t)nrow, GatherRegions{d_list.p, ngen, d_rg_start.p, d_rg_len.p, d_starts.p, d_lens.p});
        }
        // event counters: kSlices counters one 64-byte line apart, then the error word of the batch (read back together)
        const size_t ncounter = (size_t)kSlices * kSliceStride + 8;
        ensure(d_counter, ncounter);
        uint32_t* d_err = (uint32_t*)(d_counter.p + (size_t)kSlices * kSliceStride);

        // -- reference index + repeat lengths
        ensure(d_slots, (size_t)tsize); ensure(d_filter, (size_t)fwords);
        ensure(d_next, (size_t)std::max<int64_t>(npos, 1)); ensure(d_rep, (size_t)std::max<int64_t>(npos, 1));
        ensure(d_epm, (size_t)std::max<int64_t>(npos, 1) + 1);     // + the verdict word of a sharded run
        // everything the call clears before it starts, in one launch: presence filter, slots (0xff = empty), event counters + error
        // word, the repeated-K-mer bits, the coarse index, the closing words of the two prefix scans, the byte counters
        const int64_t nwv = (npos + 63) / 64;
        const bool from_store_ = gb && gb->store_ids;
        ensure(d_repeated, (size_t)(npos / 32 + 2));       // (SeedExtend reads the word after a position's own)
        ensure(d_ucount, (size_t)npairs + 1); ensure(d_uoff, (size_t)npairs + 1);
        ensure(d_coarse, (size_t)std::max<int64_t>(centries, 1));
        ensure(d_wmask, (size_t)nwv + 1); ensure(d_wcount, (size_t)nwv + 1); ensure(d_woff, (size_t)nwv + 1);
        if (from_store_) ensure(d_alg, 8 * (size_t)kAlgSets);
        {
            const ClearJob jobs[] = {
                {d_filter.p, sizeof(uint32_t) * (size_t)fwords, 0}, {d_slots.p, sizeof(uint64_t) * (size_t)tsize, 0xff}, {d_counter.p, 8 * ncounter, 0},
                {d_repeated.p, 4 * (size_t)(npos / 32 + 2), 0}, {d_coarse.p, 4 * (size_t)std::max<int64_t>(centries, 1), 0},
                {d_ucount.p + npairs, 8, 0}, {d_wcount.p + nwv, 8, 0}, {from_store_ ? d_alg.p : nullptr, from_store_ ? (size_t)64 * kAlgSets : 0, 0}};
            be.clear_many(jobs, (int)(sizeof jobs / sizeof jobs[0]));
        }
        // rows the device derived itself (gaps of the anchor table, rows of the region store) were never seen by the host: the
        // same checks the explicit rows get above, raised through the batch's error word
        if (gb) be.launch("check_rows", (int64_t)nrow, CheckRows{d_R.p, d_starts.p, d_lens.p, ngen, d_glen, d_err});
        last_alg[0] = last_alg[1] = last_alg[2] = 0;
        if (from_store) {      // the algorithmic bytes of a search whose rows the host never saw (read back with the event counters)
            be.launch_wave("alg_bytes", (nreg * nq + kAlgPairs - 1) / kAlgPairs, AlgBytes{d_R.p, d_lens.p, ngen, nreg * nq, d_alg.p});
        }
        be.mark("index");
        be.launch("index_insert", npos, IndexInsert{P, d_R.p, nreg, d_posbase.p, d_slots.p, d_next.p, d_filter.p, d_rep.p});
        be.mark("repeat");
        ensure(d_run, (size_t)std::max<int64_t>(npos, 1));
        be.launch("run_length", npos, RunLength{P, d_R.p, nreg, d_posbase.p, d_run.p});
        be.launch("repeat_length", npos, RepeatLength{P, d_R.p, nreg, d_posbase.p, d_slots.p, d_filter.p, d_next.p, d_run.p, d_rep.p, d_repeated.p, d_err, work_budget});

        // -- work units (pairs that fit 128 bases on both sides go to SmallPairEvents instead)
        be.mark("units");
        be.launch("count_units", npairs, CountUnits{d_R.p, d_lens.p, ngen, d_ucount.p, g_first, g_last, no_small});
        be.exclusive_scan(d_ucount.p, d_uoff.p, (size_t)npairs + 1);
        // the host never saw the rows of a store search: the unit count is the device's.  The call used to wait for it (8 bytes); now it
        // launches over a CAPACITY from the last call of the shape, the kernels read the live count, and the count comes back with the
        // event counters (a capacity that was too small repeats the event search, like a full event buffer)
        // (which call of the step this is: the anchor call is number 0, the searches of store regions follow it -- a step that repeats
        // makes the same calls in the same order, and the capacities of a call come from the same call of the step before)
        if (!gb && nreg == 1) call_ordinal = 0; else if (gb) call_ordinal++;
        const size_t ord = gb || nreg == 1 ? (size_t)call_ordinal : (size_t)-1;      // (-1: a batch of the host route -- it waits for its counts)
        if (ord != (size_t)-1 && call_hints.size() <= ord) call_hints.resize(ord + 1);
        CallHint none;
        CallHint& hint = ord != (size_t)-1 ? call_hints[ord] : none;
        bool units_known = !gb;
        if (gb && !(fast_tail && hint.units > 0)) {
            be.d2h(&nunits, d_uoff.p + npairs, 8);
            units_known = true;
        } else if (gb) nunits = hint.units + hint.units / 8 + 64;
        if (nunits >= (1ll << 31)) { error = "too many work units in one batch"; return -5; }
        if (gb) ev_guess += (size_t)nunits * 16;

        // -- events: kSlices append buffers (retry with larger ones on overflow), gathered, then sorted by (pair, l, strand)
        ensure(d_sliceoff, (size_t)kSlices + 1);
        uint64_t nev = 0;
        size_t slice_cap = (std::min<size_t>(std::max<size_t>(ev_cap_hint, ev_guess), (size_t)1 << 31) + kSlices - 1) / kSlices + 64;
        std::vector<uint64_t> counts(ncounter);
        uint32_t errbits = 0;
        // samples that SeedExtend hands to SeedRest: kSlices sub-queues sized from what earlier calls needed (first guess: one
        // sample in sixteen), repeated with the exact size if one overflows (the counters keep counting past the capacity)
        ensure(d_qcount, (size_t)kSlices * kSliceStride);
        size_t queue_cap = std::max<size_t>(std::max<size_t>(rest_cap_hint[nreg == 1], (size_t)nunits * kUnitSamples / 16 / kSlices), 64) + 64;      // (anchor-shaped and recursion-shaped calls alternate)
        uint64_t nrest = 0;
        uint32_t sticky = 0;
        std::vector<uint64_t> qcounts((size_t)kSlices * kSliceStride);
        // the small regions of a recursion batch: their events once per distinct piece, straight into the head of the final array
        const bool grouping = group_small && !no_small && !want_events && nreg >= 2 && nq <= kGrpGenomes;
        const size_t kGrpSlot = (size_t)kSlices * kSliceStride + 2;      // (d_counter: the block counter of the grouped events)
        size_t grp_cap = grouping ? std::max<size_t>(grp_cap_hint, (size_t)npairs * 3) + 64 : 0;
        uint64_t ngrp = 0;
        if (grouping) { ensure(d_gflag, (size_t)nreg); ensure(d_glo, (size_t)npairs); }
        int64_t nunits_live = nunits;
        uint64_t worst_slice = 0;      // events of the fullest sub-buffer
        for (bool again = false;; again = true) {
            ensure(d_units, (size_t)std::max<int64_t>(nunits, 1));
            be.launch("fill_units", nunits, FillUnits{P, d_starts.p, d_lens.p, ngen, d_uoff.p, d_ucount.p, npairs, d_units.p});
            ensure(d_evkey, slice_cap * kSlices); ensure(d_evval, slice_cap * kSlices); ensure(d_rest, queue_cap * kSlices);
            if (again) be.memset(d_counter.p, 0, 8 * ncounter);      // event counters, error word, grouped events
            if (grouping) {
                ensure(d_evkey3, grp_cap + ev_cap_hint); ensure(d_evval3, grp_cap + ev_cap_hint);
                ensure(d_state, grp_cap + ev_cap_hint); ensure(d_emax, grp_cap + ev_cap_hint);
                be.mark("grouped_events");
                be.launch_wave("grouped_pair_events", xcd_grid(nreg),
                               GroupedPairEvents{P, d_R.p, d_starts.p, d_lens.p, ngen, d_rep.p, d_evkey3.p, d_evval3.p, d_counter.p + kGrpSlot, (uint64_t)grp_cap, lbits,
                                                 d_glo.p, g_first, g_last, d_gflag.p, d_state.p, d_emax.p, d_epm.p, (int64_t)nreg});
            }
            be.memset(d_qcount.p, 0, 8 * (size_t)kSlices * kSliceStride);
            be.mark("seed_extend");
            be.launch("seed_extend", nunits * 64,
                      SeedExtend{P, d_R.p, d_units.p, d_slots.p, d_filter.p, d_next.p, d_rep.p, d_repeated.p,
                                 d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, d_err, work_budget, d_rest.p, d_qcount.p, (uint64_t)queue_cap, d_uoff.p + npairs});
            if (nunits > 0)      // one lane per queued sample; lanes past a sub-queue's count leave at once (the counts stay on the device)
                be.launch("seed_rest", (int64_t)(queue_cap * kSlices),
                          SeedRest{P, d_R.p, d_units.p, d_rest.p, d_qcount.p, (uint64_t)queue_cap, d_slots.p, d_filter.p, d_next.p, d_rep.p,
                                   d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, d_err, work_budget});
            if (!no_small)
                be.launch("small_pair_events", npairs * 2,
                          SmallPairEvents{P, d_R.p, d_starts.p, d_lens.p, ngen, d_rep.p, d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, g_first, g_last, grouping ? d_gflag.p : nullptr});
            be.mark("sort");
            be.launch_wave("slice_offsets", 1, SliceOffsets{d_counter.p, d_sliceoff.p});
            be.d2h_async(qcounts.data(), d_qcount.p, 8 * qcounts.size());
            if (from_store) be.d2h_async(alg_sets, d_alg.p, sizeof alg_sets);
            if (!units_known) be.d2h_async(&nunits_live, d_uoff.p + npairs, 8);
            be.d2h(counts.data(), d_counter.p, 8 * counts.size());            // round trip 1: event counts + error word (+ the queues' lengths, + the unit count)
            if (!units_known) {
                units_known = true;
                if (nunits_live >= (1ll << 31)) { error = "too many work units in one batch"; return -5; }
                if (nunits_live > nunits) {      // more units than the capacity: the event search again, over all of them
                    nunits = nunits_live;
                    queue_cap = std::max<size_t>(queue_cap, (size_t)nunits * kUnitSamples / 16 / kSlices + 64);
                    tail_repeats++;
                    continue;
                }
            }
            uint64_t worst = 0, qworst = 0;
            nev = 0; nrest = 0;
            for (int sl = 0; sl < kSlices; sl++) {
                const uint64_t c = counts[(size_t)sl * kSliceStride]; nev += c; worst = std::max(worst, c);
                const uint64_t q = qcounts[(size_t)sl * kSliceStride]; nrest += q; qworst = std::max(qworst, q);
            }
            errbits = (uint32_t)counts[(size_t)kSlices * kSliceStride];
            ngrp = grouping ? counts[kGrpSlot] : 0;
            worst_slice = worst;
            if (worst <= slice_cap && qworst <= queue_cap && ngrp <= grp_cap) break;
            if (ngrp > grp_cap) grp_cap = (size_t)(ngrp + ngrp / 8 + 64);
            if (worst > slice_cap) slice_cap = (size_t)(worst + worst / 8 + 64);
            if (qworst > queue_cap) queue_cap = (size_t)(qworst + qworst / 8 + 64);
            sticky |= errbits & (kErrWork | kErrRows);      // (RepeatLength's and CheckRows' verdicts: the word is cleared with the counters)
        }
        errbits |= sticky;
        if (from_store) {
            alg_raw[0] = alg_raw[1] = alg_raw[2] = 0;
            for (int x = 0; x < kAlgSets; x++) for (int k = 0; k < 3; k++) alg_raw[k] += alg_sets[8 * x + k];
        }
        if (from_store) { last_alg[0] = (double)alg_raw[0] / 4.0; last_alg[1] = (double)alg_raw[1] / 2.0; last_alg[2] = (double)alg_raw[2] / 2.0; }
        if (errbits & kErrRows) { error = "region outside its genome"; return -2; }
        rest_cap_hint[nreg == 1] = queue_cap - 64;      // (the margin is added again by the next call: storing it with the margin let the queues grow by 64 items a call, and every ~20 steps the 100 MB block was reallocated -- an 8-16 ms step)
        if (gb) hint.units = std::max<int64_t>(nunits_live, 1);
        last_rest = (int64_t)nrest;
        ev_cap_hint = (size_t)(nev + nev / 4);
        if (grouping) grp_cap_hint = (size_t)(ngrp + ngrp / 4);
        last_events = (int64_t)(nev + ngrp);
        last_grouped = (int64_t)ngrp;
        // a rank of a sharded run that ran out of budget must not leave the others waiting in the collectives: the
        // verdict travels with the first exchange (below) and every rank returns the error together
        const bool sharded = coll.world > 1 || coll.device;      // (a one-rank RCCL session still runs the exchanges: that is how a 1-GPU box tests them)
        if ((errbits & kErrWork) && !sharded) { budget_exceeded = true; error = "per-thread work budget exceeded (degenerate repeat structure in a region)"; return -5; }

        ensure(d_evkey2, std::max<size_t>(nev, 1)); ensure(d_evval2, std::max<size_t>(nev, 1));
        // the final array: [ grouped events, contiguous and ordered per pair | the other events, sorted by (pair, l, strand) ]
        ensure_keep(d_evkey3, std::max<size_t>(ngrp + nev, 1), (size_t)ngrp); ensure_keep(d_evval3, std::max<size_t>(ngrp + nev, 1), (size_t)ngrp);
        const int keybits = bits_for((uint64_t)npairs) + lbits + 1;
        uint64_t *skey = d_evkey3.p, *sval = d_evval3.p;
        // the events in order: by buckets (pair, 256-position block) whose scanned counts are the readers' coarse table as well, or
        // (tune bucket_sort = 0: rounds 1-5) gathered, radix-sorted, and the table from a pass over the sorted keys
        const int64_t nbuckets = (int64_t)nq * nchunks;
        const bool by_buckets = bucket_sort && nev > 0 && nbuckets < ((int64_t)1 << 31);
        ensure(d_lo, (size_t)npairs + 1);
        if (by_buckets) {
            ensure(d_bcount, (size_t)nbuckets + 1); ensure(d_bbegin, (size_t)nbuckets + 1); ensure(d_pbase, (size_t)npairs);
            int shift = 6;
            while (((uint64_t)1 << shift) < worst_slice) shift++;      // (the fullest slice's count: known since the wait above)
            be.memset(d_bcount.p, 0, 8 * ((size_t)nbuckets + 1));
            be.launch("pair_bucket_base", npairs, PairBucketBase{nq, d_cbase.p, d_pbase.p});
            be.launch("event_bucket_count", (int64_t)kSlices << shift, EventBucketCount{d_evkey.p, d_counter.p, (uint64_t)slice_cap, shift, lbits, d_pbase.p, d_bcount.p});
            be.exclusive_scan(d_bcount.p, d_bbegin.p, (size_t)nbuckets + 1);
            be.launch("event_place", (int64_t)kSlices << shift,
                      EventPlace{d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, shift, lbits, d_pbase.p, d_bbegin.p, d_bcount.p, d_evkey3.p + ngrp, d_evval3.p + ngrp});
            be.launch("event_order", nbuckets, EventOrder{d_bbegin.p, nbuckets, d_evkey3.p + ngrp, d_evval3.p + ngrp});
        } else if (nev > 0) {
            be.launch("compact_events", (int64_t)nev, CompactEvents{d_evkey.p, d_evval.p, d_sliceoff.p, (uint64_t)slice_cap, d_evkey2.p, d_evval2.p});
            be.sort_pairs(d_evkey2.p, d_evkey3.p + ngrp, d_evval2.p, d_evval3.p + ngrp, (size_t)nev, keybits);
        }
        be.mark("scan");
        const int64_t nev_sorted = (int64_t)nev;
        nev += ngrp;
        if (by_buckets) be.launch("coarse_from_buckets", centries, CoarseFromBuckets{d_bbegin.p, d_cbase.p, nreg, nq, (int64_t)ngrp, d_coarse.p, d_lo.p, npairs, nev_sorted});
        else be.launch("pair_bounds", npairs, PairBounds{skey, (int64_t)nev, lbits, npairs, d_lo.p, (int64_t)ngrp});
        if (grouping) be.launch("grouped_bounds", npairs, GroupedBounds{d_gflag.p, d_glo.p, nq, d_lo.p});
        // (the grouped events came with their states: the scan runs over the sorted part, its event numbers relative to it)
        ensure_keep(d_state, std::max<size_t>(nev, 1), (size_t)ngrp); ensure_keep(d_emax, std::max<size_t>(nev, 1), (size_t)ngrp);
        const int64_t nsorted = (int64_t)(nev - ngrp);
        if (nsorted > 0) {
            const int64_t nwaves = (nsorted + kWaveEvents - 1) / kWaveEvents;
            ensure(d_wsummary, (size_t)nwaves);
            const WaveScanCore core{skey + ngrp, sval + ngrp, nsorted, lbits, d_R.p, nq, d_rep.p};
            be.launch_wave("wave_summary", nwaves, WaveSummary{core, d_wsummary.p});
            be.launch_wave("wave_scan", nwaves, WaveScan{core, d_wsummary.p, d_lo.p, (int64_t)ngrp, d_state.p + ngrp, d_emax.p + ngrp});
        }

        if (want_events) {   // parity hook (pm_find_events): sorted events + rep'
            ev_key_h.resize((size_t)nev); ev_val_h.resize((size_t)nev); rep_h.resize((size_t)npos);
            if (nev) { be.d2h(ev_key_h.data(), skey, 8 * (size_t)nev); be.d2h(ev_val_h.data(), sval, 8 * (size_t)nev); }
            if (npos) be.d2h(rep_h.data(), d_rep.p, 4 * (size_t)npos);
            ev_lbits = lbits;
        }

        if (mumi) {
            be.mark("mumi_coverage");
            ensure(d_cov, (size_t)npairs);
            be.launch("mumi_coverage", npairs, MumiCoverage{d_R.p, skey, sval, d_lo.p, d_state.p, d_rep.p, lbits, ngen, d_cov.p});
            mumi_covered.resize((size_t)npairs);
            be.d2h(mumi_covered.data(), d_cov.p, 8 * (size_t)npairs);
            if (sharded) {   // every rank computed its own genome block: gather the blocks (+ the ranks' error words)
                int widest = 0;
                for (int r = 0; r < coll.world; r++) { int a, b; shard_range(ngen, r, coll.world, &a, &b); widest = std::max(widest, b - a); }
                const size_t per = (size_t)std::max(widest, 1) + 1;
                std::vector<int64_t> send(per, 0), recv(per * (size_t)coll.world);
                for (int g = g_first; g < g_last; g++) send[(size_t)(g - g_first)] = mumi_covered[(size_t)(g - 1)];
                send[per - 1] = errbits;
                if (allgather_host(send.data(), (int64_t)(8 * send.size()), recv.data())) { error = "all-gather of MUMi coverage failed"; return -4; }
                for (int r = 0; r < coll.world; r++) {
                    int a, b; shard_range(ngen, r, coll.world, &a, &b);
                    for (int g = a; g < b; g++) mumi_covered[(size_t)(g - 1)] = recv[(size_t)r * per + (size_t)(g - a)];
                    errbits |= (uint32_t)recv[(size_t)r * per + per - 1];
                }
                if (errbits & kErrWork) { budget_exceeded = true; error = "per-thread work budget exceeded on some rank (degenerate repeat structure in a region)"; return -5; }
            }
            be.mark(nullptr);
            collect_timing();
            return 0;
        }

        // -- Master.EP, candidates
        be.mark("master_ep");
        // (by buckets: the table came with the order -- but for the grouped regions' pairs, whose events never were in a bucket)
        if (!by_buckets) be.launch("coarse_fill", (int64_t)nev, CoarseFill{skey, (int64_t)nev, lbits, d_lo.p, d_R.p, d_cbase.p, nq, d_coarse.p});
        else if (ngrp > 0) be.launch("coarse_fill", (int64_t)ngrp, CoarseFill{skey, (int64_t)ngrp, lbits, d_lo.p, d_R.p, d_cbase.p, nq, d_coarse.p});
        if (master_seg) be.launch_wave("master_ep_seg", xcd_grid(nchunks), MasterEPSeg{d_R.p, nreg, ngen, skey, d_lo.p, d_emax.p, lbits, d_epm.p, d_cbase.p, d_coarse.p, g_first, g_last, grouping ? d_gflag.p : nullptr, nchunks});
        else be.launch_wave("master_ep", xcd_grid(nchunks), MasterEP{d_R.p, nreg, ngen, skey, d_lo.p, d_emax.p, lbits, d_epm.p, d_cbase.p, d_coarse.p, g_first, g_last, grouping ? d_gflag.p : nullptr, nchunks});
        int32_t verdict = 0;
        if (sharded && coll.device) {   // exchange 1 on the device: RCCL all-reduce(min) of Master.EP in place, + the error verdict word
            be.mark("exchange_ep");
            be.fill32(d_epm.p + npos, (errbits & kErrWork) ? -1 : 0);
            if (be.allreduce_min_i32_dev(d_epm.p, npos + 1)) { error = "RCCL all-reduce of Master.EP failed: " + be.error(); return -4; }
            be.d2h_async(&verdict, d_epm.p + npos, 4);       // read with the candidate count below
        } else if (sharded) {   // exchange 1: Master.EP = min over the ranks' genome blocks; the last word carries the error verdict
            be.mark("exchange_ep");
            std::vector<int32_t> h((size_t)npos + 1);
            if (npos) be.d2h(h.data(), d_epm.p, 4 * (size_t)npos);
            h[(size_t)npos] = (errbits & kErrWork) ? -1 : 0;
            if (coll.allreduce_min_i32(coll.ctx, h.data(), npos + 1)) { error = "all-reduce of Master.EP failed"; return -4; }
            verdict = h[(size_t)npos];
            if (verdict >= 0 && npos) be.h2d(d_epm.p, h.data(), 4 * (size_t)npos);
        }
        be.mark("candidates");
        be.launch_wave("cand_mark", nwv, CandMark{d_R.p, nreg, d_posbase.p, npos, d_epm.p, d_wmask.p, d_wcount.p});
        be.exclusive_scan(d_wcount.p, d_woff.p, (size_t)nwv + 1);
        const int64_t* ncand_p = d_woff.p + nwv;      // the candidate count, where the device keeps it
        // The rest of the call -- candidates listed, folded over the genomes, the accepted ones compacted into rows, the overlap flags
        // of a long list -- used to wait twice for the device, for the candidate count and for the accepted count, because the
        // launches and buffers were sized with them.  A call whose rows stay on the device (the resident route's anchor call and its
        // store searches) no longer does: launches and buffers take CAPACITIES from what the last call of the same shape needed,
        // every kernel reads the live count from device memory, and both counts come back with the results, in the call's one last
        // wait.  A capacity that turns out too small (or an anchor list that turns out short) repeats the tail the exact way.
        const bool anchor_guess = nreg == 1 && !gb && hint.nok >= dirty_min;
        const bool may_skip_waits = !sharded && want_rows && resident && hint.ncand > 0 && (from_store || anchor_guess) && fast_tail;
        for (int attempt = may_skip_waits ? 0 : 1;; attempt++) {
            const bool exact = attempt > 0;
            int64_t ncand_i = 0;
            if (exact) be.d2h(&ncand_i, ncand_p, 8);                                   // round trip 2: candidate count
            if (verdict < 0) { budget_exceeded = true; error = "per-thread work budget exceeded on some rank (degenerate repeat structure in a region)"; return -5; }
            if (exact) {
                last_candidates = ncand_i;
                if (ncand_i == 0) { hint.ncand = 1; hint.nok = 0; if (resident && from_store) out->store_base = ms_count; be.mark(nullptr); collect_timing(); return 0; }
            }
            // capacity of the candidate list: the count itself when it is known
            const uint64_t ncand = exact ? (uint64_t)ncand_i : (uint64_t)(hint.ncand + hint.ncand / 8 + 256);
            ensure(d_cand, (size_t)ncand);
            be.launch("cand_write", nwv * 64, CandWrite{d_R.p, nreg, d_posbase.p, d_wmask.p, d_woff.p, d_cand.p, ncand});
            const uint64_t* scand = d_cand.p;        // in (region, k) order by construction

            // -- per-candidate genome fold
            be.mark("fold");
            ensure(d_ok, (size_t)ncand); ensure(d_ok_k, (size_t)ncand); ensure(d_ok_lon, (size_t)ncand);
            ensure(d_osp, (size_t)ncand * (size_t)nq); ensure(d_ofwd, (size_t)ncand * (size_t)nq);
            const GenomeAtK* at = nullptr;
            if (sharded) {   // exchange 2: every rank contributes the (EP,UP,SP) columns of its genome block
                ensure(d_at, (size_t)ncand * (size_t)nq);
                be.launch("state_at_candidate", (int64_t)ncand * nq, StateAtCandidate{d_R.p, scand, ngen, skey, sval, d_lo.p, d_state.p, d_rep.p, lbits, d_at.p, d_cbase.p, d_coarse.p});
                be.mark("exchange_states");
                const size_t nqz = (size_t)nq, cz = (size_t)ncand;
                int widest = 0;
                for (int r = 0; r < coll.world; r++) { int a, b; shard_range(ngen, r, coll.world, &a, &b); widest = std::max(widest, b - a); }
                widest = std::max(widest, 1);
                const size_t blk = cz * (size_t)widest;
                if (coll.device) {      // pack this rank's columns, RCCL all-gather on the engine's stream, spread the blocks: nothing leaves the device
                    ensure(d_xsend, blk); ensure(d_xrecv, blk * (size_t)coll.world);
                    be.launch("pack_states", (int64_t)blk, PackStates{d_at.p, nq, g_first - 1, g_last - g_first, widest, d_xsend.p});
                    if (be.allgather_dev(d_xsend.p, (int64_t)(sizeof(GenomeAtK) * blk), d_xrecv.p)) { error = "RCCL all-gather of candidate states failed: " + be.error(); return -4; }
                    be.launch("unpack_states", (int64_t)(cz * nqz), UnpackStates{d_xrecv.p, nq, coll.world, widest, (int64_t)ncand, d_at.p});
                } else {
                    std::vector<GenomeAtK> all(cz * nqz);
                    be.d2h(all.data(), d_at.p, sizeof(GenomeAtK) * all.size());
                    std::vector<GenomeAtK> send(blk), recv(blk * (size_t)coll.world);
                    const size_t mine = (size_t)(g_last - g_first);
                    for (size_t c = 0; c < cz; c++)
                        for (size_t x = 0; x < mine; x++) send[c * (size_t)widest + x] = all[c * nqz + (size_t)(g_first - 1) + x];
                    if (coll.allgather(coll.ctx, send.data(), (int64_t)(sizeof(GenomeAtK) * send.size()), recv.data())) { error = "all-gather of candidate states failed"; return -4; }
                    for (int r = 0; r < coll.world; r++) {
                        int a, b; shard_range(ngen, r, coll.world, &a, &b);
                        const GenomeAtK* src = recv.data() + (size_t)r * send.size();
                        for (size_t c = 0; c < cz; c++)
                            for (int x = 0; x < b - a; x++) all[c * nqz + (size_t)(a - 1 + x)] = src[c * (size_t)widest + (size_t)x];
                    }
                    be.h2d(d_at.p, all.data(), sizeof(GenomeAtK) * all.size());
                }
                be.mark("fold");
                at = d_at.p;
            }
            be.launch_wave("fold_candidates", xcd_grid((int64_t)ncand),
                           FoldCandidates{d_R.p, scand, ngen, at, skey, sval, d_lo.p, d_state.p, d_rep.p, lbits, d_cbase.p, d_coarse.p,
                                          d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_ok.p, (int64_t)ncand, ncand_p});

            // -- accepted candidates, compacted on the device and downloaded straight into the result's blocks
            be.mark("compact");
            ensure(d_okcnt, (size_t)ncand + 1); ensure(d_okpos, (size_t)ncand + 1);
            be.launch("ok_count", (int64_t)ncand + 1, OkCount{d_ok.p, ncand_p, d_okcnt.p});
            be.exclusive_scan(d_okcnt.p, d_okpos.p, (size_t)ncand + 1);
            const int64_t* nok_p = d_okpos.p + ncand;      // the accepted count, where the device keeps it
            int64_t nok = 0;
            if (exact) be.d2h(&nok, nok_p, 8);                                    // round trip 3: accepted count
            // capacity of the accepted rows: the count itself when it is known, else every candidate could be accepted
            const size_t nokz = exact ? (size_t)nok : (size_t)ncand, nqz2 = (size_t)nq, ngz = (size_t)ngen;
            ensure(d_creg, std::max<size_t>(nokz, 1)); ensure(d_ck, std::max<size_t>(nokz, 1)); ensure(d_clon, std::max<size_t>(nokz, 1));
            std::vector<int32_t> reg_h(nokz);
            out->release();
            out->kb = pool->take(4 * nokz); out->lonb = pool->take(4 * nokz);
            out->rows = want_rows; out->dirty_known = false; out->table_id = 0; out->store_base = -1;
            const int64_t ms_before = ms_count, rg_before = rg_count, lay_before = layout_rows;
            bool anchor_call = false;
            if (!want_rows) {
                ensure(d_csp, std::max<size_t>(nokz * nqz2, 1)); ensure(d_cfwd, std::max<size_t>(nokz * nqz2, 1));
                be.launch("compact_sp", (int64_t)ncand * nq,
                          CompactSp{scand, d_ok.p, d_okpos.p, nq, d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_creg.p, d_ck.p, d_clon.p, d_csp.p, d_cfwd.p, ncand_p});
                be.mark("download");
                out->spb = pool->take(4 * nokz * nqz2); out->fwdb = pool->take(nokz * nqz2);
                be.d2h_async(out->spb.p, d_csp.p, 4 * nokz * nqz2);
                be.d2h_async(out->fwdb.p, d_cfwd.p, nokz * nqz2);
            } else {
                // Resident mode (pm_session_rows(s, 2)): the rows of the anchor call, and of every search of regions of the region store,
                // stay on the device as rows of the MUM store -- the host receives the per-row scalars only (flags, k, length).  The
                // anchor call's rows are the session's anchor table (= rows [0, A) of the store) in either mode.  The compaction
                // writes such rows straight into the store (a copy of the 60 MB anchor table is half a millisecond).
                anchor_call = nreg == 1 && !gb && (exact ? nok >= dirty_min : anchor_guess);
                const bool keep_rows = resident && (anchor_call || from_store);
                const bool to_store = anchor_call || keep_rows;
                int32_t* p_start; uint8_t* p_strand; int32_t* p_lon; uint32_t* p_flags;
                if (to_store) {
                    if (anchor_call) { ms_count = 0; rg_count = 0; layout_rows = -1; }
                    const size_t base = (size_t)ms_count, upto = base + nokz;
                    ensure_keep(d_anchor_start, std::max<size_t>(upto * ngz, 1), base * ngz); ensure_keep(d_ms_strand, std::max<size_t>(upto * ngz, 1), base * ngz);
                    ensure_keep(d_anchor_lon, std::max<size_t>(upto, 1), base); ensure_keep(d_anchor_flags, std::max<size_t>(upto, 1), base);
                    ensure_keep(d_ms_shift, std::max<size_t>(upto, 1), base); ensure_keep(d_ms_len, std::max<size_t>(upto, 1), base); ensure_keep(d_ms_state, std::max<size_t>(upto, 1), base);
                    p_start = d_anchor_start.p + base * ngz; p_strand = d_ms_strand.p + base * ngz; p_lon = d_anchor_lon.p + base; p_flags = d_anchor_flags.p + base;
                    out->store_base = (int64_t)base;
                    ms_count = (int64_t)upto;      // (capacity: set to base + the accepted count once that is known)
                } else {
                    ensure(d_csp, std::max<size_t>(nokz * ngz, 1)); ensure(d_cfwd, std::max<size_t>(nokz * ngz, 1)); ensure(d_cflags, std::max<size_t>(nokz, 1));
                    p_start = d_csp.p; p_strand = d_cfwd.p; p_lon = d_clon.p; p_flags = d_cflags.p;
                }
                const bool long_list = nreg == 1 && (exact ? nok >= dirty_min : anchor_guess);
                if (long_list) ensure(d_dirty, nokz);
                {      // the flags of the rows (OR-ed into), the cheap overlap test's marks, shift and state of the new store rows: one launch
                    const ClearJob jobs[] = {{p_flags, 4 * nokz, 0}, {long_list ? d_dirty.p : nullptr, long_list ? 4 * nokz : 0, 0},
                                             {to_store ? d_ms_shift.p + (ms_count - (int64_t)nokz) : nullptr, to_store ? 4 * nokz : 0, 0},
                                             {to_store ? d_ms_state.p + (ms_count - (int64_t)nokz) : nullptr, to_store ? nokz : 0, 0}};
                    be.clear_many(jobs, 4);
                }
                be.launch("compact_candidates", (int64_t)ncand * ngen,
                          CompactCandidates{scand, d_ok.p, d_okpos.p, ngen, d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_starts.p, d_lens.p, d_glen,
                                            d_creg.p, d_ck.p, p_lon, p_start, p_strand, p_flags, ncand_p});
                if (long_list) {     // a long list of one region (the anchor call): the cheap overlap test on the device
                    const int64_t nblocks = ((int64_t)nokz + kDirtyBlock - 1) / kDirtyBlock, groups = (ngen + 63) / 64;
                    ensure(d_bmax, (size_t)nblocks * ngz); ensure(d_bmin, (size_t)nblocks * ngz);
                    be.launch_wave("dirty_extent", nblocks * groups, DirtyExtent{p_start, p_lon, p_flags, nok_p, ngen, d_bmax.p, d_bmin.p});
                    be.launch_wave("dirty_prefix", groups, DirtyPrefix{nblocks, ngen, d_bmax.p, d_bmin.p});
                    be.launch_wave("dirty_mark", nblocks * groups, DirtyMark{p_start, p_lon, nok_p, ngen, d_bmax.p, d_bmin.p, p_flags, d_dirty.p});
                    be.launch("dirty_merge", (int64_t)nokz, DirtyMerge{d_dirty.p, p_flags, nok_p});
                    out->dirty_known = true;
                }
                if (to_store) {
                    be.d2d(d_ms_len.p + (ms_count - (int64_t)nokz), p_lon, 4 * nokz);
                    if (anchor_call) out->table_id = ++table_counter;
                }
                be.mark("download");
                out->flagsb = pool->take(4 * nokz);
                be.d2h_async(out->flagsb.p, p_flags, 4 * nokz);
                if (!keep_rows) {
                    out->startb = pool->take(4 * nokz * ngz); out->strandb = pool->take(nokz * ngz);
                    be.d2h_async(out->startb.p, p_start, 4 * nokz * ngz);
                    be.d2h_async(out->strandb.p, p_strand, nokz * ngz);
                }
                be.d2h_async(out->lonb.p, p_lon, 4 * nokz);
            }
            if (!want_rows) be.d2h_async(out->lonb.p, d_clon.p, 4 * nokz);
            be.d2h_async(reg_h.data(), d_creg.p, 4 * nokz);
            be.d2h_async(out->kb.p, d_ck.p, 4 * nokz);
            int64_t counts_h[2] = {ncand_i, nok};
            if (!exact) { be.d2h_async(&counts_h[0], ncand_p, 8); be.d2h_async(&counts_h[1], nok_p, 8); }
            be.mark(nullptr);
            be.sync();                                                         // round trip 4: the results (and, where the call did not wait for them, the two counts)
            if (!exact) {
                ncand_i = counts_h[0]; nok = counts_h[1];
                const bool fits = (uint64_t)ncand_i <= ncand && anchor_call == (nreg == 1 && !gb && nok >= dirty_min);
                if (!fits) {      // the capacity was too small, or the list is not the long list it was taken for: again, with the counts
                    ms_count = ms_before; rg_count = rg_before; layout_rows = lay_before;
                    tail_repeats++;
                    be.mark("candidates");
                    continue;
                }
                last_candidates = ncand_i;
                if (out->store_base >= 0) ms_count = out->store_base + nok;
            }
            hint.ncand = std::max<int64_t>(ncand_i, 1); hint.nok = nok;
            last_accepted = nok;
            if (anchor_call) { anchor_table_rows = nok; anchor_table_id = out->table_id; }
            for (int64_t w = 0; w < nok; w++) out->off[(size_t)reg_h[(size_t)w] + 1]++;
            for (int64_t r = 0; r < nreg; r++) out->off[(size_t)r + 1] += out->off[(size_t)r];
            out->total = out->off[(size_t)nreg];
            if (out->table_id && out->store_base == 0) anchor_flags_h.assign(out->flags(), out->flags() + nok);      // (settle() lists the flagged rows from it)
            break;
        }
        collect_timing();
        return 0;
    }


    // =====================================================================================================================
    // The resident route (store_kernels.h; include/parsnp_mum.h: pm_store_*): what the reference does with the candidate lists
    // AFTER csgmum produced them, on rows that never leave the device.  Every method is one call of the C ABI.
    // =====================================================================================================================
    bool resident = false;            // pm_session_rows(s, 2)
    int64_t ms_count = 0;             // rows of the MUM store: [0, anchor_table_rows) = the anchor table, then every search of store regions
    int64_t rg_count = 0;             // regions of the region store
    std::vector<RegInfo> rg_info_h;   // host mirror of what the device said about each region (by region id)
    std::vector<uint32_t> anchor_flags_h;
    static constexpr int kAgain = -6; // PM_EAGAIN: the resident route does not apply; the caller takes the host route

    void begin_store_call() { timing.clear(); last_events = last_rest = last_positions = last_candidates = last_accepted = last_grouped = 0; last_alg[0] = last_alg[1] = last_alg[2] = 0; }      // (counts of the last search travel with pm_last_timing)
    Store store_view() { return Store{d_anchor_start.p, d_ms_strand.p, d_anchor_lon.p, d_anchor_flags.p, d_ms_shift.p, d_ms_len.p, d_ms_state.p, ngen}; }
    // coherent: the reader must see marks made while its kernel runs (a wavefront that validates candidates in order)
    Layout layout_view(uint64_t* image, bool coherent = true) { return Layout{image, d_lay_off.p, d_lay_bits.p, coherent ? 1 : 0}; }
    // geometry of the layout image: genome j has glen[j] + 1 bits (the last one the sentinel, src/parsnp.cpp:3184-3185)
    size_t layout_geometry() {
        const size_t ngz = (size_t)ngen;
        if (lay_words == 0) {
            std::vector<int64_t> off(ngz + 1, 0), bits(ngz);
            for (size_t j = 0; j < ngz; j++) { bits[j] = glen_h[j] + 1; off[j + 1] = off[j] + (bits[j] + 63) / 64 + 1; }
            ensure(d_lay_off, ngz + 1); ensure(d_lay_bits, ngz);
            be.h2d(d_lay_off.p, off.data(), 8 * (ngz + 1)); be.h2d(d_lay_bits.p, bits.data(), 8 * ngz);
            lay_words = (size_t)off[ngz];
            lay_off_h = off;
        }
        return lay_words;
    }
    int need_resident(int64_t table_id) {
        if (!resident) { error = "the session is not in resident mode"; return -2; }
        if (table_id == 0 || table_id != anchor_table_id) { error = "the anchor table of this request is no longer resident"; return -2; }
        return 0;
    }
    // Second half of setMums1 (src/parsnp.cpp:1713-1841) for the resident anchor list, into an EMPTY layout: rows that overlap
    // nothing earlier are settled at once and marked; the flagged ones against those marks -- all at once where they meet no
    // other flagged row, in list order where they do.  out[c] for every row of the table.
    int store_settle(int64_t table_id, RowInfo* out) {
        if (int rc = settle_launch(table_id)) return rc;
        return store_info(0, anchor_table_rows, out);
    }
    // the launches of store_settle (nothing is waited for)
    int settle_launch(int64_t table_id) {
        if (int rc = need_resident(table_id)) return rc;
        begin_store_call();
        const int64_t rows = anchor_table_rows;
        // (the clears are queued first: the device wipes 2 x 126 MB while the host looks through the list's flags)
        const size_t words = layout_geometry();
        ensure(d_image, words);
        be.mark("settle");
        ensure(d_foreign_count, 2);
        be.memset(d_foreign_count.p, 0, 16);      // (candidates noted for store_order_check: none yet)
        ms_key_rows = 0; foreign_seen = 0;
        be.memset(d_image.p, 0, 8 * words);
        ensure(d_recwords, words);
        be.memset(d_recwords.p, 0, words);      // (where the recursion marks: ClusterValidate, for the order check)
        const Store S = store_view();
        const Layout L = layout_view(d_image.p);
        be.launch_wave("settle_clean", rows, SettleClean{S, P});
        std::vector<int32_t> fl;
        for (int64_t c = 0; c < rows; c++) { const uint32_t f = anchor_flags_h[(size_t)c]; if (!(f & (kRowBad | kRowOutside)) && (f & kRowDirty)) fl.push_back((int32_t)c); }
        // The cheap running-extent test flags every row of an inverted or moved block, whether it overlaps anything or not: a
        // population with a few inversions has a third of its rows flagged and a handful tangled, a set with 10 % of every one of its
        // 500 genomes rearranged has all of them flagged and thousands tangled.  The flagged rows that meet no other flagged row cost a
        // wavefront each, side by side; the TANGLED ones are settled in list order where they meet one another -- in rounds, every
        // row as soon as the tangled rows before it that share a 64-base word with it are done (TangleOwner / TangleSettle), and what
        // kTangleRounds rounds leave by one wavefront.  With more than one row in flagged_div flagged the tangled rows are counted
        // first (one 8-byte read-back) and the list is declined above tangled_max of them (PM_EAGAIN: the host route's exact overlap
        // test and threads).  flagged_div = 1: never declined (tests).
        const bool many_flagged = flagged_div > 1 && fl.size() * (size_t)flagged_div > (size_t)rows;
        // do the rows that can be accepted untrimmed lie in list order in every genome (none starts before the end of an earlier
        // row: the engine's kRowEarly bit)?  Then their marks need no atomics (StoreMarkOrdered)
        bool ordered = true;
        for (int64_t c = 0; c < rows && ordered; c++) { const uint32_t f = anchor_flags_h[(size_t)c]; ordered = !((f & kRowEarly) && !(f & (kRowBad | kRowOutside | kRowDirty))); }
        if (ordered && !force_atomic_marks) be.launch_wave("store_mark_ordered", ((rows + kMarkRows - 1) / kMarkRows) * ((ngen + 63) / 64), StoreMarkOrdered{S, L, rows});
        else be.launch("store_mark", rows * ngen, StoreMark{S, L, 0, (uint8_t)(kStAccepted | kStFlagged), kStAccepted});
        be.launch("layout_sentinel", (int64_t)ngen, LayoutSentinel{d_lay_off.p, d_lay_bits.p, d_image.p});
        if (!fl.empty()) {
            const uint64_t* before_once = d_once.p; const uint64_t* before_twice = d_twice.p;
            ensure(d_once, words); ensure(d_twice, words); ensure(d_list, fl.size());
            if (d_once.p != before_once || d_twice.p != before_twice) { be.memset(d_once.p, 0, 8 * words); be.memset(d_twice.p, 0, 8 * words); }      // (kept all zero between calls: CollideClear)
            be.h2d(d_list.p, fl.data(), 4 * fl.size());
            be.launch("collide_mark", (int64_t)fl.size() * ngen, CollideMark{S, d_list.p, layout_view(d_once.p), d_twice.p});
            be.launch_wave("collide_test", (int64_t)fl.size(), CollideTest{S, d_list.p, layout_view(d_twice.p, false)});
            be.launch("collide_clear", (int64_t)fl.size() * ngen, CollideClear{S, d_list.p, layout_view(d_once.p), d_twice.p});
            ensure(d_t_rem, 2); ensure(d_t_done, fl.size());
            {
                const ClearJob jobs[] = {{d_t_rem.p, 16, 0}, {d_t_done.p, fl.size(), 0}};
                be.clear_many(jobs, 2);
            }
            be.launch("count_tangled", (int64_t)fl.size(), CountTangled{S, d_list.p, d_t_rem.p});
            if (many_flagged) {
                uint64_t tangled = 0;
                be.d2h(&tangled, d_t_rem.p, 8);
                if ((int64_t)tangled > tangled_max) {
                    be.mark(nullptr);
                    collect_timing_more();
                    error = "too many rows of the list overlap one another (rearranged genomes): the host route decides";
                    return kAgain;
                }
            }
            be.launch_wave("settle_flagged", (int64_t)fl.size(), SettleFlagged{S, layout_view(d_image.p, false), P, d_list.p});
            {
                const int32_t* before = d_owner.p;
                ensure(d_owner, words);
                if (d_owner.p != before) be.memset(d_owner.p, 0, 4 * words);      // (kept all zero between calls)
            }
            // as many rounds as the last step of the session needed, + 1 (a round that settles nothing costs three launches of 4 us;
            // what the rounds leave is the serial wavefront's either way): every round notes how many rows it left
            int rounds = tangle_rounds ? kTangleRounds : 0;
            if (rounds && tangle_ran > 0) {
                int last = 0;
                for (int r = 1; r < tangle_ran; r++) if (tangle_left[r] < tangle_left[r - 1]) last = r;
                rounds = std::min(kTangleRounds, last + 2);
            }
            ensure(d_t_left, kTangleRounds);
            for (int round = 0; round < rounds; round++) {
                be.launch_wave("tangle_owner", (int64_t)fl.size(), TangleOwner{S, d_list.p, d_lay_off.p, d_lay_bits.p, d_owner.p, d_t_done.p, d_t_rem.p});
                be.launch_wave("tangle_settle", (int64_t)fl.size(), TangleSettle{S, layout_view(d_image.p, false), P, d_list.p, d_owner.p, d_t_done.p, d_t_rem.p});
                be.launch_wave("tangle_clear", (int64_t)fl.size(), TangleClear{S, d_list.p, d_lay_off.p, d_lay_bits.p, d_owner.p, d_t_done.p, d_t_rem.p, d_t_left.p + round});
            }
            tangle_ran = 0;
            if (rounds) { be.d2h_async(tangle_left, d_t_left.p, 8 * (size_t)rounds); tangle_ran = rounds; }      // (lands with the call's results)
            be.launch_wave("settle_tangled", 1, SettleTangled{S, L, P, d_list.p, (int64_t)fl.size(), d_t_done.p, d_t_rem.p});
        }
        layout_rows = rows;
        return 0;
    }
    // store_settle and store_seeds in one call, without the round trip between them: the accepted anchors are listed on the device,
    // the kept regions are placed in the reference's push order (SeedCount / SeedPlace), and the per-row scalars of the table travel
    // with the regions' summaries.  infos / ids: as store_seeds leaves them.
    int store_settle_seeds(int64_t table_id, int32_t q, RowInfo* out, std::vector<RegInfo>* infos, std::vector<int32_t>* ids) {
        infos->clear(); ids->clear();
        if (int rc = settle_launch(table_id)) return rc;
        const int64_t rows = anchor_table_rows;
        rg_count = 0;
        if (rows == 0) return 0;
        const Store S = store_view();
        ensure(d_rowinfo, (size_t)rows);
        be.launch("store_info", rows, StoreInfoOut{S, 0, d_rowinfo.p});
        be.mark("seeds");
        ensure(d_ch_flag, (size_t)rows + 1); ensure(d_ch_pos, (size_t)rows + 1); ensure(d_list, (size_t)rows);
        ensure(d_sd_cnt, (size_t)rows + 1); ensure(d_sd_off, (size_t)rows + 1); ensure(d_sd_keep, (size_t)rows);
        be.launch("chain_flag", rows + 1, ChainFlag{S, rows, d_ch_flag.p});
        be.exclusive_scan(d_ch_flag.p, d_ch_pos.p, (size_t)rows + 1);
        be.launch("anchor_list", rows, AnchorList{S, d_ch_pos.p, d_list.p});
        const int64_t* nacc = d_ch_pos.p + rows;
        const Layout L = layout_view(d_image.p, false);
        be.memset(d_sd_cnt.p + rows, 0, 8);
        be.launch_wave("seed_count", xcd_grid(rows), SeedCount{S, L, P, d_list.p, nacc, q, d_sd_cnt.p, d_sd_keep.p, rows});
        be.exclusive_scan(d_sd_cnt.p, d_sd_off.p, (size_t)rows + 1);
        size_t cap = std::max<size_t>(rg_cap_hint, (size_t)rows / 4 + 1024);
        std::vector<RegInfo> raw;
        for (bool again = false;; again = true) {
            ensure(d_rg_start, cap * (size_t)ngen); ensure(d_rg_len, cap * (size_t)ngen); ensure(d_rg_info, cap);
            be.launch_wave("seed_place", xcd_grid(rows), SeedPlace{S, L, P, d_list.p, nacc, d_sd_off.p, d_sd_keep.p, d_rg_start.p, d_rg_len.p, d_rg_info.p, (uint64_t)cap, rows});
            be.mark(nullptr);
            // the summaries travel with the count: as many as the last run had (a run with more fetches the rest)
            const size_t guess = std::min(cap, rg_cap_hint);
            raw.resize(guess);
            if (guess) be.d2h_async(raw.data(), d_rg_info.p, sizeof(RegInfo) * guess);
            if (!again) be.d2h_async(out, d_rowinfo.p, sizeof(RowInfo) * (size_t)rows);
            int64_t got = 0;
            be.d2h(&got, d_sd_off.p + rows, 8);
            if ((size_t)got <= cap) {
                rg_count = got;
                raw.resize((size_t)got);
                if ((size_t)got > guess) be.d2h(raw.data() + guess, d_rg_info.p + guess, sizeof(RegInfo) * ((size_t)got - guess));
                break;
            }
            cap = (size_t)got + (size_t)got / 8 + 64;
            be.mark("seeds");
        }
        rg_cap_hint = (size_t)rg_count + (size_t)rg_count / 4;
        if (rg_info_h.size() < (size_t)rg_count) rg_info_h.resize((size_t)rg_count);
        bool sorted = true;
        for (int64_t i = 0; i < rg_count; i++) { rg_info_h[(size_t)i] = raw[(size_t)i]; if (i && raw[(size_t)i - 1].key >= raw[(size_t)i].key) sorted = false; }
        if (!sorted) { error = "seed regions out of order"; return -4; }
        infos->swap(raw);
        ids->resize((size_t)rg_count);
        for (int64_t i = 0; i < rg_count; i++) (*ids)[(size_t)i] = (int32_t)i;
        collect_timing_more();
        return 0;
    }
    // per-row scalars of store rows [first, first + count)
    int store_info(int64_t first, int64_t count, RowInfo* out) {
        if (!resident || first < 0 || count < 0 || first + count > ms_count) { error = "rows outside the MUM store"; return -2; }
        if (count == 0) return 0;
        ensure(d_rowinfo, (size_t)count);
        be.launch("store_info", count, StoreInfoOut{store_view(), first, d_rowinfo.p});
        be.mark(nullptr);
        be.d2h(out, d_rowinfo.p, sizeof(RowInfo) * (size_t)count);
        collect_timing_more();
        return 0;
    }
    // the regions the device appended since `first`, as the host wants them: sorted by key, dropped ones left out
    void take_regions(int64_t first, int64_t upto, std::vector<RegInfo>* infos, std::vector<int32_t>* ids) {
        const size_t m = (size_t)(upto - first);
        std::vector<RegInfo> raw(m);
        if (m) be.d2h(raw.data(), d_rg_info.p + first, sizeof(RegInfo) * m);
        if (rg_info_h.size() < (size_t)upto) rg_info_h.resize((size_t)upto);
        std::vector<int32_t> order;
        for (size_t i = 0; i < m; i++) { rg_info_h[(size_t)first + i] = raw[i]; if (raw[i].key >= 0) order.push_back((int32_t)i); }
        std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return raw[(size_t)x].key < raw[(size_t)y].key; });
        infos->clear(); ids->clear();
        for (int32_t i : order) { infos->push_back(raw[(size_t)i]); ids->push_back((int32_t)(first + i)); }
    }
    // setInitialClusters' seed regions (src/parsnp.cpp:2150-2172): both neighbour regions of every accepted anchor (acc: their rows,
    // in list order) by walks over the image, kept when longer than q in every genome; in the reference's push order
    int store_seeds(int64_t table_id, const int32_t* acc, int64_t nacc, int32_t q, std::vector<RegInfo>* infos, std::vector<int32_t>* ids) {
        if (int rc = need_resident(table_id)) return rc;
        if (layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        for (int64_t i = 0; i < nacc; i++) if (acc[i] < 0 || acc[i] >= anchor_table_rows) { error = "anchor outside the table"; return -2; }
        rg_count = 0;
        infos->clear(); ids->clear();
        begin_store_call();
        if (nacc == 0) return 0;
        ensure(d_list, (size_t)nacc); ensure(d_rg_count, 2);
        be.h2d(d_list.p, acc, 4 * (size_t)nacc);
        be.mark("seeds");
        size_t cap = std::max<size_t>(rg_cap_hint, (size_t)nacc / 4 + 1024);
        for (;;) {
            ensure(d_rg_start, cap * (size_t)ngen); ensure(d_rg_len, cap * (size_t)ngen); ensure(d_rg_info, cap);
            be.memset(d_rg_count.p, 0, 16);
            be.launch_wave("seed_walk", xcd_grid(nacc), SeedWalk{store_view(), layout_view(d_image.p, false), P, d_list.p, q, d_rg_start.p, d_rg_len.p, d_rg_info.p, d_rg_count.p, (uint64_t)cap, (int64_t)nacc});
            uint64_t got = 0;
            be.d2h(&got, d_rg_count.p, 8);
            if (got <= cap) { rg_count = (int64_t)got; break; }
            cap = (size_t)got + (size_t)got / 8 + 64;
        }
        rg_cap_hint = (size_t)rg_count + (size_t)rg_count / 4;
        be.mark(nullptr);
        take_regions(0, rg_count, infos, ids);
        collect_timing_more();
        return 0;
    }
    // are regions a[i] and b[i] the same in every genome (TRegion operator==)?
    int store_regions_equal(const int32_t* a, const int32_t* b, int64_t n, uint8_t* same) {
        if (!resident) { error = "the session is not in resident mode"; return -2; }
        for (int64_t i = 0; i < n; i++) if (a[i] < 0 || a[i] >= rg_count || b[i] < 0 || b[i] >= rg_count) { error = "region outside the region store"; return -2; }
        if (n == 0) return 0;
        ensure(d_list, (size_t)n); ensure(d_list2, (size_t)n); ensure(d_small8, (size_t)n);
        be.h2d(d_list.p, a, 4 * (size_t)n); be.h2d(d_list2.p, b, 4 * (size_t)n);
        be.launch_wave("regions_equal", n, RegionsEqual{d_list.p, d_list2.p, ngen, d_rg_start.p, d_rg_len.p, d_small8.p});
        be.d2h(same, d_small8.p, (size_t)n);
        return 0;
    }
    // the multi-MUM search of regions of the store (pm_multi_mum_batch on their rows); the candidates become rows
    // [*first_row + off[i], *first_row + off[i + 1]) of the MUM store
    int store_search(const int32_t* ids, const int32_t* minsize, int64_t n, int64_t* first_row, int64_t* off) {
        if (!resident) { error = "the session is not in resident mode"; return -2; }
        std::vector<int64_t> rs((size_t)n), rl((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            if (ids[i] < 0 || ids[i] >= rg_count) { error = "region outside the region store"; return -2; }
            rs[(size_t)i] = rg_info_h[(size_t)ids[i]].ref_start; rl[(size_t)i] = rg_info_h[(size_t)ids[i]].ref_len;
        }
        *first_row = ms_count;
        off[0] = 0;
        if (n == 0) return 0;
        GapBatch gb;
        gb.store_ids = ids; gb.ref_start = rs.data(); gb.ref_len = rl.data();
        BatchResult br;
        const bool keep = want_rows;
        want_rows = true;
        const int rc = run(n, nullptr, nullptr, minsize, &br, false, false, &gb);
        want_rows = keep;
        if (rc) return rc;
        *first_row = br.store_base >= 0 ? br.store_base : ms_count;
        for (int64_t i = 0; i <= n; i++) off[i] = br.off[(size_t)i];
        return 0;
    }
    // One generation of the recursion (doWork, src/parsnp.cpp:173-317) over clusters of waiting regions that the caller found
    // pairwise disjoint in every genome: candidates settled against the image and marked, children appended to the region store.
    int store_validate(const int32_t* regions, const int64_t* row0, const int32_t* cnt, int64_t nreg, const int64_t* cluster_first, int64_t ncl, int32_t q,
                       uint32_t* trouble, std::vector<RegInfo>* kids, std::vector<int32_t>* kid_ids, int64_t info_first = 0, int64_t info_count = 0, RowInfo* info = nullptr,
                       int64_t stage_first = 0, int32_t* second_stage_ran = nullptr, int32_t generation_no = 1, int32_t* done = nullptr) {
        // stage_first > 0: clusters [0, stage_first) are a generation of their own (the first pushed seed, which the reference
        // processes before anything is sorted); the rest -- the generation the caller formed on the assumption that the first leaves
        // no child region -- runs behind it in the same call if that held, and is left untouched if not (*second_stage_ran = 0)
        // done[cluster]: how many regions of the cluster were processed -- all of them, or fewer: none where the cluster meets an
        // earlier one in some genome (it waits for that one and for everything it leads to), the first few where a child sorts
        // before the next waiting region.  done == nullptr: a caller that cannot keep regions waiting; either case is trouble then
        // (bits 3 / 0) and the run must be discarded
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        kids->clear(); kid_ids->clear(); *trouble = 0;
        if (nreg == 0 || ncl == 0) return 0;
        int64_t total = 0;
        for (int64_t x = 0; x < nreg; x++) {
            if (regions[x] < 0 || regions[x] >= rg_count || row0[x] < 0 || cnt[x] < 0 || row0[x] + cnt[x] > ms_count) { error = "bad generation list"; return -2; }
            total += cnt[x];
        }
        if (cluster_first[0] != 0 || cluster_first[ncl] != nreg) { error = "bad cluster list"; return -2; }
        if (stage_first < 0 || stage_first >= ncl) { error = "bad stage boundary"; return -2; }
        begin_store_call();
        const size_t cap = (size_t)rg_count + 2 * (size_t)total + 16;       // (every accepted candidate has two neighbour regions)
        ensure_keep(d_rg_start, cap * (size_t)ngen, (size_t)rg_count * (size_t)ngen); ensure_keep(d_rg_len, cap * (size_t)ngen, (size_t)rg_count * (size_t)ngen);
        ensure_keep(d_rg_info, cap, (size_t)rg_count);
        const size_t bytes = 4 * (size_t)nreg + 8 * (size_t)nreg + 4 * (size_t)nreg + 8 * ((size_t)ncl + 1) + 64;
        uint8_t* block = (uint8_t*)be.staging(bytes);
        if (!block) { error = "cannot allocate the request staging block"; return -3; }
        int64_t* s_row0 = (int64_t*)block; int64_t* s_first = s_row0 + nreg; int32_t* s_reg = (int32_t*)(s_first + ncl + 1); int32_t* s_cnt = s_reg + nreg;
        memcpy(s_row0, row0, 8 * (size_t)nreg); memcpy(s_first, cluster_first, 8 * ((size_t)ncl + 1)); memcpy(s_reg, regions, 4 * (size_t)nreg); memcpy(s_cnt, cnt, 4 * (size_t)nreg);
        ensure(d_v_row0, (size_t)nreg); ensure(d_v_first, (size_t)ncl + 1); ensure(d_list, (size_t)nreg); ensure(d_list2, (size_t)nreg); ensure(d_rg_count, 4);
        ensure(d_v_done, (size_t)ncl); ensure(d_v_defer, (size_t)ncl); ensure(d_v_involved, (size_t)ncl);
        be.h2d_staged(d_v_row0.p, s_row0, 8 * (size_t)nreg); be.h2d_staged(d_v_first.p, s_first, 8 * ((size_t)ncl + 1));
        be.h2d_staged(d_list.p, s_reg, 4 * (size_t)nreg); be.h2d_staged(d_list2.p, s_cnt, 4 * (size_t)nreg);
        // candidates with a member outside their region are noted for store_order_check (store_kernels.h: ForeignRead); the counter is
        // zeroed by settle_launch, once per anchor list
        foreign_cap = std::min<size_t>((size_t)1 << 16, ((size_t)32 << 20) / (8 * (size_t)ngen));
        ensure(d_foreign, foreign_cap); ensure(d_foreign_masks, foreign_cap * (size_t)ngen); ensure(d_foreign_count, 2);
        ensure_keep(d_ms_key, (size_t)ms_count, (size_t)ms_key_rows); ms_key_rows = ms_count;
        // what every region was processed with (0: it waits), and the accepted members outside their region of this call (OutsideWriteCheck)
        if (rg_pkey_init > rg_count) rg_pkey_init = 0;
        ensure_keep(d_rg_pkey, cap, (size_t)rg_pkey_init);
        if (rg_pkey_init < rg_count) { be.memset(d_rg_pkey.p + rg_pkey_init, 0, 8 * (size_t)(rg_count - rg_pkey_init)); rg_pkey_init = rg_count; }
        constexpr size_t kOutwCap = 4096; constexpr int64_t kOutwWaves = 256;
        ensure(d_outw, kOutwCap); ensure(d_outw_count, 2);
        const int64_t na = stage_first > 0 ? stage_first : ncl;
        uint64_t head[4] = {(uint64_t)rg_count, 0, 0, 0};       // [0] the region counter, [1] the trouble word, [2] the gate of the second stage, [3] collinear test failed
        if (info_count > 0) {      // the scalars of the candidates just decided come back with the same round trip
            if (info_first < 0 || info_first + info_count > ms_count) { error = "rows outside the MUM store"; return -2; }
            ensure(d_rowinfo, (size_t)info_count);
        }
        std::vector<int32_t> done_h((size_t)ncl, 0);
        // the launches of the call and its one wait.  exact = false: the clusters are expected in reference order in every genome
        // (ClustersDisjoint looks; if not, nothing runs).  exact = true: which clusters meet an earlier one is worked out first
        // (ClusterExtents ... ClusterDefer, for the stage that runs clusters side by side) and those wait
        auto generation = [&](bool exact) {
            be.h2d(d_rg_count.p, head, 32);
            be.memset(d_v_done.p, 0, 4 * (size_t)ncl);
            be.memset(d_outw_count.p, 0, 16);
            const uint8_t* defer = nullptr;
            if (exact) {
                be.memset(d_v_defer.p, 0, (size_t)ncl);
                const int64_t c0 = stage_first > 0 ? na : 0, cn = ncl - c0;      // (the first stage of a two-stage call is one cluster)
                if (cn >= 2) {
                    const size_t words = layout_geometry();
                    const uint64_t* b1 = d_once.p; const uint64_t* b2 = d_twice.p; const int32_t* b3 = d_owner.p;
                    ensure(d_once, words); ensure(d_twice, words); ensure(d_owner, words);
                    if (d_once.p != b1 || d_twice.p != b2) { be.memset(d_once.p, 0, 8 * words); be.memset(d_twice.p, 0, 8 * words); }      // (kept all zero between calls)
                    if (d_owner.p != b3) be.memset(d_owner.p, 0, 4 * words);
                    be.launch_wave("cluster_extents", cn, ClusterExtents{ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, layout_view(d_once.p), d_twice.p, d_owner.p, c0, 1});
                    be.launch_wave("cluster_involved", cn, ClusterInvolved{ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, layout_view(d_twice.p, false), d_owner.p, d_v_involved.p, c0});
                    be.launch_wave("cluster_defer", cn, ClusterDefer{ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, d_lay_off.p, d_lay_bits.p, d_owner.p, d_v_involved.p, d_v_defer.p, c0});
                    be.launch_wave("cluster_extents", cn, ClusterExtents{ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, layout_view(d_once.p), d_twice.p, d_owner.p, c0, 0});
                    // ... and where a candidate's member outside its region reads beside a cluster that may mark there (the same scratch arrays,
                    // wiped in between; `twice` lends its memory to the second owner array)
                    int32_t* owner2 = (int32_t*)d_twice.p;
                    be.launch_wave("reader_mark", cn, ReaderMark{store_view(), d_v_row0.p, d_list2.p, ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, layout_view(d_once.p), d_owner.p, owner2, c0, 1});
                    be.launch_wave("marker_look", cn, MarkerLook{store_view(), d_v_row0.p, d_list2.p, ngen, d_v_first.p, layout_view(d_once.p, false), d_owner.p, owner2, d_v_defer.p, c0});
                    be.launch_wave("reader_look", cn, ReaderLook{store_view(), d_v_row0.p, d_list2.p, ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, d_lay_off.p, d_lay_bits.p, owner2, d_v_defer.p, c0});
                    be.launch_wave("reader_mark", cn, ReaderMark{store_view(), d_v_row0.p, d_list2.p, ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, layout_view(d_once.p), d_owner.p, owner2, c0, 0});
                }
                defer = d_v_defer.p;
                exact_cluster_tests++;
            } else be.launch_wave("clusters_disjoint", ncl - 1, ClustersDisjoint{ngen, d_rg_start.p, d_rg_len.p, d_list.p, d_v_first.p, d_rg_count.p + 3, stage_first, force_unsure ? 1 : 0});
            be.launch_wave("cluster_validate", xcd_grid(na),
                           ClusterValidate{store_view(), layout_view(d_image.p), P, d_rg_start.p, d_rg_len.p, d_rg_info.p, d_rg_count.p, (uint64_t)cap,
                                           d_list.p, d_v_row0.p, d_list2.p, d_v_first.p, q, (uint32_t*)(d_rg_count.p + 1), na, 0, nullptr, d_rg_count.p + 3,
                                           d_foreign.p, d_foreign_count.p, (uint64_t)foreign_cap, d_foreign_masks.p, d_ms_key.p, generation_no, d_recwords.p, defer, d_v_done.p,
                                           d_rg_pkey.p, d_outw.p, d_outw_count.p, (uint64_t)kOutwCap});
            if (stage_first > 0) {
                be.launch("stage_gate", 1, StageGate{d_rg_count.p, (uint64_t)rg_count, force_gate ? 1 : 0});
                be.launch_wave("cluster_validate", xcd_grid(ncl - na),
                               ClusterValidate{store_view(), layout_view(d_image.p), P, d_rg_start.p, d_rg_len.p, d_rg_info.p, d_rg_count.p, (uint64_t)cap,
                                               d_list.p, d_v_row0.p, d_list2.p, d_v_first.p, q, (uint32_t*)(d_rg_count.p + 1), ncl - na, na, d_rg_count.p + 2, d_rg_count.p + 3,
                                               d_foreign.p, d_foreign_count.p, (uint64_t)foreign_cap, d_foreign_masks.p, d_ms_key.p, generation_no + 1, d_recwords.p, defer, d_v_done.p,
                                               d_rg_pkey.p, d_outw.p, d_outw_count.p, (uint64_t)kOutwCap});
            }
            be.launch_wave("outside_write_check", kOutwWaves,
                           OutsideWriteCheck{store_view(), d_rg_start.p, d_rg_len.p, d_rg_info.p, d_rg_pkey.p, d_rg_count.p, (uint64_t)cap, d_list.p, d_outw.p, d_outw_count.p, (uint64_t)kOutwCap,
                                             d_ms_key.p, anchor_table_rows, (uint32_t*)(d_rg_count.p + 1), kOutwWaves});
            if (info_count > 0) be.launch("store_info", info_count, StoreInfoOut{store_view(), info_first, d_rowinfo.p});
            be.mark(nullptr);
            if (info_count > 0) be.d2h_async(info, d_rowinfo.p, sizeof(RowInfo) * (size_t)info_count);
            be.d2h_async(&foreign_seen, d_foreign_count.p, 8);
            be.d2h_async(&outside_writes_call, d_outw_count.p, 8);
            be.d2h_async(done_h.data(), d_v_done.p, 4 * (size_t)ncl);
            be.d2h(head, d_rg_count.p, 32);
        };
        be.mark("validate");
        // a session whose genomes have once held the clusters in another order (an inversion) asks the exact question at once:
        // the collinear test would fail again, at the price of a round trip
        generation(clusters_out_of_order);
        if (head[3] != 0 && !clusters_out_of_order) {
            // some genome does not hold the clusters in reference order: nothing has been validated (ClusterValidate saw the word)
            clusters_out_of_order = true;
            head[0] = (uint64_t)rg_count; head[1] = head[2] = head[3] = 0;
            be.mark("validate");
            generation(true);
        }
        *trouble = (uint32_t)head[1];
        outside_writes += (int64_t)outside_writes_call;
        const bool ran2 = stage_first > 0 && head[2] == 0;
        if (second_stage_ran) *second_stage_ran = ran2 ? 1 : 0;
        for (int64_t cl = 0; cl < ncl; cl++) {
            if (stage_first > 0 && cl >= na && !ran2) { done_h[(size_t)cl] = 0; continue; }
            const int64_t size = cluster_first[cl + 1] - cluster_first[cl];
            if (done_h[(size_t)cl] < size) {
                deferred_regions += size - done_h[(size_t)cl];
                if (!done && !*trouble) *trouble |= done_h[(size_t)cl] == 0 ? 8u : 1u;
            }
        }
        if (done) memcpy(done, done_h.data(), 4 * (size_t)ncl);
        if (head[0] > cap) { error = "region store overflow"; return -4; }
        const int64_t before = rg_count;
        rg_count = (int64_t)head[0];
        take_regions(before, rg_count, kids, kid_ids);
        collect_timing_more();
        return 0;
    }
    // chain()'s test of MUM cur[i] against the open chain's last MUM back[i] (setFinalClusters :2596-2700): verdict 0 = every
    // genome's gap inside [0, d] (min_gap / max_gap for the caller's ratio test), 1 = the chain closes, 2 = a reverse member: from the rows
    int store_judge(const int32_t* cur, const int32_t* back, int64_t n, int32_t d, int32_t* min_gap, int32_t* max_gap, uint8_t* verdict) {
        if (!resident) { error = "the session is not in resident mode"; return -2; }
        for (int64_t i = 0; i < n; i++) if (cur[i] < 0 || cur[i] >= ms_count || back[i] < 0 || back[i] >= ms_count) { error = "rows outside the MUM store"; return -2; }
        if (n == 0) return 0;
        begin_store_call();
        ensure(d_list, (size_t)n); ensure(d_list2, (size_t)n); ensure(d_j_min, (size_t)n); ensure(d_j_max, (size_t)n); ensure(d_small8, (size_t)n);
        int32_t* block = (int32_t*)be.staging(8 * (size_t)n);
        if (!block) { error = "cannot allocate the request staging block"; return -3; }
        memcpy(block, cur, 4 * (size_t)n); memcpy(block + n, back, 4 * (size_t)n);
        be.h2d_staged(d_list.p, block, 4 * (size_t)n); be.h2d_staged(d_list2.p, block + n, 4 * (size_t)n);
        be.mark("judge");
        be.launch_wave("judge_pairs", n, JudgePairs{store_view(), d_list.p, d_list2.p, d, d_j_min.p, d_j_max.p, d_small8.p});
        be.mark(nullptr);
        be.d2h_async(min_gap, d_j_min.p, 4 * (size_t)n); be.d2h_async(max_gap, d_j_max.p, 4 * (size_t)n);
        be.d2h(verdict, d_small8.p, (size_t)n);
        collect_timing_more();
        return 0;
    }
    // the listed MUMs leave the layout (filterRandom1 :415-418, filterRandomClustersSimple1 :460-466)
    int store_unmark(const int32_t* rows, int64_t n) {
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        for (int64_t i = 0; i < n; i++) if (rows[i] < 0 || rows[i] >= ms_count) { error = "rows outside the MUM store"; return -2; }
        if (n == 0) return 0;
        ensure(d_list, (size_t)n);
        be.h2d(d_list.p, rows, 4 * (size_t)n);
        be.launch("store_unmark", n * ngen, StoreUnmark{store_view(), layout_view(d_image.p), d_list.p});
        return 0;
    }
    // setInterClusterRegions (:2389-2460) for consecutive LCBs given by (last MUM of one, first MUM of the next); rows of the
    // fillers that are made, [n_made][ngen] each, in pair order
    int store_fill(const int32_t* last_of, const int32_t* first_of_next, int64_t n, uint8_t* add, std::vector<int64_t>* starts, std::vector<int64_t>* ends) {
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        for (int64_t i = 0; i < n; i++) if (last_of[i] < 0 || last_of[i] >= ms_count || first_of_next[i] < 0 || first_of_next[i] >= ms_count) { error = "rows outside the MUM store"; return -2; }
        starts->clear(); ends->clear();
        if (n == 0) return 0;
        begin_store_call();
        const size_t ngz = (size_t)ngen;
        ensure(d_list, (size_t)n); ensure(d_list2, (size_t)n); ensure(d_small8, (size_t)n); ensure(d_f_start, (size_t)n * ngz); ensure(d_f_end, (size_t)n * ngz);
        be.h2d(d_list.p, last_of, 4 * (size_t)n); be.h2d(d_list2.p, first_of_next, 4 * (size_t)n);
        be.mark("fill");
        be.launch_wave("fill_between", n, FillBetween{store_view(), layout_view(d_image.p, false), P, d_list.p, d_list2.p, d_small8.p, d_f_start.p, d_f_end.p});
        be.mark(nullptr);
        be.d2h(add, d_small8.p, (size_t)n);
        std::vector<int32_t> which;
        for (int64_t i = 0; i < n; i++) if (add[i] == 1) which.push_back((int32_t)i);
        if (!which.empty()) {      // the rows of the fillers, packed on the device: one copy
            const size_t m = which.size();
            ensure(d_list, m); ensure(d_f_pack, 2 * m * ngz);
            be.h2d(d_list.p, which.data(), 4 * m);
            be.launch("fill_gather", (int64_t)m * ngen, FillGather{d_list.p, ngen, d_f_start.p, d_f_end.p, d_f_pack.p});
            std::vector<int64_t> pack(2 * m * ngz);
            be.d2h(pack.data(), d_f_pack.p, 8 * pack.size());
            starts->resize(m * ngz); ends->resize(m * ngz);
            for (size_t k = 0; k < m; k++) {
                memcpy(starts->data() + k * ngz, pack.data() + (2 * k) * ngz, 8 * ngz);
                memcpy(ends->data() + k * ngz, pack.data() + (2 * k + 1) * ngz, 8 * ngz);
            }
        }
        collect_timing_more();
        return 0;
    }
    // Phases C-D where their list logic is order-free (store_kernels.h, "phases C-D on the store"): the accepted rows sorted by
    // reference start, chained, the short LCBs dissolved, chained again, the fillers counted -- one queue of launches, no round
    // trip.  Two halves: begin() queues the work and the copies, end() waits for them; the caller may work in between.
    // expect: the number of accepted rows (the caller's MUM list).
    struct ChainInfo { int64_t n_in, lcbs_first, lcbs_dissolved, mums_dissolved, n_mums, n_lcbs, n_fillers; uint64_t trouble; };
    // After the last generation: every candidate that read marks outside its region, decided again the way the reference's order
    // had them (store_kernels.h: ForeignRead and what follows it).  *trouble != 0: the order would show.  store_chain_begin runs
    // the same check first (bit kChainOrder of pm_chain_info.trouble); store_order_check is for a caller that does not queue phases C-D.
    void order_check_launch(uint32_t* word, uint32_t bit) {
        const int64_t cnt = (int64_t)std::min<uint64_t>(foreign_seen, (uint64_t)foreign_cap);
        const int64_t rows = ms_count - layout_rows;
        if (cnt <= 0 || rows <= 0) return;      // (nothing was noted)
        ensure(d_hits, kHitCap); ensure(d_hit_count, 2); ensure(d_hit_owned, kHitCap * (size_t)ngen); ensure(d_hit_present, kHitCap * (size_t)ngen);
        be.memset(d_hit_count.p, 0, 16);
        const Store S = store_view();
        const Layout L = layout_view(d_image.p, false);
        be.launch_wave("foreign_bound", cnt, ForeignBound{S, L, P, d_rg_start.p, d_rg_len.p, d_foreign.p, cnt, d_foreign_masks.p, d_recwords.p, d_hits.p, d_hit_count.p, (uint64_t)kHitCap,
                                                         d_hit_owned.p, d_hit_present.p, word, bit});
        const int64_t chunks = (rows + kScanRows - 1) / kScanRows, waves = 8192;
        be.launch_wave("foreign_scan", waves, ForeignScan{S, d_rg_start.p, d_rg_len.p, d_ms_key.p, d_foreign.p, d_hits.p, d_hit_count.p, (uint64_t)kHitCap, d_hit_owned.p, d_hit_present.p,
                                                         layout_rows, ms_count, chunks, waves});
        be.launch_wave("foreign_decide_hits", (int64_t)kHitCap, ForeignDecideHits{S, L, P, d_rg_start.p, d_rg_len.p, d_foreign.p, d_foreign_masks.p, d_hits.p, d_hit_count.p, (uint64_t)kHitCap,
                                                                                 d_hit_owned.p, d_hit_present.p, word, bit});
        if (order_debug) {      // (pm_session_tune "order_debug": waits for the queue and says what the check had to do)
            uint64_t nh = 0;
            be.d2h(&nh, d_hit_count.p, 8);
            fprintf(stderr, "[order check] %ld noted candidates, %ld left to the scan (launched for %ld), %ld recursion rows\n", (long)cnt, (long)nh, (long)kHitCap, (long)rows);
        }
    }
    int store_order_check(uint32_t* trouble) {
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        *trouble = 0;
        begin_store_call();
        ensure(d_rg_count, 4);
        be.mark("validate");
        be.memset(d_rg_count.p + 1, 0, 8);
        order_check_launch((uint32_t*)(d_rg_count.p + 1), 1u);
        be.mark(nullptr);
        uint64_t word = 0;
        be.d2h(&word, d_rg_count.p + 1, 8);
        *trouble = (uint32_t)word;
        collect_timing_more();
        return 0;
    }
    int store_chain_begin(int64_t expect, int32_t d, float diag_diff, int64_t c) {
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        if (expect <= 0 || expect > ms_count) { error = "bad MUM count"; return -2; }
        if (chain_event) { be.event_wait(chain_event); be.event_release(chain_event); chain_event = nullptr; }
        begin_store_call();
        const size_t rows = (size_t)ms_count, cap = (size_t)expect;
        ensure(d_ch_flag, rows + 1); ensure(d_ch_pos, rows + 1);
        ensure(d_ch_key, cap); ensure(d_ch_val, cap); ensure(d_ch_skey, cap); ensure(d_ch_srow, cap); ensure(d_ch_key2, cap); ensure(d_ch_row2, cap);
        ensure(d_ch_verdict, cap); ensure(d_ch_head, cap + 1); ensure(d_ch_hpos, cap + 1); ensure(d_ch_survive, cap + 1); ensure(d_ch_spos, cap + 1);
        ensure(d_ch_lcblen, cap); ensure(d_ch_hdr, kChWords); ensure(d_ch_outrow, cap); ensure(d_ch_outhead, cap);
        const size_t host_bytes = 8 * (size_t)kChWords + 4 * cap + cap;
        if (host_bytes > chain_host_cap) {
            if (chain_host) be.pinned_free(chain_host);
            chain_host_cap = 0;
            chain_host = (uint8_t*)be.pinned_alloc(host_bytes + host_bytes / 4 + 64);
            if (!chain_host) { error = "cannot allocate the chain result block"; return -3; }
            chain_host_cap = host_bytes + host_bytes / 4 + 64;
        }
        chain_cap = (int64_t)cap;
        const Store S = store_view();
        be.mark("chain");
        {
            const ClearJob jobs[] = {{d_ch_hdr.p, 8 * (size_t)kChWords, 0}, {d_ch_lcblen.p, 8 * cap, 0}};
            be.clear_many(jobs, 2);
        }
        order_check_launch((uint32_t*)(d_ch_hdr.p + kChTrouble), (uint32_t)kChainOrder);      // (before anything leaves the layout)
        be.launch("chain_flag", (int64_t)rows + 1, ChainFlag{S, (int64_t)rows, d_ch_flag.p});
        be.exclusive_scan(d_ch_flag.p, d_ch_pos.p, rows + 1);
        be.launch("chain_keys", (int64_t)rows, ChainKeys{S, d_ch_pos.p, d_ch_key.p, d_ch_val.p, (int64_t)cap});
        be.sort_pairs(d_ch_key.p, d_ch_skey.p, d_ch_val.p, d_ch_srow.p, cap, bits_for((uint64_t)glen_h[0]));
        const int64_t* n1 = d_ch_pos.p + rows;
        int64_t* hdr = d_ch_hdr.p;
        uint64_t* trouble = (uint64_t*)(hdr + kChTrouble);
        // first pass: verdicts, chains, their lengths, the dissolved ones out of the layout and out of the list
        be.launch_wave("chain_judge", (int64_t)cap, ChainJudge{S, d_ch_skey.p, d_ch_srow.p, n1, d, diag_diff, d_ch_verdict.p, trouble, force_chain_tie ? 1 : 0});
        be.launch("chain_judge_reverse", (int64_t)cap, ChainJudgeReverse{S, d_ch_srow.p, n1, d, diag_diff, d_ch_verdict.p});
        be.launch("chain_heads", (int64_t)cap + 1, ChainHeads{d_ch_verdict.p, n1, d_ch_head.p});
        be.exclusive_scan(d_ch_head.p, d_ch_hpos.p, cap + 1);
        be.launch("chain_lcb_sum", (int64_t)cap, ChainLcbSum{S, d_ch_srow.p, n1, d_ch_hpos.p, d_ch_lcblen.p});
        be.launch("chain_dissolve", (int64_t)cap + 1, ChainDissolve{n1, d_ch_hpos.p, d_ch_head.p, d_ch_lcblen.p, c, d_ch_survive.p, hdr});
        be.launch_wave("chain_unmark", (int64_t)cap, ChainUnmark{S, layout_view(d_image.p), d_ch_srow.p, n1, d_ch_survive.p});
        be.exclusive_scan(d_ch_survive.p, d_ch_spos.p, cap + 1);
        be.launch("chain_compact", (int64_t)cap, ChainCompact{n1, d_ch_survive.p, d_ch_spos.p, d_ch_skey.p, d_ch_srow.p, d_ch_key2.p, d_ch_row2.p, hdr});
        // second pass (the reference chains again after dissolving, :3261-3268), then the fillers between the final LCBs
        const int64_t* n2 = hdr + kChN2;
        be.launch_wave("chain_judge", (int64_t)cap, ChainJudge{S, d_ch_key2.p, d_ch_row2.p, n2, d, diag_diff, d_ch_verdict.p, trouble, 0});
        be.launch("chain_judge_reverse", (int64_t)cap, ChainJudgeReverse{S, d_ch_row2.p, n2, d, diag_diff, d_ch_verdict.p});
        be.launch("chain_heads", (int64_t)cap + 1, ChainHeads{d_ch_verdict.p, n2, d_ch_head.p});
        be.exclusive_scan(d_ch_head.p, d_ch_hpos.p, cap + 1);
        if (cap > 1) be.launch_wave("chain_fill", (int64_t)cap - 1, ChainFill{S, layout_view(d_image.p, false), P, d_ch_row2.p, n2, d_ch_head.p, hdr});
        be.launch("chain_out", (int64_t)cap, ChainOut{d_ch_row2.p, n2, d_ch_head.p, d_ch_hpos.p, d_ch_outrow.p, d_ch_outhead.p, hdr});
        be.mark(nullptr);
        be.d2h_async_pinned(chain_host, hdr, 8 * (size_t)kChWords);
        be.d2h_async_pinned(chain_host + 8 * (size_t)kChWords, d_ch_outrow.p, 4 * cap);
        be.d2h_async_pinned(chain_host + 8 * (size_t)kChWords + 4 * cap, d_ch_outhead.p, cap);
        chain_event = be.event_record();
        chain_pending = true;
        return 0;
    }
    int store_chain_end(ChainInfo* info, const int32_t** rows, const uint8_t** heads) {
        if (!chain_pending) { error = "no chain call in flight"; return -2; }
        chain_pending = false;
        if (chain_event) { be.event_wait(chain_event); be.event_release(chain_event); chain_event = nullptr; }
        else be.sync();
        const int64_t* h = (const int64_t*)chain_host;
        *info = ChainInfo{h[kChN1], h[kChLcb1], h[kChLcbDissolved], h[kChMumDissolved], h[kChN2], h[kChLcb2], h[kChFill], (uint64_t)h[kChTrouble]};
        *rows = (const int32_t*)(chain_host + 8 * (size_t)kChWords);
        *heads = chain_host + 8 * (size_t)kChWords + 4 * (size_t)chain_cap;
        collect_timing_more();
        if (info->n_in != chain_cap) { error = "the MUM list of the caller and the accepted rows of the store differ"; return -2; }
        return 0;
    }
    // rows of the store for the host (the XMFA writer after phase D; a caller that falls back to the host route): start per genome
    // with the trim applied (raw: as the search delivered them) + strand byte.  rows == nullptr: rows [first, first + n)
    int store_rows(const int32_t* rows, int64_t first, int64_t n, bool raw, int32_t* out_start, uint8_t* out_strand) {
        if (!resident) { error = "the session is not in resident mode"; return -2; }
        if (n == 0) return 0;
        const size_t ngz = (size_t)ngen;
        if (!rows) {
            if (first < 0 || first + n > ms_count) { error = "rows outside the MUM store"; return -2; }
            if (raw) {      // contiguous and untouched: straight copies
                be.d2h_async(out_start, d_anchor_start.p + (size_t)first * ngz, 4 * (size_t)n * ngz);
                be.d2h(out_strand, d_ms_strand.p + (size_t)first * ngz, (size_t)n * ngz);
                return 0;
            }
        } else for (int64_t i = 0; i < n; i++) if (rows[i] < 0 || rows[i] >= ms_count) { error = "rows outside the MUM store"; return -2; }
        // in pieces: the gathered copy of a 60 000-row list is 60 MB of device memory otherwise
        const int64_t piece = std::max<int64_t>(1, (int64_t)((32u << 20) / (5 * ngz)));
        ensure(d_list, (size_t)std::min(n, piece)); ensure(d_o_start, (size_t)std::min(n, piece) * ngz); ensure(d_o_strand, (size_t)std::min(n, piece) * ngz);
        std::vector<int32_t> seq;
        for (int64_t at = 0; at < n; at += piece) {
            const int64_t m = std::min(piece, n - at);
            const int32_t* src = rows ? rows + at : nullptr;
            if (!src) { seq.resize((size_t)m); for (int64_t i = 0; i < m; i++) seq[(size_t)i] = (int32_t)(first + at + i); src = seq.data(); }
            be.h2d(d_list.p, src, 4 * (size_t)m);
            be.launch("store_rows", m * ngen, StoreRowsOut{store_view(), d_list.p, d_o_start.p, d_o_strand.p, raw ? 1 : 0});
            be.d2h_async(out_start + (size_t)at * ngz, d_o_start.p, 4 * (size_t)m * ngz);
            be.d2h(out_strand + (size_t)at * ngz, d_o_strand.p, (size_t)m * ngz);
        }
        return 0;
    }
    // the layout as the host keeps it (write_unaligned reads it): one block, words of genome j at off[j]
    int store_layout(uint64_t* out, int64_t words) {
        if (!resident || layout_rows < 0) { error = "the anchor list has not been settled"; return -2; }
        if ((size_t)words != layout_geometry()) { error = "layout size does not match"; return -2; }
        be.d2h(out, d_image.p, 8 * (size_t)words);
        return 0;
    }
    std::vector<int64_t> lay_off_h;
    void collect_timing_more() { for (const PhaseTime& t : be.collect()) timing.push_back(t); }

    // small host-side all-gather (calcmumi's per-genome results): through device staging when the collectives are RCCL
    int allgather_host(const void* send, int64_t bytes, void* recv) {
        if (!coll.device) return coll.allgather(coll.ctx, send, bytes, recv);
        ensure(d_hsend, (size_t)bytes); ensure(d_hrecv, (size_t)bytes * (size_t)coll.world);
        be.h2d(d_hsend.p, send, (size_t)bytes);
        if (be.allgather_dev(d_hsend.p, bytes, d_hrecv.p)) return 1;
        be.d2h(recv, d_hrecv.p, (size_t)bytes * (size_t)coll.world);
        return 0;
    }

    // parity hook output (valid after run(..., want_events = true))
    std::vector<uint64_t> ev_key_h, ev_val_h;
    std::vector<int32_t> rep_h;
    int ev_lbits = 0;
    // tunables of a session (pm_session_tune): per-thread work budget of the index walks; shortest one-region candidate list
    // that gets the device overlap test and stays resident as the anchor table (the host's threshold for its long-list routes)
    int64_t work_budget = 1 << 22;
    int64_t dirty_min = 4096;
    int64_t flagged_div = 8;      // store_settle: above one flagged row in flagged_div the tangled rows are counted before the list is taken (1: never declined)
    int64_t tangled_max = 1 << 17;   // ... and the list declined (PM_EAGAIN) with more tangled rows than this
    bool fast_tail = true;        // a call whose rows stay on the device does not wait for its candidate and accepted counts (run_once); false: it does (tests, before/after measurements)
    int64_t tail_repeats = 0;     // tails repeated the exact way (a capacity too small, an anchor list that turned out short)
    bool tangle_rounds = true;    // tangled rows settled in rounds (TangleOwner / TangleSettle) before the one wavefront that takes what is left; false: that wavefront takes them all (tests)
    bool group_small = true;      // the events of a recursion batch's small regions once per distinct piece (GroupedPairEvents)
    int64_t last_grouped = 0;
    bool force_atomic_marks = false;      // (tests) store_settle marks with atomic ORs although the list is in order
    int filter_factor = 8;                // presence-filter bits per reference position, before rounding up to a power of two
    int slot_factor = 2;                  // index slots per reference position, before rounding up to a power of two (a measurement switch)
    bool bucket_sort = true;              // the events put in order by (pair, 256-position block) buckets (EventBucketCount ... CoarseFromBuckets); false: gathered and radix-sorted
    bool master_seg = true;               // Master.EP from the genomes' segments (MasterEPSeg); false: every lane against every staged event (MasterEP)
    bool force_gate = false;              // (tests) the second stage of a two-stage store_validate never runs
    bool force_chain_tie = false;         // (tests) store_chain_begin reports two MUMs with one reference start
    bool force_unsure = false;            // (tests) store_validate's collinear test of the clusters reports failure: the exact test decides
    bool order_debug = false;             // the order check prints its counts (noted candidates, candidates left to the scan) to stderr
    int64_t outside_writes = 0;           // accepted reverse-strand members outside their region that OutsideWriteCheck looked at
    int64_t exact_cluster_tests = 0;      // generations validated with the exact test of their clusters (ClusterExtents ... ClusterDefer)
    int64_t deferred_regions = 0;         // regions a validation call left on the caller's work list (their cluster met an earlier one, or a child sorted first)
    bool clusters_out_of_order = false;   // a generation of this session failed the collinear test of its clusters: the exact test is asked at once from then on
    bool phase_timing = true;             // HIP events around the phases of a call (pm_last_timing); off: the marks cost nothing
    bool tune(const std::string& key, int64_t value) {
        if (key == "flagged_div" && value >= 1) { flagged_div = value; return true; }
        if (key == "tangled_max" && value >= 0) { tangled_max = value; return true; }
        if (key == "tangle_rounds") { tangle_rounds = value != 0; return true; }
        if (key == "fast_tail") { fast_tail = value != 0; return true; }
        if (key == "atomic_marks") { force_atomic_marks = value != 0; return true; }
        if (key == "master_seg") { master_seg = value != 0; return true; }
        if (key == "bucket_sort") { bucket_sort = value != 0; return true; }
        if (key == "filter_factor") { filter_factor = value < 1 ? 1 : (int)value; return true; }
        if (key == "slot_factor") { slot_factor = value < 1 ? 1 : (int)value; return true; }
        if (key == "stage_gate") { force_gate = value != 0; return true; }
        if (key == "chain_tie") { force_chain_tie = value != 0; return true; }
        if (key == "cluster_unsure") { force_unsure = value != 0; return true; }
        if (key == "order_debug") { order_debug = value != 0; return true; }
        if (key == "timing") { phase_timing = value != 0; be.timing_on = phase_timing; return true; }
        if (key == "group_small") { group_small = value != 0; return true; }
        if (key == "work_budget" && value > 0) { work_budget = value; return true; }
        if (key == "dirty_min" && value >= 0) { dirty_min = value; return true; }
        return false;
    }

    void release() {
        if (chain_event) { be.event_wait(chain_event); be.event_release(chain_event); chain_event = nullptr; }
        for (BufBase* b : all_bufs) { if (b->raw) be.free(b->raw); b->raw = nullptr; b->cap = 0; }
        if (chain_host) be.pinned_free(chain_host);
        chain_host = nullptr; chain_host_cap = 0; chain_pending = false;
        if (blk) be.free(blk);
        if (d_goff) be.free(d_goff);
        if (d_glen) be.free(d_glen);
        blk = nullptr; d_goff = nullptr; d_glen = nullptr;
    }

    // device allocation failure inside run(): mapped to PM_ENOMEM at the C ABI (abi_glue.h), never abort()
    struct DeviceOutOfMemory { size_t bytes; };

private:
    B& be;
    // every device buffer of the engine registers itself here, so release() cannot miss one
    struct BufBase { void* raw = nullptr; size_t cap = 0; };
    std::vector<BufBase*> all_bufs;
    template <class T> struct Buf : BufBase {
        T* p = nullptr;
    };
    template <class T> void ensure(Buf<T>& b, size_t n) {
        if (n <= b.cap && b.p) return;
        if (std::find(all_bufs.begin(), all_bufs.end(), (BufBase*)&b) == all_bufs.end()) all_bufs.push_back(&b);
        if (b.raw) be.free(b.raw);
        b.raw = nullptr; b.p = nullptr; b.cap = 0;
        const size_t want = n + n / 4 + 16;
        b.raw = be.alloc(want * sizeof(T));
        if (!b.raw) throw DeviceOutOfMemory{want * sizeof(T)};
        b.p = (T*)b.raw; b.cap = want;
    }
    // the same, keeping the first `keep` elements (the stores grow while the run appends to them)
    template <class T> void ensure_keep(Buf<T>& b, size_t n, size_t keep) {
        if (n <= b.cap && b.p) return;
        if (std::find(all_bufs.begin(), all_bufs.end(), (BufBase*)&b) == all_bufs.end()) all_bufs.push_back(&b);
        const size_t want = n + n / 2 + 16;
        void* raw = be.alloc(want * sizeof(T));
        if (!raw) throw DeviceOutOfMemory{want * sizeof(T)};
        if (b.p && keep) { be.d2d(raw, b.p, keep * sizeof(T)); be.sync(); }
        if (b.raw) be.free(b.raw);
        b.raw = raw; b.p = (T*)raw; b.cap = want;
    }
    void collect_timing() { timing = be.collect(); }

    SeqBlock* blk = nullptr; int64_t* d_goff = nullptr; int64_t* d_glen = nullptr;
    int64_t total_words = 0;
    Packed P{};
    size_t ev_cap_hint = 0, cand_cap_hint = 0, rest_cap_hint[2] = {0, 0}, grp_cap_hint = 0;
    // what call number `i` of the last step needed (i = 0: the anchor call; then the searches of store regions in their order): the
    // capacities call i of this step launches over instead of waiting for its counts
    struct CallHint { int64_t units = 0, ncand = 0, nok = 0; };
    std::vector<CallHint> call_hints;
    int64_t call_ordinal = 0;      // candidates / accepted rows of the last call of a shape ([1]: one region = an anchor call): capacities of the next one's tail
    Buf<uint8_t> d_gflag; Buf<int64_t> d_glo;
    Buf<RegionInfo> d_R; Buf<int64_t> d_starts, d_lens, d_posbase;
    Buf<uint64_t> d_slots; Buf<uint32_t> d_filter, d_repeated; Buf<int32_t> d_next, d_rep, d_run, d_epm;
    Buf<int64_t> d_ucount, d_uoff; Buf<UnitRec> d_units;
    Buf<uint64_t> d_counter, d_evkey, d_evval, d_evkey2, d_evval2, d_evkey3, d_evval3; Buf<int64_t> d_sliceoff;
    Buf<int64_t> d_lo, d_cov; Buf<EventAtK> d_state; Buf<PairState> d_wsummary; Buf<int32_t> d_emax;
    Buf<uint64_t> d_cand; Buf<GenomeAtK> d_at; Buf<uint8_t> d_ok; Buf<int32_t> d_ok_k, d_ok_lon, d_osp; Buf<uint8_t> d_ofwd;
    Buf<uint64_t> d_wmask; Buf<int64_t> d_wcount, d_woff;
    Buf<int64_t> d_bcount, d_bbegin, d_pbase;      // the event buckets: counts, where each begins, every pair's first
    Buf<int64_t> d_okcnt, d_okpos, d_cbase; Buf<int32_t> d_coarse; Buf<int32_t> d_creg, d_ck, d_clon, d_csp; Buf<uint8_t> d_cfwd;
    Buf<uint32_t> d_cflags, d_dirty; Buf<int32_t> d_bmax, d_bmin;
    Buf<GenomeAtK> d_xsend, d_xrecv; Buf<uint8_t> d_hsend, d_hrecv;
    Buf<RestItem> d_rest; Buf<uint64_t> d_qcount;
    Buf<int32_t> d_anchor_start, d_anchor_lon; Buf<uint32_t> d_anchor_flags;      // the MUM store's rows as the searches delivered them
    int64_t table_counter = 0;
    Buf<uint64_t> d_image;
    Buf<uint8_t> d_ms_strand, d_ms_state, d_small8, d_o_strand; Buf<int32_t> d_ms_shift, d_ms_len, d_list, d_list2, d_j_min, d_j_max, d_o_start;
    Buf<int64_t> d_rg_start, d_rg_len, d_lay_off, d_lay_bits, d_v_row0, d_v_first, d_f_start, d_f_end, d_f_pack; Buf<RegInfo> d_rg_info; Buf<uint64_t> d_rg_count, d_once, d_twice;
    Buf<RowInfo> d_rowinfo; Buf<uint64_t> d_alg;
    Buf<int64_t> d_sd_cnt, d_sd_off; Buf<uint8_t> d_sd_keep;      // store_settle_seeds
    Buf<uint64_t> d_t_rem; Buf<uint8_t> d_t_done;      // settle_launch: tangled rows left, settled flags per flagged row
    Buf<int32_t> d_v_done, d_owner; Buf<uint8_t> d_v_defer, d_v_involved;      // store_validate: regions processed per cluster; the exact test of the clusters
    Buf<uint64_t> d_t_left; uint64_t tangle_left[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int tangle_ran = 0;      // settle_launch: what each round of the tangled rows left, last step
    Buf<int64_t> d_rg_pkey; Buf<OutsideWrite> d_outw; Buf<uint64_t> d_outw_count; int64_t rg_pkey_init = 0; uint64_t outside_writes_call = 0;      // store_validate: OutsideWriteCheck
    Buf<ForeignRead> d_foreign; Buf<uint64_t> d_foreign_count, d_foreign_masks; Buf<int64_t> d_ms_key;       // store_validate / store_order_check: candidates with a member outside their region
    size_t foreign_cap = 0; int64_t ms_key_rows = 0; uint64_t foreign_seen = 0;      // foreign_seen: the device's counter as of the last validation call
    static constexpr size_t kHitCap = 256;      // noted candidates that the bounds of ForeignBound do not decide
    Buf<uint64_t> d_hit_count, d_hit_owned, d_hit_present; Buf<int32_t> d_hits; Buf<uint8_t> d_recwords;      // d_recwords: one byte per image word, set where the recursion has marked
    // store_chain_begin / _end
    Buf<int64_t> d_ch_flag, d_ch_pos, d_ch_head, d_ch_hpos, d_ch_survive, d_ch_spos, d_ch_hdr;
    Buf<uint64_t> d_ch_key, d_ch_val, d_ch_skey, d_ch_srow, d_ch_key2, d_ch_row2, d_ch_lcblen;
    Buf<uint8_t> d_ch_verdict, d_ch_outhead; Buf<int32_t> d_ch_outrow;
    uint8_t* chain_host = nullptr; size_t chain_host_cap = 0; void* chain_event = nullptr; bool chain_pending = false; int64_t chain_cap = 0;
    size_t lay_words = 0, rg_cap_hint = 0;
    int64_t layout_rows = -1;       // >= 0: the image holds the layout of the current anchor table (store_settle ran)
};

}  // namespace pm
