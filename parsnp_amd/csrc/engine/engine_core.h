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

struct BatchResult {
    int64_t nregions = 0, total = 0;
    int nq = 0;
    std::vector<int64_t> off;
    HostPool::Block kb, lonb, spb, fwdb;     // int32 k[total], int32 lon[total], int32 sp[total*nq], uint8 fwd[total*nq]
    // a session in MUM-row mode (pm_session_rows) fills these instead of sp / fwd:
    HostPool::Block startb, strandb, flagsb; // int32 start[total*(nq+1)], uint8 strand[total*(nq+1)], uint32 flags[total]
    bool rows = false, dirty_known = false;  // dirty_known: the kRowDirty bits were computed (one-region batch with a long list)
    int64_t table_id = 0;                    // != 0: the rows of this result stay on the device as the session's anchor table (run_gaps)
    std::vector<GapRef> spec_refs;           // run_spec: which gap of the anchor table each region of this result is ...
    std::vector<int32_t> spec_minsize;       // ... and the minimum length it was searched with
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
    int64_t last_events = 0, last_candidates = 0, last_rest = 0;

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
        auto resident = [&](int g) { return g == 0 || (g >= g_first && g < g_last); };
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
    // Requests given as gaps of the anchor table (kernels.h: GapRef / ExpandGaps) + explicit rows for the rest.
    struct GapBatch {
        int64_t table_id = 0;
        const GapRef* gaps = nullptr;                         // [nreg]
        const int64_t* ref_start = nullptr; const int64_t* ref_len = nullptr;   // [nreg]: the reference column of every region (sizes the index)
        int64_t n_explicit = 0; const int64_t* ex_starts = nullptr; const int64_t* ex_lens = nullptr;   // [n_explicit][ngen]
    };
    int run(int64_t nreg, const int64_t* starts, const int64_t* lens, const int32_t* minsize, BatchResult* out, bool want_events = false,
            bool mumi = false, const GapBatch* gb = nullptr) {
        budget_exceeded = false;
        if (gb && (gb->table_id == 0 || gb->table_id != anchor_table_id)) { error = "the anchor table of these gap requests is no longer resident"; return -2; }
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
    int64_t anchor_table_id = 0, anchor_table_rows = 0;      // the resident anchor table: rows of the last one-region call in row mode
    // The seed regions of the resident anchor table worked out on the device (GapSeeds) and searched in one batch: what the
    // host will ask for once it has validated the anchors, computed beside that validation (include/parsnp_mum.h:
    // pm_multi_mum_batch_spec).  out->spec_refs says which gap each region of the result is.
    int run_spec(int64_t table_id, int32_t q, int64_t ref_len_limit, const int32_t* minsize_by_length, int64_t table_len, BatchResult* out) {
        if (table_id == 0 || table_id != anchor_table_id) { error = "the anchor table of this speculation is no longer resident"; return -2; }
        if (!minsize_by_length || table_len < 1) { error = "bad minimum-length table"; return -2; }
        const int64_t rows = anchor_table_rows;
        const size_t cap = 2 * (size_t)(rows + 1);
        ensure(d_spec, cap); ensure(d_speccount, 1); ensure(d_mintable, (size_t)table_len);
        be.memset(d_speccount.p, 0, 8);
        be.h2d(d_mintable.p, minsize_by_length, 4 * (size_t)table_len);
        be.launch_wave("gap_seeds", rows + 1, GapSeeds{d_anchor_start.p, d_anchor_lon.p, d_anchor_flags.p, rows, ngen, d_glen, q, ref_len_limit, d_mintable.p, table_len,
                                                       d_spec.p, d_speccount.p, (uint64_t)cap});
        uint64_t n = 0;
        be.d2h(&n, d_speccount.p, 8);
        if (n > cap) n = cap;
        std::vector<SpecRegion> sr((size_t)n);
        if (n) be.d2h(sr.data(), d_spec.p, sizeof(SpecRegion) * (size_t)n);
        // (the wavefronts append in any order: sorted, so that a run is reproducible event for event)
        std::sort(sr.begin(), sr.end(), [](const SpecRegion& x, const SpecRegion& y) {
            if (x.ref.next != y.ref.next) return (uint32_t)x.ref.next < (uint32_t)y.ref.next;      // (-1 = the end: last)
            if (x.ref.prev != y.ref.prev) return x.ref.prev < y.ref.prev;
            return x.ref.side < y.ref.side; });
        std::vector<GapRef> refs((size_t)n); std::vector<int64_t> rs((size_t)n), rl((size_t)n); std::vector<int32_t> ms((size_t)n);
        for (size_t i = 0; i < (size_t)n; i++) { refs[i] = sr[i].ref; rs[i] = sr[i].ref_start; rl[i] = sr[i].ref_len; ms[i] = sr[i].minsize; }
        GapBatch gb;
        gb.table_id = table_id; gb.gaps = refs.data(); gb.ref_start = rs.data(); gb.ref_len = rl.data();
        const int rc = run((int64_t)n, nullptr, nullptr, ms.data(), out, false, false, &gb);
        out->spec_refs.swap(refs); out->spec_minsize.swap(ms);
        return rc;
    }
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
        // (gap requests: the rows part holds only the explicit rows, followed by the GapRef table)
        const size_t nrow = (size_t)nreg * (size_t)ngen;
        const size_t regz = (size_t)nreg;
        const size_t bytes_R = sizeof(RegionInfo) * regz, bytes_pre = 8 * (regz + 1);
        const size_t nrow_staged = gb ? (size_t)gb->n_explicit * (size_t)ngen : nrow;
        uint8_t* block = (uint8_t*)be.staging(bytes_R + 2 * bytes_pre + 16 * nrow_staged + (gb ? sizeof(GapRef) * regz : 0) + 64);
        if (!block) { error = "cannot allocate the request staging block"; return -3; }
        RegionInfo* R = (RegionInfo*)block;
        int64_t* posbase = (int64_t*)(block + bytes_R);
        int64_t* cbase = posbase + regz + 1;
        int64_t* stage = cbase + regz + 1;
        std::vector<size_t> guess_r(regz, 0);
        std::vector<int64_t> units_r(regz, 0);
        if (gb) {
            if (gb->n_explicit < 0 || (gb->n_explicit > 0 && (!gb->ex_starts || !gb->ex_lens))) { error = "bad explicit rows"; return -2; }
            for (int64_t r = 0; r < nreg; r++) {
                const GapRef& g = gb->gaps[r];
                if (g.explicit_row >= gb->n_explicit || (g.explicit_row < 0 && (g.prev >= anchor_table_rows || g.next >= anchor_table_rows || (g.prev < 0 && g.next < 0) || g.side < 0 || g.side > 1)))
                    { error = "gap request outside the anchor table"; return -2; }
                if (gb->ref_start[r] < 0 || gb->ref_len[r] < 0 || gb->ref_start[r] + gb->ref_len[r] > glen_h[0]) { error = "region outside its genome"; return -2; }
                if (gb->ref_len[r] >= (1ll << 31)) { error = "region longer than 2^31"; return -5; }
            }
            for (int64_t x = 0; x < gb->n_explicit; x++)
                for (int g = 0; g < ngen; g++) {
                    const int64_t st = gb->ex_starts[x * ngen + g], ln = gb->ex_lens[x * ngen + g];
                    if (st < 0 || ln < 0 || st + ln > glen_h[(size_t)g]) { error = "region outside its genome"; return -2; }
                    if (ln >= (1ll << 31)) { error = "region longer than 2^31"; return -5; }
                }
            if (nrow_staged) { memcpy(stage, gb->ex_starts, 8 * nrow_staged); memcpy(stage + nrow_staged, gb->ex_lens, 8 * nrow_staged); }
            memcpy(stage + 2 * nrow_staged, gb->gaps, sizeof(GapRef) * regz);
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
                    const int K = minlen < 16 ? minlen : 16, stride = minlen - K + 1;
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
            ri.K = ri.minlen < 16 ? ri.minlen : 16;
            ri.stride = ri.minlen - ri.K + 1;
            int64_t slots = 16;
            while (2 * slots < 3 * (int64_t)ri.nR) slots <<= 1;   // load factor <= 2/3
            ri.tmask = (uint32_t)(slots - 1);
            ri.tbase = tsize; tsize += slots;
            int64_t fbits = 64;
            while (fbits < 8 * (int64_t)ri.nR) fbits <<= 1;
            ri.fmask = (uint32_t)(fbits - 1); ri.fbase = fwords; fwords += fbits / 32; ri.pad_ = 0;
            ri.posbase = npos; posbase[(size_t)r] = npos; npos += ri.nR;
            cbase[(size_t)r + 1] = cbase[(size_t)r] + (((int64_t)ri.nR + kChunkPos - 1) >> kCoarseShift) + 1;
            max_nr = std::max(max_nr, ri.nR);
            ev_guess += guess_r[(size_t)r];
            nunits += units_r[(size_t)r];
        }
        posbase[regz] = npos;
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
        } else {
            ensure(d_exstarts, std::max<size_t>(nrow_staged, 1)); ensure(d_exlens, std::max<size_t>(nrow_staged, 1)); ensure(d_gaps, regz);
            if (nrow_staged) { be.h2d_staged(d_exstarts.p, stage, 8 * nrow_staged); be.h2d_staged(d_exlens.p, stage + nrow_staged, 8 * nrow_staged); }
            be.h2d_staged(d_gaps.p, stage + 2 * nrow_staged, sizeof(GapRef) * regz);
            be.launch("expand_gaps", (int64_t)nrow, ExpandGaps{d_gaps.p, ngen, d_anchor_start.p, d_anchor_lon.p, d_glen, d_exstarts.p, d_exlens.p, d_starts.p, d_lens.p});
        }
        // event counters: kSlices counters one 64-byte line apart, then the error word of the batch (read back together)
        const size_t ncounter = (size_t)kSlices * kSliceStride + 8;
        ensure(d_counter, ncounter);
        uint32_t* d_err = (uint32_t*)(d_counter.p + (size_t)kSlices * kSliceStride);

        // -- reference index + repeat lengths
        ensure(d_slots, (size_t)tsize); ensure(d_filter, (size_t)fwords);
        be.memset(d_filter.p, 0, sizeof(uint32_t) * (size_t)fwords);
        ensure(d_next, (size_t)std::max<int64_t>(npos, 1)); ensure(d_rep, (size_t)std::max<int64_t>(npos, 1));
        ensure(d_epm, (size_t)std::max<int64_t>(npos, 1) + 1);     // + the verdict word of a sharded run
        be.memset(d_slots.p, 0xff, sizeof(uint64_t) * (size_t)tsize);
        be.memset(d_counter.p, 0, 8 * ncounter);
        be.mark("index");
        be.launch("index_insert", npos, IndexInsert{P, d_R.p, nreg, d_posbase.p, d_slots.p, d_next.p, d_filter.p});
        be.mark("repeat");
        ensure(d_run, (size_t)std::max<int64_t>(npos, 1));
        be.launch("run_length", npos, RunLength{P, d_R.p, nreg, d_posbase.p, d_run.p});
        ensure(d_repeated, (size_t)(npos / 32 + 2));       // (SeedExtend reads the word after a position's own)
        be.memset(d_repeated.p, 0, 4 * (size_t)(npos / 32 + 2));
        be.launch("repeat_length", npos, RepeatLength{P, d_R.p, nreg, d_posbase.p, d_slots.p, d_filter.p, d_next.p, d_run.p, d_rep.p, d_repeated.p, d_err, work_budget});

        // -- work units (pairs that fit 128 bases on both sides go to SmallPairEvents instead)
        be.mark("units");
        ensure(d_ucount, (size_t)npairs + 1); ensure(d_uoff, (size_t)npairs + 1);
        be.launch("count_units", npairs, CountUnits{d_R.p, d_lens.p, ngen, d_ucount.p, g_first, g_last, no_small});
        be.memset(d_ucount.p + npairs, 0, 8);
        be.exclusive_scan(d_ucount.p, d_uoff.p, (size_t)npairs + 1);
        if (gb) {      // the host never saw the rows: the unit count comes back from the device (8 bytes)
            be.d2h(&nunits, d_uoff.p + npairs, 8);
            if (nunits >= (1ll << 31)) { error = "too many work units in one batch"; return -5; }
            ev_guess += (size_t)nunits * 16;
        }
        ensure(d_units, (size_t)std::max<int64_t>(nunits, 1));
        be.launch("fill_units", nunits, FillUnits{P, d_starts.p, d_lens.p, ngen, d_uoff.p, d_ucount.p, npairs, d_units.p});

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
        for (bool again = false;; again = true) {
            ensure(d_evkey, slice_cap * kSlices); ensure(d_evval, slice_cap * kSlices); ensure(d_rest, queue_cap * kSlices);
            if (again) be.memset(d_counter.p, 0, 8 * ((size_t)kSlices * kSliceStride + 1));      // event counters and error word
            be.memset(d_qcount.p, 0, 8 * (size_t)kSlices * kSliceStride);
            be.mark("seed_extend");
            be.launch("seed_extend", nunits * 64,
                      SeedExtend{P, d_R.p, d_units.p, d_slots.p, d_filter.p, d_next.p, d_rep.p, d_repeated.p,
                                 d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, d_err, work_budget, d_rest.p, d_qcount.p, (uint64_t)queue_cap});
            if (nunits > 0)      // one lane per queued sample; lanes past a sub-queue's count leave at once (the counts stay on the device)
                be.launch("seed_rest", (int64_t)(queue_cap * kSlices),
                          SeedRest{P, d_R.p, d_units.p, d_rest.p, d_qcount.p, (uint64_t)queue_cap, d_slots.p, d_filter.p, d_next.p, d_rep.p,
                                   d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, d_err, work_budget});
            if (!no_small)
                be.launch("small_pair_events", npairs * 2,
                          SmallPairEvents{P, d_R.p, d_starts.p, d_lens.p, ngen, d_rep.p, d_evkey.p, d_evval.p, d_counter.p, (uint64_t)slice_cap, lbits, g_first, g_last});
            be.mark("sort");
            be.launch_wave("slice_offsets", 1, SliceOffsets{d_counter.p, d_sliceoff.p});
            be.d2h_async(qcounts.data(), d_qcount.p, 8 * qcounts.size());
            be.d2h(counts.data(), d_counter.p, 8 * counts.size());            // round trip 1: event counts + error word (+ the queues' lengths)
            uint64_t worst = 0, qworst = 0;
            nev = 0; nrest = 0;
            for (int sl = 0; sl < kSlices; sl++) {
                const uint64_t c = counts[(size_t)sl * kSliceStride]; nev += c; worst = std::max(worst, c);
                const uint64_t q = qcounts[(size_t)sl * kSliceStride]; nrest += q; qworst = std::max(qworst, q);
            }
            errbits = (uint32_t)counts[(size_t)kSlices * kSliceStride];
            if (worst <= slice_cap && qworst <= queue_cap) break;
            if (worst > slice_cap) slice_cap = (size_t)(worst + worst / 8 + 64);
            if (qworst > queue_cap) queue_cap = (size_t)(qworst + qworst / 8 + 64);
            sticky |= errbits & kErrWork;      // (RepeatLength's verdict: the word is cleared with the counters)
        }
        errbits |= sticky;
        rest_cap_hint[nreg == 1] = queue_cap;
        last_rest = (int64_t)nrest;
        ev_cap_hint = (size_t)(nev + nev / 4);
        last_events = (int64_t)nev;
        // a rank of a sharded run that ran out of budget must not leave the others waiting in the collectives: the
        // verdict travels with the first exchange (below) and every rank returns the error together
        const bool sharded = coll.world > 1 || coll.device;      // (a one-rank RCCL session still runs the exchanges: that is how a 1-GPU box tests them)
        if ((errbits & kErrWork) && !sharded) { budget_exceeded = true; error = "per-thread work budget exceeded (degenerate repeat structure in a region)"; return -5; }

        ensure(d_evkey2, std::max<size_t>(nev, 1)); ensure(d_evval2, std::max<size_t>(nev, 1));
        ensure(d_evkey3, std::max<size_t>(nev, 1)); ensure(d_evval3, std::max<size_t>(nev, 1));
        const int keybits = bits_for((uint64_t)npairs) + lbits + 1;
        uint64_t *skey = d_evkey3.p, *sval = d_evval3.p;
        if (nev > 0) {
            be.launch("compact_events", (int64_t)nev, CompactEvents{d_evkey.p, d_evval.p, d_sliceoff.p, (uint64_t)slice_cap, d_evkey2.p, d_evval2.p});
            be.sort_pairs(d_evkey2.p, d_evkey3.p, d_evval2.p, d_evval3.p, (size_t)nev, keybits);
        }
        be.mark("scan");
        ensure(d_lo, (size_t)npairs + 1);
        be.launch("pair_bounds", npairs, PairBounds{skey, (int64_t)nev, lbits, npairs, d_lo.p});
        ensure(d_state, std::max<size_t>(nev, 1)); ensure(d_emax, std::max<size_t>(nev, 1));
        const int64_t nscan = ((int64_t)nev + kChunk - 1) / kChunk;
        ensure(d_summary, (size_t)std::max<int64_t>(nscan, 1)); ensure(d_startshere, (size_t)std::max<int64_t>(nscan, 1));
        be.launch("chunk_reduce", nscan, ChunkReduce{skey, sval, (int64_t)nev, lbits, d_summary.p, d_startshere.p});
        be.launch("chunk_scan", nscan, ChunkScan{skey, sval, (int64_t)nev, lbits, d_summary.p, d_startshere.p, d_state.p, d_emax.p});

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
        ensure(d_coarse, (size_t)std::max<int64_t>(centries, 1));
        be.memset(d_coarse.p, 0, 4 * (size_t)std::max<int64_t>(centries, 1));
        be.launch("coarse_fill", (int64_t)nev, CoarseFill{skey, (int64_t)nev, lbits, d_lo.p, d_R.p, d_cbase.p, nq, d_coarse.p});
        be.launch_wave("master_ep", nchunks, MasterEP{d_R.p, nreg, ngen, skey, d_lo.p, d_emax.p, lbits, d_epm.p, d_cbase.p, d_coarse.p, g_first, g_last});
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
        const int64_t nwv = (npos + 63) / 64;
        ensure(d_wmask, (size_t)nwv + 1); ensure(d_wcount, (size_t)nwv + 1); ensure(d_woff, (size_t)nwv + 1);
        be.launch_wave("cand_mark", nwv, CandMark{d_R.p, nreg, d_posbase.p, npos, d_epm.p, d_wmask.p, d_wcount.p});
        be.memset(d_wcount.p + nwv, 0, 8);
        be.exclusive_scan(d_wcount.p, d_woff.p, (size_t)nwv + 1);
        int64_t ncand_i = 0;
        be.d2h(&ncand_i, d_woff.p + nwv, 8);                                   // round trip 2: candidate count
        if (verdict < 0) { budget_exceeded = true; error = "per-thread work budget exceeded on some rank (degenerate repeat structure in a region)"; return -5; }
        const uint64_t ncand = (uint64_t)ncand_i;
        last_candidates = (int64_t)ncand;
        if (ncand == 0) { be.mark(nullptr); collect_timing(); return 0; }
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
        be.launch_wave("fold_candidates", (int64_t)ncand,
                       FoldCandidates{d_R.p, scand, ngen, at, skey, sval, d_lo.p, d_state.p, d_rep.p, lbits, d_cbase.p, d_coarse.p,
                                      d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_ok.p});

        // -- accepted candidates, compacted on the device and downloaded straight into the result's blocks
        be.mark("compact");
        ensure(d_okcnt, (size_t)ncand + 1); ensure(d_okpos, (size_t)ncand + 1);
        be.launch("ok_count", (int64_t)ncand + 1, OkCount{d_ok.p, (int64_t)ncand, d_okcnt.p});
        be.exclusive_scan(d_okcnt.p, d_okpos.p, (size_t)ncand + 1);
        int64_t nok = 0;
        be.d2h(&nok, d_okpos.p + ncand, 8);                                    // round trip 3: accepted count
        const size_t nokz = (size_t)nok, nqz2 = (size_t)nq, ngz = (size_t)ngen;
        ensure(d_creg, std::max<size_t>(nokz, 1)); ensure(d_ck, std::max<size_t>(nokz, 1)); ensure(d_clon, std::max<size_t>(nokz, 1));
        std::vector<int32_t> reg_h(nokz);
        out->kb = pool->take(4 * nokz); out->lonb = pool->take(4 * nokz);
        out->rows = want_rows; out->dirty_known = false;
        if (!want_rows) {
            ensure(d_csp, std::max<size_t>(nokz * nqz2, 1)); ensure(d_cfwd, std::max<size_t>(nokz * nqz2, 1));
            be.launch("compact_sp", (int64_t)ncand * nq,
                      CompactSp{scand, d_ok.p, d_okpos.p, nq, d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_creg.p, d_ck.p, d_clon.p, d_csp.p, d_cfwd.p});
            be.mark("download");
            out->spb = pool->take(4 * nokz * nqz2); out->fwdb = pool->take(nokz * nqz2);
            be.d2h_async(out->spb.p, d_csp.p, 4 * nokz * nqz2);
            be.d2h_async(out->fwdb.p, d_cfwd.p, nokz * nqz2);
        } else {
            ensure(d_csp, std::max<size_t>(nokz * ngz, 1)); ensure(d_cfwd, std::max<size_t>(nokz * ngz, 1)); ensure(d_cflags, std::max<size_t>(nokz, 1));
            be.memset(d_cflags.p, 0, 4 * std::max<size_t>(nokz, 1));
            be.launch("compact_candidates", (int64_t)ncand * ngen,
                      CompactCandidates{scand, d_ok.p, d_okpos.p, ngen, d_ok_k.p, d_ok_lon.p, d_osp.p, d_ofwd.p, d_starts.p, d_lens.p, d_glen,
                                        d_creg.p, d_ck.p, d_clon.p, d_csp.p, d_cfwd.p, d_cflags.p});
            if (nreg == 1 && nok >= dirty_min) {     // a long list of one region (the anchor call): the cheap overlap test on the device
                const int64_t nblocks = (nok + kDirtyBlock - 1) / kDirtyBlock, groups = (ngen + 63) / 64;
                ensure(d_bmax, (size_t)nblocks * ngz); ensure(d_bmin, (size_t)nblocks * ngz);
                be.launch_wave("dirty_extent", nblocks * groups, DirtyExtent{d_csp.p, d_clon.p, d_cflags.p, nok, ngen, d_bmax.p, d_bmin.p});
                be.launch_wave("dirty_prefix", groups, DirtyPrefix{nblocks, ngen, d_bmax.p, d_bmin.p});
                ensure(d_dirty, nokz);
                be.memset(d_dirty.p, 0, 4 * nokz);
                be.launch_wave("dirty_mark", nblocks * groups, DirtyMark{d_csp.p, d_clon.p, nok, ngen, d_bmax.p, d_bmin.p, d_cflags.p, d_dirty.p});
                be.launch("dirty_merge", nok, DirtyMerge{d_dirty.p, d_cflags.p});
                out->dirty_known = true;
            }
            if (nreg == 1 && !gb && nok >= dirty_min) {      // a long list of one region (the anchor call): its rows stay on the device as the session's anchor table (run(..., gb))
                ensure(d_anchor_start, std::max<size_t>(nokz * ngz, 1)); ensure(d_anchor_lon, std::max<size_t>(nokz, 1)); ensure(d_anchor_flags, std::max<size_t>(nokz, 1));
                be.d2d(d_anchor_start.p, d_csp.p, 4 * nokz * ngz);
                be.d2d(d_anchor_lon.p, d_clon.p, 4 * nokz);
                be.d2d(d_anchor_flags.p, d_cflags.p, 4 * nokz);
                ensure(d_anchor_accept, std::max<size_t>(nokz, 1));      // (the overlap flags are in: same condition above)
                be.launch("anchor_accept", nok, AnchorAccept{d_cflags.p, d_clon.p, d_cfwd.p, ngen, d_anchor_accept.p});
                anchor_table_rows = nok;
                out->table_id = anchor_table_id = ++table_counter;
            }
            be.mark("download");
            out->startb = pool->take(4 * nokz * ngz); out->strandb = pool->take(nokz * ngz); out->flagsb = pool->take(4 * nokz);
            be.d2h_async(out->flagsb.p, d_cflags.p, 4 * nokz);
            be.d2h_async(out->startb.p, d_csp.p, 4 * nokz * ngz);
            be.d2h_async(out->strandb.p, d_cfwd.p, nokz * ngz);
        }
        be.d2h_async(reg_h.data(), d_creg.p, 4 * nokz);
        be.d2h_async(out->kb.p, d_ck.p, 4 * nokz);
        be.d2h_async(out->lonb.p, d_clon.p, 4 * nokz);
        be.mark(nullptr);
        be.sync();                                                         // round trip 4: the results
        for (size_t w = 0; w < nokz; w++) out->off[(size_t)reg_h[w] + 1]++;
        for (int64_t r = 0; r < nreg; r++) out->off[(size_t)r + 1] += out->off[(size_t)r];
        out->total = out->off[(size_t)nreg];
        collect_timing();
        return 0;
    }

    // ---- the layout after the anchor call, built from the resident anchor table (kernels.h: LayoutMark; pm_layout_image)
    // accept[c] != 0: row c of the table is marked as it stands (accept == nullptr: the rows AnchorAccept chose when the table
    // was made); rows the host changed (trimmed) come as extra rows.  The
    // image is copied to a page-locked block of the session on a second stream, beside whatever is queued next; the caller
    // reads it after layout_wait().  The block is rewritten by the next call: the previous copy is awaited first.
    int layout_image(int64_t table_id, const int64_t* nbits, const uint8_t* accept, int64_t nrows, const int32_t* xstart, const int32_t* xlon, int64_t nx,
                     uint64_t** image) {
        be.bind();
        if (table_id == 0 || table_id != anchor_table_id) { error = "the anchor table of this layout is no longer resident"; return -2; }
        if (nrows != anchor_table_rows || !nbits || nx < 0 || (nx > 0 && (!xstart || !xlon))) { error = "bad layout request"; return -2; }
        const size_t ngz = (size_t)ngen, nxz = (size_t)nx, nrz = (size_t)nrows;
        std::vector<int64_t> off(ngz + 1, 0);
        for (size_t j = 0; j < ngz; j++) {
            if (nbits[j] < 0 || nbits[j] > glen_h[j] + 64) { error = "layout size does not fit its genome"; return -2; }
            off[j + 1] = off[j] + (nbits[j] + 63) / 64 + 1;
        }
        const size_t words = (size_t)off[ngz];
        be.side_wait();
        if (words > image_words) {
            if (image_h) be.pinned_free(image_h);
            image_h = (uint64_t*)be.pinned_alloc(8 * words);
            image_words = image_h ? words : 0;
            if (!image_h) { error = "cannot allocate the page-locked layout block"; return -3; }
        }
        // parameters: [ word_off | nbits | extra lengths | extra rows | accept ] in a page-locked block of their own (the
        // request block of run() is rewritten by the next call, which may be queued before these copies have run)
        const size_t bytes = 8 * (ngz + 1) + 8 * ngz + 4 * nxz + 4 * nxz * ngz + nrz + 64;
        if (bytes > image_stage_cap) {
            if (image_stage) be.pinned_free(image_stage);
            image_stage = (uint8_t*)be.pinned_alloc(bytes + bytes / 4);
            image_stage_cap = image_stage ? bytes + bytes / 4 : 0;
            if (!image_stage) { error = "cannot allocate the layout staging block"; return -3; }
        }
        int64_t* s_off = (int64_t*)image_stage; int64_t* s_bits = s_off + ngz + 1;
        int32_t* s_xlon = (int32_t*)(s_bits + ngz); int32_t* s_xstart = s_xlon + nxz; uint8_t* s_acc = (uint8_t*)(s_xstart + nxz * ngz);
        memcpy(s_off, off.data(), 8 * (ngz + 1)); memcpy(s_bits, nbits, 8 * ngz);
        if (nxz) { memcpy(s_xlon, xlon, 4 * nxz); memcpy(s_xstart, xstart, 4 * nxz * ngz); }
        if (accept) memcpy(s_acc, accept, nrz);
        ensure(d_image, words); ensure(d_imgoff, ngz + 1); ensure(d_imgbits, ngz); ensure(d_accept, std::max<size_t>(nrz, 1));
        ensure(d_xstart, std::max<size_t>(nxz * ngz, 1)); ensure(d_xlon, std::max<size_t>(nxz, 1));
        be.h2d_staged(d_imgoff.p, s_off, 8 * (ngz + 1)); be.h2d_staged(d_imgbits.p, s_bits, 8 * ngz);
        if (accept) be.h2d_staged(d_accept.p, s_acc, nrz);
        if (nxz) { be.h2d_staged(d_xlon.p, s_xlon, 4 * nxz); be.h2d_staged(d_xstart.p, s_xstart, 4 * nxz * ngz); }
        be.memset(d_image.p, 0, 8 * words);
        be.launch("layout_sentinel", (int64_t)ngen, LayoutSentinel{d_imgoff.p, d_imgbits.p, d_image.p});
        be.launch("layout_mark", nrows * ngen, LayoutMark{d_anchor_start.p, d_anchor_lon.p, accept ? d_accept.p : d_anchor_accept.p, ngen, d_imgoff.p, d_imgbits.p, d_image.p});
        be.launch("layout_mark", nx * ngen, LayoutMark{d_xstart.p, d_xlon.p, nullptr, ngen, d_imgoff.p, d_imgbits.p, d_image.p});
        be.d2h_side(image_h, d_image.p, 8 * words);
        *image = image_h;
        return 0;
    }
    void layout_wait() { be.bind(); be.side_wait(); }      // (may be called by a helper thread while another call is running)

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
    bool tune(const std::string& key, int64_t value) {
        if (key == "work_budget" && value > 0) { work_budget = value; return true; }
        if (key == "dirty_min" && value >= 0) { dirty_min = value; return true; }
        return false;
    }

    void release() {
        for (BufBase* b : all_bufs) { if (b->raw) be.free(b->raw); b->raw = nullptr; b->cap = 0; }
        be.side_wait();
        if (image_h) be.pinned_free(image_h);
        if (image_stage) be.pinned_free(image_stage);
        image_h = nullptr; image_stage = nullptr; image_words = 0; image_stage_cap = 0;
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
    void collect_timing() { timing = be.collect(); }

    SeqBlock* blk = nullptr; int64_t* d_goff = nullptr; int64_t* d_glen = nullptr;
    int64_t total_words = 0;
    Packed P{};
    size_t ev_cap_hint = 0, cand_cap_hint = 0, rest_cap_hint[2] = {0, 0};
    Buf<RegionInfo> d_R; Buf<int64_t> d_starts, d_lens, d_posbase;
    Buf<uint64_t> d_slots; Buf<uint32_t> d_filter, d_repeated; Buf<int32_t> d_next, d_rep, d_run, d_epm;
    Buf<int64_t> d_ucount, d_uoff; Buf<UnitRec> d_units;
    Buf<uint64_t> d_counter, d_evkey, d_evval, d_evkey2, d_evval2, d_evkey3, d_evval3; Buf<int64_t> d_sliceoff;
    Buf<int64_t> d_lo, d_cov; Buf<EventState> d_state, d_summary; Buf<int32_t> d_emax; Buf<uint8_t> d_startshere;
    Buf<uint64_t> d_cand; Buf<GenomeAtK> d_at; Buf<uint8_t> d_ok; Buf<int32_t> d_ok_k, d_ok_lon, d_osp; Buf<uint8_t> d_ofwd;
    Buf<uint64_t> d_wmask; Buf<int64_t> d_wcount, d_woff;
    Buf<int64_t> d_okcnt, d_okpos, d_cbase; Buf<int32_t> d_coarse; Buf<int32_t> d_creg, d_ck, d_clon, d_csp; Buf<uint8_t> d_cfwd;
    Buf<uint32_t> d_cflags, d_dirty; Buf<int32_t> d_bmax, d_bmin;
    Buf<GenomeAtK> d_xsend, d_xrecv; Buf<uint8_t> d_hsend, d_hrecv;
    Buf<RestItem> d_rest; Buf<uint64_t> d_qcount;
    Buf<int32_t> d_anchor_start, d_anchor_lon; Buf<uint32_t> d_anchor_flags; Buf<uint8_t> d_anchor_accept; Buf<GapRef> d_gaps; Buf<int64_t> d_exstarts, d_exlens;
    Buf<SpecRegion> d_spec; Buf<uint64_t> d_speccount; Buf<int32_t> d_mintable;
    int64_t table_counter = 0;
    Buf<uint64_t> d_image; Buf<int64_t> d_imgoff, d_imgbits; Buf<uint8_t> d_accept; Buf<int32_t> d_xstart, d_xlon;
    uint64_t* image_h = nullptr; size_t image_words = 0;      // page-locked: the layout image as the host reads it
    uint8_t* image_stage = nullptr; size_t image_stage_cap = 0;
};

}  // namespace pm
