/* include/parsnp_mum.h -- C ABI of the MI355X multi-MUM engine (libparsnp_hip.so).
 *
 * This is the drop-in boundary for the one hot path of marbl/parsnp that this project replaces: the
 * csgmum calls made by Aligner::setMums1.  The reference has no FFI for it (csgmum is #included C,
 * src/parsnp.cpp:73-78); the entry points below are what a binding for that path would bind, one per
 * reference call site.  Plain pointers and sizes only; every function returns 0 on success or a negative
 * PM_E* code (no exceptions cross the boundary).  pm_last_error() gives a message for the calling thread.
 *
 * Sequences are ASCII over {A,C,G,T,N}, exactly the strings parsnp_core holds after ingest
 * (src/parsnp.cpp:2999-3141): any other byte is treated as N.  'N' matches 'N' (src/csgmum/csg.c:13-25).
 * Coordinates are 0-based; all arithmetic is integer (int32 reference positions, int64 query positions).
 */
#ifndef PARSNP_MUM_H
#define PARSNP_MUM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_ENODEV (-1)   /* no usable gfx950 device / HIP runtime error at start-up */
#define PM_EINVAL (-2)   /* bad argument */
#define PM_ENOMEM (-3)   /* host or device allocation failed */
#define PM_EHIP (-4)     /* a HIP call or kernel failed */
#define PM_ELIMIT (-5)   /* a size limit of the engine was exceeded (see pm_limits) */
#define PM_EAGAIN (-6)   /* the resident route (pm_store_*) does not apply to this input: the caller takes the host route */

typedef struct pm_session pm_session; /* genomes resident in HBM (2-bit + N-mask, both strands) */
typedef struct pm_result pm_result;   /* output of one batch call, owned by the library until pm_result_free */

const char* pm_last_error(void);
/* "hip" for libparsnp_hip.so. (The test-only CPU provider under oracle/ answers "oracle".) */
const char* pm_provider(void);

/* Optional: start the HIP runtime of this process (device discovery, context, code objects -- ~0.15 s when nothing else has
 * touched the GPU yet).  Thread-safe; meant to be called from a side thread while the caller is still reading its input,
 * as parsnp_core does during FASTA ingest.  device < 0: PARSNP_DEVICE or the current device. */
int pm_warmup(int device);

/* Upload the n genomes once (genome 0 = reference).  Replaces the per-call substr()+reversec()+strcpy of
 * src/parsnp.cpp:1540-1561: regions are addressed by (start,len) into these resident copies.
 * device < 0 selects the current HIP device. */
int pm_session_create(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens);
/* Sharded run over the GPUs of one node: rank r of `world` SEARCHES a contiguous block of the query genomes
 * (block r of world equal blocks of genomes 1..n-1, in ini order) against the reference; every rank keeps all genomes
 * resident (one byte per base and strand -- memory is not what a 288 GB GPU runs out of -- so that the validation of the
 * resident route can check any member of a MUM against its sequence).  Every rank makes the same sequence of calls with the
 * same arguments and receives the same results.  Per batch the engine exchanges
 *   (1) Master.EP, reduced with min over the ranks           (allreduce_min, one int32 per reference position), and
 *   (2) the per-genome (EP,UP,SP) columns at the candidates  (allgather of equal-size blocks),
 * through the two callbacks (host buffers; 0 = success), which the embedding process implements with
 * torch.distributed -- RCCL over xGMI on the GPUs, gloo in the CPU tests.  The strand/UP fold then runs over all
 * genomes in ini order on every rank, so results are bit-identical to the unsharded run. */
typedef int (*pm_allreduce_min_i32_fn)(void* ctx, int32_t* buf, int64_t count);
typedef int (*pm_allgather_fn)(void* ctx, const void* send, int64_t send_bytes, void* recv /* world * send_bytes */);
int pm_session_create_sharded(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens,
                              int rank, int world, pm_allreduce_min_i32_fn allreduce_min, pm_allgather_fn allgather, void* ctx);
/* The same sharded run with the two exchanges ON THE DEVICE: the library owns an RCCL communicator (one rank per process
 * and GPU, xGMI inside a node) and runs all-reduce(min) on Master.EP in place and the all-gather of the candidate columns
 * on its own stream, between its kernels -- no host staging, no callback.  Rank 0 obtains an id with pm_rccl_unique_id
 * and hands its 128 bytes to the other ranks by whatever channel launched them (a file, MPI, torch.distributed's store);
 * every rank then calls pm_session_create_rccl with its rank.  RCCL is loaded on demand from PARSNP_RCCL_LIB or
 * /opt/rocm/lib/librccl.so.1; a one-rank communicator (world = 1) is legal and runs the same exchange code. */
#define PM_RCCL_ID_BYTES 128
int pm_rccl_unique_id(uint8_t* id /* PM_RCCL_ID_BYTES */);
int pm_session_create_rccl(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens,
                           int rank, int world, const uint8_t* id /* PM_RCCL_ID_BYTES */);
/* number of ranks of the session's RCCL communicator as RCCL reports it (ncclCommCount); 0 for a session without one */
int pm_session_rccl_ranks(const pm_session* s);
void pm_session_destroy(pm_session* s);
int pm_session_genomes(const pm_session* s);

/* Multi-MUM candidates for a batch of regions.  For region r (r = 0..n_regions-1) and genome g:
 *   starts[r*n_genomes+g], lens[r*n_genomes+g] : the substring of genome g   (TRegion start/length, src/LCR.cpp:12-37;
 *                                                 the reference chunk of src/parsnp.cpp:1519-1547 when g == 0)
 *   minsize[r]                                 : minimum MUM length          (src/parsnp.cpp:1502-1514)
 * Per region this is the whole of src/parsnp.cpp:1563-1695:
 *   new_CSG/build_CSG/find_leaves (csg.c:105-575)  -> device index of the reference substring
 *   Find_UM x 2 per query (mum.c:177-250)          -> R-unique maximal matches, both strands
 *   Intersect_UM x 2 (mum.c:125-175)               -> forward carry + fold into Master
 *   Merge_Master (mum.c:92-123)                    -> strand choice in ini order, ties to reverse
 *   extraction loop (parsnp.cpp:1633-1695)         -> candidates (k, LON, per-genome SP and strand)
 * The candidate list is bit-identical to the reference's list_mums for the same region. */
int pm_multi_mum_batch(pm_session* s, int64_t n_regions, const int64_t* starts, const int64_t* lens,
                       const int32_t* minsize, pm_result** out);

/* Result accessors.  Candidates of region r are c in [off[r], off[r+1]).  For candidate c:
 *   k[c]   reference offset inside the region (Mum.DSP[0]-1-ini_region, parsnp.cpp:1671)
 *   lon[c] length (Mum.LON, :1687)
 *   sp[c*(n_genomes-1)+g-1]  start inside genome g's substring, on the chosen strand's string (SPF[g-1].MSP[k], :1681; region lengths are < 2^31, so int32)
 *   fwd[c*(n_genomes-1)+g-1] 1 = forward, 0 = reverse complement (SPF[g-1].forward[k], :1678)            */
int64_t pm_result_regions(const pm_result* r);
int64_t pm_result_total(const pm_result* r);
const int64_t* pm_result_offsets(const pm_result* r);
const int32_t* pm_result_k(const pm_result* r);
const int32_t* pm_result_lon(const pm_result* r);
const int32_t* pm_result_sp(const pm_result* r);
const uint8_t* pm_result_fwd(const pm_result* r);
void pm_result_free(pm_result* r);

/* MUM-row mode.  After pm_session_rows(s, 1) the results of pm_multi_mum_batch carry, instead of sp / fwd (those accessors
 * then return NULL), what the second half of setMums1 derives from them per candidate (src/parsnp.cpp:1717-1780 and the
 * TMum constructor src/TMum.cpp:25-60), built on the device; n = n_genomes, genome 0 = the reference:
 *   start[c*n+j]   start of the MUM in genome j on the genome's forward coordinates: window start + sp for a forward
 *                  member, genome length - (window start + sp + lon) for a reverse one (flipped against the WHOLE genome)
 *   strand[c*n+j]  1 forward, 0 reverse (genome 0: always 1)
 *   flags[c]       PM_ROW_BAD      a start position outside its window: the reference skips the candidate (:1723)
 *                  PM_ROW_OUTSIDE  start < 0 or start + lon > genome length in some genome (`notgood`)
 *                  PM_ROW_REVERSE  some member is on the reverse strand
 *                  PM_ROW_DIRTY    (only when pm_result_dirty_known) overlaps an earlier candidate of the list in some
 *                                  genome by the running-extent test of the anchor validation; computed for one-region
 *                                  batches with at least "dirty_min" (4096) accepted candidates
 *                  PM_ROW_EARLY    (with PM_ROW_DIRTY's condition) starts, in some genome, before the end of an earlier
 *                                  candidate of the list: candidates WITHOUT it lie in list order in every genome
 * The window of genome j is the starts/lens row the caller passed, so the rows equal the reference's only for requests
 * whose rows ARE the region (one unclamped reference chunk); a caller with clamped or chunked rows keeps to sp / fwd.
 * The blocks are writable: the caller may trim the rows in place (Aligner::trim) and keep them as its MUM table. */
#define PM_ROW_BAD 1u
#define PM_ROW_OUTSIDE 2u
#define PM_ROW_REVERSE 4u
#define PM_ROW_DIRTY 8u
#define PM_ROW_EARLY 16u
int pm_session_rows(pm_session* s, int enable);
int32_t* pm_result_start(pm_result* r);
uint8_t* pm_result_strand(pm_result* r);
const uint32_t* pm_result_flags(const pm_result* r);
int pm_result_dirty_known(const pm_result* r);

/* The rows of a long ONE-region result in row mode (the anchor call: at least "dirty_min" candidates) stay on the device as the
 * session's ANCHOR TABLE; pm_result_table_id() names it (0: this result left no table).  The resident route below works on it. */
int64_t pm_result_table_id(const pm_result* r);
/* Tunables of a session (tests lower the thresholds of the long-list routes so that small inputs take them):
 *   "work_budget"  per-thread step budget of the index walks (default 2^22; a batch that exhausts it is repeated once with
 *                  256 times as much, then PM_ELIMIT)
 *   "dirty_min"    shortest one-region candidate list that gets the overlap / order flags and stays on the device as the
 *                  anchor table (default 4096)
 *   "flagged_div"  pm_store_settle takes an anchor table of which at most one row in flagged_div is flagged (overlaps an earlier
 *                  one by the cheap running-extent test: default 8) as it is; above that it counts the TANGLED rows first (flagged rows
 *                  that overlap one another: settled in list order where they meet, in rounds) and answers PM_EAGAIN with more than
 *                  "tangled_max" of them (default 131072).  1: never PM_EAGAIN (tests)
 *   "tangle_rounds" 0: the tangled rows are settled by ONE wavefront in list order instead of in rounds (default 1; the same result)
 *   "fast_tail"    0: a search whose rows stay on the device waits for its unit, candidate and accepted counts before it sizes the
 *                  launches that need them, as every other search does (default 1: capacities from the last call of the shape, the
 *                  kernels read the live counts, the counts come back with the results; the same result)
 *   "atomic_marks" != 0: pm_store_settle marks the layout with atomic ORs even where the list's order allows plain stores (tests)
 *   "group_small"  0: the events of a recursion batch's small regions are found pair by pair and sorted with the others, instead of
 *                  once per distinct query piece (default 1; both give the same events, tests compare the two)
 *   "slot_factor"  index slots per reference position before rounding up to a power of two (default 2: load <= 1/2; 1 = rounds 1-5,
 *                  load <= 2/3 -- the wavefronts of IndexInsert and of the probes run as long as their longest probe sequence);
 *   "filter_factor" presence-filter bits per reference position before rounding up (default 8; 4 ... 32 measured: flat)
 *   "bucket_sort"  0: the events of a search are gathered and radix-sorted by (pair, l, strand) and the readers' table of events per
 *                  256-position block comes from a pass over the sorted keys (rounds 1-5), instead of a counting sort by (pair, block)
 *                  buckets whose scanned counts ARE that table (default 1; the same events in the same order up to equal keys)
 *   "master_seg"   0: Master.EP by the round-4 kernel (every lane tests every staged event: MasterEP) instead of from the genomes'
 *                  segments (MasterEPSeg; default 1; the same values, tests compare the two)
 *   "stage_gate"   != 0: the second stage of a two-stage pm_store_validate is never run (tests: the caller forms it again)
 *   "cluster_unsure" != 0: pm_store_validate's collinear test of the clusters reports failure although they are in order (tests:
 *                  the exact test -- extents ORed into a scratch image -- must find them disjoint and the same bytes result)
 *   "order_debug"  != 0: the order check (pm_store_order_check / pm_store_chain_begin) waits for its kernels and prints to stderr
 *                  how many candidates it had noted and how many its bounds left to the scan
 *   "chain_tie"    != 0: pm_store_chain_begin reports two MUMs with one reference start although there is none (tests: the
 *                  caller's own list logic must give the same bytes)
 *   "timing"       0: no HIP events around the phases of a call (pm_last_timing then reports counts only)
 * PM_EINVAL for an unknown key or a value out of range. */
int pm_session_tune(pm_session* s, const char* key, int64_t value);

/* ---------------------------------------------------------------------------------------------------------------------------
 * The RESIDENT route.  What Aligner does with a candidate list after csgmum produced it -- validation and trimming (second half
 * of setMums1, src/parsnp.cpp:1713-1841; Aligner::trim :1399-1477), the neighbour regions (determineRegion :1199-1290), the
 * generations of the work list (doWork :173-317), the pairwise test of the LCB chaining (setFinalClusters :2596-2700) and the
 * inter-LCB fillers (setInterClusterRegions :2389-2460) -- on rows that never leave the device.  One entry point per reference
 * function; the order-dependent list logic (work-list order, ties, the chain walk) stays with the caller, which receives a few
 * bytes per MUM instead of its rows.
 *
 * pm_session_rows(s, 2) switches the session to resident mode: the rows of a long one-region result (the anchor call: the
 * condition of the anchor table, above) stay on the device as rows [0, A) of the session's MUM STORE -- pm_result_start /
 * _strand of such a result are NULL, pm_result_store_base() is 0 -- and every pm_store_search appends its candidates.  A store
 * row has, besides the row the search delivered (never rewritten), `shift` (bases trimmed on the left: TMum::trimleft moves the
 * start in EVERY genome), `len` (current length) and a state.  The layout (mumlayout, :3181-3186) is an image in device memory.
 * The REGION STORE holds the request rows of seed and child regions.  A later anchor call of the session starts all three over.
 * Every function returns PM_OK, PM_EAGAIN (the caller falls back to the host route: pm_store_rows(raw) gives it the rows it
 * did not receive) or an error; a session is single-threaded. */
#define PM_ST_BUILT 1u      /* the reference constructs a TMum for the candidate (no PM_ROW_BAD) */
#define PM_ST_OK 2u         /* inside every genome (no PM_ROW_OUTSIDE) */
#define PM_ST_FLAGGED 4u    /* (anchor list) overlapped an earlier row: settled against the marks of the others */
#define PM_ST_TANGLED 8u    /* ... and another flagged row: settled in list order */
#define PM_ST_ACCEPTED 16u  /* a MUM of the run: marked in the layout */
typedef struct { int32_t start0, len, shift; uint32_t state_flags; } pm_row_info;   /* start on the reference (trim applied), length, left trim, state | PM_ROW_* << 8 */
typedef struct { int64_t key, ref_start, ref_len; int32_t slength, parent; } pm_region_info;   /* push order, reference column, shortest length, store row of the MUM it lies next to */
int64_t pm_result_store_base(const pm_result* r);      /* first store row of a result whose rows stayed resident, else -1 */
/* setMums1, second half, for the anchor table `table_id` into an EMPTY layout (:1781-1841): rows that overlap nothing earlier
 * are settled and marked at once; the flagged ones (PM_ROW_DIRTY) are trimmed against those marks (Aligner::trim) -- side by
 * side where they meet no other flagged row, in list order where they do (rounds: a tangled row is settled once the tangled
 * rows before it that share a 64-base word with it are).  rows[c] for every row of the table.  PM_EAGAIN only above
 * "tangled_max" tangled rows (the exact overlap test and the threads of the host route decide); rearranged sets are taken. */
int pm_store_settle(pm_session* s, int64_t table_id, pm_row_info* rows);
int pm_store_info(pm_session* s, int64_t first, int64_t count, pm_row_info* out);
/* setInitialClusters' seed regions (:2150-2172): determineRegion on both sides of every accepted anchor (anchors[]: their store
 * rows in list order), kept when longer than q in every genome.  The kept regions are regions [0, *n_regions) of the region
 * store; pm_store_new_regions / _ids list them in the reference's push order. */
int pm_store_seeds(pm_session* s, int64_t table_id, const int32_t* anchors, int64_t n_anchors, int32_t q, int64_t* n_regions);
/* pm_store_settle and pm_store_seeds in one call, without the round trip between them: the accepted anchors are listed on the
 * device and the kept regions arrive in the reference's push order (pm_store_new_regions / _ids).  PM_EAGAIN as pm_store_settle. */
int pm_store_settle_seeds(pm_session* s, int64_t table_id, int32_t q, pm_row_info* rows, int64_t* n_regions);
const pm_region_info* pm_store_new_regions(const pm_session* s);   /* of the last pm_store_seeds / pm_store_validate, valid until the next */
const int32_t* pm_store_new_region_ids(const pm_session* s);
int pm_store_regions_equal(pm_session* s, const int32_t* a, const int32_t* b, int64_t n, uint8_t* same);   /* TRegion operator== (LCR.cpp:48-58) */
/* pm_multi_mum_batch on the rows of the listed regions; the candidates of region i become store rows
 * [*first_row + offsets[i], *first_row + offsets[i + 1]) (offsets[n + 1]). */
int pm_store_search(pm_session* s, const int32_t* regions, const int32_t* minsize, int64_t n, int64_t* first_row, int64_t* offsets);
/* One generation of doWork (:173-317).  The caller lists the waiting regions in the reference's order (reference start), with the
 * store rows of their candidates, cut into clusters (cluster c = regions [cluster_first[c], cluster_first[c + 1])) that it has
 * formed on the reference (maximal runs that overlap or touch there); the clusters are validated side by side, each in order: candidates settled against
 * the layout and marked (setMums1 second half), the neighbour regions of every new MUM longer than q appended to the region
 * store (:215-254; pm_store_new_regions lists them parent by parent, in push order; one equal to a region still waiting in its
 * cluster is dropped as the work list would, :294-306).  The caller clusters on the reference only; what the other genomes do
 * to the clusters is the call's business, and its answer is done[c] (n_clusters values): how many regions of cluster c were
 * processed.  All of them where the cluster meets no EARLIER cluster in any genome (clusters in reference order follow each
 * other with a base between in every genome -- or, where a genome holds them in another order, their extents are tested
 * exactly, and so is what their CANDIDATES touch: a member outside its region reads where another cluster may mark); NONE where it does: the reference (doWork pops the smallest reference start, and children lie inside their parents)
 * finishes the earlier cluster and everything it leads to first, so the cluster must wait -- the caller keeps its regions on
 * the work list for the next generation, where the question is put again; the FIRST FEW where a child of a processed region
 * sorts before (or ties with) the next waiting region of the cluster: the rest waits as well, the next generation sorts it
 * with the children (:291-292).  Every call processes at least the first region of its first cluster.  done == NULL: a caller
 * that cannot keep regions waiting -- either case is then reported as trouble (bits 3 / 0) instead.
 * *trouble != 0: the reference's order would show and the caller must discard the run and take the host route -- bit 1 (2): a
 * reverse-strand member outside its region (TMum.cpp:33-35 flips it against the whole genome) was accepted where a region on the
 * wrong side of the order covers its marks (one walked out or processed by a MUM the reference takes LATER, or one that still
 * waits and sorts EARLIER; an accepted member whose marks fall on ground no such region covers is harmless and stays); bit 2 (4): a
 * region with 2^22 candidates or more, or more candidates with a member outside their region (or one longer than 64 bases) than
 * the engine notes for pm_store_order_check; bits 0 (1) and 3 (8): only with done == NULL, see above.
 * info_count > 0: the per-row scalars (pm_store_info) of store rows [info_first, info_first + info_count) -- the candidates
 * just decided -- come back with the same round trip.
 * stage_first > 0: TWO generations in one call.  Clusters [0, stage_first) are validated first (the first pushed seed, which the
 * reference processes before its work list is ever sorted, :194-195 before :291-292); clusters [stage_first, n_clusters) -- the
 * generation the caller formed on the assumption that the first stage pushes no child region -- are validated behind them if
 * that held (*second_stage_ran = 1) and are left untouched if not (0: the caller forms the generation again, with the children;
 * their done[] is 0).
 * generation: the number of the (first) generation of the call, 0 = the first pushed seed alone; with the regions' reference
 * starts it orders the regions as the reference's work list does (pm_store_order_check). */
int pm_store_validate(pm_session* s, const int32_t* regions, const int64_t* row_first, const int32_t* row_count, int64_t n_regions,
                      const int64_t* cluster_first, int64_t n_clusters, int32_t q, uint32_t* trouble, int64_t* n_children,
                      int64_t info_first, int64_t info_count, pm_row_info* info, int64_t stage_first, int32_t* second_stage_ran, int32_t generation, int32_t* done);
/* After the LAST generation.  A candidate with a reverse-strand member outside its region reads layout marks in another
 * cluster's territory, and what is marked there when the reference looks depends on its order (doWork :173-317: the first pushed
 * seed, then always the waiting region with the smallest reference start).  pm_store_validate notes every such candidate with
 * the marks it saw; this call decides each of them again with the marks the reference's order had in place -- the anchors' and
 * those of the recursion's MUMs whose region comes earlier in that order -- and reports *trouble != 0 when a verdict, a shift or
 * a length differs from what the generations stored: the caller must discard the run and take the host route.
 * pm_store_chain_begin runs the same check ahead of phases C-D (bit 2 of pm_chain_info.trouble): a caller that queues them
 * does not call this. */
int pm_store_order_check(pm_session* s, uint32_t* trouble);
/* The test of setFinalClusters (:2596-2700) of MUM cur[i] against the open chain's last MUM back[i]: verdict[i] = 0 every
 * genome's gap lies in [0, d] (min_gap / max_gap: what the ratio test :2693 reads), 1 the chain closes, 2 a reverse-strand
 * member (the strand rules of :2604-2625 depend on the genome order): the caller judges the pair from its rows. */
int pm_store_judge(pm_session* s, const int32_t* cur, const int32_t* back, int64_t n, int32_t d, int32_t* min_gap, int32_t* max_gap, uint8_t* verdict);
int pm_store_unmark(pm_session* s, const int32_t* rows, int64_t n);      /* the MUMs leave the layout (:415-418, :460-466) */
/* setInterClusterRegions (:2389-2460) for consecutive LCBs (last MUM of one, first MUM of the next): add[i] = 1 a filler is made
 * -- its rows, [n_genomes] each, follow one another in pm_store_fill_starts / _ends --, 0 none, 2 the reference's bookkeeping
 * would overrun (:2419-2433). */
int pm_store_fill(pm_session* s, const int32_t* last_of, const int32_t* first_of_next, int64_t n, uint8_t* add);
const int64_t* pm_store_fill_starts(const pm_session* s);
const int64_t* pm_store_fill_ends(const pm_session* s);
/* Phases C-D in one queue of launches, for a run whose list logic is order-free -- all reference starts of the accepted MUMs
 * differ, diag_diff <= 1 (the ratio test joins or closes, :2693), no MUM short enough for filterRandom1 (:338-425): the sort by
 * reference start (:338, :2571), the chain walk of setFinalClusters (:2563-2719; the pairwise test of pm_store_judge, pairs with
 * reverse-strand members included, with the ratio test applied on the device in the reference's float / double mix), the
 * dissolving of LCBs no longer than c by filterRandomClustersSimple1 (:433-497; the last LCB is never examined, :447; their
 * MUMs leave the layout), the second chaining pass (:3261-3268) and the fillers of setInterClusterRegions (:2389-2460; counted:
 * a filler is never printed, it shifts the numbers of the LCBs behind it).  _begin queues the work and returns; _end waits
 * for it: n_mums store rows in reference order (rows), a flag per MUM that begins an LCB (heads) -- both valid until the
 * session's next chain call -- and the counters of the log.  trouble != 0: bit 0 two MUMs share a reference start (the
 * reference's unstable sort decides: nothing on the device has changed, the caller runs pm_store_judge / _unmark / _fill with
 * its own list logic); bit 1 the reference's filler bookkeeping would overrun (:2419-2433, pm_store_fill's add = 2); bit 2 the
 * order check (pm_store_order_check) failed: nothing has changed, the caller discards the run and takes the host route.
 * n_expected: the number of accepted store rows (the caller's MUM list); a mismatch is an error. */
typedef struct { int64_t n_in, lcbs_first, lcbs_dissolved, mums_dissolved, n_mums, n_lcbs, n_fillers; uint64_t trouble; } pm_chain_info;
int pm_store_chain_begin(pm_session* s, int64_t n_expected, int32_t d, float diag_diff, int64_t c);
int pm_store_chain_end(pm_session* s, pm_chain_info* info, const int32_t** rows, const uint8_t** heads);
/* Rows for the host (the XMFA writer after the LCBs are final; a caller falling back to the host route): start[i * n_genomes + j]
 * with the trim applied (raw != 0: as the search delivered it), strand byte.  rows == NULL: store rows [first, first + n). */
int pm_store_rows(pm_session* s, const int32_t* rows, int64_t first, int64_t n, int raw, int32_t* start, uint8_t* strand);
/* the layout image: genome j = words [word_off[j], word_off[j + 1]) (word_off[n_genomes + 1] filled in), n_j + 1 bits, returns the
 * number of words; pm_store_layout copies it out */
int64_t pm_store_layout_words(pm_session* s, int64_t* word_off);
int pm_store_layout(pm_session* s, uint64_t* out, int64_t words);
/* bytes this session has moved over the host link so far (every copy the engine issued, both directions) */
int pm_session_traffic(const pm_session* s, uint64_t* h2d_bytes, uint64_t* d2h_bytes);

/* calcmumi mode (Aligner::setMumi, src/parsnp.cpp:1869-2115): every query genome ALONE against the reference chunk
 * starts[0],lens[0] (query g: starts[g],lens[g]).  covered[g-1] = number of reference positions covered by the
 * pairwise MUMs of length >= 15 of query g, i.e. total_M_LON of :2064-2069 before the length-ratio and clamp rules
 * (:2074-2077), which stay with the caller together with the "%d:%f" distance (:2080).
 * Replaces, per query: Find_UM x2, Intersect_UM x2, Merge_Master and the coverage loop (:2015-2063). */
int pm_mumi_coverage(pm_session* s, const int64_t* starts, const int64_t* lens, int64_t* covered);

/* Single-strand entry point mirroring Find_UM (src/csgmum/mum.c:177-250) for parity tests: the event stream of one
 * query strand against ref -- every maximal exact match (j, l, len) of length >= min_len whose reference side is unique
 * in ref -- in increasing l.  strand 0: the query as given, 1: its reverse complement (Aligner::reversec).
 * rep[i] is the uniqueness point pos_label - l of mum.c:219-224 (longest repeated prefix of ref[l..)) where it is
 * >= min(min_len,16) and 0 below that (values that small cannot change any MUM).
 * At most cap events are written; *count receives the total. */
int pm_find_events(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int32_t min_len, int strand,
                   int64_t cap, int64_t* count, int64_t* ev_j, int64_t* ev_l, int32_t* ev_len, int32_t* ev_rep);

/* Inter-MUM gap alignment for a whole batch of gaps on the device.  Replaces MuscleInterface::CallMuscleFast
 * (src/MuscleInterface.cpp:37-78; call site src/parsnp.cpp:854-855): libMUSCLE 3.7 with SEQTYPE_DNA, one iteration,
 * ClustalW weights, over the n_seqs[j] gap strings of gap j -- the rows it returns are byte-identical to MUSCLE's.
 *   seq_off[0 .. total_seqs]   offsets into chars of all sequences of all jobs, job after job (ASCII, upper case)
 *   max_cols[j], row_off[j]    row capacity of job j and where its rows go: sequence i of job j is written to
 *                              out_rows[row_off[j] + i*max_cols[j] ..], cols[j] columns of it
 *   cols[j] = -1               the device declined the job (more than 512 sequences, an intermediate alignment wider than
 *                              96 columns or than max_cols[j], an empty sequence, a lower-case letter or a 'U' -- its rows
 *                              are kept as one code byte per column, which round-trips upper-case DNA without 'U' -- or a
 *                              case in which MUSCLE itself quits): the caller aligns it on the host
 * device < 0: PARSNP_DEVICE or the current device.  Returns PM_OK or a PM_E* code (pm_gap_last_error()). */
int pm_gap_align_batch(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                       const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols);
/* The same batch in n_groups groups of consecutive jobs (group g = jobs [group_end[g-1], group_end[g]), group_end[n_groups-1]
 * = n_jobs): everything is uploaded once, the groups are aligned one after the other, and as soon as the rows and column
 * counts of group g are in the caller's arrays `done(ctx, g)` is called from the calling thread -- the caller may use them
 * while the later groups are still being aligned (the XMFA writer lays out and writes the records of the LCBs whose gaps
 * are done).  On an error return the groups after the last one reported have not been reported.  done may be NULL.
 * The rows of a group come back in ONE copy of the span of out_rows that its accepted jobs cover: the row areas of the
 * groups must not interleave (row_off ascending with the job number, as the batch form lays them out), and the rows of a
 * declined job (cols[j] = -1) that lie inside such a span hold unspecified bytes afterwards.
 * pm_gap_last_error(): the message of the last failed call of this process, from any thread. */
int pm_gap_align_groups(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                        const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols,
                        int n_groups, const int64_t* group_end, void (*done)(void* ctx, int group), void* ctx);
const char* pm_gap_last_error(void);

/* Device-side timing of the last pm_multi_mum_batch on this session (HIP events on the engine's stream):
 * names[i] / ms[i] for i < *count (count in: capacity, out: filled).  Used by bench.py's roofline line. */
int pm_last_timing(const pm_session* s, int* count, const char** names, float* ms);

#ifdef __cplusplus
}
#endif
#endif
