/* oracle/pm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 * CPU provider of the include/parsnp_mum.h ABI built on oracle/mum_oracle.c.  It exists so that the host-side
 * logic (parsnp_amd/csrc/host) can be exercised and compared with the reference binary on machines without a GPU
 * (tests -m "not gpu").  It is linked ONLY into oracle/_ref/parsnp_core_oracle and loaded ONLY by tests/;
 * the product binary links libparsnp_hip.so and has no path to this file. */
#include "../include/parsnp_mum.h"
#include "mum_oracle.h"
#include <stdlib.h>
#include <string.h>

struct pm_session { int n; uint8_t** seq; int64_t* len; };
struct pm_result { int64_t nreg, total; int nq; int64_t* off; int32_t* k; int32_t* lon; int32_t* sp; uint8_t* fwd; };
static const char* g_err = "";
const char* pm_last_error(void) { return g_err; }
const char* pm_provider(void) { return "oracle"; }

int pm_session_create(pm_session** out, int device, int n, const uint8_t* const* seqs, const int64_t* lens) {
    (void)device;
    if (!out || n < 1) { g_err = "bad argument"; return PM_EINVAL; }
    pm_session* s = calloc(1, sizeof *s);
    s->n = n; s->seq = calloc((size_t)n, sizeof *s->seq); s->len = calloc((size_t)n, sizeof *s->len);
    for (int i = 0; i < n; i++) {
        s->len[i] = lens[i]; s->seq[i] = malloc((size_t)lens[i] + 1);
        for (int64_t j = 0; j < lens[i]; j++) { uint8_t c = seqs[i][j]; s->seq[i][j] = (c=='A'||c=='C'||c=='G'||c=='T') ? c : 'N'; }
    }
    *out = s; return PM_OK;
}
int pm_session_create_sharded(pm_session** out, int device, int n, const uint8_t* const* seqs, const int64_t* lens, int rank, int world,
                              pm_allreduce_min_i32_fn ar, pm_allgather_fn ag, void* ctx) {
    (void)rank; (void)ar; (void)ag; (void)ctx;
    if (world != 1) { g_err = "the CPU checker is not sharded"; return PM_EINVAL; }
    return pm_session_create(out, device, n, seqs, lens);
}
void pm_session_destroy(pm_session* s) { if (!s) return; for (int i = 0; i < s->n; i++) free(s->seq[i]); free(s->seq); free(s->len); free(s); }
int pm_session_genomes(const pm_session* s) { return s->n; }

int pm_multi_mum_batch(pm_session* s, int64_t nreg, const int64_t* starts, const int64_t* lens, const int32_t* minsize, pm_result** out) {
    int n = s->n, q = n - 1;
    pm_result* r = calloc(1, sizeof *r);
    r->nreg = nreg; r->nq = q; r->off = calloc((size_t)nreg + 1, sizeof(int64_t));
    int64_t cap = 0;
    const uint8_t** ptr = malloc(sizeof(*ptr) * (size_t)n);
    for (int64_t x = 0; x < nreg; x++) {
        for (int g = 0; g < n; g++) {
            int64_t st = starts[x * n + g], ln = lens[x * n + g];
            if (st < 0 || ln < 0 || st + ln > s->len[g]) { g_err = "region out of range"; return PM_EINVAL; }
            ptr[g] = s->seq[g] + st;
        }
        int64_t c, *k, *sp; int32_t* lon; uint8_t* fw;
        int ml = minsize[x] < 1 ? 1 : minsize[x];
        if (oracle_multi_mum(n, ptr, lens + x * n, minsize[x], ml, &c, &k, &lon, &sp, &fw, NULL, NULL)) { g_err = "oracle failed"; return PM_ENOMEM; }
        int64_t base = r->off[x];
        if (base + c > cap) {
            cap = (base + c) * 2 + 16;
            r->k = realloc(r->k, sizeof(int32_t) * (size_t)cap); r->lon = realloc(r->lon, sizeof(int32_t) * (size_t)cap);
            r->sp = realloc(r->sp, sizeof(int32_t) * (size_t)(cap * (q > 0 ? q : 1))); r->fwd = realloc(r->fwd, (size_t)(cap * (q > 0 ? q : 1)));
        }
        for (int64_t i = 0; i < c; i++) { r->k[base + i] = (int32_t)k[i]; r->lon[base + i] = lon[i]; }
        for (int64_t i = 0; i < c * q; i++) r->sp[base * q + i] = (int32_t)sp[i];
        memcpy(r->fwd + base * q, fw, (size_t)(c * q));
        r->off[x + 1] = base + c;
        oracle_free(k); oracle_free(lon); oracle_free(sp); oracle_free(fw);
    }
    free(ptr);
    r->total = r->off[nreg]; *out = r; return PM_OK;
}
int64_t pm_result_regions(const pm_result* r) { return r->nreg; }
int64_t pm_result_total(const pm_result* r) { return r->total; }
const int64_t* pm_result_offsets(const pm_result* r) { return r->off; }
const int32_t* pm_result_k(const pm_result* r) { return r->k; }
const int32_t* pm_result_lon(const pm_result* r) { return r->lon; }
const int32_t* pm_result_sp(const pm_result* r) { return r->sp; }
const uint8_t* pm_result_fwd(const pm_result* r) { return r->fwd; }
void pm_result_free(pm_result* r) { if (!r) return; free(r->off); free(r->k); free(r->lon); free(r->sp); free(r->fwd); free(r); }

int pm_find_events(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int32_t min_len, int strand,
                   int64_t cap, int64_t* count, int64_t* ev_j, int64_t* ev_l, int32_t* ev_len, int32_t* ev_rep) {
    uint8_t* q = malloc((size_t)m + 1);
    for (int64_t i = 0; i < m; i++) {
        uint8_t c = strand ? query[m - 1 - i] : query[i];
        if (strand) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
        q[i] = c;
    }
    int64_t c = oracle_events(ref, n, q, m, min_len < 1 ? 1 : min_len, cap, ev_j, ev_l, ev_len, ev_rep);
    free(q);
    if (c < 0) return PM_ENOMEM;
    *count = c;
    return PM_OK;
}
int pm_mumi_coverage(pm_session* s, const int64_t* starts, const int64_t* lens, int64_t* covered) {
    for (int g = 1; g < s->n; g++) {
        int64_t c = oracle_mumi_coverage(s->seq[0] + starts[0], lens[0], s->seq[g] + starts[g], lens[g], 15);
        if (c < 0) return PM_ENOMEM;
        covered[g - 1] = c;
    }
    return PM_OK;
}
/* MUM-row mode is a device feature of the HIP engine; this checker keeps to sp / fwd and says so */
int pm_session_rows(pm_session* s, int enable) { (void)s; return enable ? PM_EINVAL : PM_OK; }
int32_t* pm_result_start(pm_result* r) { (void)r; return 0; }
uint8_t* pm_result_strand(pm_result* r) { (void)r; return 0; }
const uint32_t* pm_result_flags(const pm_result* r) { (void)r; return 0; }
int pm_result_dirty_known(const pm_result* r) { (void)r; return 0; }
int64_t pm_result_table_id(const pm_result* r) { (void)r; return 0; }
int pm_session_tune(pm_session* s, const char* key, int64_t value) { (void)s; (void)key; (void)value; return PM_OK; }      /* nothing to tune here */
/* the device gap aligner belongs to the HIP library; this checker declines every job, so the host aligner runs */
int pm_gap_align_batch(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                       const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols) {
    (void)device; (void)n_seqs; (void)seq_off; (void)chars; (void)max_cols; (void)row_off; (void)out_rows; (void)out_bytes;
    for (int64_t j = 0; j < n_jobs; j++) cols[j] = -1;
    return PM_OK;
}
int pm_gap_align_groups(int device, int64_t n_jobs, const int32_t* n_seqs, const int64_t* seq_off, const uint8_t* chars,
                        const int32_t* max_cols, const int64_t* row_off, uint8_t* out_rows, int64_t out_bytes, int32_t* cols,
                        int n_groups, const int64_t* group_end, void (*done)(void* ctx, int group), void* ctx) {
    (void)device; (void)n_seqs; (void)seq_off; (void)chars; (void)max_cols; (void)row_off; (void)out_rows; (void)out_bytes; (void)group_end;
    for (int64_t j = 0; j < n_jobs; j++) cols[j] = -1;
    for (int g = 0; done && g < n_groups; g++) done(ctx, g);
    return PM_OK;
}
const char* pm_gap_last_error(void) { return ""; }
int pm_warmup(int device) { (void)device; return PM_OK; }
/* RCCL sessions exist in the HIP library only */
int pm_rccl_unique_id(uint8_t* id) { (void)id; return PM_EINVAL; }
int pm_session_rccl_ranks(const pm_session* s) { (void)s; return 0; }
int pm_session_create_rccl(pm_session** out, int device, int n_genomes, const uint8_t* const* seqs, const int64_t* lens, int rank, int world, const uint8_t* id) {
    (void)out; (void)device; (void)n_genomes; (void)seqs; (void)lens; (void)rank; (void)world; (void)id; return PM_EINVAL;
}
int pm_last_timing(const pm_session* s, int* count, const char** names, float* ms) { (void)s; (void)names; (void)ms; if (count) *count = 0; return PM_OK; }

/* the resident route (pm_store_*) is the HIP engine's: this provider has no MUM store, its callers take the host route */
int64_t pm_result_store_base(const pm_result* r) { (void)r; return -1; }
int pm_store_settle(pm_session* s, int64_t table_id, pm_row_info* rows) { (void)s; (void)table_id; (void)rows; return PM_EINVAL; }
int pm_store_info(pm_session* s, int64_t first, int64_t count, pm_row_info* out) { (void)s; (void)first; (void)count; (void)out; return PM_EINVAL; }
int pm_store_seeds(pm_session* s, int64_t table_id, const int32_t* anchors, int64_t n_anchors, int32_t q, int64_t* n_regions) { (void)s; (void)table_id; (void)anchors; (void)n_anchors; (void)q; (void)n_regions; return PM_EINVAL; }
const pm_region_info* pm_store_new_regions(const pm_session* s) { (void)s; return 0; }
const int32_t* pm_store_new_region_ids(const pm_session* s) { (void)s; return 0; }
int pm_store_regions_equal(pm_session* s, const int32_t* a, const int32_t* b, int64_t n, uint8_t* same) { (void)s; (void)a; (void)b; (void)n; (void)same; return PM_EINVAL; }
int pm_store_search(pm_session* s, const int32_t* regions, const int32_t* minsize, int64_t n, int64_t* first_row, int64_t* offsets) { (void)s; (void)regions; (void)minsize; (void)n; (void)first_row; (void)offsets; return PM_EINVAL; }
int pm_store_validate(pm_session* s, const int32_t* regions, const int64_t* row_first, const int32_t* row_count, int64_t n_regions, const int64_t* cluster_first, int64_t n_clusters, int32_t q, uint32_t* trouble, int64_t* n_children,
                      int64_t info_first, int64_t info_count, pm_row_info* info, int64_t stage_first, int32_t* second_stage_ran, int32_t generation, int32_t* done) {
    (void)s; (void)generation; (void)done; (void)stage_first; (void)second_stage_ran; (void)regions; (void)row_first; (void)row_count; (void)n_regions; (void)cluster_first; (void)n_clusters; (void)q; (void)trouble; (void)n_children; (void)info_first; (void)info_count; (void)info; return PM_EINVAL;
}
int pm_store_settle_seeds(pm_session* s, int64_t table_id, int32_t q, pm_row_info* rows, int64_t* n_regions) { (void)s; (void)table_id; (void)q; (void)rows; (void)n_regions; return PM_EINVAL; }
int pm_store_order_check(pm_session* s, uint32_t* trouble) { (void)s; (void)trouble; return PM_EINVAL; }
int pm_store_chain_begin(pm_session* s, int64_t n_expected, int32_t d, float diag_diff, int64_t c) { (void)s; (void)n_expected; (void)d; (void)diag_diff; (void)c; return PM_EINVAL; }
int pm_store_chain_end(pm_session* s, pm_chain_info* info, const int32_t** rows, const uint8_t** heads) { (void)s; (void)info; (void)rows; (void)heads; return PM_EINVAL; }
int pm_store_judge(pm_session* s, const int32_t* cur, const int32_t* back, int64_t n, int32_t d, int32_t* min_gap, int32_t* max_gap, uint8_t* verdict) { (void)s; (void)cur; (void)back; (void)n; (void)d; (void)min_gap; (void)max_gap; (void)verdict; return PM_EINVAL; }
int pm_store_unmark(pm_session* s, const int32_t* rows, int64_t n) { (void)s; (void)rows; (void)n; return PM_EINVAL; }
int pm_store_fill(pm_session* s, const int32_t* last_of, const int32_t* first_of_next, int64_t n, uint8_t* add) { (void)s; (void)last_of; (void)first_of_next; (void)n; (void)add; return PM_EINVAL; }
const int64_t* pm_store_fill_starts(const pm_session* s) { (void)s; return 0; }
const int64_t* pm_store_fill_ends(const pm_session* s) { (void)s; return 0; }
int pm_store_rows(pm_session* s, const int32_t* rows, int64_t first, int64_t n, int raw, int32_t* start, uint8_t* strand) { (void)s; (void)rows; (void)first; (void)n; (void)raw; (void)start; (void)strand; return PM_EINVAL; }
int64_t pm_store_layout_words(pm_session* s, int64_t* word_off) { (void)s; (void)word_off; return 0; }
int pm_store_layout(pm_session* s, uint64_t* out, int64_t words) { (void)s; (void)out; (void)words; return PM_EINVAL; }
int pm_session_traffic(const pm_session* s, uint64_t* h2d_bytes, uint64_t* d2h_bytes) { (void)s; if (h2d_bytes) *h2d_bytes = 0; if (d2h_bytes) *d2h_bytes = 0; return PM_OK; }
