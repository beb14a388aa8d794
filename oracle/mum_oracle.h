/* oracle/mum_oracle.h -- TEST INFRASTRUCTURE ONLY (see mum_oracle.c). */
#ifndef MUM_ORACLE_H
#define MUM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* rep[l] = length of the longest prefix of ref[l..n) that also occurs at another position of ref. */
int oracle_rep(const uint8_t* ref, int64_t n, int32_t* rep);

/* Raw per-reference-position (UP,EP,SP) of one query strand == Find_UM (src/csgmum/mum.c:177-250):
 * R-unique maximal exact matches of length >= min_len, applied with Test_UM (mum.c:27-45) in query order.
 * Arrays have n entries and are fully overwritten (0 where no event). */
int oracle_find_um(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_len,
                   int32_t* UP, int32_t* EP, int64_t* SP);

/* The forward carry of Intersect_UM (mum.c:125-175) on one strand's raw arrays, in place (top-2 fold form). */
int oracle_propagate(int64_t n, int32_t* UP, int32_t* EP, int64_t* SP);

/* Events only: all R-unique MEMs (j,l,len) with len >= min_len, in query order. Returns count (<= cap written). */
int64_t oracle_events(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_len,
                      int64_t cap, int64_t* ev_j, int64_t* ev_l, int32_t* ev_len, int32_t* ev_rep);

/* One region of Aligner::setMums1 (src/parsnp.cpp:1570-1695): seqs[0] = reference substring, seqs[1..cnt-1] =
 * query substrings (ASCII ACGTN; reverse complements are derived here as Aligner::reversec does).
 * Candidates c = 0..*out_c-1: k[c], lon[c], sp[c*(cnt-1)+g], fwd[c*(cnt-1)+g] for query g = 0..cnt-2.
 * Outputs are malloc'ed; release with oracle_free. min_event_len = 1 reproduces the reference event stream,
 * min_event_len = minsize is the reduced stream the GPU uses (same candidate list). */
int oracle_multi_mum(int cnt, const uint8_t* const* seqs, const int64_t* lens, int minsize, int min_event_len,
                     int64_t* out_c, int64_t** out_k, int32_t** out_lon, int64_t** out_sp, uint8_t** out_fwd,
                     int32_t* masterUP, int32_t* masterEP);
void oracle_free(void* p);

/* calcmumi (Aligner::setMumi, src/parsnp.cpp:1977-2069): number of reference positions covered by the pairwise MUMs
 * (length >= 15) of one query genome against ref, both strands merged as Merge_Master does. */
int64_t oracle_mumi_coverage(const uint8_t* ref, int64_t n, const uint8_t* query, int64_t m, int min_event_len);

/* minsize = int(ceil(Calculator(Converter(expr), S))) for the expression forms the driver emits
 * (src/Converter.cpp:11-286, src/parsnp.cpp:1502-1514). Returns INT32_MIN on a parse error. */
int32_t oracle_min_length(const char* expr, int64_t S);

#ifdef __cplusplus
}
#endif
#endif
