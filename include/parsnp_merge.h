/* parsnp_merge.h -- C ABI of partition mode's merge step.
 *
 * Replaces, in the reference driver, the four partition.py calls between "Computing intersection of all partition
 * LCBs..." and the merged parsnp.xmfa (/root/reference/parsnp:1601-1615):
 *     partition.get_chunked_intervals      partition.py:507-536
 *     partition.get_intersected_intervals  partition.py:539-583
 *     partition.trim_xmfas                 partition.py:586-648   (trim :99-216)
 *     partition.merge_xmfas                partition.py:683-736   (merge_blocks :320-433, headers :245-318)
 * Host code (no GPU): implemented in parsnp_amd/csrc/host/partition_merge.cpp, exported by libparsnp_core.so and by the
 * parsnp_merge executable.  INTEGRATION.md shows the ctypes stub a maintainer of the reference driver would add.
 */
#ifndef PARSNP_MERGE_H
#define PARSNP_MERGE_H
#ifdef __cplusplus
extern "C" {
#endif

/* xmfa_paths[n_xmfas]: the alignment of every partition that ran to completion, in chunk-label order (the order in which
 *     the reference sorts its good chunks, parsnp:1599).
 * out_path: the merged alignment (<outdir>/parsnp.xmfa).
 * min_interval_size: intersected reference intervals shorter than this are dropped (the reference's default: 10).
 * keep_trimmed != 0: also write <xmfa>.trimmed next to every input, as trim_single_xmfa does.
 * clusters / sequences / ref_bases (each may be NULL): blocks and sequences of the merged alignment, reference bases
 *     covered by the intersection.
 * Returns 0, or 1 with a message in err[err_cap] (malformed input; partitions that disagree after trimming --
 * partition.py:644-646 raises there too). */
int parsnp_partition_merge(int n_xmfas, const char* const* xmfa_paths, const char* out_path, long min_interval_size, int threads,
                           int keep_trimmed, long* clusters, long* sequences, long* ref_bases, char* err, long err_cap);

/* Where the last merge OF THE CALLING THREAD could depend on the insertion aligner (the counts are kept per call: merges may run
 * side by side in one process) (the reference re-aligns runs of columns that are
 * insertions relative to the reference with spoa.poa, partition.py:386; this library with the gap aligner of the XMFA writer):
 * runs of such columns; those that collected bases from more than one sequence (a single sequence is its own alignment either
 * way); those among them whose sequences are not all the same string; and the merged columns of the shared runs. */
void parsnp_partition_merge_insertions(long* runs, long* shared, long* shared_diverse, long* shared_columns);

#ifdef __cplusplus
}
#endif
#endif
