// gapalign.h -- the inter-MUM gap aligner of the XMFA writer.
// The reference hands every multi-column gap between two adjacent MUMs of an LCB to its vendored libMUSCLE 3.7
// (src/parsnp.cpp:854-855 -> src/MuscleInterface.cpp:37-78).  gap_align() restates that one configuration of MUSCLE
// (DNA, one iteration, stable order, ClustalW weights) in the same float32 arithmetic, so the rows it returns are the
// rows the reference writes.
#pragma once
#include <string>
#include <vector>

namespace parsnp {

// seqs: >= 2 non-empty sequences.  rows: one aligned row per input sequence, in input order, all of one length.
// Returns false (rows untouched) for input the aligner does not take (fewer than two sequences, an empty one).
bool gap_align(const std::vector<std::string>& seqs, std::vector<std::string>* rows);

}  // namespace parsnp
