// minlen.h -- minimum MUM length from the ini expression, e.g. "1.1*(Log(S))".
// Same arithmetic as the reference's Converter()/Calculator() pair (src/Converter.cpp:11-286) as called from
// Aligner::setMums1 (src/parsnp.cpp:1502-1514): every operand and intermediate is float32, Log(x) is
// float(double(logf(x))/log(2.0)) (Converter.cpp:268-270), the result is ceil()ed.  Supports the well-formed
// infix subset (numbers, S, + - * / ^, parentheses, Log(...)); anything else is reported as an error
// instead of reproducing the reference's undefined stack behaviour.
#pragma once
#include <string>

namespace parsnp {
// returns false on a malformed expression
bool min_mum_length(const std::string& expr, long S, int* out);
}
