// oracle/muscle_harness.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the REFERENCE's inter-MUM gap aligner exactly as Aligner::writeOutput does (src/parsnp.cpp:854-855 ->
// src/MuscleInterface.cpp:37-78 -> vendored libMUSCLE 3.7), compiled from the sources where they lie.
// stdin: blocks of sequences, one per line, blocks separated by an empty line.  stdout: the aligned rows of every block
// in input order, blocks separated by an empty line.
#include <iostream>
#include <string>
#include <vector>
#include "MuscleInterface.h"
int main() {
    std::vector<std::string> block;
    std::string line;
    auto flush = [&]() {
        if (block.empty()) return;
        std::vector<std::string> out;
        MuscleInterface gmi = MuscleInterface();
        gmi.CallMuscleFast(out, block);
        for (auto& r : out) std::cout << r << "\n";
        std::cout << "\n";
        block.clear();
    };
    while (std::getline(std::cin, line)) {
        if (line.empty()) flush(); else block.push_back(line);
    }
    flush();
    return 0;
}
