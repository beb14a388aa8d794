// oracle/calc_harness.cpp -- TEST INFRASTRUCTURE ONLY.
// Prints minsize = int(ceil(Calculator(Converter(expr), S))) exactly as Aligner::setMums1 does
// (src/parsnp.cpp:1502-1514), using the REFERENCE's Converter.cpp compiled where it lies.
// usage: calc_ref '<expr>' S [S ...]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include "Converter.h"
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    for (int i = 2; i < argc; i++) {
        std::string out;
        Converter(std::string(argv[1]), out, 80);
        float limit = Calculator(out, out.length(), (float)atol(argv[i]));
        printf("%s %d\n", argv[i], int(ceil(limit)));
    }
    return 0;
}
