// tests/emu/steps_main.cpp -- TEST INFRASTRUCTURE ONLY.
// parsnp_core's phases A-D run several times in ONE process (what bench.py and the C API do with a resident session), then
// the output once: the driver of the sanitizer runs (scripts/sanitize_host.sh) of the host code over the kernel emulation.
//   steps_main <file.ini> [steps]
#include <cstdio>
#include <cstdlib>

#include "../../parsnp_amd/csrc/host/core.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: steps_main <file.ini> [steps]\n"); return 2; }
    const int steps = argc > 2 ? atoi(argv[2]) : 3;
    parsnp::CoreRun run;
    int rc = run.open(argv[1]);
    if (rc) return rc;
    parsnp::StepReport rep;
    for (int i = 0; i < steps; i++) rep = run.step();
    bool note = false;
    run.write(&note);
    fprintf(stderr, "steps_main: %d steps, %ld anchors, %ld MUMs, %ld LCBs, resident route %ld (left and repeated on the host route: %ld)\n", steps, rep.anchors, rep.mums, rep.lcbs,
            rep.host.resident, rep.host.resident_retry);
    return 0;
}
