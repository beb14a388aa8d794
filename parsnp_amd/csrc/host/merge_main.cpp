// merge_main.cpp -- parsnp_merge: the merge step of partition mode as an executable (include/parsnp_merge.h).
//   parsnp_merge [-t threads] [-m min_interval_size] [--keep-trimmed] <out.xmfa> <partition1.xmfa> <partition2.xmfa> ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/parsnp_merge.h"

int main(int argc, char** argv) {
    int threads = 1, keep = 0; long min_size = 10;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-t") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-m") && i + 1 < argc) min_size = atol(argv[++i]);
        else if (!strcmp(argv[i], "--keep-trimmed")) keep = 1;
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) { fprintf(stderr, "usage: parsnp_merge [-t threads] [-m min_interval_size] [--keep-trimmed] <out.xmfa> <partition.xmfa>...\n"); return 2; }
    long clusters = 0, sequences = 0, bases = 0;
    char err[512] = "";
    const int rc = parsnp_partition_merge((int)pos.size() - 1, pos.data() + 1, pos[0], min_size, threads, keep, &clusters, &sequences, &bases, err, sizeof err);
    if (rc) { fprintf(stderr, "parsnp_merge: %s\n", err); return 1; }
    printf("%ld reference bases over %ld clusters, %ld sequences\n", bases, clusters, sequences);
    return 0;
}
