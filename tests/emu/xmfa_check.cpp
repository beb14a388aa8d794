// tests/emu/xmfa_check.cpp -- TEST INFRASTRUCTURE ONLY.
// Size-independent self-check of an XMFA against the genomes it was made from, for files too large for the Python
// version (tests/xmfa_util.py::consistency, whose counts it reproduces; tests compare the two on small sets):
//   per block all rows have one length; MUM (lower-case) columns of the first row hold no gap and the same base in every
//   row; every record, gaps removed, spells genome[start-1:end] of the sequence its index names in the header
//   (reverse-complemented for '-' records).  Records whose block was overlap-trimmed by the writer
//   (src/parsnp.cpp:928-952 shifts the start by a column count, a reference quirk) are counted in `shifted`.
// --merged: the file is a partition merge -- a column that is a MUM column in one partition need not be one in another,
//   so a row is only held to the first row's base where the row itself is lower case.
// Usage: xmfa_check [--merged] [--intervals out.txt] <xmfa> <genome dir> [threads]
//   genomes: single-record FASTA files named as the header's ##SequenceFile entries.  Prints one JSON object.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct Rec { int idx; long a, b; char strand; const char* seq; size_t bytes; };      // seq: the lines, newlines included
struct Block { std::vector<Rec> recs; };

static std::string unwrap(const Rec& r) {
    std::string s; s.reserve(r.bytes);
    for (size_t i = 0; i < r.bytes; i++) if (r.seq[i] != '\n') s.push_back(r.seq[i]);
    return s;
}
static bool load_fasta(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::string raw((size_t)n, 0);
    if (fread(&raw[0], 1, (size_t)n, f) != (size_t)n) { fclose(f); return false; }
    fclose(f);
    out->clear(); out->reserve((size_t)n);
    size_t i = 0;
    while (i < raw.size()) {
        size_t e = raw.find('\n', i); if (e == std::string::npos) e = raw.size();
        if (raw[i] != '>') for (size_t k = i; k < e; k++) if (raw[k] != '\r') out->push_back((char)toupper((unsigned char)raw[k]));
        i = e + 1;
    }
    return true;
}
static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; } }

int main(int argc, char** argv) {
    bool merged = false; const char* ivout = nullptr;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--merged")) merged = true;
        else if (!strcmp(argv[i], "--intervals") && i + 1 < argc) ivout = argv[++i];
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) { fprintf(stderr, "usage: xmfa_check [--merged] [--intervals out.txt] <xmfa> <genome dir> [threads]\n"); return 2; }
    const int threads = pos.size() > 2 ? atoi(pos[2]) : 8;
    const int fd = open(pos[0], O_RDONLY);
    if (fd < 0) { fprintf(stderr, "cannot open %s\n", pos[0]); return 2; }
    struct stat st; fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const char* p0 = (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p0 == MAP_FAILED) { fprintf(stderr, "cannot map %s\n", pos[0]); return 2; }
    const char* p = p0; const char* const end = p0 + n;
    std::map<int, std::string> file_of;
    int cur_idx = 0;
    while (p < end && *p == '#') {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const std::string line(p, nl ? nl : end);
        if (!line.compare(0, 16, "##SequenceIndex ")) cur_idx = atoi(line.c_str() + 16);
        else if (!line.compare(0, 15, "##SequenceFile ")) file_of[cur_idx] = line.substr(15);
        p = nl ? nl + 1 : end;
    }
    std::vector<Block> blocks;
    Block cur;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (*p == '>') {
            Rec r; char strand = '+'; int idx = 0; long a = 0, b = 0;
            char hdr[256];      // (sscanf on the mapping itself would measure the rest of the file with strlen for every record)
            const size_t hl = std::min<size_t>((size_t)(le - p), sizeof hdr - 1);
            memcpy(hdr, p, hl); hdr[hl] = 0;
            if (sscanf(hdr, "> %d:%ld-%ld %c", &idx, &a, &b, &strand) != 4) { fprintf(stderr, "malformed record header at byte %zu\n", (size_t)(p - p0)); return 2; }
            r.idx = idx; r.a = a; r.b = b; r.strand = strand;
            const char* s = nl ? nl + 1 : end;
            const char* q = s;
            while (q < end && *q != '>' && *q != '=') { const char* e2 = (const char*)memchr(q, '\n', (size_t)(end - q)); q = e2 ? e2 + 1 : end; }
            r.seq = s; r.bytes = (size_t)(q - s);
            cur.recs.push_back(r);
            p = q;
            continue;
        }
        if (*p == '=' && !cur.recs.empty()) { blocks.push_back(std::move(cur)); cur = Block(); }
        p = le < end ? le + 1 : end;
    }
    if (!cur.recs.empty()) blocks.push_back(std::move(cur));
    // genomes
    std::vector<int> ids;
    for (auto& kv : file_of) ids.push_back(kv.first);
    std::map<int, std::string> genome;
    for (int id : ids) genome[id];
    long missing = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : missing)
    for (long i = 0; i < (long)ids.size(); i++)
        if (!load_fasta(std::string(pos[1]) + "/" + file_of[ids[(size_t)i]], &genome[ids[(size_t)i]])) missing++;
    long lcbs = 0, records = 0, bad_length = 0, bad_mum = 0, bad_seq = 0, shifted = 0, reverse = 0, ref_bases = 0, mum_cols = 0, min_rows = 1 << 30, max_rows = 0, no_genome = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : lcbs, records, bad_length, bad_mum, bad_seq, shifted, reverse, ref_bases, mum_cols, no_genome) reduction(min : min_rows) reduction(max : max_rows)
    for (long bi = 0; bi < (long)blocks.size(); bi++) {
        const Block& b = blocks[(size_t)bi];
        lcbs++;
        min_rows = std::min<long>(min_rows, (long)b.recs.size()); max_rows = std::max<long>(max_rows, (long)b.recs.size());
        std::vector<std::string> rows;
        for (const Rec& r : b.recs) rows.push_back(unwrap(r));
        bool one_len = true;
        for (auto& s : rows) one_len = one_len && s.size() == rows[0].size();
        if (!one_len) bad_length++;
        const std::string& first = rows[0];
        std::vector<size_t> lower;
        for (size_t i = 0; i < first.size(); i++) if (islower((unsigned char)first[i])) lower.push_back(i);
        mum_cols += (long)lower.size();
        for (const std::string& s : rows) {
            if (s.size() != first.size()) continue;
            bool bad = false;
            if (!merged) { for (size_t i : lower) if (!islower((unsigned char)s[i]) || s[i] != first[i]) { bad = true; break; } }
            else { for (size_t i : lower) if (islower((unsigned char)s[i]) && s[i] != first[i]) { bad = true; break; } }
            if (bad) bad_mum++;
        }
        for (size_t k = 0; k < b.recs.size(); k++) {
            const Rec& r = b.recs[k];
            records++;
            if (k == 0) ref_bases += r.b - (r.a - 1);
            auto it = genome.find(r.idx);
            if (it == genome.end() || it->second.empty()) { no_genome++; continue; }
            const std::string& g = it->second;
            std::string got;
            for (char c : rows[k]) if (c != '-') got.push_back((char)toupper((unsigned char)c));
            std::string want;
            if (r.a >= 1 && r.b <= (long)g.size() && r.b >= r.a - 1) want = g.substr((size_t)(r.a - 1), (size_t)(r.b - (r.a - 1)));
            if (r.strand == '-') { reverse++; std::reverse(want.begin(), want.end()); for (char& c : want) c = comp(c); }
            if (got != want) {
                const bool suffix = (want.size() >= got.size() && !want.compare(want.size() - got.size(), got.size(), got)) ||
                                    (got.size() >= want.size() && !got.compare(got.size() - want.size(), want.size(), want));
                if (suffix) shifted++; else bad_seq++;
            }
        }
    }
    if (ivout) {
        FILE* f = fopen(ivout, "w");
        if (f) { for (const Block& b : blocks) fprintf(f, "%ld %ld %c\n", b.recs[0].a, b.recs[0].b, b.recs[0].strand); fclose(f); }
    }
    if (blocks.empty()) { min_rows = 0; }
    printf("{\"lcbs\": %ld, \"records\": %ld, \"bad_length\": %ld, \"bad_mum_column\": %ld, \"bad_sequence\": %ld, \"shifted\": %ld, \"reverse\": %ld, "
           "\"ref_bases\": %ld, \"mum_columns\": %ld, \"min_rows\": %ld, \"max_rows\": %ld, \"sequences\": %zu, \"missing_genomes\": %ld, \"records_without_genome\": %ld}\n",
           lcbs, records, bad_length, bad_mum, bad_seq, shifted, reverse, ref_bases, mum_cols, min_rows, max_rows, file_of.size(), missing, no_genome);
    return 0;
}
