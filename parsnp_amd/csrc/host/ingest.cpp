// ingest.cpp -- FASTA -> in-memory genome, behaviour of the reference's main() loop (src/parsnp.cpp:2913-3160).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>

#include "aligner.h"

namespace parsnp {

namespace {
// what one sequence byte does (switch at parsnp.cpp:3003-3132, evaluated on toupper(ch))
enum Act : uint8_t { SKIP = 0, BASE_A, BASE_C, BASE_G, BASE_T, BASE_U, BASE_N, HEADER };
struct Table {
    uint8_t act[256];
    Table() {
        memset(act, SKIP, sizeof act);
        for (int c = 0; c < 256; c++) {
            switch (toupper(c)) {
                case 'A': act[c] = BASE_A; break;
                case 'C': act[c] = BASE_C; break;
                case 'G': act[c] = BASE_G; break;
                case 'T': act[c] = BASE_T; break;
                case 'U': act[c] = BASE_U; break;
                case 'X': case 'Y': case 'S': case 'W': case 'K': case 'H': case 'R': case 'M': case 'V': case 'D':
                case 'B': case '-': case 'N': act[c] = BASE_N; break;
                case '>': act[c] = HEADER; break;
                default: break;   // '\n', '\t', ' ' and every other byte are dropped
            }
        }
    }
};
const Table kTable;
}  // namespace

bool ingest(const std::string& path, bool is_ref, bool reverse, int d, Genome* g, std::string* console) {
    std::ostringstream con;
    g->path = path;
    size_t slash = path.rfind('/');
    g->fname = slash == std::string::npos ? path : path.substr(slash + 1);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (is_ref) con << " Cannot open reference file ! " << std::endl;
        else con << " Cannot open query file: " << path << std::endl;
        *console = con.str();
        return false;
    }
    std::string data;
    {   // the whole file in one read (a stream iterator costs more per byte than the parse below)
        struct Close { FILE* f; ~Close() { fclose(f); } } closer{f};
        long size = (fseek(f, 0, SEEK_END) == 0) ? ftell(f) : -1;
        if (size >= 0 && fseek(f, 0, SEEK_SET) == 0) {
            data.resize((size_t)size);
            size_t got = size ? fread(&data[0], 1, (size_t)size, f) : 0;
            data.resize(got);
        } else {            // not seekable: read in pieces
            char buf[1 << 16];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
        }
    }

    const size_t kLine = 2499;   // getline(buf, 2500): at most 2499 bytes, longer lines put the stream in fail state
    size_t pos = 0;
    bool failed = false;
    auto read_line = [&](std::string* out) {
        size_t nl = data.find('\n', pos);
        size_t end = nl == std::string::npos ? data.size() : nl;
        if (end - pos > kLine) { if (out) *out = data.substr(pos, kLine); failed = true; pos += kLine; return; }
        if (out) *out = data.substr(pos, end - pos);
        if (nl == std::string::npos) { if (end == pos) failed = true; pos = data.size(); }   // EOF with nothing read: failbit
        else pos = nl + 1;
    };
    read_line(&g->header);
    g->pos2hdr.clear();
    g->pos2hdr[1] = "s1";

    // what a byte appends (0: nothing) and which counter it feeds; headers are the only bytes that need more than that
    unsigned char emit[256]; uint8_t cls[256];
    for (int ch = 0; ch < 256; ch++) {
        cls[ch] = kTable.act[ch];
        switch (kTable.act[ch]) {
            case BASE_A: emit[ch] = reverse ? 'T' : 'A'; break;
            case BASE_C: emit[ch] = reverse ? 'G' : 'C'; break;
            case BASE_G: emit[ch] = reverse ? 'C' : 'G'; break;
            case BASE_T: emit[ch] = reverse ? 'A' : 'T'; break;
            case BASE_U: emit[ch] = 'T'; break;                 // not complemented (parsnp.cpp:3066-3069)
            case BASE_N: emit[ch] = 'N'; break;
            default: emit[ch] = 0; break;
        }
    }
    long long count[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // per Act
    int padding = 0;
    unsigned seqcount = 1;
    std::string& s = g->seq;
    s.clear();
    s.resize(data.size() > pos ? data.size() - pos : 0);      // a byte appends at most one base; contig padding grows it below
    size_t w = 0;
    const unsigned char* in = (const unsigned char*)data.data();
    // Sequence bytes between two header lines go through a loop without branches: the byte's emission is stored and the
    // write index moves on if it was one; what a byte is (for the base counts) comes out of a histogram of the raw bytes
    // afterwards.  (Four histograms: consecutive bytes are often equal, and one counter would wait for its own store.)
    uint32_t hist[4][256];
    memset(hist, 0, sizeof hist);
    auto flush_hist = [&] {
        for (int ch = 0; ch < 256; ch++) { count[cls[ch]] += (long long)hist[0][ch] + hist[1][ch] + hist[2][ch] + hist[3][ch]; }
        memset(hist, 0, sizeof hist);
    };
    while (pos < data.size() && !failed) {
        const void* hit = memchr(in + pos, '>', data.size() - pos);
        const size_t stop = hit ? (size_t)((const unsigned char*)hit - in) : data.size();
        char* out = &s[0];
        size_t x = pos;
        for (; x + 4 <= stop; x += 4) {
            const unsigned char c0 = in[x], c1 = in[x + 1], c2 = in[x + 2], c3 = in[x + 3];
            hist[0][c0]++; hist[1][c1]++; hist[2][c2]++; hist[3][c3]++;
            const unsigned char e0 = emit[c0], e1 = emit[c1], e2 = emit[c2], e3 = emit[c3];
            out[w] = (char)e0; w += e0 != 0;
            out[w] = (char)e1; w += e1 != 0;
            out[w] = (char)e2; w += e2 != 0;
            out[w] = (char)e3; w += e3 != 0;
        }
        for (; x < stop; x++) { const unsigned char c = in[x]; hist[0][c]++; const unsigned char e = emit[c]; out[w] = (char)e; w += e != 0; }
        pos = stop;
        if (!hit) break;
        pos++;                                                 // the '>' itself: a header line follows
        read_line(nullptr);
        flush_hist();
        if (!is_ref) {                                         // :3114-3118
            const size_t pad = (size_t)(d + 10);
            if (s.size() < w + pad + (data.size() - pos) + 1) s.resize(w + pad + (data.size() - pos) + 1);
            memset(&s[w], 'N', pad); w += pad;
            count[BASE_N] += d + 10; padding += d + 10;
        }
        seqcount++;
        g->pos2hdr[(int)(count[BASE_N] + count[BASE_C] + count[BASE_T] + count[BASE_U] + count[BASE_A] + count[BASE_G])] = "s" + std::to_string(seqcount);
    }
    flush_hist();
    s.resize(w);
    const long long a = count[BASE_A], c = count[BASE_C], gg = count[BASE_G], t = count[BASE_T] + count[BASE_U], nn = count[BASE_N];
    if (reverse) std::reverse(s.begin(), s.end());
    g->size_nopad = (int)s.size() - padding;
    g->gc = float(gg) + float(c);
    g->at = float(a) + float(t);
    con << g->fname << ",Len:" << s.size() << ",GC:" << ((float(gg) + float(c)) / float(s.size() - nn)) * 100 << std::endl;
    *console = con.str();
    return true;
}

std::string reverse_complement(const std::string& in) {
    std::string out;
    out.reserve(in.size());
    for (char ch : in) {
        switch (toupper((unsigned char)ch)) {
            case 'A': out.push_back('T'); break;
            case 'C': out.push_back('G'); break;
            case 'G': out.push_back('C'); break;
            case 'T': case 'U': out.push_back(toupper((unsigned char)ch) == 'T' ? 'A' : 'T'); break;
            case '\r': case '\n': case '\t': case ' ': case '>': case '.': break;   // dropped (parsnp.cpp:1369-1380)
            default: out.push_back('N'); break;
        }
    }
    std::reverse(out.begin(), out.end());
    return out;
}

}  // namespace parsnp
