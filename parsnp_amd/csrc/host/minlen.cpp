#include "minlen.h"

#include <cctype>
#include <cmath>
#include <cstdlib>

namespace parsnp {
namespace {
struct Parser {
    const char* s;
    float S;
    bool bad = false;
    void ws() { while (*s == ' ' || *s == '\t') s++; }
    float base() {
        ws();
        if (*s == '(') {
            s++;
            float v = sum();
            ws();
            if (*s == ')') s++; else bad = true;
            return v;
        }
        if (isdigit((unsigned char)*s)) {   // digits with embedded '.', as Calculator scans them (Converter.cpp:195-207)
            const char* b = s;
            while (isdigit((unsigned char)*s)) { s++; if (*s == '.') s++; }
            return (float)atof(std::string(b, s).c_str());
        }
        if (*s == 'S' || *s == 's') { s++; return S; }
        if (s[0] == 'L' && s[1] == 'o' && s[2] == 'g') {
            s += 3; ws();
            if (*s != '(') { bad = true; return 0; }
            float x = base();
            return (float)((double)logf(x) / log(2.0));
        }
        bad = true;
        return 0;
    }
    float power() { float v = base(); for (ws(); *s == '^'; ws()) { s++; float x = base(); v = powf(v, x); } return v; }
    float product() {
        float v = power();
        for (ws(); *s == '*' || *s == '/'; ws()) {
            char op = *s++; float x = power();
            if (op == '*') v = v * x; else if (x == 0) bad = true; else v = v / x;
        }
        return v;
    }
    float sum() {
        float v = product();
        for (ws(); *s == '+' || *s == '-'; ws()) { char op = *s++; float x = product(); v = op == '+' ? v + x : v - x; }
        return v;
    }
};
}  // namespace

bool min_mum_length(const std::string& expr, long S, int* out) {
    Parser p{expr.c_str(), (float)S};
    float v = p.sum();
    p.ws();
    if (p.bad || *p.s) return false;
    float limit = ceilf(v);              // Calculator's own ceil (Converter.cpp:283-284)
    *out = int(ceil(limit));             // setMums1's ceil + int (parsnp.cpp:1506)
    return true;
}
}  // namespace parsnp
