#include "ini.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>

namespace parsnp {

std::string IniFile::lower(std::string s) {
    for (auto& c : s) c = (char)tolower((unsigned char)c);
    return s;
}

long IniFile::find_section(const std::string& name) const {
    const std::string want = lower(name);
    for (size_t i = 0; i < sections_.size(); i++)
        if (lower(sections_[i].name) == want) return (long)i;
    return -1;
}

bool IniFile::read(const std::string& path) {
    std::ifstream f(path.c_str());
    if (!f) return false;
    std::string line;
    long cur = -1;   // section that '=' lines are filed under
    std::string cur_name;
    while (std::getline(f, line)) {
        if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
        if (line.empty()) continue;
        if (!isprint((unsigned char)line[0])) {
            printf("Failing on char %d\n", line[0]);
            return false;
        }
        size_t at = line.find_first_of(";#[=");
        if (at == std::string::npos) continue;
        if (line[at] == '[') {
            size_t close = line.find_last_of(']');
            if (close != std::string::npos && close > at) {
                cur_name = line.substr(at + 1, close - at - 1);
                sections_.push_back(Section{cur_name, {}, {}});   // AddKeyName never merges duplicates
            }
        } else if (line[at] == '=') {
            // SetValue(keyname, ...) looks the section up by name (first match), creating it if absent
            cur = find_section(cur_name);
            if (cur < 0) { sections_.push_back(Section{cur_name, {}, {}}); cur = (long)sections_.size() - 1; }
            Section& s = sections_[(size_t)cur];
            const std::string name = line.substr(0, at), value = line.substr(at + 1), lname = lower(name);
            size_t i = 0;
            for (; i < s.names.size(); i++) if (lower(s.names[i]) == lname) break;
            if (i == s.names.size()) { s.names.push_back(name); s.values.push_back(value); }
            else s.values[i] = value;
        }
        // ';' and '#': comments
    }
    return !sections_.empty();
}

std::string IniFile::get(const std::string& section, const std::string& name, const std::string& def) const {
    long k = find_section(section);
    if (k < 0) return def;
    const Section& s = sections_[(size_t)k];
    const std::string lname = lower(name);
    for (size_t i = 0; i < s.names.size(); i++) if (lower(s.names[i]) == lname) return s.values[i];
    return def;
}

int IniFile::get_int(const std::string& section, const std::string& name, int def) const {
    char buf[64]; snprintf(buf, sizeof buf, "%d", def);
    return atoi(get(section, name, buf).c_str());
}

double IniFile::get_double(const std::string& section, const std::string& name, double def) const {
    char buf[64]; snprintf(buf, sizeof buf, "%f", def);
    return atof(get(section, name, buf).c_str());
}

unsigned IniFile::count(const std::string& section) const {
    long k = find_section(section);
    return k < 0 ? 0u : (unsigned)sections_[(size_t)k].names.size();
}

}  // namespace parsnp
