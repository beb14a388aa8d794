// ini.h -- reader for the parsnp_core parameter file.
// Grammar and lookup rules follow the reference's CIniFile (src/ext/iniFile.cpp:35-97 ReadFile,
// :120-149 FindKey/FindValue, :201-222 SetValue, :263-289 GetValue*, :476-483 CheckCase):
//   * a line is classified by the FIRST of ';', '#', '[', '=' found in it;
//   * '[' ... last ']' opens a section, '=' splits "name=value" with NO trimming;
//   * section and value names compare case-insensitively; a repeated name overwrites;
//   * a trailing '\r' is dropped; a line whose first byte is not printable aborts the read.
#pragma once
#include <string>
#include <vector>

namespace parsnp {

class IniFile {
public:
    bool read(const std::string& path);
    std::string get(const std::string& section, const std::string& name, const std::string& def = "") const;
    int get_int(const std::string& section, const std::string& name, int def = 0) const;      // atoi, iniFile.cpp:274-280
    double get_double(const std::string& section, const std::string& name, double def = 0.0) const;  // atof, :282-289
    bool get_bool(const std::string& section, const std::string& name, bool def = false) const { return get_int(section, name, int(def)) != 0; }
    unsigned count(const std::string& section) const;   // NumValues, iniFile.cpp:173-179

private:
    struct Section { std::string name; std::vector<std::string> names, values; };
    std::vector<Section> sections_;
    long find_section(const std::string& name) const;
    static std::string lower(std::string s);
};

}  // namespace parsnp
