// cli_util.h — the two small pieces of the `match` command line that mirror reference code: the flat token search of
// matching/argparser.h:5-24 and a reader for the flat JSON object of afis.config (main.cpp:41-44 reads it with nlohmann/json).
// Shared by match_main.cpp and match_selftest.cpp (the CPU-side checks of tests/ pin both against the reference's own headers).
#pragma once
#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace afis {

// argparser.h:5-24 — flat token search
struct ArgParser {
    std::vector<std::string> tokens;
    ArgParser(int argc, char** argv) { for (int i = 1; i < argc; ++i) tokens.push_back(argv[i]); }
    const std::string& getCmdOption(const std::string& option) const
    {
        static const std::string empty;
        auto it = std::find(tokens.begin(), tokens.end(), option);
        if (it != tokens.end() && ++it != tokens.end()) return *it;
        return empty;
    }
    bool cmdOptionExists(const std::string& option) const { return std::find(tokens.begin(), tokens.end(), option) != tokens.end(); }
};

// afis.config is a flat JSON object of string values (afis.config:1-17); this reads exactly that.
inline std::map<std::string, std::string> read_flat_json(const std::string& path)
{
    std::map<std::string, std::string> kv;
    std::ifstream in(path);
    if (!in) return kv;
    std::stringstream ss; ss << in.rdbuf();
    const std::string s = ss.str();
    size_t i = 0;
    auto read_string = [&](std::string& out) -> bool {
        while (i < s.size() && s[i] != '"') ++i;
        if (i >= s.size()) return false;
        ++i; out.clear();
        while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) ++i; out.push_back(s[i++]); }
        ++i;
        return true;
    };
    std::string k, v;
    while (read_string(k)) {
        while (i < s.size() && s[i] != ':' ) ++i;
        if (i >= s.size()) break;
        ++i;
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        if (i < s.size() && s[i] == '"') { if (!read_string(v)) break; kv[k] = v; }
    }
    return kv;
}

}  // namespace afis
