// configer.hpp -- setting.conf access with the reference's semantics (happy.coding.io.FileConfiger / LineConfiger over
// java.util.Properties): a value is split on [,\t ]; the first token not starting with '-' is the
// main parameter; a token starting with '-' that is NOT numeric opens an option key; other tokens append to the current
// key; isOn(v) <=> v in {on,true}; getPath(k) = k, else k.lins (k.wins on Windows).  Floats are Java floats promoted.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace carskit {

inline double java_float(const std::string &s) { return (double)std::strtof(s.c_str(), nullptr); }

inline bool is_numeric(const std::string &t) {
    if (t.empty()) return false;
    char *end = nullptr;
    std::strtod(t.c_str(), &end);
    return end && *end == '\0';
}

inline std::string lower(std::string s) {
    for (char &c : s)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return s;
}

class LineConfiger {
  public:
    LineConfiger() {}
    explicit LineConfiger(const std::string &line) {
        std::string tok;
        std::string cur;
        bool have_cur = false;
        auto flush = [&]() {
            if (tok.empty()) return;
            if (tok[0] == '-' && !is_numeric(tok)) {
                cur = tok;
                have_cur = true;
                params_[cur];
            } else if (!have_cur && !has_main_) {
                main_ = tok;
                has_main_ = true;
            } else if (have_cur) {
                params_[cur].push_back(tok);
            }
            tok.clear();
        };
        for (char c : line) {
            if (c == ',' || c == '\t' || c == ' ' || c == '\r' || c == '\n') flush();
            else tok.push_back(c);
        }
        flush();
    }
    bool hasMainParam() const { return has_main_; }
    const std::string &getMainParam() const { return main_; }
    bool isMainOn() const { return lower(main_) == "on" || lower(main_) == "true"; }
    bool contains(const std::string &k) const { return params_.count(k) > 0; }
    std::string getString(const std::string &k, const std::string &def = "") const {
        auto it = params_.find(k);
        return it != params_.end() && !it->second.empty() ? it->second[0] : def;
    }
    bool hasValue(const std::string &k) const {
        auto it = params_.find(k);
        return it != params_.end() && !it->second.empty();
    }
    double getFloat(const std::string &k, double def) const { return hasValue(k) ? java_float(getString(k)) : def; }
    double getDouble(const std::string &k, double def) const { return hasValue(k) ? std::strtod(getString(k).c_str(), nullptr) : def; }
    long getLong(const std::string &k, long def) const { return hasValue(k) ? std::strtol(getString(k).c_str(), nullptr, 10) : def; }
    int getInt(const std::string &k, int def) const { return (int)getLong(k, def); }
    bool isOn(const std::string &k, bool def) const {
        if (!hasValue(k)) return def;
        const std::string v = lower(getString(k));
        return v == "on" || v == "true";
    }

  private:
    std::string main_;
    bool has_main_ = false;
    std::map<std::string, std::vector<std::string>> params_;
};

class FileConfiger {
  public:
    explicit FileConfiger(const std::string &path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open configuration file " + path);
        std::string raw;
        while (std::getline(f, raw)) {
            size_t b = 0;
            while (b < raw.size() && (raw[b] == ' ' || raw[b] == '\t' || raw[b] == '\f')) ++b;
            if (b >= raw.size() || raw[b] == '#' || raw[b] == '!') continue;
            std::string key, val;
            size_t i = b;
            for (; i < raw.size(); ++i) { // key up to an unescaped '=', ':' or whitespace
                char c = raw[i];
                if (c == '\\' && i + 1 < raw.size()) {
                    key.push_back(unescape(raw[++i]));
                    continue;
                }
                if (c == '=' || c == ':' || c == ' ' || c == '\t') break;
                key.push_back(c);
            }
            while (i < raw.size() && (raw[i] == ' ' || raw[i] == '\t')) ++i;
            if (i < raw.size() && (raw[i] == '=' || raw[i] == ':')) ++i;
            while (i < raw.size() && (raw[i] == ' ' || raw[i] == '\t')) ++i;
            for (; i < raw.size(); ++i) {
                if (raw[i] == '\\' && i + 1 < raw.size()) val.push_back(unescape(raw[++i]));
                else if (raw[i] != '\r') val.push_back(raw[i]);
            }
            props_[key] = val;
        }
    }
    bool contains(const std::string &k) const { return props_.count(k) > 0; }
    std::string getString(const std::string &k, const std::string &def = "") const {
        auto it = props_.find(k);
        if (it == props_.end()) return def;
        std::string v = it->second;
        size_t b = 0, e = v.size();
        while (b < e && (unsigned char)v[b] <= 0x20) ++b;
        while (e > b && (unsigned char)v[e - 1] <= 0x20) --e;
        return v.substr(b, e - b);
    }
    int getInt(const std::string &k, int def) const { return contains(k) ? std::atoi(getString(k).c_str()) : def; }
    bool hasOptions(const std::string &k) const { return contains(k); }
    LineConfiger getParamOptions(const std::string &k) const { return LineConfiger(getString(k)); }
    std::string getPath(const std::string &k) const { return contains(k) ? getString(k) : getString(k + ".lins"); }

  private:
    static char unescape(char c) {
        switch (c) {
        case 't': return '\t';
        case 'n': return '\n';
        case 'r': return '\r';
        case 'f': return '\f';
        default: return c;
        }
    }
    std::map<std::string, std::string> props_;
};

} // namespace carskit
