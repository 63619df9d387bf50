// json.h — a small recursive-descent JSON reader for the glTF chunk of a GLB file.
// (The reference uses tinygltf + nlohmann json from deps/, which are third-party and not part of
// this repo; glTF needs only objects, arrays, strings, numbers, booleans and null.)
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rthost {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj; // insertion order kept (extension iteration order)

    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_number() const { return kind == Number; }
    const Json* find(const std::string& key) const {
        if (kind != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    const Json& at(const std::string& key) const {
        const Json* j = find(key);
        if (!j) throw std::runtime_error("glTF: missing key '" + key + "'");
        return *j;
    }
    const Json& at(size_t i) const {
        if (kind != Array || i >= arr.size()) throw std::runtime_error("glTF: array index out of range");
        return arr[i];
    }
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    double number_or(const std::string& key, double dflt) const {
        const Json* j = find(key);
        return j && j->kind == Number ? j->num : dflt;
    }
    // the casts below are undefined for NaN / out-of-range doubles, and every number here comes from an untrusted file
    int int_or(const std::string& key, int dflt) const {
        const double v = number_or(key, (double)dflt);
        if (!(v >= -2147483648.0 && v <= 2147483647.0)) throw std::runtime_error("glTF: '" + key + "' is not a valid integer");
        return (int)v;
    }
    // a byte offset / count / index: finite, non-negative, at most `limit`
    size_t size_or(const std::string& key, size_t dflt, size_t limit) const {
        const Json* j = find(key);
        if (!j || j->kind != Number) return dflt;
        if (!(j->num >= 0.0 && j->num <= (double)limit)) throw std::runtime_error("glTF: '" + key + "' out of range");
        return (size_t)j->num;
    }
    size_t index_value(size_t limit) const { // this number as an index < limit
        if (kind != Number || !(num >= 0.0 && num < (double)limit)) throw std::runtime_error("glTF: index out of range");
        return (size_t)num;
    }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string& text) : s_(text) {}
    Json parse() {
        Json j = value();
        ws();
        if (p_ != s_.size()) fail("trailing characters");
        return j;
    }

  private:
    const std::string& s_;
    size_t p_ = 0;
    int depth_ = 0; // nesting of arrays / objects: a file of "[[[[..." must not overflow the stack
    static constexpr int kMaxDepth = 256;
    struct Nest {
        JsonParser& p;
        explicit Nest(JsonParser& q) : p(q) {
            if (++p.depth_ > kMaxDepth) p.fail("nesting deeper than 256 levels");
        }
        ~Nest() { --p.depth_; }
    };
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("JSON parse error at byte ") + std::to_string(p_) + ": " + what);
    }
    void ws() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_;
    }
    Json value() {
        ws();
        if (p_ >= s_.size()) fail("unexpected end");
        char c = s_[p_];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') {
            Json j;
            j.kind = Json::String;
            j.str = string();
            return j;
        }
        if (!std::strncmp(s_.c_str() + p_, "true", 4)) { p_ += 4; Json j; j.kind = Json::Bool; j.b = true; return j; }
        if (!std::strncmp(s_.c_str() + p_, "false", 5)) { p_ += 5; Json j; j.kind = Json::Bool; return j; }
        if (!std::strncmp(s_.c_str() + p_, "null", 4)) { p_ += 4; return Json{}; }
        if (!(c == '-' || (c >= '0' && c <= '9'))) fail("bad value"); // strtod alone would take "nan", "inf", "0x10", "+1"
        char* end = nullptr;
        double v = std::strtod(s_.c_str() + p_, &end);
        if (end == s_.c_str() + p_) fail("bad value");
        p_ = (size_t)(end - s_.c_str());
        Json j;
        j.kind = Json::Number;
        j.num = v;
        return j;
    }
    std::string string() {
        std::string out;
        ++p_; // opening quote
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\') {
                if (p_ >= s_.size()) fail("bad escape");
                char e = s_[p_++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': { // keep BMP code points as UTF-8
                    if (p_ + 4 > s_.size()) fail("bad \\u escape");
                    unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    break;
                }
                default: out += e;
                }
            } else {
                out += c;
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        ++p_;
        return out;
    }
    Json array() {
        Nest nest(*this);
        Json j;
        j.kind = Json::Array;
        ++p_;
        ws();
        if (p_ < s_.size() && s_[p_] == ']') { ++p_; return j; }
        for (;;) {
            j.arr.push_back(value());
            ws();
            if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
            if (p_ < s_.size() && s_[p_] == ']') { ++p_; return j; }
            fail("expected , or ]");
        }
    }
    Json object() {
        Nest nest(*this);
        Json j;
        j.kind = Json::Object;
        ++p_;
        ws();
        if (p_ < s_.size() && s_[p_] == '}') { ++p_; return j; }
        for (;;) {
            ws();
            if (p_ >= s_.size() || s_[p_] != '"') fail("expected key");
            std::string k = string();
            ws();
            if (p_ >= s_.size() || s_[p_] != ':') fail("expected :");
            ++p_;
            j.obj.emplace_back(std::move(k), value());
            ws();
            if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
            if (p_ < s_.size() && s_[p_] == '}') { ++p_; return j; }
            fail("expected , or }");
        }
    }
};

} // namespace rthost
