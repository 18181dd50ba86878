// m2s_json.h — minimal JSON DOM for the glTF chunk of a .glb file (the reference uses nlohmann json
// through tiny_gltf; neither is copied here).  Strict enough for glTF 2.0: objects, arrays, strings with
// escapes (\uXXXX -> UTF-8), numbers, true/false/null.  Header-only, host-only.
#pragma once
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace m2s_json {

struct Value;
using Array = std::vector<Value>;
using Object = std::map<std::string, Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Object> obj;

    bool is_object() const { return kind == Obj; }
    bool is_array() const { return kind == Arr; }
    bool is_number() const { return kind == Number; }
    bool is_string() const { return kind == String; }
    bool has(const std::string& k) const { return kind == Obj && obj->count(k) != 0; }
    const Value& operator[](const std::string& k) const {
        static const Value null_value;
        if (kind != Obj) return null_value;
        auto it = obj->find(k);
        return it == obj->end() ? null_value : it->second;
    }
    const Value& operator[](size_t i) const {
        static const Value null_value;
        return (kind == Arr && i < arr->size()) ? (*arr)[i] : null_value;
    }
    size_t size() const { return kind == Arr ? arr->size() : kind == Obj ? obj->size() : 0; }
    double number_or(double d) const { return kind == Number ? num : d; }
    // a number that no 64-bit integer can hold (or NaN): callers reject it where an index / size is expected
    bool bad_int() const { return kind == Number && !(num > -9.0e18 && num < 9.0e18); }
    // NaN / out-of-range numbers (llround would be undefined) read as "absent"
    long long int_or(long long d) const {
        if (kind != Number || !(num > -9.0e18 && num < 9.0e18)) return d;
        return (long long)std::llround(num);
    }
    std::string string_or(const std::string& d) const { return kind == String ? str : d; }
};

class Parser {
public:
    Parser(const char* p, size_t n) : s_(p), e_(p + n) {}
    Value parse() {
        Value v = value();
        ws();
        if (s_ != e_) fail("trailing characters");
        return v;
    }

private:
    const char* s_;
    const char* e_;
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("JSON: ") + m); }
    void ws() { while (s_ < e_ && (*s_ == ' ' || *s_ == '\t' || *s_ == '\n' || *s_ == '\r')) ++s_; }
    bool lit(const char* w) {
        const char* p = s_;
        while (*w) { if (p >= e_ || *p != *w) return false; ++p; ++w; }
        s_ = p;
        return true;
    }
    static void utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (e_ - s_ < 4) fail("bad \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *s_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        if (s_ >= e_ || *s_ != '"') fail("expected string");
        ++s_;
        std::string o;
        while (s_ < e_ && *s_ != '"') {
            char c = *s_++;
            if (c != '\\') { o += c; continue; }
            if (s_ >= e_) fail("bad escape");
            char x = *s_++;
            switch (x) {
                case '"': o += '"'; break;
                case '\\': o += '\\'; break;
                case '/': o += '/'; break;
                case 'b': o += '\b'; break;
                case 'f': o += '\f'; break;
                case 'n': o += '\n'; break;
                case 'r': o += '\r'; break;
                case 't': o += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && e_ - s_ >= 6 && s_[0] == '\\' && s_[1] == 'u') {
                        s_ += 2;
                        unsigned lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(o, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        if (s_ >= e_) fail("unterminated string");
        ++s_;
        return o;
    }
    int depth_ = 0;
    static constexpr int kMaxDepth = 512;      // nesting limit: value() recurses, "[[[[..." must not exhaust the stack
    struct DepthGuard { int& d; explicit DepthGuard(int& x) : d(x) { ++d; } ~DepthGuard() { --d; } };
    Value value() {
        ws();
        if (s_ >= e_) fail("unexpected end");
        DepthGuard guard(depth_);
        if (depth_ > kMaxDepth) fail("nesting too deep");
        Value v;
        char c = *s_;
        if (c == '{') {
            ++s_;
            v.kind = Value::Obj;
            v.obj = std::make_shared<Object>();
            ws();
            if (s_ < e_ && *s_ == '}') { ++s_; return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (s_ >= e_ || *s_ != ':') fail("expected ':'");
                ++s_;
                (*v.obj)[k] = value();
                ws();
                if (s_ < e_ && *s_ == ',') { ++s_; continue; }
                if (s_ < e_ && *s_ == '}') { ++s_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            ++s_;
            v.kind = Value::Arr;
            v.arr = std::make_shared<Array>();
            ws();
            if (s_ < e_ && *s_ == ']') { ++s_; return v; }
            for (;;) {
                v.arr->push_back(value());
                ws();
                if (s_ < e_ && *s_ == ',') { ++s_; continue; }
                if (s_ < e_ && *s_ == ']') { ++s_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = string();
        } else if (lit("true")) { v.kind = Value::Bool; v.b = true; }
        else if (lit("false")) { v.kind = Value::Bool; v.b = false; }
        else if (lit("null")) { v.kind = Value::Null; }
        else {
            // std::from_chars: correctly rounded like glibc's strtod, but independent of the embedding process's
            // LC_NUMERIC (under a comma-decimal locale strtod reads "1.5" as 1)
            const auto r = std::from_chars(s_, e_, v.num);
            if (r.ec == std::errc::result_out_of_range) v.num = (*s_ == '-') ? -HUGE_VAL : HUGE_VAL;   // strtod's answer
            else if (r.ec != std::errc() || r.ptr == s_) fail("unexpected character");
            s_ = r.ptr;
            v.kind = Value::Number;
        }
        return v;
    }
};

inline Value parse(const char* p, size_t n) { return Parser(p, n).parse(); }

}  // namespace m2s_json
