// Minimal JSON reader for scene.json (the reference uses serde_json 1.0.64; only the value model the
// loader observes is reproduced: objects, arrays, strings, bools, null, and numbers that remember
// whether they were written as integers, because serde_json::Number::as_i64 is None for "15.0").
#pragma once
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "vecmath.hpp"

namespace rrt {

struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0.0;
  bool is_int = false;     // written without fraction/exponent and fits i64
  long long inum = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // insertion order; duplicate keys: last wins on get()

  const Json* get(const std::string& key) const {
    if (kind != Obj) return nullptr;
    const Json* found = nullptr;
    for (auto& kv : obj)
      if (kv.first == key) found = &kv.second;
    return found;
  }
};

class JsonParser {
 public:
  explicit JsonParser(const std::string& text) : s_(text) {}
  Json parse() {
    Json v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
  int depth_ = 0;   // serde_json (the reference's parser) refuses nesting beyond 128 levels ("recursion limit exceeded"); so does this one
  struct Nest {
    JsonParser& p;
    explicit Nest(JsonParser& q) : p(q) { if (++p.depth_ > 128) p.fail("recursion limit exceeded"); }
    ~Nest() { p.depth_--; }
  };
  [[noreturn]] void fail(const char* what) {
    throw ParseError("scene json: " + std::string(what) + " at byte " + std::to_string(i_));
  }
  void ws() {
    while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) i_++;
  }
  Json value() {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    char c = s_[i_];
    if (c == '{') { Nest n(*this); return object(); }
    if (c == '[') { Nest n(*this); return array(); }
    if (c == '"') { Json j; j.kind = Json::Str; j.str = string(); return j; }
    if (c == 't') { lit("true"); Json j; j.kind = Json::Bool; j.b = true; return j; }
    if (c == 'f') { lit("false"); Json j; j.kind = Json::Bool; j.b = false; return j; }
    if (c == 'n') { lit("null"); return Json(); }
    return number();
  }
  void lit(const char* w) {
    size_t n = strlen(w);
    if (s_.compare(i_, n, w) != 0) fail("bad literal");
    i_ += n;
  }
  Json number() {
    size_t st = i_;
    if (i_ < s_.size() && s_[i_] == '-') i_++;
    bool digits = false, is_int = true;
    while (i_ < s_.size() && isdigit((unsigned char)s_[i_])) { i_++; digits = true; }
    if (i_ < s_.size() && s_[i_] == '.') { is_int = false; i_++; while (i_ < s_.size() && isdigit((unsigned char)s_[i_])) i_++; }
    if (i_ < s_.size() && (s_[i_] == 'e' || s_[i_] == 'E')) {
      is_int = false; i_++;
      if (i_ < s_.size() && (s_[i_] == '+' || s_[i_] == '-')) i_++;
      while (i_ < s_.size() && isdigit((unsigned char)s_[i_])) i_++;
    }
    if (!digits) fail("bad number");
    std::string tok = s_.substr(st, i_ - st);
    Json j;
    j.kind = Json::Num;
    j.num = strtod(tok.c_str(), nullptr);
    if (is_int) {
      errno = 0;
      long long v = strtoll(tok.c_str(), nullptr, 10);
      if (errno == 0) { j.is_int = true; j.inum = v; }
    }
    return j;
  }
  std::string string() {
    std::string out;
    i_++;  // opening quote
    while (true) {
      if (i_ >= s_.size()) fail("unterminated string");
      char c = s_[i_++];
      if (c == '"') break;
      if (c == '\\') {
        if (i_ >= s_.size()) fail("bad escape");
        char e = s_[i_++];
        switch (e) {
          case '"': out += '"'; break;
          case '\\': out += '\\'; break;
          case '/': out += '/'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'n': out += '\n'; break;
          case 'r': out += '\r'; break;
          case 't': out += '\t'; break;
          case 'u': {
            if (i_ + 4 > s_.size()) fail("bad \\u");
            unsigned cp = (unsigned)strtoul(s_.substr(i_, 4).c_str(), nullptr, 16);
            i_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: fail("bad escape");
        }
      } else {
        out += c;
      }
    }
    return out;
  }
  Json array() {
    Json j;
    j.kind = Json::Arr;
    i_++;
    ws();
    if (i_ < s_.size() && s_[i_] == ']') { i_++; return j; }
    while (true) {
      j.arr.push_back(value());
      ws();
      if (i_ >= s_.size()) fail("unterminated array");
      if (s_[i_] == ',') { i_++; continue; }
      if (s_[i_] == ']') { i_++; break; }
      fail("expected , or ]");
    }
    return j;
  }
  Json object() {
    Json j;
    j.kind = Json::Obj;
    i_++;
    ws();
    if (i_ < s_.size() && s_[i_] == '}') { i_++; return j; }
    while (true) {
      ws();
      if (i_ >= s_.size() || s_[i_] != '"') fail("expected key");
      std::string k = string();
      ws();
      if (i_ >= s_.size() || s_[i_] != ':') fail("expected :");
      i_++;
      j.obj.emplace_back(k, value());
      ws();
      if (i_ >= s_.size()) fail("unterminated object");
      if (s_[i_] == ',') { i_++; continue; }
      if (s_[i_] == '}') { i_++; break; }
      fail("expected , or }");
    }
    return j;
  }
};

}  // namespace rrt
