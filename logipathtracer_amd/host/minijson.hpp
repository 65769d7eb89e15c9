// Small JSON reader for glTF 2.0 documents (RFC 8259 subset: no surrogate-pair
// decoding beyond pass-through, numbers parsed with strtod).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace minijson {

class Value {
 public:
  enum Kind { Null, Bool, Number, String, Array, Object };
  Kind kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Value> arr;
  std::map<std::string, Value> obj;

  bool isNull() const { return kind == Null; }
  bool isObject() const { return kind == Object; }
  bool isArray() const { return kind == Array; }
  bool isNumber() const { return kind == Number; }
  bool isString() const { return kind == String; }
  bool has(const std::string& key) const { return kind == Object && obj.count(key) != 0; }
  // Missing members read as a shared null value, so lookups can be chained.
  const Value& operator[](const std::string& key) const {
    static const Value nullValue;
    if (kind != Object) return nullValue;
    auto it = obj.find(key);
    return it == obj.end() ? nullValue : it->second;
  }
  const Value& operator[](size_t i) const {
    static const Value nullValue;
    return (kind == Array && i < arr.size()) ? arr[i] : nullValue;
  }
  size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
  double number(double fallback) const { return kind == Number ? num : fallback; }
  long long integer(long long fallback) const { return kind == Number ? (long long)num : fallback; }
  std::string string(const std::string& fallback = "") const { return kind == String ? str : fallback; }
  bool boolean(bool fallback) const { return kind == Bool ? b : fallback; }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : s_(text) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != s_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t p_ = 0;
  int depth_ = 0;  // open arrays and objects: bounded, so a file of nothing but '[' cannot exhaust the stack
  static constexpr int kMaxDepth = 200;

  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error(std::string("JSON parse error at byte ") + std::to_string(p_) + ": " + what);
  }
  void ws() {
    while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_;
  }
  bool lit(const char* w) {
    size_t n = std::char_traits<char>::length(w);
    if (s_.compare(p_, n, w) == 0) {
      p_ += n;
      return true;
    }
    return false;
  }
  Value value() {
    ws();
    if (p_ >= s_.size()) fail("unexpected end");
    char c = s_[p_];
    Value v;
    if ((c == '{' || c == '[') && depth_ >= kMaxDepth) fail("nesting too deep");
    if (c == '{') {
      v.kind = Value::Object;
      ++p_;
      ws();
      if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
      ++depth_;
      for (;;) {
        ws();
        if (p_ >= s_.size() || s_[p_] != '"') fail("expected member name");
        std::string key = str();
        ws();
        if (p_ >= s_.size() || s_[p_] != ':') fail("expected ':'");
        ++p_;
        v.obj[key] = value();
        ws();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == '}') { ++p_; break; }
        fail("expected ',' or '}'");
      }
      --depth_;
    } else if (c == '[') {
      v.kind = Value::Array;
      ++p_;
      ws();
      if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
      ++depth_;
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
      --depth_;
    } else if (c == '"') {
      v.kind = Value::String;
      v.str = str();
    } else if (lit("true")) {
      v.kind = Value::Bool;
      v.b = true;
    } else if (lit("false")) {
      v.kind = Value::Bool;
    } else if (lit("null")) {
    } else {
      const char* begin = s_.c_str() + p_;
      char* end = nullptr;
      double d = std::strtod(begin, &end);
      if (end == begin) fail("unexpected character");
      p_ += (size_t)(end - begin);
      v.kind = Value::Number;
      v.num = d;
    }
    return v;
  }
  std::string str() {
    std::string out;
    ++p_;  // opening quote
    while (p_ < s_.size() && s_[p_] != '"') {
      char c = s_[p_++];
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= s_.size()) fail("bad escape");
      char e = s_[p_++];
      switch (e) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          if (p_ + 4 > s_.size()) fail("bad \\u escape");
          unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
          p_ += 4;
          if (cp < 0x80) out.push_back((char)cp);
          else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
          else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
          break;
        }
        default: out.push_back(e); break;  // \" \\ \/
      }
    }
    if (p_ >= s_.size()) fail("unterminated string");
    ++p_;  // closing quote
    return out;
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace minijson
