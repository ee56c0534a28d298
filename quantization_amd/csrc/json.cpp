// See json.hpp.  A recursive-descent reader over the bytes of the file; depth is bounded (the metadata nests three deep).
#include "json.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace qamd {
namespace {

struct Reader {
    const std::string &s;
    size_t p = 0;
    std::string err;

    bool fail(const char *what) {
        if (err.empty()) err = std::string(what) + " at byte " + std::to_string(p);
        return false;
    }
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++;
    }
    bool literal(const char *word) {
        const size_t n = strlen(word);
        if (s.compare(p, n, word) != 0) return fail("expected value");
        p += n;
        return true;
    }
    static void utf8(uint32_t cp, std::string &out) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    bool hex4(uint32_t &v) {
        if (p + 4 > s.size()) return fail("EOF while parsing a string");
        v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = s[p++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else return fail("invalid escape");
        }
        return true;
    }
    bool string(std::string &out) {  // at the opening quote
        p++;
        out.clear();
        for (;;) {
            if (p >= s.size()) return fail("EOF while parsing a string");
            const unsigned char c = (unsigned char)s[p++];
            if (c == '"') return true;
            if (c < 0x20) return fail("control character (\\u0000-\\u001F) found while parsing a string");
            if (c != '\\') {
                out += (char)c;
                continue;
            }
            if (p >= s.size()) return fail("EOF while parsing a string");
            const char e = s[p++];
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
                    uint32_t cp = 0;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00) {  // a surrogate pair
                        uint32_t lo = 0;
                        if (p + 2 > s.size() || s[p] != '\\' || s[p + 1] != 'u') return fail("unexpected end of hex escape");
                        p += 2;
                        if (!hex4(lo)) return false;
                        if (lo < 0xDC00 || lo > 0xDFFF) return fail("lone leading surrogate in hex escape");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp < 0xE000) {
                        return fail("lone trailing surrogate in hex escape");
                    }
                    utf8(cp, out);
                    break;
                }
                default: return fail("invalid escape");
            }
        }
    }
    bool number(JsonValue &v) {  // the JSON grammar: -? (0 | [1-9][0-9]*) (\.[0-9]+)? ([eE][+-]?[0-9]+)?
        const size_t start = p;
        v.kind = JsonValue::Number;
        v.negative = s[p] == '-';
        if (v.negative) p++;
        if (p >= s.size() || s[p] < '0' || s[p] > '9') return fail("invalid number");
        if (s[p] == '0') {
            p++;
            if (p < s.size() && s[p] >= '0' && s[p] <= '9') return fail("invalid number");  // leading zero
        } else {
            while (p < s.size() && s[p] >= '0' && s[p] <= '9') p++;
        }
        v.integer = true;
        if (p < s.size() && s[p] == '.') {
            v.integer = false;
            p++;
            if (p >= s.size() || s[p] < '0' || s[p] > '9') return fail("invalid number");
            while (p < s.size() && s[p] >= '0' && s[p] <= '9') p++;
        }
        if (p < s.size() && (s[p] == 'e' || s[p] == 'E')) {
            v.integer = false;
            p++;
            if (p < s.size() && (s[p] == '+' || s[p] == '-')) p++;
            if (p >= s.size() || s[p] < '0' || s[p] > '9') return fail("invalid number");
            while (p < s.size() && s[p] >= '0' && s[p] <= '9') p++;
        }
        const std::string tok = s.substr(start, p - start);
        errno = 0;
        v.number = strtod(tok.c_str(), nullptr);
        if (!std::isfinite(v.number)) return fail("number out of range");  // serde_json: f64 overflow is an error
        v.uint = 0;
        if (v.integer && !v.negative) {
            errno = 0;
            const unsigned long long u = strtoull(tok.c_str(), nullptr, 10);
            if (errno == ERANGE) v.integer = false;  // larger than u64: serde_json falls back to f64
            else v.uint = u;
        }
        return true;
    }
    bool value(JsonValue &v, int depth) {
        if (depth > 64) return fail("recursion limit exceeded");
        ws();
        if (p >= s.size()) return fail("EOF while parsing a value");
        const char c = s[p];
        if (c == '{') {
            v.kind = JsonValue::Object;
            p++;
            ws();
            if (p < s.size() && s[p] == '}') { p++; return true; }
            for (;;) {
                ws();
                if (p >= s.size()) return fail("EOF while parsing an object");
                if (s[p] != '"') return fail("key must be a string");
                std::string key;
                if (!string(key)) return false;
                ws();
                if (p >= s.size() || s[p] != ':') return fail("expected `:`");
                p++;
                v.members.emplace_back(std::move(key), JsonValue{});
                if (!value(v.members.back().second, depth + 1)) return false;
                ws();
                if (p >= s.size()) return fail("EOF while parsing an object");
                if (s[p] == ',') { p++; continue; }
                if (s[p] == '}') { p++; return true; }
                return fail("expected `,` or `}`");
            }
        }
        if (c == '[') {
            v.kind = JsonValue::Array;
            p++;
            ws();
            if (p < s.size() && s[p] == ']') { p++; return true; }
            for (;;) {
                v.items.emplace_back();
                if (!value(v.items.back(), depth + 1)) return false;
                ws();
                if (p >= s.size()) return fail("EOF while parsing a list");
                if (s[p] == ',') { p++; continue; }
                if (s[p] == ']') { p++; return true; }
                return fail("expected `,` or `]`");
            }
        }
        if (c == '"') {
            v.kind = JsonValue::String;
            return string(v.text);
        }
        if (c == 't') { v.kind = JsonValue::Bool; v.boolean = true; return literal("true"); }
        if (c == 'f') { v.kind = JsonValue::Bool; v.boolean = false; return literal("false"); }
        if (c == 'n') { v.kind = JsonValue::Null; return literal("null"); }
        if (c == '-' || (c >= '0' && c <= '9')) return number(v);
        return fail("expected value");
    }
};

const char *kind_name(const JsonValue &v) {
    switch (v.kind) {
        case JsonValue::Null: return "null";
        case JsonValue::Bool: return "a boolean";
        case JsonValue::Number: return v.integer ? "an integer" : "a floating point number";
        case JsonValue::String: return "a string";
        case JsonValue::Array: return "a sequence";
        default: return "a map";
    }
}

}  // namespace

bool json_parse(const std::string &text, JsonValue &out, std::string &err) {
    Reader r{text};
    out = JsonValue{};
    if (!r.value(out, 0)) {
        err = r.err;
        return false;
    }
    r.ws();
    if (r.p != text.size()) {
        err = "trailing characters at byte " + std::to_string(r.p);
        return false;
    }
    return true;
}

const JsonValue *json_field(const JsonValue &obj, const char *key, std::string &err) {
    if (obj.kind != JsonValue::Object) {
        err = std::string("invalid type: ") + kind_name(obj) + ", expected a struct";
        return nullptr;
    }
    const JsonValue *found = nullptr;
    for (const auto &m : obj.members) {
        if (m.first != key) continue;
        if (found) {
            err = std::string("duplicate field `") + key + "`";
            return nullptr;
        }
        found = &m.second;
    }
    if (!found) err = std::string("missing field `") + key + "`";
    return found;
}

bool json_usize(const JsonValue &obj, const char *key, uint64_t &out, std::string &err) {
    const JsonValue *v = json_field(obj, key, err);
    if (!v) return false;
    if (v->kind != JsonValue::Number || !v->integer || v->negative) {
        err = std::string("invalid type: ") + kind_name(*v) + ", expected usize (field `" + key + "`)";
        return false;
    }
    out = v->uint;
    return true;
}

bool json_number_as_f32(const JsonValue &v, float &out, std::string &err) {
    if (v.kind != JsonValue::Number) {
        err = std::string("invalid type: ") + kind_name(v) + ", expected f32";
        return false;
    }
    out = (float)v.number;  // serde_json: the token as f64 (or u64 / i64), then `as f32`
    return true;
}

bool json_f32_field(const JsonValue &obj, const char *key, float &out, std::string &err) {
    const JsonValue *v = json_field(obj, key, err);
    if (!v) return false;
    if (!json_number_as_f32(*v, out, err)) {
        err += std::string(" (field `") + key + "`)";
        return false;
    }
    return true;
}

bool json_bool(const JsonValue &obj, const char *key, bool &out, std::string &err) {
    const JsonValue *v = json_field(obj, key, err);
    if (!v) return false;
    if (v->kind != JsonValue::Bool) {
        err = std::string("invalid type: ") + kind_name(*v) + ", expected a boolean (field `" + key + "`)";
        return false;
    }
    out = v->boolean;
    return true;
}

bool json_string(const JsonValue &obj, const char *key, std::string &out, std::string &err) {
    const JsonValue *v = json_field(obj, key, err);
    if (!v) return false;
    if (v->kind != JsonValue::String) {
        err = std::string("invalid type: ") + kind_name(*v) + ", expected a string (field `" + key + "`)";
        return false;
    }
    out = v->text;
    return true;
}

}  // namespace qamd
