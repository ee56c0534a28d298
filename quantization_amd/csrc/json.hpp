// A small JSON reader for the reference's metadata files (serde_json::to_vec(&Metadata):
// quantization/src/encoded_vectors_u8.rs:24-31,263-288, encoded_vectors_pq.rs:39-44,498-523,
// encoded_vectors_binary.rs:21-24,260-286).  `load` must accept whatever serde_json (with ryu for the floats) writes and
// whatever serde_json::from_str would read back: objects with their keys in any order, nested objects and arrays,
// whitespace, numbers in every JSON form (1e-7, 1.0, -0.0, 1.17549435e-38, integers), strings with escapes.  What serde's
// derived Deserialize rejects is rejected here too, with serde's wording where that is short: a missing or duplicate
// field, null or a string where a number belongs (serde_json WRITES NaN / inf as null and cannot read that back into an
// f32), a fraction / exponent / sign in a usize, an unknown enum variant, trailing characters.  Unknown fields are skipped,
// as serde does by default.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace qamd {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean = false;
    double number = 0.0;       // the token read as f64 (serde_json parses every number as f64 / u64 / i64 first)
    bool integer = false;      // the token had no fraction and no exponent
    bool negative = false;     // the token started with '-'
    uint64_t uint = 0;         // its value when `integer && !negative` and it fits 64 bits
    std::string text;          // String
    std::vector<JsonValue> items;                               // Array
    std::vector<std::pair<std::string, JsonValue>> members;     // Object, in file order (duplicates kept: see get())
};

// Parses one JSON document (RFC 8259) that fills `text` but for whitespace.  false: `err` says what and where.
bool json_parse(const std::string &text, JsonValue &out, std::string &err);

// Field access with serde's struct semantics: the member named `key` of object `obj`; a missing field, a duplicate
// field or a non-object is an error.
const JsonValue *json_field(const JsonValue &obj, const char *key, std::string &err);
// Typed reads of a field, serde's conversions: usize from a non-negative integer token only; f32 from any number token
// as f64 and then `as f32` (serde_json's visit_f64 -> f32 path; overflow becomes +-inf); bool; string.
bool json_usize(const JsonValue &obj, const char *key, uint64_t &out, std::string &err);
bool json_f32_field(const JsonValue &obj, const char *key, float &out, std::string &err);
bool json_bool(const JsonValue &obj, const char *key, bool &out, std::string &err);
bool json_string(const JsonValue &obj, const char *key, std::string &out, std::string &err);
bool json_number_as_f32(const JsonValue &v, float &out, std::string &err);

}  // namespace qamd
