// Weights-file ingest: streaming JSON walk + base85 + fp16 helpers.
//
// Behaviour follows src/libdg_nn/loader.rs:36-116 (what is accepted, what is ignored, which error is
// returned) and src/libdg_utils/b85.rs:101-132 (alphabet, big-endian groups of five, silent drop of a
// trailing partial group).  The file format is written by contrib/trainer/dream_tf/hooks/dump.py:54-61.
#include "dgnn_internal.h"

#include <cstdio>
#include <cstring>

namespace dgnn {

// ------------------------------------------------------------------------------------------ fp16

float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;

    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int shift = 0;
            while ((man & 0x400u) == 0) { man <<= 1; ++shift; }
            bits = sign | ((uint32_t)(113 - shift) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }

    float f;
    std::memcpy(&f, &bits, sizeof(f));
    return f;
}

uint16_t float_to_half(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, sizeof(x));

    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7fffffffu;

    if (mag > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((mag >> 13) & 0x3ffu));  // nan
    if (mag >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                           // overflow / inf
    if (mag <= 0x33000000u) return (uint16_t)sign;                                        // underflow to zero

    const int e = (int)(mag >> 23) - 127;
    uint32_t sig = (mag & 0x7fffffu) | 0x800000u;
    int drop = 13;               // bits of the 24-bit significand that do not fit
    uint32_t base = 0;

    if (e < -14) {
        drop += -14 - e;         // subnormal result
    } else {
        base = (uint32_t)(e + 15) << 10;
        sig &= 0x7fffffu;
    }

    uint32_t kept = sig >> drop;
    const uint32_t lost = sig & ((1u << drop) - 1u);
    const uint32_t tie = 1u << (drop - 1);
    if (lost > tie || (lost == tie && (kept & 1u))) ++kept;   // nearest, ties to even; carry is exact

    return (uint16_t)(sign | (base + kept));
}

// ------------------------------------------------------------------------------------------ tensor

size_t HostTensor::element_size() const
{
    switch (dtype) {
        case DataType::Int8: return 1;
        case DataType::Half: return 2;
        default: return 4;
    }
}

float HostTensor::as_f32() const
{
    float v = 0.0f;
    if (bytes.size() >= 4) std::memcpy(&v, bytes.data(), 4);
    return v;
}

int32_t HostTensor::as_i32() const
{
    int32_t v = 0;
    if (bytes.size() >= 4) std::memcpy(&v, bytes.data(), 4);
    return v;
}

// ------------------------------------------------------------------------------------------ base85

namespace {

struct B85Table {
    int8_t value[256];

    B85Table()
    {
        static const char alphabet[] =
            "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz!#$%&()*+-;<=>?@^_`{|}~";
        for (int i = 0; i < 256; ++i) value[i] = -1;
        for (int i = 0; i < 85; ++i) value[(unsigned char)alphabet[i]] = (int8_t)i;
    }
};

const B85Table& b85_table()
{
    static const B85Table table;
    return table;
}

}  // namespace

bool b85_decode(const char* text, size_t len, std::vector<uint8_t>& out)
{
    const B85Table& table = b85_table();
    const size_t groups = len / 5;   // a trailing partial group is dropped, b85.rs:111-123

    out.clear();
    out.reserve(groups * 4);

    // the reference validates characters as it consumes them, including those of a partial tail
    for (size_t i = groups * 5; i < len; ++i)
        if (table.value[(unsigned char)text[i]] < 0) return false;

    for (size_t g = 0; g < groups; ++g) {
        uint32_t acc = 0;
        for (int i = 0; i < 5; ++i) {
            const int8_t de = table.value[(unsigned char)text[5 * g + i]];
            if (de < 0) return false;
            acc = 85u * acc + (uint32_t)de;
        }
        out.push_back((uint8_t)(acc >> 24));
        out.push_back((uint8_t)(acc >> 16));
        out.push_back((uint8_t)(acc >> 8));
        out.push_back((uint8_t)acc);
    }

    return true;
}

// ------------------------------------------------------------------------------------------ JSON walk

namespace {

// A cursor over the document.  Only what the weights schema needs: objects whose values are strings
// or (one level down) objects of strings.  Anything else is malformed, like the `_ =>` arm of
// loader.rs:96.
struct Cursor {
    const char* p;
    const char* end;

    void skip_ws()
    {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }

    bool eat(char c)
    {
        skip_ws();
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }

    // Reads a string token; returns a view of its raw contents (escapes are kept verbatim: neither
    // tensor names nor base85 payloads contain any).
    bool string(const char*& s, size_t& n)
    {
        skip_ws();
        if (p >= end || *p != '"') return false;
        ++p;
        s = p;
        while (p < end && *p != '"') {
            if (*p == '\\') ++p;
            ++p;
        }
        if (p >= end) return false;
        n = (size_t)(p - s);
        ++p;
        return true;
    }
};

int parse_tensor(Cursor& c, HostTensor& tensor)
{
    // positioned after '{'
    if (c.eat('}')) return DGNN_OK;

    for (;;) {
        const char *key, *val;
        size_t key_len, val_len;

        if (!c.string(key, key_len) || !c.eat(':')) return DGNN_MALFORMED_WEIGHTS;
        if (!c.string(val, val_len)) return DGNN_MALFORMED_WEIGHTS;

        if (key_len == 1 && key[0] == 's') {
            std::vector<uint8_t> raw;
            if (!b85_decode(val, val_len, raw) || raw.size() < 4) return DGNN_MALFORMED_WEIGHTS;
            std::memcpy(&tensor.scale, raw.data(), 4);
        } else if (key_len == 1 && key[0] == 't') {
            if (val_len != 2) return DGNN_MALFORMED_WEIGHTS;
            if (val[0] == 'i' && val[1] == '1') tensor.dtype = DataType::Int8;
            else if (val[0] == 'i' && val[1] == '4') tensor.dtype = DataType::Int32;
            else if (val[0] == 'f' && val[1] == '2') tensor.dtype = DataType::Half;
            else if (val[0] == 'f' && val[1] == '4') tensor.dtype = DataType::Float;
            else return DGNN_MALFORMED_WEIGHTS;
        } else if (key_len == 1 && key[0] == 'v') {
            if (!b85_decode(val, val_len, tensor.bytes)) return DGNN_MALFORMED_WEIGHTS;
        } else {
            return DGNN_MALFORMED_WEIGHTS;   // loader.rs:92-94
        }

        if (c.eat(',')) continue;
        if (c.eat('}')) return DGNN_OK;
        return DGNN_MALFORMED_WEIGHTS;
    }
}

}  // namespace

int parse_weights(const char* text, size_t len, TensorMap& out)
{
    Cursor c{text, text + len};
    out.clear();

    c.skip_ws();
    if (c.p >= c.end) return DGNN_MISSING_WEIGHTS;     // "an empty result-set is an error", loader.rs:98-104
    if (!c.eat('{')) return DGNN_MALFORMED_WEIGHTS;

    if (!c.eat('}')) {
        for (;;) {
            const char* key;
            size_t key_len;

            if (!c.string(key, key_len) || !c.eat(':')) return DGNN_MALFORMED_WEIGHTS;

            c.skip_ws();
            if (c.p < c.end && *c.p == '"') {
                // top-level string values (e.g. "model_name:0") are ignored, loader.rs:47
                const char* val;
                size_t val_len;
                if (!c.string(val, val_len)) return DGNN_MALFORMED_WEIGHTS;
            } else if (c.eat('{')) {
                HostTensor tensor;
                const int status = parse_tensor(c, tensor);
                if (status != DGNN_OK) return status;
                out[std::string(key, key_len)] = std::move(tensor);
            } else {
                return DGNN_MALFORMED_WEIGHTS;
            }

            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return DGNN_MALFORMED_WEIGHTS;
        }
    }

    c.skip_ws();
    if (c.p != c.end) return DGNN_MALFORMED_WEIGHTS;

    return out.empty() ? DGNN_MISSING_WEIGHTS : DGNN_OK;
}

int load_weights_file(const char* path, TensorMap& out)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return DGNN_MISSING_WEIGHTS;                // loader.rs:110-115

    std::vector<char> text;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) text.insert(text.end(), chunk, chunk + got);
    std::fclose(f);

    return parse_weights(text.data(), text.size(), out);
}

}  // namespace dgnn
