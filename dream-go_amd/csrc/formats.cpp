// On-disk / wire formats either side of the path (scope row f-4): base85 of fp16 vectors and the per-move record of a
// self-play game.
//   b85::encode   src/libdg_utils/b85.rs:141-165 (pairs of fp32 -> fp16 -> 5 characters of the RFC 1924 alphabet)
//   b85::decode   src/libdg_utils/b85.rs:101-132 (loader.cpp holds the decoder the weights loader uses)
//   Played::fmt   src/libdg_mcts/self_play.rs:187-213 (";B[dd]C[..]TR[..]TV[n]P[<b85 policy>]V[0.1234]")
//   CGoban::to_sgf src/libdg_go/utils/sgf.rs:36-44
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dgnn_internal.h"

namespace dgnn {
namespace {

const char kAlphabet[] = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz!#$%&()*+-;<=>?@^_`{|}~";

uint16_t byteswap16(uint16_t v) { return (uint16_t)((v << 8) | (v >> 8)); }

std::string b85_encode_f16(const float* values, size_t n)
{
    std::string out;
    out.reserve(n / 2 * 5);
    for (size_t i = 0; i + 1 < n; i += 2) {
        // b85.rs:150-156: cast to half precision, big-endian 16-bit words, first value in the high half
        const uint32_t a = byteswap16(float_to_half(values[i]));
        const uint32_t b = byteswap16(float_to_half(values[i + 1]));
        const uint32_t acc = (a << 16) | b;
        out.push_back(kAlphabet[acc / 52200625u]);
        out.push_back(kAlphabet[(acc / 614125u) % 85u]);
        out.push_back(kAlphabet[(acc / 7225u) % 85u]);
        out.push_back(kAlphabet[(acc / 85u) % 85u]);
        out.push_back(kAlphabet[acc % 85u]);
    }
    return out;
}

std::string sgf_point(int packed_index)   // CGoban::to_sgf: pass is the empty string
{
    if (packed_index < 0 || packed_index >= DGNN_NUM_POINTS) return std::string();
    std::string s;
    s.push_back((char)('a' + packed_index % 19));
    s.push_back((char)('a' + packed_index / 19));
    return s;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" {

size_t dgnn_b85_encode_f16(const float* values, size_t count, char* out, size_t capacity)
{
    if (!values || count % 2 != 0) return 0;
    const size_t need = count / 2 * 5;
    if (!out || capacity < need + 1) return need + 1;    // size query
    const std::string s = b85_encode_f16(values, count);
    std::memcpy(out, s.data(), need);
    out[need] = '\0';
    return need + 1;
}

int dgnn_b85_decode(const char* text, size_t len, uint8_t* out, size_t capacity, size_t* num_bytes)
{
    if (!text || !num_bytes) return DGNN_BAD_ARGUMENT;
    std::vector<uint8_t> bytes;
    if (!b85_decode(text, len, bytes)) return DGNN_MALFORMED_WEIGHTS;    // invalid character (b85.rs:117-119)
    *num_bytes = bytes.size();
    if (out) {
        if (capacity < bytes.size()) return DGNN_BAD_ARGUMENT;
        std::memcpy(out, bytes.data(), bytes.size());
    }
    return DGNN_OK;
}

size_t dgnn_sgf_played(char* out, size_t capacity, int color, int packed_index, int num_rollout, const float* softmax,
                       int has_value, float value, int prior_index, const char* explain)
{
    std::string s = ";";
    s += color == 0 ? "B" : "W";                                            // color.rs:62-69
    s += "[" + sgf_point(packed_index) + "]";
    if (explain && *explain) {                                              // self_play.rs:191-193
        std::string e(explain);
        for (char& c : e) if (c == '\n') c = '\r';
        s += "C[" + e + "]";
    }
    if (prior_index >= 0 && prior_index < DGNN_NUM_POINTS) s += "TR[" + sgf_point(prior_index) + "]";   // :195-197
    if (num_rollout > 1 && softmax) {                                       // :199-206
        s += "TV[" + std::to_string(num_rollout) + "]P[" + b85_encode_f16(softmax, DGNN_NUM_MOVES) + "]";
    }
    if (has_value) {                                                        // :174-184, :208-212: always from black's side
        const float normalized = color == 0 ? 2.0f * value - 1.0f : -2.0f * value + 1.0f;
        char buf[32];
        std::snprintf(buf, sizeof(buf), "V[%.4f]", (double)normalized);
        s += buf;
    }
    if (out && capacity > s.size()) std::memcpy(out, s.c_str(), s.size() + 1);
    return s.size() + 1;
}

}  // extern "C"
