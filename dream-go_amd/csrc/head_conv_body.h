// head_conv_board_body: both 3x3 head convolutions (C -> S policy planes, C -> 2 value planes, policy_head.rs:92 / value_head.rs:76) as ONE
// convolution on the padded slot-major layout, one board (or a third of its rows) per 256-thread workgroup.  A header because two
// kernels run it: head_conv_board_kernel (heads.hip: a launch of its own) and -- round 5 -- tower_board_kernel, whose workgroup walks
// its board's layers and then, without leaving the CU, this body on the board it has just finished (conv_board_tower.hip): one launch
// less per batch, and the head convolution of a board overlaps the other boards' last layers instead of waiting for a free CU.
#pragma once
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace dgnn {
namespace headc {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kPoints = 361;

// ---- geometry shared with conv_board.hip ----
constexpr int kRows = kBoardRows;
constexpr int kHalo = 21;
constexpr int kPieces = 7;
constexpr int kSlotRows = 480;                         // 21 + 7 * 64 = 469 used; the k-slot stride is a multiple of 256 bytes (the B
constexpr int kSlotBytes = kSlotRows * 16;             // fragment of v_mfma_f32_16x16x32_f16 takes 16 rows from each of four k-slots)
constexpr int kXBufBytes = 4 * kSlotBytes;             // 30720: one 32-channel chunk of activations
constexpr int kHeadThreads = 256;

__device__ __forceinline__ _Float16 round_to_half(float t)
{
    asm volatile("" : "+v"(t));          // fma and cvt stay separate instructions (two roundings, like the oracle)
    return (_Float16)t;
}

__device__ __forceinline__ float half_bits_to_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t float_to_half_bits(float f) { return __builtin_bit_cast(uint16_t, round_to_half(f)); }

// Round 3: v_mfma_f32_16x16x32_f16 on NT tiles of 16 output channels (S + 2 = 10 of them at the reference's 8 samples: ONE tile
// instead of a 32-row tile that was two thirds padding), and -- small batches, where a workgroup per board leaves most CUs idle
// and the kernel is a latency chain over the chunks -- a board's rows over THREE workgroups (grid.y = 3: rows 128 q .. 128 q + 127,
// four of the seven activation pieces per k-slot).  4 waves; wave w owns TPW 16-row tiles: rows 96 w .. (a workgroup per board) or
// 128 q + 32 w .. (row split).  Per 32-channel chunk the activations and the chunk's filter rows for all 9 taps (9 * NT pieces)
// are brought into one of two LDS buffers; one barrier per chunk.  The arithmetic per output does not depend on the split.
template <int NT, int TPW>
constexpr int head_conv_lds_bytes() { return 2 * (kXBufBytes + 9 * NT * 1024); }

template <int NT, int TPW>
__device__ __forceinline__ void head_conv_board_body(const HeadConvLaunch& a, char* lds, int b, int q)
{
    constexpr int kWBufBytes = 9 * NT * 1024;          // one 32-channel chunk of the filter, all taps
    constexpr int kBufBytes = kXBufBytes + kWBufBytes;
    constexpr bool SPLIT = TPW == 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = a.cin / 32;
    const int row0 = SPLIT ? 128 * q + 32 * wave : 96 * wave;       // first board row of this wave
    // 64-row activation pieces this workgroup needs per k-slot: all seven, or those covering rows 128 q - 21 .. 128 q + 148
    const int piece0 = SPLIT ? (q == 0 ? 0 : q == 1 ? 1 : 3) : 0;
    constexpr int NPIECE = SPLIT ? 4 : kPieces;
    constexpr int TOTAL = 4 * NPIECE + 9 * NT;

    if (tid < 2 * 4 * kHalo) {           // halo rows of both activation buffers
        const int buf = tid / (4 * kHalo);
        const int rem = tid - buf * (4 * kHalo);
        const int slot = rem / kHalo;
        const int row = rem - slot * kHalo;
        *reinterpret_cast<uint4*>(lds + buf * kBufBytes + slot * kSlotBytes + row * 16) = make_uint4(0, 0, 0, 0);
    }

    const uint16_t* const x_board = a.x + (long)b * a.cin * kRows + lane * 8;
    const uint16_t* const w_lane = a.w + lane * 8;

    auto issue = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < (TOTAL + 3) / 4; ++j) {
            const int p = wave + 4 * j;          // activation pieces, then 9 * NT filter pieces
            if (p < 4 * NPIECE) {
                const int s = p / NPIECE;
                const int i = p - s * NPIECE + piece0;
                const uint16_t* src = x_board + ((long)(chunk * 4 + s) * kRows + i * 64) * 8;
                char* const dst = lds + buf * kBufBytes + s * kSlotBytes + (kHalo + i * 64) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else if (p < TOTAL) {
                const int f = p - 4 * NPIECE;    // tap * NT + tile
                const uint16_t* src = w_lane + ((long)chunk * 9 * NT + f) * 512;
                char* const dst = lds + buf * kBufBytes + kXBufBytes + f * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };

    floatx4 acc[NT][TPW];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][t][r] = 0.0f;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // B fragment: lane l reads row (l & 15) of its 16-row tile from k-slot l >> 4; A fragment: lane-linear
    const int x_lane = (lane >> 4) * kSlotBytes + (row0 + (lane & 15)) * 16;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) issue(chunk + 1, buf ^ 1);
        const char* const xb = lds + buf * kBufBytes + x_lane;
        const char* const wb = lds + buf * kBufBytes + kXBufBytes + lane * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int shift = (tap / 3 - 1) * 20 + (tap % 3 - 1);
            half8 af[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) af[nt] = *reinterpret_cast<const half8*>(wb + (tap * NT + nt) * 1024);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const half8 bf = *reinterpret_cast<const half8*>(xb + t * 256 + (kHalo + shift) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt], bf, acc[nt][t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next chunk landed; this chunk's reads retired
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // D[i][j] of a 16 x 16 tile: j = lane & 15 = row within the tile, i = 4 * (lane >> 4) + r = output channel within the tile.
    // Channels 0 .. S-1 are the policy planes, S and S + 1 the value planes; both outputs are compact NHWC.
    const int samples = a.samples;
    const int h4 = lane >> 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int row = row0 + t * 16 + (lane & 15);
        const int y = row / 20, x = row - y * 20;
        if (x == 19 || y >= 19) continue;
        const long pos = (long)b * kPoints + y * 19 + x;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = nt * 16 + 4 * h4 + r;
                if (c >= samples + 2) continue;
                float tv = acc[nt][t][r] + half_bits_to_float(a.bias[c]);
                tv = __builtin_fmaxf(tv, 0.0f);                        // conv2d.rs:119-125: ReLU on both head convolutions
                const uint16_t bits = float_to_half_bits(tv);
                if (c < samples) a.policy_features[pos * samples + c] = bits;
                else a.value_features[pos * 2 + (c - samples)] = bits;
            }
    }
}

}  // namespace headc
}  // namespace dgnn
