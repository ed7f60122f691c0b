// features::V1 on the device (scope row f-3 moved across the boundary): compact board records -> the 32 input planes.
//
// The reference builds the planes on the CPU, once per leaf, on the search worker's thread
// (src/libdg_go/utils/features.rs:99-251 called from src/libdg_mcts/pool/event.rs:46-59): 10-45 us of host time and 23 KB per
// leaf, which -- not xGMI, not the GPU -- is what caps a node of eight MI355X.  Here a workgroup per board computes them from
// DGNN_BOARD_RECORD_BYTES of state the host keeps incrementally anyway (include/dgnn.h: dgnn_forward_boards):
//
//   phase 1  every stone adds itself to its block's stone set and its four neighbours to its block's neighbour set
//            (two 361-bit sets per block id in LDS, LDS atomics)
//   phase 2  liberties of a block = |neighbour set & empty points|          (BoardFast::get_n_liberty, board_fast.rs:169-173)
//   phase 3  per point the 32-bit plane mask: liberty thresholds of stones (features.rs:130-140), validity and liberties after
//            the move for both colours on empty points (is_valid board_fast.rs:213-241; get_n_liberty_if :484-541 in its
//            general form: the union of the point's neighbourhood and the neighbourhoods of the own blocks it joins, counted over
//            empty points and the stones of opponent blocks it captures), the two most recent moves, and the host-provided
//            ko / ladder bits gated by validity (features.rs:161-185)
//   phase 4  symmetry, the board-wide planes (colour / komi, any-ko), the packed words of dgnn_forward_packed, and -- for the
//            board kernels -- the padded slot-major fp16 input of the up-layer itself
//
// HBM-bound integer work: 896 B in, 1.5 KB (+ 12.8 KB of fp16 planes) out per board; a few microseconds per batch next to a
// 3 ms tower.  Bit-identical to go.cpp's features_v1 (tests/test_features_gpu.py: every position of the three committed games x
// 8 symmetries x both colours); go.cpp's record_to_packed is the host statement of exactly these steps.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dgnn {
namespace {

constexpr int kPoints = 361;
constexpr int kWords = 12;          // 361 bits
constexpr int kRecordWords = 224;   // DGNN_BOARD_RECORD_BYTES / 4
constexpr int kCellWords = 184;     // u16 cell[368]

__device__ __forceinline__ int symmetry_apply(int t, int p)       // symmetry.rs:39-62 in the order of symmetry::ALL (:119-128)
{
    const int x = p % 19 - 9, y = p / 19 - 9;
    int tx = x, ty = y;
    switch (t) {
        case 1: tx = -x; ty = y; break;
        case 2: tx = x; ty = -y; break;
        case 3: tx = y; ty = x; break;
        case 4: tx = -y; ty = -x; break;
        case 5: tx = y; ty = -x; break;
        case 6: tx = -x; ty = -y; break;
        case 7: tx = -y; ty = x; break;
        default: break;
    }
    return (ty + 9) * 19 + (tx + 9);
}

__device__ __forceinline__ int neighbours_of(int p, int* out)
{
    const int x = p % 19, y = p / 19;
    int n = 0;
    if (x < 18) out[n++] = p + 1;
    if (y > 0) out[n++] = p - 19;
    if (x > 0) out[n++] = p - 1;
    if (y < 18) out[n++] = p + 19;
    return n;
}

__device__ __forceinline__ uint4 expand_mask(unsigned mask, unsigned scalars, int slot)   // == conv_board.hip: expand_packed
{
    const unsigned bits = mask >> (slot * 8);
    unsigned h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (bits >> e) & 1u ? 0x3c00u : 0u;
    if (slot == 0) {
        if (bits & 1u) h[0] = scalars & 0xffffu;
        if (bits & 2u) h[1] = scalars >> 16;
    }
    return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

struct Shared {
    unsigned record[kRecordWords];          // the record as loaded
    unsigned nb[kPoints][kWords];           // per block id: every neighbour of its stones
    unsigned st[kPoints][kWords];           // per block id: its stones
    unsigned empty[kWords];
    unsigned word[368];                     // plane masks in OUTPUT (symmetry-applied) order
    unsigned short libs[kPoints + 3];       // per block id
    unsigned any_ko;
};

__global__ __launch_bounds__(256)
void board_records_kernel(const unsigned* __restrict__ records, unsigned* __restrict__ packed, uint16_t* __restrict__ planes_board, int batch)
{
    __shared__ Shared s;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= batch) return;

    if (t < kRecordWords) s.record[t] = records[(size_t)b * kRecordWords + t];
    for (int i = t; i < kPoints * kWords; i += 256) { (&s.nb[0][0])[i] = 0; (&s.st[0][0])[i] = 0; }
    if (t < kWords) s.empty[t] = 0;
    if (t == 0) s.any_ko = 0;
    __syncthreads();

    const unsigned short* cell = reinterpret_cast<const unsigned short*>(s.record);
    const unsigned* ko = s.record + kCellWords;
    const unsigned* ladder_capture = ko + kWords;
    const unsigned* ladder_escape = ladder_capture + kWords;
    const unsigned header = s.record[kCellWords + 3 * kWords];
    const int to_move = header & 0xff, symmetry = (header >> 8) & 7;
    const unsigned komi = header >> 16;
    const unsigned last_word = s.record[kCellWords + 3 * kWords + 1];
    const int last0 = (short)(last_word & 0xffff), last1 = (short)(last_word >> 16);

    // phase 1
    for (int p = t; p < kPoints; p += 256) {
        const unsigned c = cell[p];
        const unsigned bit = 1u << (p & 31);
        if ((c & 3u) == 0) { atomicOr(&s.empty[p >> 5], bit); continue; }
        const int h = c >> 2;
        atomicOr(&s.st[h][p >> 5], bit);
        int q[4];
        const int n = neighbours_of(p, q);
        for (int i = 0; i < n; ++i) atomicOr(&s.nb[h][q[i] >> 5], 1u << (q[i] & 31));
    }
    __syncthreads();

    // phase 2
    for (int p = t; p < kPoints; p += 256) {
        const unsigned c = cell[p];
        if ((c & 3u) == 0 || (int)(c >> 2) != p) continue;
        int n = 0;
#pragma unroll
        for (int i = 0; i < kWords; ++i) n += __popc(s.nb[p][i] & s.empty[i]);
        s.libs[p] = (unsigned short)n;
    }
    __syncthreads();

    // phase 3
    for (int p = t; p < kPoints; p += 256) {
        const unsigned c = cell[p];
        const int stone = c & 3u;
        unsigned m = 0;
        if (stone != 0) {
            const int start = stone == to_move + 1 ? 5 : 17;
            const int n = min((int)s.libs[c >> 2], 6);
            m = ((1u << n) - 1u) << start;
        } else {
            int q[4];
            const int nq = neighbours_of(p, q);
            unsigned qc[4];
            for (int i = 0; i < nq; ++i) qc[i] = cell[q[i]];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int color = side == 0 ? to_move : 1 - to_move;
                bool valid = false;                                   // board_fast.rs:213-241
                for (int i = 0; i < nq; ++i) {
                    const int qs = qc[i] & 3u;
                    if (qs == 0) valid = true;
                    else if ((qs == color + 1) == (s.libs[qc[i] >> 2] >= 2)) valid = true;
                }
                if (!valid) continue;
                unsigned reach[kWords], counted[kWords];              // board_fast.rs:484-541
#pragma unroll
                for (int i = 0; i < kWords; ++i) { reach[i] = 0; counted[i] = s.empty[i]; }
                for (int i = 0; i < nq; ++i) {
                    const int qs = qc[i] & 3u, h = qc[i] >> 2;
                    const unsigned qbit = 1u << (q[i] & 31);
#pragma unroll
                    for (int w = 0; w < kWords; ++w) if (w == (q[i] >> 5)) reach[w] |= qbit;
                    if (qs == color + 1) {
#pragma unroll
                        for (int w = 0; w < kWords; ++w) reach[w] |= s.nb[h][w];
                    } else if (qs == 2 - color && s.libs[h] == 1) {
#pragma unroll
                        for (int w = 0; w < kWords; ++w) counted[w] |= s.st[h][w];
                    }
                }
                int n = 0;
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    unsigned r = reach[w];
                    if (w == (p >> 5)) r &= ~(1u << (p & 31));
                    n += __popc(r & counted[w]);
                }
                m |= ((1u << min(n, 6)) - 1u) << (side == 0 ? 11 : 23);
                if (side == 0) {                                      // features.rs:161-185: only where the side to move may play
                    const unsigned bit = 1u << (p & 31);
                    if (ko[p >> 5] & bit) { m |= 1u << 29; atomicOr(&s.any_ko, 1u); }
                    if (ladder_capture[p >> 5] & bit) m |= 1u << 30;
                    if (ladder_escape[p >> 5] & bit) m |= 1u << 31;
                }
            }
        }
        if (p == last0) m |= 1u << 3;
        if (p == last1) m |= 1u << 4;
        s.word[symmetry_apply(symmetry, p)] = m;
    }
    __syncthreads();

    // phase 4
    const bool komi_set = (komi & 0x7fffu) != 0;
    const unsigned global = (komi_set ? 1u << to_move : 0u) | (s.any_ko ? 4u : 0u);
    const unsigned scalars = komi_set ? komi << (16 * to_move) : 0u;
    for (int i = t; i < kPackedWords; i += 256)
        packed[(size_t)b * kPackedWords + i] = i < kPoints ? (s.word[i] | global) : i == kPoints ? scalars : 0u;
    if (planes_board != nullptr) {                                    // [B][4][400][8], pad row / column zero
        for (int i = t; i < 4 * kBoardRows; i += 256) {
            const int row = i % kBoardRows, slot = i / kBoardRows;
            const int y = row / 20, x = row % 20;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x < 19 && y < 19) v = expand_mask(s.word[y * 19 + x] | global, scalars, slot);
            *reinterpret_cast<uint4*>(planes_board + ((size_t)b * 4 * kBoardRows + i) * 8) = v;
        }
    }
}

}  // namespace

hipError_t launch_board_records(const uint8_t* records, uint32_t* packed, uint16_t* planes_board, int batch, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(board_records_kernel, dim3((unsigned)batch), dim3(256), 0, stream, reinterpret_cast<const unsigned*>(records), packed,
                       planes_board, batch);
    return hipGetLastError();
}

}  // namespace dgnn
