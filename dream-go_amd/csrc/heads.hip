// The two heads of the network (policy_head.rs:43-103, value_head.rs:40-84 of the reference) after the tower.
//
//   head_conv_board_kernel   both 3x3 head convolutions (C -> S policy planes, C -> 2 value planes) as ONE convolution
//                            with S + 2 <= 32 output channels on the padded slot-major layout of conv_board.hip, one
//                            board per workgroup.  HBM/L2-bound: the tower output is read once (the generic path read
//                            it twice, once per head, and ran two launches of 100 us each).
//   policy_gemm_kernel       dense 361 * S -> 362 for the whole batch as an MFMA GEMM [B x K] x [K x 384] against the
//                            TRANSPOSED weight matrix (the reference transposes it on the device at first use,
//                            dense.rs:182-189; here the host does it once).  logits = fp16(tau * (w . x) + fp16(tau * b)).
//   softmax_value_kernel     soft-max over the 362 logits (cudnnSoftmaxForward ACCURATE: max-subtracted, softmax.rs:58-79),
//                            dense 722 -> 1 + tanh (value_head.rs:80-81, activation.rs:60-82); one wave per board,
//                            wave-shuffle reductions.
#include "head_conv_body.h"

namespace dgnn {
namespace {

using namespace headc;                                  // the head convolution's geometry, helpers and body: head_conv_body.h

constexpr int kMoves = 362;
constexpr int kMovesPad = 384;                         // 12 MFMA tiles of 32

template <int NT, int TPW>
__global__ __launch_bounds__(kHeadThreads)
void head_conv_board_kernel(HeadConvLaunch a)
{
    __shared__ __attribute__((aligned(16))) char lds[head_conv_lds_bytes<NT, TPW>()];
    head_conv_board_body<NT, TPW>(a, lds, blockIdx.x, TPW == 2 ? blockIdx.y : 0);
}

// C[b][m] = sum_k P[b][k] * Wt[m][k];  one 32 x 32 tile per workgroup, the eight waves split K in pieces of 64.
//
// A fragment read straight from memory is 64 lanes x 16 bytes out of 32 different rows: 64 cache-line requests per
// instruction, and the vector memory path of the ONE CU a workgroup runs on serves about one per cycle -- 2 x 181 such
// instructions made the kernel 13-14 us at ANY batch size (as long as the whole 9x128 tower at batch 8; more waves changed
// nothing).  So the operands are staged: an LDS-DMA instruction whose lanes cover 8 rows x 128 contiguous bytes (8 line
// requests) drops a 32-row x 64-k tile into LDS, 16-byte segments XOR-swizzled by the row so that the fragment reads
// (32 rows at a 128-byte stride) are two-way conflicts at worst.  Double buffered per wave.  Rows whose length is not a
// multiple of 8 (dgnn_layer_dense with arbitrary shapes) and the ragged last piece take the element-wise path.
constexpr int kGemmWaves = 8;
constexpr int kGemmTileBytes = 32 * 128;      // 32 rows x 64 halfs
__global__ __launch_bounds__(kGemmWaves * 64)
void policy_gemm_kernel(PolicyGemmLaunch a)
{
    constexpr int WV = kGemmWaves;
    __shared__ __attribute__((aligned(16))) char stage[WV][2][2][kGemmTileBytes];     // [wave][buffer][P / Wt][tile]: 128 KiB
    float (*const partial)[16][64] = reinterpret_cast<float (*)[16][64]>(&stage[0][0][0][0]);   // reused behind a barrier: [WV][16][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * 32;
    const int m0 = blockIdx.y * 32;
    const int k_in = a.k_in;            // 361 * S
    const int kp = a.k_pad;             // k_in rounded up to a multiple of 64; Wt rows are zero beyond k_in

    floatx16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc2[r] = 0.0f; }

    const int nsteps = kp / 16;
    const bool staged = (k_in & 7) == 0;                  // 16-byte aligned rows
    const int nfull = staged ? k_in / 64 : 0;             // pieces of 64 that lie completely inside an input row

    // ---- staged pieces: piece p = wave, wave + WV, ...
    {
        // DMA: instruction j of a tile covers rows 8 j .. 8 j + 7; this lane fetches segment (lane & 7) ^ (row & 7) of its row
        const int drow = lane >> 3;                       // row within the group of 8 (== row & 7)
        const int dseg = (lane & 7) ^ drow;
        const uint16_t* prow_j[4];
        const uint16_t* wrow_j[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int br = b0 + 8 * j + drow;             // rows past the batch read board 0 instead (never stored)
            prow_j[j] = a.features + (long)(br < a.batch ? br : 0) * k_in + dseg * 8;
            wrow_j[j] = a.wt + (long)(m0 + 8 * j + drow) * kp + dseg * 8;
        }
        auto issue = [&](int piece, int buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(prow_j[j] + piece * 64),
                                                 (__attribute__((address_space(3))) void*)(&stage[wave][buf][0][j * 1024]), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow_j[j] + piece * 64),
                                                 (__attribute__((address_space(3))) void*)(&stage[wave][buf][1][j * 1024]), 16, 0, 0);
        };
        // fragment of k16 step t: row lane & 31, segment t * 2 + (lane >> 5), where the swizzle put it
        const int frow = lane & 31;
        int foff[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) foff[t] = frow * 128 + (((t * 2 + (lane >> 5)) ^ (frow & 7)) * 16);

        int piece = wave, buf = 0;
        if (piece < nfull) issue(piece, 0);
        for (; piece < nfull; piece += WV, buf ^= 1) {
            if (piece + WV < nfull) {
                issue(piece + WV, buf ^ 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // this piece landed; the next one may stay in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const char* const pa = &stage[wave][buf][0][0];
            const char* const pw = &stage[wave][buf][1][0];
            half8 af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *reinterpret_cast<const half8*>(pa + foff[t]);
                bf[t] = *reinterpret_cast<const half8*>(pw + foff[t]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2], bf[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[3], bf[3], acc2, 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the fragments are in registers before the buffer is refilled
        }
    }

    // ---- what is left (everything, for unaligned rows): k16 steps straight from memory, element-wise at the ragged end
    {
        const int brow = b0 + (lane & 31);
        const uint16_t* const prow = a.features + (long)(brow < a.batch ? brow : 0) * k_in;
        const uint16_t* const wrow = a.wt + (long)(m0 + (lane & 31)) * kp;
        const int kl = (lane >> 5) * 8;
        const int nmain = k_in / 16;    // k16 steps that lie completely inside the input row
        int s = nfull * 4 + wave;
        if (staged) {
            for (; s < nmain; s += WV) {
                const int k = s * 16 + kl;
                const half8 af = *reinterpret_cast<const half8*>(prow + k);
                const half8 bf = *reinterpret_cast<const half8*>(wrow + k);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
            }
        }
        for (; s < nsteps; s += WV) {   // element-wise, zero beyond k_in
            const int k = s * 16 + kl;
            half8 af;
#pragma unroll
            for (int e = 0; e < 8; ++e) af[e] = (k + e < k_in) ? __builtin_bit_cast(_Float16, prow[k + e]) : (_Float16)0.0f;
            half8 bf;
#pragma unroll
            for (int e = 0; e < 8; ++e) bf[e] = __builtin_bit_cast(_Float16, wrow[k + e]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        }
    }

    __syncthreads();                    // every wave is done with its staging buffers: they become the partial sums
#pragma unroll
    for (int r = 0; r < 16; ++r) partial[wave][r][lane] = acc[r] + acc2[r];
    __syncthreads();

    // D[i][j]: i = board row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = move = lane & 31.  1024 elements over the threads.
    constexpr int PER = 16 / WV;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int r = wave * PER + q;
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < WV; ++w) sum += partial[w][r][lane];
        const int bi = b0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = m0 + (lane & 31);
        if (bi < a.batch && m < a.n_out) {
            // dense.rs:197-220 with alpha1 = tau (policy_head.rs:54): fp16(tau * (w . x) + fp16(tau * b))
            float logit = __builtin_fmaf(a.tau, sum, half_bits_to_float(a.bias[m]));
            if (a.relu) logit = __builtin_fmaxf(logit, 0.0f);
            a.logits[(long)bi * a.ld_out + m] = float_to_half_bits(logit);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// one wave per board, 4 boards per workgroup
__global__ __launch_bounds__(256)
void softmax_value_kernel(SoftmaxValueLaunch a)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.batch) return;

    float logit[6];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int mv = lane + 64 * j;
        logit[j] = mv < kMoves ? half_bits_to_float(a.logits[(long)b * kMovesPad + mv]) : -INFINITY;
        m = fmaxf(m, logit[j]);
    }
    m = wave_max(m);
    float e[6];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        e[j] = (lane + 64 * j) < kMoves ? expf(logit[j] - m) : 0.0f;
        sum += e[j];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int mv = lane + 64 * j;
        if (mv < kMoves) a.policy_out[(long)b * kMoves + mv] = float_to_half_bits(e[j] / sum);
    }

    // value_head.rs:80-81: tanh(fp16(w . v + b))
    const uint16_t* vf = a.value_features + (long)b * (kPoints * 2);
    float part = 0.0f;
    for (int k = lane; k < kPoints * 2; k += 64)
        part = __builtin_fmaf(half_bits_to_float(vf[k]), half_bits_to_float(a.value_w[k]), part);
    part = wave_sum(part);
    if (lane == 0) {
        const _Float16 pre = round_to_half(part + half_bits_to_float(a.value_b[0]));
        if (a.value_pre_out) a.value_pre_out[b] = __builtin_bit_cast(uint16_t, pre);
        a.value_out[b] = float_to_half_bits(tanhf((float)pre));
    }
}

}  // namespace

hipError_t launch_head_conv_board(const HeadConvLaunch& args, hipStream_t stream)
{
    if (args.batch <= 0) return hipSuccess;
    // a board's rows over three workgroups where a workgroup per board would leave two thirds of the CUs idle
    const bool split = args.batch * 3 <= 256;
    const bool one_tile = args.samples + 2 <= 16;
    const dim3 grid(args.batch, split ? 3 : 1);
    if (one_tile && split) hipLaunchKernelGGL((head_conv_board_kernel<1, 2>), grid, dim3(kHeadThreads), 0, stream, args);
    else if (one_tile) hipLaunchKernelGGL((head_conv_board_kernel<1, 6>), grid, dim3(kHeadThreads), 0, stream, args);
    else if (split) hipLaunchKernelGGL((head_conv_board_kernel<2, 2>), grid, dim3(kHeadThreads), 0, stream, args);
    else hipLaunchKernelGGL((head_conv_board_kernel<2, 6>), grid, dim3(kHeadThreads), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_policy_gemm(const PolicyGemmLaunch& args, hipStream_t stream)
{
    if (args.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(policy_gemm_kernel, dim3((args.batch + 31) / 32, (args.n_out + 31) / 32), dim3(kGemmWaves * 64), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_softmax_value(const SoftmaxValueLaunch& args, hipStream_t stream)
{
    if (args.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(softmax_value_kernel, dim3((args.batch + 3) / 4), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace dgnn
