// gfx950 (CDNA4) kernels of the leaf-evaluation path.
//
// conv3x3_kernel -- replaces cudnnConvolutionBiasActivationForward as the reference configures it
//   (src/libdg_nn/layers/conv2d.rs:99-117,171-220; src/libdg_cuda/cudnn/convolution_bias_activation.rs:21-41):
//       y = act(alpha1 * (x (*) w) + alpha2 * z + bias),   3x3, pad 1, stride 1, cross-correlation,
//   x/y/z NHWC fp16 on 19x19 boards, fp32 accumulation, output rounded to fp16 (nearest even).
//   Implicit GEMM, no im2col buffer: the GEMM "K" axis runs over (tap, 32-channel chunk); for every k-step the
//   activation tile is gathered straight from the NHWC tensor at the tap-shifted board position (zero for taps
//   that leave the board), staged through LDS next to the matching filter slab, and contracted with
//   v_mfma_f32_16x16x32_f16.  The MFMA is issued "swapped" -- A operand = filter rows (output channels),
//   B operand = board positions -- so that each lane ends up with four CONSECUTIVE output channels of one
//   position and the epilogue reads bias / skip and writes y as 8-byte NHWC vectors.
//
// heads_kernel -- the tail of policy_head.rs:78-103 and value_head.rs:66-84: dense 2888->362 scaled by
//   tau = 1/temperature + softmax (cudnnSoftmaxForward ACCURATE) and dense 722->1 + tanh, one workgroup per
//   board, wave-shuffle reductions.
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace dgnn {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kPoints = 361;
constexpr int kMoves = 362;
constexpr int kBK = 32;          // channels per k-step
constexpr int kLdsStride = 40;   // halfs per LDS row: 32 + 8 of padding (80 B) against bank conflicts

__device__ __forceinline__ float half_bits_to_float(uint16_t h)
{
    return (float)__builtin_bit_cast(_Float16, h);
}


// fp32 -> fp16 at exactly one place: the fp32 value is made opaque first so that the compiler cannot fuse the
// producing fma with the conversion (v_fma_mixlo_f16 rounds once, v_fma_f32 + v_cvt_f16_f32 rounds twice; hipcc
// picks one or the other per unrolled copy, which made results depend on the fragment index).  The reference
// semantics -- and the oracle's -- are "compute in fp32, then convert": two roundings.
__device__ __forceinline__ _Float16 round_to_half(float t)
{
    asm volatile("" : "+v"(t));
    return (_Float16)t;
}

__device__ __forceinline__ uint16_t float_to_half_bits(float f)
{
    return __builtin_bit_cast(uint16_t, round_to_half(f));   // v_cvt_f16_f32, round to nearest even
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64)
void conv3x3_kernel(ConvLaunch a)
{
    constexpr int T = WM * WN * 64;
    constexpr int TM = BM / WM;                 // positions per wave
    constexpr int TN = BN / WN;                 // output channels per wave
    constexpr int FM = TM / 16;
    constexpr int FN = TN / 16;
    constexpr int XP = (BM * 4 + T - 1) / T;    // 16-byte pieces of the activation tile per thread
    constexpr int WP = (BN * 4 + T - 1) / T;    // 16-byte pieces of the filter slab per thread

    __shared__ __attribute__((aligned(16))) _Float16 lds[(BM + BN) * kLdsStride];
    _Float16* const sX = lds;
    _Float16* const sW = lds + BM * kLdsStride;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int cin = a.cin;
    const int board = a.board;
    const int chunks = cin / kBK;

    // ---- what this thread gathers every k-step (fixed over the whole K loop) ----
    long x_off[XP];        // element offset of (position, channel piece) at the centre tap
    unsigned x_taps[XP];   // bit t set <=> tap t stays on the board for this position
    int x_lds[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int idx = tid + i * T;
        const int row = idx >> 2;
        const int piece = idx & 3;
        const int pos = m0 + row;
        unsigned taps = 0;
        if (idx < BM * 4 && pos < a.num_pos) {
            const int p = pos % (board * board);
            const int yy = p / board;
            const int xx = p % board;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sy = yy + t / 3 - 1;
                const int sx = xx + t % 3 - 1;
                if (sy >= 0 && sy < board && sx >= 0 && sx < board) taps |= 1u << t;
            }
        }
        x_taps[i] = taps;
        x_off[i] = (long)pos * cin + piece * 8;
        x_lds[i] = idx < BM * 4 ? row * kLdsStride + piece * 8 : -1;
    }

    long w_off[WP];
    int w_lds[WP];
    bool w_ok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int idx = tid + i * T;
        const int row = idx >> 2;
        const int piece = idx & 3;
        w_ok[i] = idx < BN * 4 && (n0 + row) < a.cout_pad;
        w_off[i] = (long)(n0 + row) * kBK + piece * 8;
        w_lds[i] = idx < BN * 4 ? row * kLdsStride + piece * 8 : -1;
    }

    uint4 xr[XP], wr[WP];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto gather = [&](int tap, int chunk, int step) {
        const long shift = (long)((tap / 3 - 1) * board + (tap % 3 - 1)) * cin + chunk * kBK;
#pragma unroll
        for (int i = 0; i < XP; ++i)
            xr[i] = ((x_taps[i] >> tap) & 1u) ? *reinterpret_cast<const uint4*>(a.x + x_off[i] + shift) : zero4;
        const uint16_t* slab = a.w + (long)step * a.cout_pad * kBK;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            wr[i] = w_ok[i] ? *reinterpret_cast<const uint4*>(slab + w_off[i]) : zero4;
    };

    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i)
            if (x_lds[i] >= 0) *reinterpret_cast<uint4*>(sX + x_lds[i]) = xr[i];
#pragma unroll
        for (int i = 0; i < WP; ++i)
            if (w_lds[i] >= 0) *reinterpret_cast<uint4*>(sW + w_lds[i]) = wr[i];
    };

    floatx4 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    const int frag_row = lane & 15;
    const int frag_k = (lane >> 4) * 8;

    gather(0, 0, 0);
    stage();
    __syncthreads();

    const int nsteps = 9 * chunks;
    int tap = 0, chunk = 0;
    for (int step = 0; step < nsteps; ++step) {
        // prefetch the next k-step into registers while this one is contracted
        int ntap = tap, nchunk = chunk + 1;
        if (nchunk == chunks) { nchunk = 0; ntap = tap + 1; }
        const bool more = step + 1 < nsteps;
        if (more) gather(ntap, nchunk, step + 1);

        half8 wf[FN], xf[FM];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
            wf[fn] = *reinterpret_cast<const half8*>(sW + (wn * TN + fn * 16 + frag_row) * kLdsStride + frag_k);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
            xf[fm] = *reinterpret_cast<const half8*>(sX + (wm * TM + fm * 16 + frag_row) * kLdsStride + frag_k);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[fn], xf[fm], acc[fn][fm], 0, 0, 0);

        __syncthreads();
        if (more) {
            stage();
            __syncthreads();
        }
        tap = ntap;
        chunk = nchunk;
    }

    // ---- epilogue: D[i = 4 * (lane >> 4) + r][j = lane & 15], i = output channel, j = position ----
    const int cout = a.cout;
    const bool vec_ok = (cout & 3) == 0;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        const int pos = m0 + wm * TM + fm * 16 + (lane & 15);
        if (pos >= a.num_pos) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int co = n0 + wn * TN + fn * 16 + (lane >> 4) * 4;
            if (co >= cout) continue;
            const long out = (long)pos * cout + co;

            const half4 b4 = *reinterpret_cast<const half4*>(a.bias + co);   // co + 3 < cout_pad always
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = __builtin_fmaf(a.alpha1, acc[fn][fm][r], (float)b4[r]);

            if (vec_ok) {
                if (a.alpha2 != 0.0f) {
                    const half4 z4 = *reinterpret_cast<const half4*>(a.z + out);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = __builtin_fmaf(a.alpha2, (float)z4[r], t[r]);
                }
                half4 y4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (a.relu && !(t[r] > 0.0f)) t[r] = 0.0f;
                    y4[r] = round_to_half(t[r]);
                }
                *reinterpret_cast<half4*>(a.y + out) = y4;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r >= cout) continue;
                    float v = t[r];
                    if (a.alpha2 != 0.0f) v = __builtin_fmaf(a.alpha2, half_bits_to_float(a.z[out + r]), v);
                    if (a.relu && !(v > 0.0f)) v = 0.0f;
                    a.y[out + r] = float_to_half_bits(v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------

constexpr int kHeadThreads = 384;   // 6 waves: one thread per move (362) for the policy matrix

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

__global__ __launch_bounds__(kHeadThreads)
void heads_kernel(HeadsLaunch a)
{
    __shared__ float sx[kPoints * 16];
    __shared__ float red[3][kHeadThreads / 64];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int num_in = kPoints * a.samples;

    const uint16_t* pf = a.policy_features + (long)b * num_in;
    for (int k = tid; k < num_in; k += kHeadThreads) sx[k] = half_bits_to_float(pf[k]);
    __syncthreads();

    // dense.rs:197-220 with alpha1 = tau (policy_head.rs:54): logits = fp16(tau * (w . x) + fp16(tau * b))
    float logit = -INFINITY;
    if (tid < kMoves) {
        const uint16_t* w = a.policy_w + tid;
        float acc = 0.0f;
        for (int k = 0; k < num_in; ++k) acc = __builtin_fmaf(sx[k], half_bits_to_float(w[(long)k * kMoves]), acc);
        logit = (float)round_to_half(__builtin_fmaf(a.tau, acc, half_bits_to_float(a.policy_b[tid])));
    }

    // value_head.rs:80: fp16(w . v + b), partial sums per thread
    const uint16_t* vf = a.value_features + (long)b * (kPoints * 2);
    float vpart = 0.0f;
    for (int k = tid; k < kPoints * 2; k += kHeadThreads)
        vpart = __builtin_fmaf(half_bits_to_float(vf[k]), half_bits_to_float(a.value_w[k]), vpart);

    const float wmax = wave_max(logit);
    const float wval = wave_sum(vpart);
    if (lane == 0) { red[0][wave] = wmax; red[1][wave] = wval; }
    __syncthreads();

    float maxv = red[0][0], vsum = red[1][0];
#pragma unroll
    for (int w = 1; w < kHeadThreads / 64; ++w) { maxv = fmaxf(maxv, red[0][w]); vsum += red[1][w]; }

    // softmax.rs:58-79 (ACCURATE: max-subtracted), fp32 inside, fp16 out
    const float e = tid < kMoves ? expf(logit - maxv) : 0.0f;
    const float wsum = wave_sum(e);
    if (lane == 0) red[2][wave] = wsum;
    __syncthreads();

    float sum = red[2][0];
#pragma unroll
    for (int w = 1; w < kHeadThreads / 64; ++w) sum += red[2][w];

    if (tid < kMoves) a.policy_out[(long)b * kMoves + tid] = float_to_half_bits(e / sum);

    if (tid == 0) {
        const float v = (float)round_to_half(vsum + half_bits_to_float(a.value_b[0]));
        a.value_out[b] = float_to_half_bits(tanhf(v));   // activation.rs:60-82
    }
}

}  // namespace

hipError_t launch_conv3x3(const ConvLaunch& args, hipStream_t stream)
{
    if (args.num_pos <= 0) return hipSuccess;

    if (args.cout_pad <= 16) {
        constexpr int BM = 128, BN = 16;
        dim3 grid((args.num_pos + BM - 1) / BM, (args.cout_pad + BN - 1) / BN);
        hipLaunchKernelGGL((conv3x3_kernel<BM, BN, 4, 1>), grid, dim3(256), 0, stream, args);
    } else if (args.cout_pad <= 64) {
        constexpr int BM = 128, BN = 64;
        dim3 grid((args.num_pos + BM - 1) / BM, (args.cout_pad + BN - 1) / BN);
        hipLaunchKernelGGL((conv3x3_kernel<BM, BN, 4, 1>), grid, dim3(256), 0, stream, args);
    } else {
        constexpr int BM = 128, BN = 128;
        dim3 grid((args.num_pos + BM - 1) / BM, (args.cout_pad + BN - 1) / BN);
        hipLaunchKernelGGL((conv3x3_kernel<BM, BN, 2, 2>), grid, dim3(256), 0, stream, args);
    }
    return hipGetLastError();
}

hipError_t launch_heads(const HeadsLaunch& args, hipStream_t stream)
{
    if (args.batch <= 0) return hipSuccess;
    hipLaunchKernelGGL(heads_kernel, dim3(args.batch), dim3(kHeadThreads), 0, stream, args);
    return hipGetLastError();
}

#ifndef DGNN_ABLATIONS
// The NHWC tile kernels of round 1 (conv_tile.hip: 40 % of the MFMA roof against the board kernel's 58 %) are A/B baselines, not
// product: they are compiled only into a library built with DGNN_ABLATIONS=1.  Without them the selector never offers them and
// DGNN_CONV=tile1|tile2|tile3 falls through to the generic kernel.
bool conv_tile_supported(int, int) { return false; }
int conv_tile_pick_fm(int, int) { return 8; }
hipError_t launch_conv3x3_tile(const ConvTileLaunch&, int, int, hipStream_t) { return hipErrorNotSupported; }
#endif

}  // namespace dgnn
