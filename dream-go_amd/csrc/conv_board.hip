// conv3x3_board_kernel: one launch per layer of the board kernel (conv_board_body.h), the launcher's rules, and the layout
// conversions at the two ends of the tower.  The tower kernels are in conv_board_tower.hip.
#include "conv_board_launch.h"

namespace dgnn {
namespace {

__global__ __launch_bounds__(256)
void nhwc_to_board_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int batch, int c)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int slots = c / 8;
    const long total = (long)batch * slots * kRows;
    if (idx >= total) return;
    const int row = (int)(idx % kRows);
    const int slot = (int)((idx / kRows) % slots);
    const long b = idx / ((long)kRows * slots);
    const int y = row / 20, x = row % 20;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (x < 19 && y < 19) v = *reinterpret_cast<const uint4*>(in + ((b * 361 + y * 19 + x) * c + slot * 8));
    *reinterpret_cast<uint4*>(out + idx * 8) = v;
}

// Packed features (include/dgnn.h: DGNN_PACKED_WORDS words per board: 361 plane masks, then fp16(v1) << 16 | fp16(v0)) ->
// the 32-channel input of the up-layer.  Channel c of a point is 0 where bit c of its mask is clear, else 1.0 -- except
// channels 0 and 1, which carry the board-wide komi constant of features.rs:236-239 (v0 / v1).
__device__ __forceinline__ uint4 expand_packed(unsigned mask, unsigned scalars, int slot)
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

// -> [B][4][400][8] (padded slot-major; pad rows / columns zero): replaces the upload of 23 KB of fp16 per board AND nhwc_to_board_kernel
__global__ __launch_bounds__(256)
void packed_to_board_kernel(const uint32_t* __restrict__ packed, uint16_t* __restrict__ out, int batch)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 4 * kRows) return;
    const int row = (int)(idx % kRows);
    const int slot = (int)((idx / kRows) % 4);
    const long b = idx / (4 * kRows);
    const int y = row / 20, x = row % 20;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (x < 19 && y < 19) v = expand_packed(packed[b * kPackedWords + y * 19 + x], packed[b * kPackedWords + 361], slot);
    *reinterpret_cast<uint4*>(out + idx * 8) = v;
}

// -> [B][361][32] NHWC (the layout dgnn_forward takes), for the NHWC kernels
__global__ __launch_bounds__(256)
void packed_to_nhwc_kernel(const uint32_t* __restrict__ packed, uint16_t* __restrict__ out, int batch)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)batch * 361 * 4) return;
    const int slot = (int)(idx % 4);
    const int pos = (int)((idx / 4) % 361);
    const long b = idx / (4 * 361);
    *reinterpret_cast<uint4*>(out + idx * 8) = expand_packed(packed[b * kPackedWords + pos], packed[b * kPackedWords + 361], slot);
}

// [B][C / 8][400][8] -> [B][361][C] NHWC
__global__ __launch_bounds__(256)
void board_to_nhwc_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int batch, int c)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int slots = c / 8;
    const long total = (long)batch * 361 * slots;
    if (idx >= total) return;
    const int slot = (int)(idx % slots);
    const int pos = (int)((idx / slots) % 361);
    const long b = idx / ((long)361 * slots);
    const int row = (pos / 19) * 20 + pos % 19;
    *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(in + ((b * slots + slot) * kRows + row) * 8);
}

}  // namespace

bool conv_board_supported(int cin, int cout)
{
    return cin >= 32 && cin % 32 == 0 && cout >= 64 && cout % 64 == 0;
}

// (launch_board_variant<NS>: conv_board_launch.h; the splits <4> and <2> are instantiated in conv_board_b.hip)
extern template void launch_board_variant<4>(const ConvBoardLaunch&, int, bool, int, hipStream_t);
extern template void launch_board_variant<2>(const ConvBoardLaunch&, int, bool, int, hipStream_t);

// Slices per workgroup: the largest that divides C_out / 32, unless a smaller one is needed to put a workgroup on most
// of the 256 CUs (a launch has batch * C_out / (32 * NS) workgroups).  DGNN_BOARD_NS=1|2|4|8 overrides.
int conv_board_pick_ns(int batch, int cout, int forced_ns)
{
    const int slices = cout / 32;
    int best = slices % 8 == 0 ? 8 : slices % 4 == 0 ? 4 : 2;
    while (best > 1 && (long)batch * (slices / best) < 192) best /= 2;
    if ((forced_ns == 1 || forced_ns == 2 || forced_ns == 4 || forced_ns == 8) && slices % forced_ns == 0) best = forced_ns;
    if (forced_ns == 16 && slices % 8 == 0) best = 8;
    // eight slices per workgroup are always the wide split (Split<16>: four waves x two 32-channel tiles).  The eight-wave form of
    // rounds 1-2 (Split<8>) is gone: with v_mfma_f32_16x16x32_f16 its skip-folding variants no longer fit 256 registers, and an
    // accumulator the compiler spills behind an asm MFMA is read before the matrix pipe has written it.
    return best == 8 ? 16 : best;
}

BoardSwitches read_board_switches()
{
    BoardSwitches s;
    if (const char* v = std::getenv("DGNN_BOARD_NS")) s.ns = std::atoi(v);
    if (const char* v = std::getenv("DGNN_BOARD_SKIP")) s.epilogue_skip = std::atoi(v) == 1;   // =1: keep the skip in the epilogue (A/B, tests)
    if (const char* v = ablation_env("DGNN_CONV_ABLATE")) s.ablate = std::atoi(v);
    return s;
}

hipError_t launch_conv3x3_board(const ConvBoardLaunch& args, const BoardSwitches& env, hipStream_t stream)
{
    if (args.batch <= 0) return hipSuccess;
    if (!conv_board_supported(args.cin, args.cout)) return hipErrorInvalidValue;
    const bool relu = args.relu != 0;
    // skip folded into the K loop needs two chunks to spread its row tiles over and a usable alpha2 / alpha1
    const int skip = args.alpha2 == 0.0f ? 0 : (args.cin >= 64 && skip_scale_foldable(args.alpha1, args.alpha2) && !env.epilogue_skip) ? 2 : 1;
    const int ablate = env.ablate;
    switch (conv_board_pick_ns(args.batch, args.cout, env.ns)) {
        case 16: launch_board_variant<16>(args, skip, relu, 0, stream); break;
        case 4: launch_board_variant<4>(args, skip, relu, ablate, stream); break;
        case 2: launch_board_variant<2>(args, skip, relu, 0, stream); break;
        default: launch_board_variant<1>(args, skip, relu, ablate, stream); break;
    }
    return hipGetLastError();
}

hipError_t launch_nhwc_to_board(const uint16_t* in, uint16_t* out, int batch, int channels, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    const long total = (long)batch * (channels / 8) * kRows;
    hipLaunchKernelGGL(nhwc_to_board_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, batch, channels);
    return hipGetLastError();
}

hipError_t launch_packed_features(const uint32_t* packed, uint16_t* out, int batch, bool board_layout, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    if (board_layout) {
        const long total = (long)batch * 4 * kRows;
        hipLaunchKernelGGL(packed_to_board_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, packed, out, batch);
    } else {
        const long total = (long)batch * 361 * 4;
        hipLaunchKernelGGL(packed_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, packed, out, batch);
    }
    return hipGetLastError();
}

hipError_t launch_board_to_nhwc(const uint16_t* in, uint16_t* out, int batch, int channels, hipStream_t stream)
{
    if (batch <= 0) return hipSuccess;
    const long total = (long)batch * 361 * (channels / 8);
    hipLaunchKernelGGL(board_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, out, batch, channels);
    return hipGetLastError();
}

}  // namespace dgnn
