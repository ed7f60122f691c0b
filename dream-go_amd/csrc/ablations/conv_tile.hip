// conv3x3_tile_kernel -- the tuned residual-tower convolution for gfx950 (C_out a multiple of 256, C_in a multiple
// of 64).  Same contract as conv3x3_kernel (kernels.hip):  y = act(alpha1 * conv3x3(x, w) + alpha2 * z + bias).
//
// Implicit GEMM, output tile = BM board positions x 256 output channels per 512-thread workgroup (8 waves, 4 along
// the channels x 2 along the positions; wave tile 64 channels x BM/2 positions), K loop over (64-channel chunk, tap).
//
//   HBM/L2 -> LDS   __builtin_amdgcn_global_load_lds, 16 B per lane (no VGPR round trip, no ds_write pass).
//                   Activations: every lane gathers its 16 B straight from the NHWC tensor at the tap-shifted board
//                   position; taps that leave the board (and rows past the end) read a zero page instead, so the
//                   kernel contains no masking arithmetic.  Filter: the host packs every k-step's [256][64] slab in
//                   exactly the LDS image order, so its copy is a linear 32 KiB stream.
//   LDS layout      row-major [rows][64 halfs] (128 B rows), 16-byte slots XOR-swizzled with (row >> 1) & 7 so that
//                   the 16 rows x one k-slot pattern of an MFMA fragment read (ds_read_b128) touches 16 distinct
//                   slots of the 256 B bank row: conflict-free.  The DMA destination is lane-linear, so the swizzle
//                   is applied to the per-lane SOURCE address (activations) / baked into the packed filter.
//   pipeline        two LDS buffers; ONE workgroup barrier per k-step, placed in the MIDDLE of the step's MFMA
//                   stream: before it every read of the current buffer has completed, after it the current buffer is
//                   refilled for step + 2 and the first fragments of step + 1 are read while the step's last MFMAs
//                   (operands already in registers) keep the matrix pipe busy.  A tile's loads therefore have a full
//                   k-step (~1500 cycles) to land.
//   MFMA            v_mfma_f32_16x16x32_f16, A operand = filter rows (output channels), B operand = positions: a lane
//                   ends up with 4 consecutive output channels of one position -> 8-byte NHWC epilogue accesses.
#include "../kernels.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

namespace dgnn {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kBN = 256;                  // output channels per workgroup
constexpr int kBK = 64;                   // channels per k-step
constexpr int kRowBytes = kBK * 2;        // 128
constexpr int kWTileBytes = kBN * kRowBytes;
constexpr int kThreads = 512;

// fp32 -> fp16 at exactly one place: the fp32 value is made opaque first so that the compiler cannot fuse the
// producing fma with the conversion (v_fma_mixlo_f16 rounds once, v_fma_f32 + v_cvt_f16_f32 rounds twice; hipcc
// picks one or the other per unrolled copy, which made results depend on the fragment index).  The reference
// semantics -- and the oracle's -- are "compute in fp32, then convert": two roundings.
__device__ __forceinline__ _Float16 round_to_half(float t)
{
    asm volatile("" : "+v"(t));
    return (_Float16)t;
}

template <int FM>   // 16-position fragments per wave along the positions; BM = 32 * FM
__global__ __launch_bounds__(kThreads)
void conv3x3_tile_kernel(ConvTileLaunch a)
{
    constexpr int BM = 32 * FM;
    constexpr int XTileBytes = BM * kRowBytes;
    constexpr int BufBytes = XTileBytes + kWTileBytes;
    constexpr int XI = BM / 64;           // activation DMA instructions per wave per k-step (8 rows each)
    constexpr int WI = kBN / 64;          // filter DMA instructions per wave per k-step
    constexpr int FN = 4;

    __shared__ __attribute__((aligned(16))) char lds[2 * BufBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3;
    const int wm = wave >> 2;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * kBN;
    const int cin = a.cin;
    const int board = a.board;
    const int chunks = cin / kBK;
    const int nsteps = 9 * chunks;

    // ---- per-lane gather description of the activation tile (fixed over the K loop) ----
    const uint16_t* x_ptr[XI];    // address of this lane's 16 bytes at the centre tap, chunk 0
    unsigned x_taps[XI];          // bit t set <=> tap t stays on the board
    const int slot = lane & 7;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = (wave * XI + i) * 8 + (lane >> 3);
        const int pos = m0 + row;
        const int piece = slot ^ ((row >> 1) & 7);
        unsigned taps = 0;
        if (pos < a.num_pos) {
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
        x_ptr[i] = a.x + (long)pos * cin + piece * 8;
    }
    const uint16_t* const zero_src = a.zero_page + slot * 8;
    const uint16_t* const w_src = a.w + (long)blockIdx.y * nsteps * (kWTileBytes / 2) + (wave * WI) * 512 + lane * 8;

    // issue the DMA of k-step `step` (chunk-major, tap-minor) into LDS buffer `buf`
    auto issue = [&](int step, int buf) {
        const int chunk = step / 9;
        const int tap = step - chunk * 9;
        const long shift = (long)((tap / 3 - 1) * board + (tap % 3 - 1)) * cin + chunk * kBK;
        char* const xb = lds + buf * BufBytes + (wave * XI) * 1024;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const uint16_t* src = ((x_taps[i] >> tap) & 1u) ? x_ptr[i] + shift : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xb + i * 1024), 16, 0, 0);
        }
        char* const wb = lds + buf * BufBytes + XTileBytes + (wave * WI) * 1024;
        const uint16_t* ws = w_src + (long)step * (kWTileBytes / 2);
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws + i * 512),
                                             (__attribute__((address_space(3))) void*)(wb + i * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses: row = base + (lane & 15), k-slot = (kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7) ----
    const int s0 = (lane >> 4) ^ ((lane >> 1) & 7);
    const int x_frag = (wm * (BM / 2) + (lane & 15)) * kRowBytes;                 // + fm * 2048 + slot * 16
    const int w_frag = XTileBytes + (wn * 64 + (lane & 15)) * kRowBytes;          // + fn * 2048 + slot * 16
    const int k_off[2] = {s0 * 16, (s0 ^ 4) * 16};

    auto read_w = [&](int buf, int kk, int fn) -> half8 {
        return *reinterpret_cast<const half8*>(lds + buf * BufBytes + w_frag + fn * 2048 + k_off[kk]);
    };
    auto read_x = [&](int buf, int kk, int fm) -> half8 {
        return *reinterpret_cast<const half8*>(lds + buf * BufBytes + x_frag + fm * 2048 + k_off[kk]);
    };

    floatx4 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- prologue: tile 0 landed and visible, tile 1 in flight, first fragments of tile 0 in registers ----
    issue(0, 0);
    __syncthreads();                       // vmcnt(0) + barrier: buffer 0 complete for every wave
    if (nsteps > 1) issue(1, 1);

    half8 wf[FN], xf[FM];                  // fragments of the (step, kk) being contracted
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) wf[fn] = read_w(0, 0, fn);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) xf[fm] = read_x(0, 0, fm);

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;

        // kk = 0 of this step (operands were read at the end of the previous iteration / in the prologue);
        // the kk = 1 fragments are read underneath these MFMAs
        half8 wg[FN], xg[FM];
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) wg[fn] = read_w(cur, 1, fn);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) xg[fm] = read_x(cur, 1, fm);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[fn], xf[fm], acc[fn][fm], 0, 0, 0);

        // first half of kk = 1
#pragma unroll
        for (int fm = 0; fm < FM / 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wg[fn], xg[fm], acc[fn][fm], 0, 0, 0);

        // ---- the step's only barrier: every read of buffer `cur` is complete (the MFMAs above consumed them or
        // the waitcnt of __syncthreads covers them); the DMA of step + 1 (issued one step ago) has landed ----
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);

        if (step + 2 < nsteps) issue(step + 2, cur);        // buffer `cur` is free again
        if (step + 1 < nsteps) {                            // first fragments of the next step, from the other buffer
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) wf[fn] = read_w(cur ^ 1, 0, fn);
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) xf[fm] = read_x(cur ^ 1, 0, fm);
        }

        // second half of kk = 1: operands already in registers, covers the latency of the reads just issued
#pragma unroll
        for (int fm = FM / 2; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wg[fn], xg[fm], acc[fn][fm], 0, 0, 0);
    }

    // ---- epilogue: D[i = 4 * (lane >> 4) + r][j = lane & 15], i = output channel, j = position ----
    // Loads are unconditional (positions clamped) and issued a whole fragment row ahead of their use so that
    // they overlap; only the stores are predicated.
    const int cout = a.cout;
    const int co0 = n0 + wn * 64 + (lane >> 4) * 4;
    const bool skip = a.alpha2 != 0.0f;        // wave-uniform
    const bool relu = a.relu != 0;
    half4 b4[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) b4[fn] = *reinterpret_cast<const half4*>(a.bias + co0 + fn * 16);

    auto finish = [&](int fm, const half4 (&z4)[FN]) {
        const int pos = m0 + wm * (BM / 2) + fm * 16 + (lane & 15);
        const long out = (long)pos * cout + co0;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            half4 y4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __builtin_fmaf(a.alpha1, acc[fn][fm][r], (float)b4[fn][r]);
                if (skip) t = __builtin_fmaf(a.alpha2, (float)z4[fn][r], t);
                if (relu && !(t > 0.0f)) t = 0.0f;
                y4[r] = round_to_half(t);
            }
            if (pos < a.num_pos) *reinterpret_cast<half4*>(a.y + out + fn * 16) = y4;
        }
    };

    if (skip) {
        half4 z4[FM][FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            int pos = m0 + wm * (BM / 2) + fm * 16 + (lane & 15);
            pos = pos < a.num_pos ? pos : a.num_pos - 1;
            const uint16_t* zp = a.z + (long)pos * cout + co0;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) z4[fm][fn] = *reinterpret_cast<const half4*>(zp + fn * 16);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) finish(fm, z4[fm]);
    } else {
        const half4 none[FN] = {};
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) finish(fm, none);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// conv3x3_tile2_kernel -- same tile, same LDS image, same MFMA mapping as conv3x3_tile_kernel, but the eight waves
// run as TWO GROUPS STAGGERED BY ONE BARRIER (group g = wave >> 2 owns the position half g of the tile; waves w and
// w + 4 share a SIMD).  Every k-step is split into a load phase L and a matrix phase M:
//
//     L(s)  issue this group's share of the DMA for later tiles, read ALL fragments of tile s into registers
//     M(s)  48 MFMAs (FM = 6) out of registers, nothing else
//
//   interval      2s        2s+1      2s+2      2s+3
//   group 0       L(s)      M(s)      L(s+1)    M(s+1)
//   group 1       M(s-1)    L(s)      M(s)      L(s+1)
//
// so on every SIMD one wave feeds the matrix pipe while its partner pays the DMA issue cost and the LDS reads.
// DMA duties are chosen so that every byte has at least one whole interval to land before its first reader:
//     group 0 in L(s):  filter slab of tile s+1            -> buffer (s+1)&1   (read from interval 2s+2)
//     group 1 in L(s):  positions half 0 of tile s+2       -> buffer  s   &1   (group 0 finished it in interval 2s,
//                                                                               read again from interval 2s+4)
//                       positions half 1 of tile s+1       -> buffer (s+1)&1   (read by group 1 in interval 2s+3)
// Ordering rules used (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  a wave drains its own DMA (s_waitcnt vmcnt(0)) at the end of its M phase, i.e. one interval after issuing
//        it, then passes a barrier; every reader of that data starts at a later barrier.
//   WAR  a region is overwritten only in an interval that starts at a barrier which every reader of the previous
//        contents passed after retiring its reads (s_waitcnt lgkmcnt(0) closes every L phase).
// ABL (measurement only, results are garbage when != 0): bit 0 drops the MFMAs, bit 1 the in-loop DMA, bit 2 the
// in-loop fragment reads.  Used by tools/ablate.sh to price each pipeline component on its own.
//
// SCHED = 3 rebalances the two intervals of a k-step so that each carries HALF of the step's MFMAs:
//     b(s)  read all fragments of tile s, 24 MFMAs (kk = 0) as the reads return, drain this wave's DMA
//     a(s)  issue this group's DMA, 24 MFMAs (kk = 1) out of registers
//   group 0:  b(s) in interval 2s, a(s) in 2s+1;   group 1 one interval later.
//   DMA duties (every byte still has one whole interval to land, same RAW / WAR argument as above):
//     group 0 in a(s):  positions half 0 of tile s+2 -> buffer s&1     (group 0 read it in interval 2s)
//                       positions half 1 of tile s+1 -> buffer (s+1)&1 (group 1 read it in interval 2s-1)
//     group 1 in a(s):  filter slab of tile s+2      -> buffer s&1     (last read by group 1 in interval 2s+1)
template <int FM, int SCHED = 2, int ABL = 0>
__global__ __launch_bounds__(kThreads)
void conv3x3_tile2_kernel(ConvTileLaunch a)
{
    static_assert(FM % 2 == 0, "position half must be a whole number of DMA instructions per wave");
    constexpr int BM = 32 * FM;
    constexpr int XTileBytes = BM * kRowBytes;
    constexpr int BufBytes = XTileBytes + kWTileBytes;
    constexpr int XH = FM / 2;            // DMA instructions per wave for one position half (16 * FM rows)
    constexpr int FN = 4;

    __shared__ __attribute__((aligned(16))) char lds[2 * BufBytes];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3;
    const int g = wave >> 2;              // group = position half
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * kBN;
    const int cin = a.cin;
    const int board = a.board;
    const int nsteps = 9 * (cin / kBK);

    // ---- gather description: rows (wn * XH + i) * 8 + (lane >> 3) of position half h ----
    const uint16_t* x_ptr[2][XH];
    unsigned x_taps[2][XH];
    const int slot = lane & 7;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < XH; ++i) {
            const int row = h * (BM / 2) + (wn * XH + i) * 8 + (lane >> 3);
            const int pos = m0 + row;
            const int piece = slot ^ ((row >> 1) & 7);
            unsigned taps = 0;
            if (pos < a.num_pos) {
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
            x_taps[h][i] = taps;
            x_ptr[h][i] = a.x + (long)pos * cin + piece * 8;
        }
    const uint16_t* const zero_src = a.zero_page + slot * 8;
    const uint16_t* const w_base = a.w + (long)blockIdx.y * nsteps * (kWTileBytes / 2) + lane * 8;

    auto issue_x = [&](int step, int h, int buf) {
        const int chunk = step / 9;
        const int tap = step - chunk * 9;
        const long shift = (long)((tap / 3 - 1) * board + (tap % 3 - 1)) * cin + chunk * kBK;
        char* const dst = lds + buf * BufBytes + (h * (BM / 16) + wn * XH) * 1024;
#pragma unroll
        for (int i = 0; i < XH; ++i) {
            const uint16_t* src = ((x_taps[h][i] >> tap) & 1u) ? x_ptr[h][i] + shift : zero_src;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    // filter slab instructions [first, first + COUNT) of the 32 that make up one k-step
    auto issue_w = [&](int step, int buf, int first, auto count) {
        char* const dst = lds + buf * BufBytes + XTileBytes + first * 1024;
        const uint16_t* src = w_base + (long)step * (kWTileBytes / 2) + first * 512;
#pragma unroll
        for (int i = 0; i < decltype(count)::value; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 512),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    const int s0 = (lane >> 4) ^ ((lane >> 1) & 7);
    const int x_frag = (g * (BM / 2) + (lane & 15)) * kRowBytes;
    const int w_frag = XTileBytes + (wn * 64 + (lane & 15)) * kRowBytes;
    const int k_off[2] = {s0 * 16, (s0 ^ 4) * 16};

    floatx4 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- prologue (all waves together): tile 0 complete, positions half 0 of tile 1 ----
    issue_w(0, 0, wave * 4, std::integral_constant<int, 4>());
    if constexpr (SCHED == 3) issue_w(1, 1, wave * 4, std::integral_constant<int, 4>());
    issue_x(0, g, 0);
    if (g == 0) issue_x(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    if (g == 1) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0
    __builtin_amdgcn_sched_barrier(0);

    unsigned long long t_seg[4] = {0, 0, 0, 0};       // ABL & 8: cycles in phase 1, barrier 1, phase 2, barrier 2
    unsigned long long t_mark = 0;
    auto stamp = [&](int seg) {
        if constexpr (ABL & 8) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (seg >= 0) t_seg[seg] += now - t_mark;
            t_mark = now;
        }
    };
    stamp(-1);

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        half8 wf[2][FN], xf[2][FM];
        const char* const base = lds + cur * BufBytes;

        auto read_fragments = [&]() {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
                    wf[kk][fn] = *reinterpret_cast<const half8*>(base + w_frag + fn * 2048 + k_off[kk]);
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    xf[kk][fm] = *reinterpret_cast<const half8*>(base + x_frag + fm * 2048 + k_off[kk]);
            }
        };
        auto contract = [&](int kk) {
            if constexpr (ABL & 1) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) asm volatile("" ::"v"(wf[kk][fn]));
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) asm volatile("" ::"v"(xf[kk][fm]));
            } else {
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn)
                        acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][fn], xf[kk][fm], acc[fn][fm], 0, 0, 0);
            }
        };

        if constexpr (SCHED == 2) {
            // ---------------- L(step) ----------------
            if constexpr (!(ABL & 2)) {
                if (g == 0) {
                    if (step + 1 < nsteps) issue_w(step + 1, cur ^ 1, wn * 8, std::integral_constant<int, 8>());
                } else {
                    if (step + 2 < nsteps) issue_x(step + 2, 0, cur);
                    if (step + 1 < nsteps) issue_x(step + 1, 1, cur ^ 1);
                }
            }
            read_fragments();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of this tile has retired
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            __builtin_amdgcn_sched_barrier(0);

            // ---------------- M(step) ----------------
            __builtin_amdgcn_s_setprio(1);
            contract(0);
            contract(1);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA of the previous L phase has landed
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            if (!(g == 1 && step == nsteps - 1)) __builtin_amdgcn_s_barrier();
            stamp(3);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // ---------------- b(step): reads + first half of the MFMAs ----------------
            read_fragments();
            contract(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // DMA of the previous a phase landed; reads retired
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            __builtin_amdgcn_s_barrier();
            stamp(1);
            __builtin_amdgcn_sched_barrier(0);

            // ---------------- a(step): DMA issue + second half of the MFMAs ----------------
            if constexpr (!(ABL & 2)) {
                if (g == 0) {
                    if (step + 2 < nsteps) issue_x(step + 2, 0, cur);
                    if (step + 1 < nsteps) issue_x(step + 1, 1, cur ^ 1);
                } else {
                    if (step + 2 < nsteps) issue_w(step + 2, cur, wn * 8, std::integral_constant<int, 8>());
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            contract(1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            if (!(g == 1 && step == nsteps - 1)) __builtin_amdgcn_s_barrier();
            stamp(3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (ABL & 8) {
        if (a.timing != nullptr && lane == 0) {
            unsigned long long* out = a.timing + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = t_seg[i];
        }
    }

    // ---- epilogue (identical to conv3x3_tile_kernel) ----
    const int cout = a.cout;
    const int co0 = n0 + wn * 64 + (lane >> 4) * 4;
    const bool skip = a.alpha2 != 0.0f;
    const bool relu = a.relu != 0;
    half4 b4[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) b4[fn] = *reinterpret_cast<const half4*>(a.bias + co0 + fn * 16);

    auto finish = [&](int fm, const half4 (&z4)[FN]) {
        const int pos = m0 + g * (BM / 2) + fm * 16 + (lane & 15);
        const long out = (long)pos * cout + co0;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            half4 y4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __builtin_fmaf(a.alpha1, acc[fn][fm][r], (float)b4[fn][r]);
                if (skip) t = __builtin_fmaf(a.alpha2, (float)z4[fn][r], t);
                if (relu && !(t > 0.0f)) t = 0.0f;
                y4[r] = round_to_half(t);
            }
            if (pos < a.num_pos) *reinterpret_cast<half4*>(a.y + out + fn * 16) = y4;
        }
    };

    if (skip) {
        half4 z4[FM][FN];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            int pos = m0 + g * (BM / 2) + fm * 16 + (lane & 15);
            pos = pos < a.num_pos ? pos : a.num_pos - 1;
            const uint16_t* zp = a.z + (long)pos * cout + co0;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) z4[fm][fn] = *reinterpret_cast<const half4*>(zp + fn * 16);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) finish(fm, z4[fm]);
    } else {
        const half4 none[FN] = {};
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) finish(fm, none);
    }
}

}  // namespace

bool conv_tile_supported(int cin, int cout)
{
    return cin >= 64 && cin % 64 == 0 && cout >= 256 && cout % 256 == 0;
}

// Picks the position-tile height that wastes the fewest CU-rounds (256 CUs, one workgroup per CU).
int conv_tile_pick_fm(int num_pos, int cout)
{
    const int candidates[3] = {8, 6, 4};
    int best = 8;
    double best_cost = 1e30;
    for (int fm : candidates) {
        const long bm = 32 * fm;
        const long tiles = ((num_pos + bm - 1) / bm) * (cout / 256);
        const long rounds = (tiles + 255) / 256;
        // time ~ rounds * tile work; smaller tiles pay a little more per FLOP for filter traffic
        const double cost = (double)rounds * bm * (1.0 + 8.0 / bm);
        if (cost < best_cost) { best_cost = cost; best = fm; }
    }
    return best;
}

hipError_t launch_conv3x3_tile(const ConvTileLaunch& args, int fm, int variant, hipStream_t stream)
{
    if (args.num_pos <= 0) return hipSuccess;
    const int bm = 32 * fm;
    dim3 grid((args.num_pos + bm - 1) / bm, args.cout / 256);
    if (variant == 1) {
        switch (fm) {
            case 8: hipLaunchKernelGGL((conv3x3_tile_kernel<8>), grid, dim3(kThreads), 0, stream, args); break;
            case 6: hipLaunchKernelGGL((conv3x3_tile_kernel<6>), grid, dim3(kThreads), 0, stream, args); break;
            case 4: hipLaunchKernelGGL((conv3x3_tile_kernel<4>), grid, dim3(kThreads), 0, stream, args); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (fm) {
            case 8:
                if (variant == 3) hipLaunchKernelGGL((conv3x3_tile2_kernel<8, 3>), grid, dim3(kThreads), 0, stream, args);
                else hipLaunchKernelGGL((conv3x3_tile2_kernel<8, 2>), grid, dim3(kThreads), 0, stream, args);
                break;
            case 6: {
                static const int ablate = std::getenv("DGNN_CONV_ABLATE") ? std::atoi(std::getenv("DGNN_CONV_ABLATE")) : 0;
                if (ablate == 8 && variant == 3) {
                    hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 3, 8>), grid, dim3(kThreads), 0, stream, args);
                } else if (ablate == 8) {
                    hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 2, 8>), grid, dim3(kThreads), 0, stream, args);
                } else if (variant == 3) {
                    switch (ablate) {
                        case 1: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 3, 1>), grid, dim3(kThreads), 0, stream, args); break;
                        case 2: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 3, 2>), grid, dim3(kThreads), 0, stream, args); break;
                        case 3: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 3, 3>), grid, dim3(kThreads), 0, stream, args); break;
                        default: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 3, 0>), grid, dim3(kThreads), 0, stream, args); break;
                    }
                } else {
                    switch (ablate) {
                        case 1: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 2, 1>), grid, dim3(kThreads), 0, stream, args); break;
                        case 2: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 2, 2>), grid, dim3(kThreads), 0, stream, args); break;
                        case 3: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 2, 3>), grid, dim3(kThreads), 0, stream, args); break;
                        default: hipLaunchKernelGGL((conv3x3_tile2_kernel<6, 2, 0>), grid, dim3(kThreads), 0, stream, args); break;
                    }
                }
                break;
            }
            case 4:
                if (variant == 3) hipLaunchKernelGGL((conv3x3_tile2_kernel<4, 3>), grid, dim3(kThreads), 0, stream, args);
                else hipLaunchKernelGGL((conv3x3_tile2_kernel<4, 2>), grid, dim3(kThreads), 0, stream, args);
                break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

}  // namespace dgnn
