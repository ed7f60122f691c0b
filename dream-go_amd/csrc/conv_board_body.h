// Shared by conv_board.hip (one launch per layer, layout conversions) and conv_board_tower.hip (the tower kernels): the body of the
// residual-tower convolution.  Two translation units so that the two halves of the instantiations compile side by side.
#pragma once
// conv3x3_board_kernel -- residual-tower convolution on the PADDED SLOT-MAJOR activation layout, one 19x19 board
// per workgroup (C_out a multiple of 64, C_in a multiple of 32).  Same contract as conv3x3_kernel
// (kernels.hip):  y = act(alpha1 * conv3x3(x, w) + alpha2 * z + bias)   [conv2d.rs:171-220 of the reference].
//
// Activation layout (x, z, y):  [board][C / 8][400 rows][8 channels] fp16.  A board is 20 rows x 20 columns, row =
// 20 * y + x, with column 19 and row 19 ZERO.  Consequences:
//   * a 3x3 tap is the constant row shift 20 * dy + dx, and every tap that leaves the board lands on a zero that
//     is physically there (x = -1 is column 19 of the row above, y = -1 is row 19 of the previous board / the
//     zero halo) -- the K loop contains no masking and no per-tap address arithmetic;
//   * a 16-byte k-slot (8 channels) of 64 consecutive rows is 1 KiB contiguous in HBM: one LDS-DMA instruction,
//     perfectly coalesced, and 16 or 32 consecutive rows of one k-slot are contiguous in LDS: conflict-free
//     ds_read_b128 fragment reads without a swizzle.
//
// Arithmetic: v_mfma_f32_16x16x32_f16 from inline asm, A = a 16-channel x 32-k filter fragment, B = 16 board rows x the chunk's 32
// input channels (the lane's 16-byte row of k-slot lane >> 4): a 16 x 16 accumulator tile per MFMA, one 32-deep dot product per
// (chunk, tap).  The part charges 14 % less for this shape than for 32x32x16 at its power cap (tools/mfma_energy.hip).
// Work splits (struct Split below): the shipped one at 256 channels is four waves x (64 channels x all 384 rows) = 96 accumulator
// tiles = 384 registers per wave, one wave per SIMD; smaller splits share a board between 2 .. 16 workgroups.
//
// K loop: chunk-major (32 input channels), tap-minor.
//   activations  one chunk = 4 k-slots x 400 rows is DMA'd into LDS ONCE and serves all 9 taps (the tap is an
//                immediate offset of the fragment read).  Double buffered; the next chunk's 28 DMA pieces are spread
//                over the first taps of the current chunk; ONE workgroup barrier per chunk (8 per layer at C_in = 256).
//   filter       pre-packed by the host in MFMA fragment order.  Wide and four-slice splits: straight from memory into the
//                registers of the NEXT k-step (DirectA); one- and two-slice splits: the workgroup's rows of a chunk beside the
//                chunk's activations in the same double buffer (Chunked).  Every wave's ordering is its own counted s_waitcnt.
// rows 384..399 are pad row 19 and are never computed.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>
#include <utility>

namespace dgnn {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int kRows = kBoardRows;                      // 400 padded rows per board
constexpr int kHalo = 21;                              // rows above row 0 that a tap can reach (20 + 1)
constexpr int kPieces = 7;                             // DMA instructions per k-slot (64 rows each, rows 400..447 unused)
constexpr int kSlotRows = 480;                         // 21 + 7 * 64 = 469 rows used; the k-slot stride is a multiple of 256 bytes:
                                                       // a B fragment takes 16 rows from each of the chunk's four k-slots, and the
                                                       // lane groups a ds_read_b128 is served in ({0-3, 12-15, 20-27}, ...) then
                                                       // fall on 64 different banks
static_assert(kSlotRows >= kHalo + kPieces * 64 && kSlotRows % 16 == 0, "k-slot stride");
constexpr int kSlotBytes = kSlotRows * 16;             // 7680
constexpr int kXBufBytes = 4 * kSlotBytes;             // 30720: one 32-channel chunk
constexpr int kRing = 3;                               // filter slabs per wave in flight / in use
constexpr int kWBase = 2 * kXBufBytes;
constexpr int kThreads = 512;

// two fp32 -> packed fp16 (round to nearest even, like v_cvt_f16_f32) and max(packed fp16, 0): the epilogue's arithmetic at
// a third of the instructions.  As asm so that the fp32 fma in front stays a separate instruction with its own rounding
// (the oracle rounds twice; hipcc would otherwise fuse fma + cvt into v_fma_mixlo_f16).
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi)
{
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned pk_relu_f16(unsigned v)
{
    unsigned r;
    asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ds_read_b128 at a VGPR byte address + immediate; completion is the caller's business (wait_lgkm)
template <int OFF>
__device__ __forceinline__ half8 lds_read(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    half8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// s_waitcnt lgkmcnt(N).  The readers of the fragments it releases are asm volatile statements too (the MFMAs), and volatile asm
// statements keep their program order, so the wait names no operands: a wait that "produces" a fragment ("+v") makes hipcc put
// an s_nop between it and the MFMA behind it (its hazard rule for a register an asm statement defines) -- 24 of them per k-step.
template <int N, class... Frag>
__device__ __forceinline__ void wait_lgkm(Frag&...)
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// The same waits where COMPILER code reads the fragments next (the copies at a chunk boundary): there the wait must "produce" them,
// or hipcc is free to move its v_mov above the wait.
template <int N>
__device__ __forceinline__ void wait_lgkm_for(half8& f0, half8& f1)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm_for(half8& f0, half8& f1, half8& f2, half8& f3)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm_for(half8& a0, half8& a1)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_for(half8& a0, half8& a1, half8& a2, half8& a3)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>());
}

// Software-pipelined K loop: number of LDS reads issued AFTER the ones the MFMA of pair I needs and before that MFMA.
// Program order of a step: [first step of a chunk: B reads of pairs 0..BD-2] then for every pair p: [p == AP: the two A reads
// of the NEXT step] [the B read of sequence index p + BD - 1: this step's pair, or the next step's unless `last`] [MFMA p].
// A position p < 0 is pair NP + p of the previous step (for a first step: the preload slot).
template <int NP, int BD, int AP>
constexpr int lds_younger(int I, bool first, bool last)
{
    const int pb = I - (BD - 1);                      // position whose B read MFMA I consumes
    const int pa = first ? -1000 : AP - NP;           // position (previous step) that requested this step's A fragments
    int n = 0;
    for (int p = (pb > pa ? pb : pa); p <= I; ++p) {
        const int q = p < 0 ? p + NP : p;
        const bool in_prev = p < 0;
        const bool has_a = q == AP && !(in_prev && first);
        const int t = p + BD - 1;
        const bool has_b = in_prev ? true : (t < NP || !last);
        if (p == pb && p == pa) {}                                   // both needed reads sit here: nothing of it is younger
        else if (p == pb) {}                                         // its B is the needed one; its A came before that
        else if (p == pa) { if (has_b) n += 1; }                     // its A is the needed one; its B is younger
        else { if (has_a) n += 2; if (has_b) n += 1; }
    }
    return n;
}

// The same count for the order [MFMA p] [p == AP: A reads of the next step] [B read of sequence index p + BD]: the read that
// refills a register is issued right BEHIND the MFMA that consumed it (a read that has to queue for the LDS then never stands
// between a wave and an MFMA whose operands are already there), and the rotation is BD deep instead of BD - 1.
template <int NP, int BD, int AP, bool A_NOT_AT_LAST = false, int NA = 2>
constexpr int lds_younger_mfma_first(int I, bool first, bool last)
{
    const int pb = I - BD;
    const int pa = first ? -1000 : AP - NP;
    int n = 0;
    auto events = [&](int p, bool only_b) {
        const int q = p < 0 ? p + NP : p;
        const bool in_prev = p < 0;
        const bool has_a = q == AP && !(in_prev && first) && !(A_NOT_AT_LAST && last && !in_prev);
        const int t = p + BD;
        const bool has_b = in_prev ? true : (t < NP || !last);
        return (has_a && !only_b ? NA : 0) + (has_b ? 1 : 0);
    };
    if (pa > pb) n += events(pa, true);                       // the needed A reads sit at pa; that position's B read is younger
    for (int p = (pa > pb ? pa : pb) + 1; p <= I - 1; ++p) n += events(p, false);
    return n;
}

// v_mfma_f32_32x32x16_f16 with the accumulator's register file chosen by the caller (AGPR: "a" / VGPR: "v").  For the wide
// variant only (Split<16>): hipcc neither schedules nor pads what is inside -- the callers keep the rules of the CDNA guide's
// inline-asm section: operands come from LDS reads behind their own s_waitcnt; consecutive MFMAs on one accumulator are an
// accumulate chain (no wait states); any OTHER reader or writer of an accumulator goes through mfma_fence() first.
template <bool AGPR>
__device__ __forceinline__ void mfma_asm(floatx4& c, const half8& a, const half8& b)
{
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// The first MFMA of a layer on an accumulator takes the inline constant 0 as its C operand: no zero-initialisation of 384 registers
// per layer (320 VALU instructions of a tower layer's prologue).  Early clobber: a destination in VGPRs must not land on A or B.
template <bool AGPR>
__device__ __forceinline__ void mfma_asm_init(floatx4& c, const half8& a, const half8& b)
{
    if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// 8-pass MFMA result -> any reader but the next MFMA of the chain: 12 wait states; VALU write -> MFMA operand: 2.  One
// statement in front of (BEFORE = true) or behind the VALU code that touches the tile, tied to it through the operand.
template <bool AGPR, bool BEFORE>
__device__ __forceinline__ void mfma_fence(floatx4& c0, floatx4& c1, floatx4& c2, floatx4& c3)
{
    if constexpr (AGPR) {
        if constexpr (BEFORE) asm volatile("s_nop 11" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
        else asm volatile("s_nop 1" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    } else {
        if constexpr (BEFORE) asm volatile("s_nop 11" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        else asm volatile("s_nop 1" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    }
}

// Work splits, template parameter NS (a workgroup covers 32 * SL output channels of one board; grid.y = C_out / (32 * SL)):
//   NS = 16  ("wide", the shipped split at 256 channels: a workgroup per board) 8 slices on FOUR waves x two 32-channel tiles x
//            all 384 rows: a B fragment feeds four 16-channel MFMA tiles, every filter byte is fetched once.  384 accumulator
//            registers per wave (one wave per SIMD, 512 registers): 32-row tiles 0..7 in AGPRs, 8..11 in VGPRs, spelled out in
//            the asm constraints.  (NS = 8, the eight-wave split of rounds 1-2 -- one 32-channel tile per wave, twice the LDS
//            reads -- is gone: DGNN_BOARD_NS=8 means 16.)
//   NS = 4   4 slices x 2 row groups of 192 rows, eight waves (6 row tiles per wave)      C_out % 128 == 0
//   NS = 2   2 slices x 4 row groups of 96 rows, eight waves (3 row tiles per wave)       C_out % 64 == 0
//   NS = 1   1 slice x 12 row groups of 32 rows (one row tile per wave), TWELVE waves, three per SIMD: the latency variant for
//            small batches (twice the workgroups per board).  [Four waves x 3 row tiles, one per SIMD, is slower: round 1 and
//            again round 3 -- 0.25 against 0.224 ms per pass at 9x128.]
//   NS = 32  the one slice of NS = 1 on HALF a board: six waves x one row tile, two workgroups per channel block (row split)
// Smaller splits = more, smaller workgroups per board: narrow networks, and filling the chip at small batch sizes.
// Filter packing (host, pack_filter_board): [k-step][C_out / 16 tiles][64 lanes][8 halfs], k-step = chunk * 9 + tap,
// lane l holds w[co = tile * 16 + (l & 15)][tap][ci = chunk * 32 + (l >> 4) * 8 .. + 8].
// NS = 1, 2, 32 are CHUNKED: the waves of such a workgroup share few slices, so wave-private fetches would bring every filter byte
// 4 .. 12 times.  Instead the workgroup's filter rows of one 32-channel chunk (9 taps x SL slices x 2 KiB) sit next to the chunk's
// activations in the SAME double buffer, fetched once by all waves together and published by the chunk barrier that is there
// anyway.  (Per-wave global loads of the fragments -- DirectA for these splits -- were measured in round 3: 0.6 -> 1.1 ms per pass.)
// The kernel runs at the POWER-LIMITED shader clock: saved cycles partly come back as a lower clock unless the joules per MFMA go
// down too.
template <int NS>
struct Split {
    // NS = 32 ("row split", round 3): the ONE slice of NS = 1 on half a board -- (board, channel block, row half) workgroups of six
    // waves x one 32-row tile.  The smallest-batch split is bound by the LDS (twelve waves read the same two filter fragments and
    // their own two activation fragments per k-step: 48 KB against 128 B per clock); taking the parallelism from the rows instead
    // of more waves halves that per CU and fills 256 CUs with 16 boards of 256 channels or 32 boards of 128.
    static constexpr int RS = NS == 32 ? 2 : 1;        // row halves a board is split into (workgroups per channel block)
    static constexpr int SL = NS == 16 ? 8 : NS == 32 ? 1 : NS;       // 32-channel slices per workgroup
    static constexpr int CT = NS == 16 ? 2 : 1;        // 32-channel MFMA tiles per wave
    static constexpr int CW = SL / CT;                 // waves along the channels
    static constexpr bool Chunked = NS <= 2 || NS == 32;
    // Every variant issues its MFMAs from inline asm with the accumulator tied to the destination ("+a" / "+v"): as a builtin the
    // four-pass v_mfma_f32_16x16x32_f16 is untied to hipcc, whose register allocator then writes results into fresh registers
    // and spills them (343 spilled registers in the K loop of the eight-wave skip variant).
    static constexpr bool AsmMfma = true;
    // the wide variant needs both register files: 32-row tiles 0..7 of both channel tiles in AGPRs (exactly 256), 8..11 in VGPRs
    static constexpr bool acc_in_agpr(int rt32) { return NS == 16 && rt32 < 8; }
    // DirectA (round 3, the wide variant): the A (filter) fragments go from memory straight into the registers of the NEXT k-step
    // (global_load_dwordx4, one k-step = 1536 MFMA-cycles ahead: more than a trip to HBM) instead of through a wave-private LDS
    // ring.  With one wave per 64 channels no two waves ever want the same filter bytes, so the ring bought nothing but prefetch
    // depth -- at the price of 16 KiB of LDS writes, four LDS-DMA instructions (the dearest instruction to issue beside MFMAs) and
    // four ds_read_b128 per k-step and wave.
    static constexpr bool DirectA = NS == 16 || NS == 4;
    static constexpr int WAVES = NS == 1 ? 12 : NS == 16 ? 4 : NS == 32 ? 6 : 8;
    static constexpr int MinWavesPerSimd = 1;
    static constexpr int Threads = WAVES * 64;
    static constexpr int XP = (4 * (NS == 32 ? 4 : NS <= 2 ? 6 : kPieces) + WAVES - 1) / WAVES;   // activation pieces per wave and chunk: 4 (one repeated), 7, 3 (row split), 2 / 3 (twelve / eight waves)
    static constexpr int RG = WAVES / CW;              // row groups of the workgroup
    static constexpr int RT = 12 / RS / RG;            // 32-row MFMA tiles per wave
    // activation pieces (64 rows) per k-slot the workgroup fetches: all seven, or -- a row half needs its 192 rows and 21 above /
    // below -- pieces 0..3 (upper half: rows 0..212 of 0..255) / 2..5 (lower half: rows 171..383; what lies below row 383 is the pad row
    // and beyond, zeros that the workgroup writes into its LDS once per layer instead of fetching them: round 3 fetched five pieces each)
    // The chunked whole-board splits (NS = 1, 2) fetch six: rows 384..447 of piece 6 are the pad row and what lies behind the board.
    static constexpr int XPS = RS == 2 ? 4 : (NS <= 2 ? 6 : kPieces);
    static constexpr int first_piece(int rh) { return RS == 2 && rh ? 2 : 0; }
    static constexpr int W = 2 * CT;                   // DMA instructions (1 KiB fragments) per filter slab: (tile, k16)
    static constexpr int SlabBytes = W * 1024;
    static constexpr int RingBytes = kRing * SlabBytes;
    static constexpr int FChunkBytes = 9 * SL * 2048;  // Chunked: the workgroup's filter rows of one chunk, [tap][slice][k16][lane][8]
    static constexpr int BufStride = Chunked ? kXBufBytes + FChunkBytes : kXBufBytes;   // one of the two chunk buffers
    static constexpr int LdsBytes = Chunked ? 2 * BufStride : DirectA ? kWBase : kWBase + WAVES * RingBytes;
    // StageF (round 4, the one-slice chunked splits): the workgroup's filter rows of the NEXT chunk do not come by LDS-DMA but by
    // ordinary 16-byte global loads into registers (FPW per wave, requested at the start of the chunk) that are written to the other
    // buffer with ds_write_b128 at the chunk boundary.  Why: per-tap stamps (tools/tap_trace.py, 9x128 batch 32) show the taps that
    // carry no DMA at 125-165 cycles and a 1 000-cycle wait at every chunk boundary for the chunk's 38 KB to land -- 256 workgroups x
    // 38 KB per 3 180 cycles is 12 B per clock and CU, 6.5 TB/s chip-wide: the rate of the LDS-DMA path itself (the guide's "all-LDS-DMA
    // prologue: 12-13 B/cyc/CU").  Half of those bytes are the filter, which the vector L1 path can carry instead.
    // MEASURED (tools/ab_stage.sh, profiles/r04/ab_stage_filter.txt, same box): no gain -- 9x128 batch 32 155.9-158.8 k against 156.1-157.4 k
    // evals/s, 20x256 batch 32 46.8 k against 49.4 k.  The limit is not the LDS-DMA path but what a CU can pull through the fabric when
    // all 256 do (11-13 B per clock and CU either way).  Off by default; bit-identical when on.
#ifndef DGNN_STAGE_FILTER
#define DGNN_STAGE_FILTER 0
#endif
    static constexpr bool StageF = DGNN_STAGE_FILTER != 0 && Chunked && SL == 1;
    static constexpr int FPW = StageF ? (18 * SL + WAVES - 1) / WAVES : 0;      // staged filter pieces (1 KiB) per wave and chunk: 3 (six waves) / 2 (twelve)
    static constexpr int Pieces = StageF ? 4 * XPS : 4 * XPS + 18 * SL;          // Chunked: DMA instructions per chunk, activations then (unless staged) filter
    static constexpr int PPW = (Pieces + WAVES - 1) / WAVES;        // ... per wave (4 at NS = 1, 8 at NS = 2; surplus ones repeat)
    // Chunked: piece j of a wave goes out behind the MFMA of pair piece_pair(j) of tap piece_tap(j) (early in the chunk: it
    // has to land before the chunk barrier)
    // One slice per workgroup (SL == 1: the smallest-batch splits): kPiecesPerTap pieces of the next chunk go out per tap.  Round 3 sent
    // ONE (the last of a wave's seven at tap 6, three taps in front of the chunk barrier that waits for it); round 4 measured 2 / 4 / 7
    // (tools/build_variant.sh, tools/ab_ppt.sh, profiles/r04/ab_pieces_per_tap.txt; same box, evals/s): 9x128 batch 32: 154.4 k ->
    // 159.4 k / 157.0 k / 155.0 k; 20x256 batch 8 / 16: 13.4 k / 26.6 k -> 14.5 k / 28.6 k with two.  Two it is: the whole chunk in
    // flight at once (seven) buys nothing -- the K loop of these splits is not waiting for bytes in flight (what bounds it: DESIGN 13.4).
#ifndef DGNN_PIECES_PER_TAP
#define DGNN_PIECES_PER_TAP 2
#endif
    static constexpr int kPiecesPerTap = DGNN_PIECES_PER_TAP;
    static constexpr int piece_tap(int j) { return SL == 1 ? j / kPiecesPerTap : j / 2; }
    static constexpr int piece_pair(int j) { return SL == 1 ? 1 : (j % 2 ? 4 : 1); }
    static constexpr int pieces_in_tap(int tap) { int n = 0; for (int j = 0; j < PPW; ++j) n += piece_tap(j) == tap ? 1 : 0; return n; }
    static constexpr int CT2 = 2 * CT;                 // 16-channel MFMA tiles per wave
    static constexpr int RT2 = 2 * RT;                 // 16-row MFMA tiles per wave
    static constexpr int NP = RT2;                     // "pairs" per k-step: one B fragment (16 rows x the chunk's 32 channels), CT2 MFMAs each
    static constexpr int BD = NS == 16 ? 3 : NS == 4 ? 4 : NS == 2 ? 5 : 2;   // B-fragment registers in rotation = pairs of read-ahead (<= NP)
    // byte offset of the B fragment of pair i = 16-row tile i relative to the lane's base row
    static constexpr int b_offset(int i) { return i * 256; }
    // pair index behind which the j-th filter piece / the activation piece of the step is issued
    static constexpr int w_pair(int j) { return NS == 16 ? 2 + 4 * j : NS == 4 ? 1 + 3 * j : j; }
    static constexpr int x_pair() { return NS == 16 ? 17 : NS == 4 ? 6 : 2; }
    // pair at which the NEXT step's A fragments are requested (behind this step's DMA instructions)
    static constexpr int a_pair() { return NS == 16 ? 19 : NS == 4 ? 8 : NS == 2 ? 3 : 0; }
};
static_assert(Split<4>::LdsBytes <= 160 * 1024 && Split<2>::LdsBytes <= 160 * 1024 && Split<16>::LdsBytes <= 160 * 1024, "LDS budget");
static_assert(Split<1>::BD <= Split<1>::NP && Split<2>::BD <= Split<2>::NP && Split<32>::BD <= Split<32>::NP, "the B rotation reaches one step ahead at most");
static_assert(Split<32>::PPW <= 9 && Split<32>::RT == 1 && Split<32>::NP == 2, "row split: a piece per tap");

// A board split over G = C_out / (32 * NS) workgroups INSIDE one launch (tower_split_kernel): what a workgroup needs to
// know about the others.  Every workgroup publishes "layer i of my channel block is in memory" by storing i + 1 into
// done[board * G + nb] after its write-through stores have drained; a consumer polls the ONE word of the producer whose
// channels it is about to fetch.  No grid-wide barrier, no fence: the payload is stored sc1 (write-through) and loaded sc1
// (past the reader's L1), the recipe of the CDNA guide's inter-workgroup rule (Guideline 16, form "sc1 both sides").
struct SplitSync {
    unsigned* done;      // [board][G]: `base` + layers published.  Never reset: every launch gets a `base` above anything an
                         // earlier launch stored (two fill kernels per pass were 9 us of a 230 us pass at 9x128, batch 32)
    unsigned* err;       // one word: set when a poll gave up (the host then repeats the pass with one launch per layer)
    int epoch;           // layers of THIS launch that precede this one: 0 = the input comes from outside, nothing to wait for
    int groups;          // G
    unsigned base;       // what the flags of this launch count from
    unsigned long long* trace;   // measurement only (DGNN_SPLIT_TRACE=<file>): [layer][board][nb][4] s_memtime stamps of wave 0, or nullptr
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one lane: wait until `flag` says the producer has published `epoch` layers.  The producer runs already or is the next workgroup
// of its queue to be given a compute unit (tickets, conv_board_tower.hip), so the wait ends as soon as what is in front of the
// launch does.  Bounded all the same, as the net under that argument: a queue must not hang on a device that something endless
// holds.  The limit is WALL time: s_memrealtime, the 100 MHz constant clock (s_memtime counts shader cycles, MI355X_MICROARCH.md).
// kGiveUpTicks = 50 ms: an order of magnitude above what a leaf pool's own towers in front of a split launch take (3 ms each at
// 20x256, two or three alive), and short enough that the three give-ups before a workspace stays per layer (engine.cpp:
// kSplitMaxFailures) add up to 0.15 s.
constexpr unsigned long long kGiveUpTicks = 5000000ull;
__device__ __forceinline__ void wait_published(unsigned* flag, unsigned epoch, unsigned* err)
{
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kGiveUpTicks) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// 16-byte activation store / load.  SPLIT: write-through to memory (sc1) so that the other workgroups of the board see it
// once this wave's vmcnt has drained, and loads that never look into this CU's L1.
template <bool SPLIT>
__device__ __forceinline__ void store_row(uint16_t* p, const uint4& v)
{
    if constexpr (SPLIT) {
        const u32x4 t = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
    } else {
        // spelled as a GLOBAL store: a pointer that came out of a descriptor in memory (tower kernels) or through an opaque
        // asm is generic to hipcc, and a flat_store / flat_load counts in vmcnt AND lgkmcnt -- hipcc then waits for a flat
        // load with "vmcnt(0) lgkmcnt(0)", i.e. for every DMA piece in flight and every LDS read of the software pipeline
        const u32x4 t = {v.x, v.y, v.z, v.w};
        *reinterpret_cast<__attribute__((address_space(1))) u32x4*>((__UINTPTR_TYPE__)p) = t;
    }
}
template <bool SPLIT>
__device__ __forceinline__ uint4 load_row(const uint16_t* p)
{
    const auto* const g = reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((__UINTPTR_TYPE__)p);   // global, not flat (store_row)
    // SPLIT (epilogue-skip variant only): the rows come from another workgroup of this launch, so the load must not be served by
    // this CU's L1.  hipcc has no builtin for an sc1 load of 16 bytes; the non-temporal load it does have (`nt`) bypasses the L1
    // as well (MI355X_MICROARCH.md, visibility table: "sc1 / sc0 sc1 / nt loads bypass L1 only") and is tracked by the compiler's
    // own waitcnt insertion, which this code path relies on.  The K-loop form (load_row_asm) spells sc1.
    const u32x4 t = SPLIT ? __builtin_nontemporal_load(g) : *g;
    return make_uint4(t[0], t[1], t[2], t[3]);
}

// The skip rows folded in during the K loop are requested from inline asm: a load hipcc tracks is waited for with the
// vmcnt it can prove -- with LDS-DMA instructions in the stream that is vmcnt(0), i.e. every filter slab and activation piece
// in flight, twelve times per skip layer.  As asm the load is waited for by the kernel's own counted vmcnt (wait_z below, which
// names the destinations so that no consumer moves above it).
template <bool SPLIT>
__device__ __forceinline__ void load_row_asm(u32x4& dst, const uint16_t* p)
{
    if constexpr (SPLIT) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");   // "sc1 both sides", like the DMA of the input chunks
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// The same load with the destination TIED ("+v"): for a load that sits under a wave-uniform branch (the skip rows of the splits in which
// a wave owns only some row tiles).  An "=v" destination is a new value; at the join behind the branch hipcc would have to merge it with
// the old one and may do so with a copy -- of a register whose data has not landed (tools/asm_lint.py finds such copies).  Tied, old and
// new value are one register and there is nothing to merge.
template <bool SPLIT>
__device__ __forceinline__ void load_row_asm_tied(u32x4& dst, const uint16_t* p)
{
    if constexpr (SPLIT) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_z(u32x4& z0, u32x4& z1)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(z0), "+v"(z1) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_z(u32x4& z0, u32x4& z1, u32x4& z2, u32x4& z3)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "n"(N) : "memory");
}

// DirectA: a filter fragment straight into registers, and the counted wait that hands four of them to the MFMAs behind it
// (scalar base + the lane's byte offset + an immediate: the k-step's base is one scalar add, no VALU address arithmetic and no
// register an asm statement "defines" in front of the load -- hipcc pads that with an s_nop)
template <int OFF>
__device__ __forceinline__ void load_a_asm(half8& dst, const uint16_t* sbase, unsigned lane_bytes)
{
    static_assert(OFF >= 0 && OFF < 4096, "global_load immediate offset");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_bytes), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_a()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");      // no "+v" operands: see wait_lgkm
}

// SKIP: 0 = no skip input, 1 = skip added in the epilogue (general), 2 = skip accumulated DURING the K loop (alpha1 != 0):
// the row tiles of z are fetched one at a time while the memory system is otherwise idle and added to the fp32
// accumulators as (alpha2 / alpha1) * z, so that the epilogue is store-only.  (With the skip read in the epilogue all
// 256 workgroups fetch their 205 KiB of z at the same moment: 52 MB at HBM speed, 8 us of a 90 us kernel.)
// ABL = ablation bits for measurements only (results are garbage): 1 drops the MFMAs, 2 the in-loop DMA, 4 the in-loop
// B-fragment reads, 8 the counted vmcnt waits of the k-steps (DMA still issued).
// One convolution of one workgroup: board `b`, channel block `nb`.  A __device__ function so that the per-layer kernel
// and the whole-tower kernel below share it; `lds` is the workgroup's S::LdsBytes of shared memory.
// SPLIT: the board is shared with other workgroups of the same launch (SplitSync above); otherwise `sync` is not looked at.
// `carry` / return value: what a layer of a tower launch leaves behind for the next one, bit kHandedOver = its last steps
// fetched the next layer's first filter slabs (chunk) and -- channel block 0 -- its epilogue left the next layer's first input
// chunk in LDS, bit kX0InFlight (SPLIT) = the next layer's first input chunk is already on its way into LDS, bit kChained = an
// earlier layer of this launch wrote this layer's input (set by the tower kernels from their second layer on), bit kStartBuffer
// (chunked kernels) = the chunk buffer the next layer starts in (the one its predecessor's last chunk did not use).
constexpr int kHandedOver = 2, kX0InFlight = 1, kChained = 4, kStartBuffer = 8;
template <int NS, int SKIP, bool RELU, int ABL, bool SPLIT = false>
__device__ __forceinline__ int conv_board_body(const ConvBoardLaunch& a, char* const lds, const int b, const int nb,
                                               half8 (&af)[2][Split<NS>::CT][2],
                                               const SplitSync sync = SplitSync{nullptr, nullptr, 0, 1, 0, nullptr}, const int carry = 0,
                                               const int rh = 0)
{
    using S = Split<NS>;
    constexpr int RS = S::RS;                        // row halves per channel block (2: `rh` says which one this workgroup is)
    static_assert(RS == 1 || (SPLIT && S::Chunked), "the row split exists inside the split tower only");
    constexpr int CT = S::CT, RT = S::RT, CT2 = S::CT2, RT2 = S::RT2, W = S::W, NP = S::NP, BD = S::BD, XP = S::XP, SL = S::SL;
    constexpr bool CHUNKED = S::Chunked;
    constexpr bool ASM = S::AsmMfma;                 // MFMAs from inline asm (Split<16>: accumulators of row tiles 0..7 in AGPRs)
    constexpr bool DIRECT_A = S::DirectA;            // filter fragments by global loads into the next step's registers, no LDS ring
    static_assert(!DIRECT_A || (ASM && !CHUNKED), "DirectA: asm MFMAs, filter not in the chunk buffer");
    constexpr int BUF = S::BufStride;
    static_assert(!CHUNKED || !(ABL & ~1), "the chunked kernels know ablation 1 (no MFMAs) only");
    const bool x0_in_flight = (carry & kX0InFlight) != 0;
    const bool carry_in = a.carry_in != 0 && (carry & kHandedOver) != 0;
    constexpr bool PIPE = !(ABL & 32);               // ABL 32: the K loop without the software pipeline across steps (A/B)
    constexpr int WI = (ABL & 64) ? 2 * W : W;       // ABL 64: every filter piece is requested TWICE (what does a DMA instruction cost?)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % S::CW;                     // channel position: slices wn * CT .. wn * CT + CT - 1 of the workgroup's SL
    const int g = wave / S::CW + rh * S::RG;         // row group OF THE BOARD: 32-row tiles g * RT .. g * RT + RT - 1
    const int cin = a.cin;
    const int cout = a.cout;
    const int nchunks = cin / 32;
    // Hand-over to the next layer of a tower launch.  Chunked: the next layer's first filter chunk is fetched into the buffer
    // the last chunk does NOT use, so that is where the next layer starts (and where its first input chunk is left).
    const bool hand_over = a.w_next != nullptr;
    const int buf0 = CHUNKED && (carry & kStartBuffer) ? 1 : 0;      // chunk c of this layer lives in buffer (buf0 + c) & 1
    const int buf_next = CHUNKED ? (buf0 + nchunks) & 1 : 0;

    const bool tracing = a.timing != nullptr || (SPLIT && sync.trace != nullptr);
    const unsigned long long t_start = tracing ? __builtin_amdgcn_s_memtime() : 0;
#ifdef DGNN_TAP_STAMPS      // measurement builds only (tools/build_variant.sh, tools/tap_trace.py): where the taps of ONE chunk start, per wave
    unsigned long long tap_stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // ---- zero the halo rows of both activation buffers (never touched by the DMA) ----
    if (tid < 2 * 4 * kHalo) {
        const int buf = tid / (4 * kHalo);
        const int rem = tid - buf * (4 * kHalo);
        const int slot = rem / kHalo;
        const int row = rem - slot * kHalo;
        *reinterpret_cast<uint4*>(lds + buf * BUF + slot * kSlotBytes + row * 16) = make_uint4(0, 0, 0, 0);
        // row split (lower half) and the chunked whole-board splits: rows 384..404 (the pad row and what a tap of row 383 reaches below it) are never fetched
        if ((RS == 2 && rh == 1) || S::XPS == 6) *reinterpret_cast<uint4*>(lds + buf * BUF + slot * kSlotBytes + (kHalo + 384 + row) * 16) = make_uint4(0, 0, 0, 0);
    }

    const uint16_t* const x_board = a.x + (long)b * cin * kRows + lane * 8;
    // one k-step of one 256-channel block is 8192 halfs; this wave's slab starts wn * W fragments of 512 halfs into it
    const int step_halfs = (cout / 32) * 1024;       // one k-step of the packed filter
    const uint16_t* const w_wave = a.w + (long)(nb * SL + wn * CT) * 1024 + lane * 8;

    // fragment f of k-step `st` of this wave's filter slice -> ring slot `slot`
    // steps past the last one belong to the next layer when there is one to hand over to (tower_board_kernel)
    const int nsteps_total = nchunks * 9;
    const uint16_t* const w_next_wave = a.w_next ? a.w_next + (long)(nb * SL + wn * CT) * 1024 + lane * 8 - (long)nsteps_total * step_halfs : w_wave;
    auto issue_w_piece = [&](int st, int slot, int f) {
        const uint16_t* src = (st >= nsteps_total ? w_next_wave : w_wave) + (long)st * step_halfs;
        char* const dst = lds + kWBase + wave * S::RingBytes + slot * S::SlabBytes;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 512),
                                         (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int st, int slot) {
#pragma unroll
        for (int f = 0; f < W; ++f) issue_w_piece(st, slot, f);
    };
    // j-th (0..3) activation piece of this wave for `chunk` -> buffer `buf`.  28 pieces, 8 waves: waves 4..7 have
    // only three and repeat their third one (same bytes to the same place), so that every wave issues the same
    // number of DMA instructions and the vmcnt bookkeeping below is one set of constants.
    auto issue_x_from = [&](const uint16_t* board, int chunk, int buf, int j) {
        int p = wave + S::WAVES * j;
        if (p >= 4 * S::XPS) p -= S::WAVES;
        const int s = p / S::XPS;
        const int i = p - s * S::XPS + S::first_piece(rh);
        const uint16_t* src = board + ((long)(chunk * 4 + s) * kRows + i * 64) * 8;
        char* const dst = lds + buf * BUF + s * kSlotBytes + (kHalo + i * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, SPLIT ? 16 : 0);   // aux 16 = sc1
    };
    auto issue_x = [&](int chunk, int buf, int j) { issue_x_from(x_board, chunk, buf, j); };
    // Chunked: piece q (0 .. 18 * NS - 1 = [tap][slice][k16]) of the workgroup's filter rows of `chunk` -> buffer `buf`.  Chunks
    // past the last one belong to the next layer when there is one to hand over to.
    const uint16_t* const w_group = a.w + (long)(nb * SL) * 1024 + lane * 8;
    const uint16_t* const w_next_group = a.w_next ? a.w_next + (long)(nb * SL) * 1024 + lane * 8 - (long)nsteps_total * step_halfs : w_group;
    auto issue_f = [&](int chunk, int buf, int q) {
        const int tap = q / (2 * SL), rest = q - tap * (2 * SL);
        const int st = chunk * 9 + tap;
        const uint16_t* src = (st >= nsteps_total ? w_next_group : w_group) + (long)st * step_halfs + rest * 512;
        char* const dst = lds + buf * BUF + kXBufBytes + q * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // Chunked: j-th DMA instruction of this wave for `chunk` (activations: the first 28 pieces, then the filter) -> buffer `buf`
    auto issue_piece = [&](int x_chunk, int f_chunk, int buf, int j) {
        int p = wave + S::WAVES * j;
        if (p >= S::Pieces) p -= S::WAVES;            // a surplus instruction repeats the wave's previous piece
        if (p < 4 * S::XPS) {
            const int sl = p / S::XPS;
            const int i = p - sl * S::XPS + S::first_piece(rh);        // a row half fetches five of the seven 64-row pieces
            const uint16_t* src = x_board + ((long)(x_chunk * 4 + sl) * kRows + i * 64) * 8;
            char* const dst = lds + buf * BUF + sl * kSlotBytes + (kHalo + i * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, SPLIT ? 16 : 0);
        } else {
            issue_f(f_chunk, buf, p - 4 * S::XPS);
        }
    };
    // StageF: this wave's FPW filter pieces of `chunk` requested into registers / written to buffer `buf` (piece q = wave + WAVES * k; a
    // surplus one repeats the previous: same bytes to the same place)
    u32x4 fstage[S::FPW > 0 ? S::FPW : 1];
    auto stage_f_load = [&](int chunk) {
        static_for<S::FPW>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            int q = wave + S::WAVES * K;
            if (q >= 18 * SL) q -= S::WAVES;
            const int tap = q / (2 * SL), rest = q - tap * (2 * SL);
            const int st = chunk * 9 + tap;
            const uint16_t* src = (st >= nsteps_total ? w_next_group : w_group) + (long)st * step_halfs + rest * 512;
            load_row_asm<false>(fstage[K], src);
        });
    };
    auto stage_f_store = [&](int buf) {       // the caller has waited for the loads (vmcnt) and names the registers
        static_for<S::FPW>([&](auto kc) {
            constexpr int K = decltype(kc)::value;
            int q = wave + S::WAVES * K;
            if (q >= 18 * SL) q -= S::WAVES;
            *reinterpret_cast<u32x4*>(lds + buf * BUF + kXBufBytes + q * 1024 + lane * 16) = fstage[K];
        });
    };
    // SPLIT: the flag of the workgroup that produces input chunk c (its channel block is c / SL).  Row split: a channel block has
    // two producers, flags [block][row half]; a consumer needs its own half of the rows and 21 of the other one, so BOTH -- its own
    // channel block included, whose other half a sibling workgroup writes.
    unsigned* const done_board = SPLIT ? sync.done + (long)b * sync.groups : nullptr;
    auto await_chunk = [&](int c) {       // lane 0 of wave 0 only; the caller's next barrier tells everybody
        const unsigned want = sync.base + (unsigned)sync.epoch;
        if constexpr (RS == 2) {
            wait_published(done_board + (c / SL) * 2 + (rh ^ 1), want, sync.err);
            if (c / SL != nb) wait_published(done_board + (c / SL) * 2 + rh, want, sync.err);
        } else {
            if (c / SL != nb) wait_published(done_board + c / SL, want, sync.err);
        }
    };
    // A look at a producer's flag costs a round trip to the L2 (some 600 cycles): it is requested EARLY -- a plain load that
    // completes while the waves compute -- and only a flag that was not up yet is polled where the data is needed.
    auto peek_chunk = [&](int c) -> unsigned {          // wave 0: the flag of chunk c's producer(s) now (own chunks: "published")
        if constexpr (RS == 2) {
            const unsigned other = __hip_atomic_load(done_board + (c / SL) * 2 + (rh ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c / SL == nb) return other;
            const unsigned same = __hip_atomic_load(done_board + (c / SL) * 2 + rh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return other < same ? other : same;
        } else {
            if (c / SL == nb) return ~0u;
            return __hip_atomic_load(done_board + c / SL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- fragment read addresses ----
    // B (rows): lane l reads row tile * 16 + (l & 15), k-slot l >> 4 (the 32 channels of the chunk); tap shift is an immediate.
    const int x_lane = (lane >> 4) * kSlotBytes + (g * (RT * 32) + (lane & 15)) * 16;
    // A (filter): fragment = 16-channel tile, lane-linear: in the wave's ring, or (Chunked) in the chunk buffer behind the activations
    const int w_lane = CHUNKED ? kXBufBytes + wn * 2048 + lane * 16 : kWBase + wave * S::RingBytes + lane * 16;
    const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)lds;
    const unsigned w_addr = lds_base + w_lane;

    floatx4 acc[CT2][RT2];                            // defined by the layer's first k-step (mfma_asm_init)
    if constexpr (ABL & 1) {
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][rt][r] = 0.0f;
    }
    // the four 16 x 16 accumulator tiles of 32-row tile R x 32-channel tile C, as one fence statement (asm variant)
    auto fence_block = [&](auto before_c, auto r_c, auto c_c) {
        constexpr bool BEFORE = decltype(before_c)::value;
        constexpr int R = decltype(r_c)::value, C = decltype(c_c)::value;
        mfma_fence<S::acc_in_agpr(R), BEFORE>(acc[2 * C][2 * R], acc[2 * C + 1][2 * R], acc[2 * C][2 * R + 1], acc[2 * C + 1][2 * R + 1]);
    };
    // ---- 16-byte accesses to rows of y / z.  The accumulator tile of 16 channels x 16 rows gives lane group h = lane >> 4 the four
    // channels 4 h .. 4 h + 3 of row lane & 15: half h & 1 of the 16-byte row of k-slot h >> 1.  For the two tiles (X, Y) of a
    // 32-channel tile one v_permlane16_swap per dword (odd lane groups of X <-> even lane groups of Y) turns "8 bytes of two
    // k-slots" into "the whole 16-byte row of ONE k-slot": lane group h then owns k-slot sigma(h) = 0, 2, 1, 3 of the tile's four.
    // The swap is its own inverse, so a 16-byte load of the same rows followed by it delivers the skip input in accumulator order.
    const int h4 = lane >> 4;
    const int slot_of_lane = ((h4 & 1) << 1) | (h4 >> 1);
    const int co_base = (nb * SL + wn * CT) * 32;
    // element offset of (this lane's k-slot of the wave's first 32-channel tile, row 0); k-slot stride = 400 rows * 8
    const long ep_base = (((long)b * (cout / 8) + co_base / 8 + slot_of_lane) * kRows) * 8;
    auto row_of = [&](int rt16) { return g * (RT * 32) + rt16 * 16 + (lane & 15); };
    u32x4 zl[CT][2];                                  // one 32-row tile of z in flight: per 32-channel tile its two 16-row halves
    // OwnerZ (round 4): in the splits where a wave owns only SOME of the board's twelve 32-row tiles (RT < 12) a wave used to request
    // zeros in the slots of tiles it does not own and WAIT for them like for real rows -- a counted vmcnt, which (memory operations
    // complete in order) also waits for every DMA piece issued before it: per-tap stamps of the row split show 500-860 cycles per tap
    // in taps 1..6 of a skip layer's first two chunks against 125-350 without (tools/tap_trace.py; a fifth of a skip layer).  Now only
    // the OWNER of a slot loads, waits and adds (wave-uniform branches); for an output the summation order is what it was (the
    // zeros the other waves used to add changed nothing: x + 0 = x, and the accumulators never hold -0).
#ifndef DGNN_OWNER_Z
#define DGNN_OWNER_Z 1
#endif
    // (The four-slice split counts its filter loads' vmcnt with the skip rows of the previous tap among the younger operations: that
    // count follows the ownership too -- wait_a below.)
    constexpr bool OWNER_Z = DGNN_OWNER_Z != 0 && SKIP == 2 && RT < 12;
    if constexpr (OWNER_Z) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { zl[ct][0] = u32x4{0, 0, 0, 0}; zl[ct][1] = u32x4{0, 0, 0, 0}; }
    }
    const uint16_t* const z_lane = a.z + ep_base + row_of(0) * 8;
    // Row tile `rt` (32 rows) of this wave -- or, for a wave that does not own board row tile T (`mine` false), ZEROS: the same
    // k-slots' pad rows 384..399, which no kernel ever writes.  A load either way, never a branch: at the join of a branch hipcc
    // copies the destination registers -- of an asm load, before the data is there (tools/asm_lint.py).
    const long z_pad_rows = (long)(384 + (lane & 15) - row_of(0)) * 8;          // from z_lane's row to a pad row of the k-slot
    auto load_z_unit = [&](int rt, bool mine) {
        const uint16_t* p = z_lane + (mine ? (long)rt * 32 * 8 : z_pad_rows);
        const int half_stride = mine ? 16 * 8 : 0;
        asm volatile("" : "+v"(p));                   // opaque per use: keeps the 24 addresses from being hoisted out of
                                                      // the K loop (they would be spilled), one 64-bit add per load instead
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int half = 0; half < 2; ++half) {    // rows rt * 32 + half * 16 .., k-slot ct * 4 + sigma(h)
                if constexpr (OWNER_Z) load_row_asm_tied<SPLIT>(zl[ct][half], p + (long)(ct * 4 * kRows) * 8 + half * half_stride);
                else load_row_asm<SPLIT>(zl[ct][half], p + (long)(ct * 4 * kRows) * 8 + half * half_stride);
            }
    };
    const float zscale = SKIP == 2 ? a.alpha2 / a.alpha1 : 0.0f;
    auto add_z_tile = [&](auto rt_c, auto ct_c) {
        constexpr int RTI = decltype(rt_c)::value, CTI = decltype(ct_c)::value;
        if constexpr (ASM) fence_block(std::true_type(), rt_c, ct_c);      // a tile's last MFMA may be the step's last
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(zl[CTI][half][0], zl[CTI][half][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(zl[CTI][half][1], zl[CTI][half][3], false, false);
            const half4 zx = __builtin_bit_cast(half4, make_uint2(s0[0], s1[0]));   // channel tile 2 * CTI
            const half4 zy = __builtin_bit_cast(half4, make_uint2(s0[1], s1[1]));   // channel tile 2 * CTI + 1
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[2 * CTI][2 * RTI + half][jj] = __builtin_fmaf(zscale, (float)zx[jj], acc[2 * CTI][2 * RTI + half][jj]);
                acc[2 * CTI + 1][2 * RTI + half][jj] = __builtin_fmaf(zscale, (float)zy[jj], acc[2 * CTI + 1][2 * RTI + half][jj]);
            }
        }
        if constexpr (ASM) fence_block(std::false_type(), rt_c, ct_c);
    };
    auto add_z_unit = [&](auto rt_c) {
        add_z_tile(rt_c, std::integral_constant<int, 0>());
        if constexpr (CT == 2) add_z_tile(rt_c, std::integral_constant<int, 1>());
    };

    // ---- prologue: chunk 0 and the first three filter slabs ----
    if constexpr (SPLIT) {
        // chunk 0 (and chunk 1, whose pieces go out during chunk 0) must have been published by their producers; the
        // workgroup that produced chunk 0 itself found it in LDS (carry_in) and, unless NS == 1, owns chunk 1 too
        const bool must_wait = sync.epoch > 0 && (((RS == 2 || nb != 0) && !x0_in_flight) || (SL == 1 && nchunks > 1));
        // (channel block 0 without a hand-over reads back its own output: no flag to wait for)
        if (must_wait) {
            if (tid == 0) {
                const unsigned f0 = peek_chunk(0);        // both requests go out before either answer is looked at
                const unsigned f1 = nchunks > 1 ? peek_chunk(1) : ~0u;
                if (f0 < sync.base + (unsigned)sync.epoch) await_chunk(0);
                if (f1 < sync.base + (unsigned)sync.epoch) await_chunk(1);
            }
            __syncthreads();
        }
    }
    // a chained layer without a hand-over fetches chunk 0 -- rows that OTHER waves of this workgroup stored a moment ago (each
    // wave has waited for its own stores only): meet them first
    if ((carry & kChained) != 0 && !carry_in) __syncthreads();
    if (!carry_in || (SPLIT && (RS == 2 || nb != 0) && !x0_in_flight)) {     // (row split: nobody finds chunk 0 in LDS, it lacks the sibling's rows)
#pragma unroll
        for (int j = 0; j < XP; ++j) issue_x(0, buf0, j);
    }
    if constexpr (DIRECT_A) {
    } else if (!carry_in) {
        if constexpr (CHUNKED) {
            for (int q = wave; q < 18 * SL; q += S::WAVES) issue_f(0, buf0, q);
        } else {
#pragma unroll
            for (int s = 0; s < kRing; ++s) issue_w(s, s);
        }
    }
    // af[step parity][32-channel tile][16-channel half]: fragment F of a slab is af[.][F / 2][F % 2].  The caller's: DirectA hands the
    // next layer's step-0 fragments over in af[0] (the last k-step of a layer fetches them instead of the padding behind its filter).
    // DirectA: fragment F (16 channels x the chunk's 32 input channels) of k-step `st` of this wave's two filter tiles -> af[PAR]; steps past the last one read
    // the next layer's filter (or the padding behind the last layer's): never used, they only keep every step alike
    const uint16_t* const w_wave_s = a.w + (long)(nb * SL + wn * CT) * 1024;      // uniform: the wave's first fragment of k-step 0
    // the step behind the last one: the next layer's step 0 when there is a layer to hand over to, else the padding behind the filter
    const uint16_t* const w_past_s = a.w_next ? a.w_next + (long)(nb * SL + wn * CT) * 1024 - (long)nsteps_total * step_halfs : w_wave_s;
    const unsigned lane_bytes = lane * 16;
    auto load_a = [&](auto par_c, int st) {
        constexpr int PAR = decltype(par_c)::value;
        const uint16_t* const src = w_wave_s + (long)st * step_halfs;
        static_for<W>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            load_a_asm<F * 1024>(af[PAR][F / 2][F % 2], src, lane_bytes);
        });
    };
    if constexpr (DIRECT_A) {
        if (!carry_in) load_a(std::integral_constant<int, 0>(), 0);
    }
    wait_vmcnt<0>();
    __syncthreads();
    const unsigned long long t_loop = tracing ? __builtin_amdgcn_s_memtime() : 0;

    // zc_c >= 0: this chunk also folds units 6 * zc_c .. 6 * zc_c + 5 of the skip input into the accumulators (SKIP == 2;
    // chunks 0 and 1 are peeled off the loop so that those VALU updates are unconditional straight-line code -- inside a
    // branch the register allocator copies whole 16-register accumulator tuples and spills)
    //
    // Software pipeline across the k-steps (PIPE): a step never starts with "wait for my slab, read A, read the first B
    // fragments, wait for them" -- that bubble (an LDS round trip per step and wave) is what kept the MFMA pipe at 75 % of a
    // step at NS = 4 and 88 % at NS = 8.  Instead
    //   * the A fragments of step s + 1 are requested at pair AP of step s (behind all DMA instructions of step s, so the
    //     counted vmcnt that guards slab s + 1 is the one the old code used at the start of step s + 1), into the other
    //     half of af[2][2] (roles alternate with the tap's parity; tap 8 -> tap 0 needs one copy per chunk);
    //   * the B-fragment rotation does not stop at the end of a step: the last BD - 1 pairs of step s request the first
    //     BD - 1 pairs of step s + 1 (the next tap's immediate).  Only the chunk boundary (x buffer flips behind a barrier)
    //     starts a fresh rotation.
    // lds_younger() counts, for every MFMA, the LDS reads issued after the ones it needs (LDS returns in order).
    // Chunked kernels: the A fragments come out of the chunk buffer (tap-major), so tap 8 cannot request its successor's
    // (other buffer, published by the chunk barrier): tap 0 requests its own in front of its first B fragments.
    half8 bf[BD];
    auto read_a = [&](auto par_c, auto off_c, unsigned addr) {       // the W fragments (tile, k16) of the slab at ring offset OFF
        constexpr int PAR = decltype(par_c)::value, OFF = decltype(off_c)::value;
        static_for<W>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            af[PAR][F / 2][F % 2] = lds_read<OFF + F * 1024>(addr);
        });
    };
    auto wait_first = [&](auto younger_c, auto par_c, half8& b0) {    // the wait in front of a step's first MFMA: B0 and all of A
        constexpr int Y = decltype(younger_c)::value, PAR = decltype(par_c)::value;
        if constexpr (CT == 1) wait_lgkm<Y>(b0, af[PAR][0][0], af[PAR][0][1]);
        else wait_lgkm<Y>(b0, af[PAR][0][0], af[PAR][0][1], af[PAR][1][0], af[PAR][1][1]);
    };
    // step 0 of the layer (slab 0 landed: the prologue waited for everything)
    if constexpr (PIPE && !CHUNKED && !DIRECT_A) read_a(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), w_addr);
    auto run_chunk = [&](int chunk, auto zc_c, auto first_c) {
        constexpr int ZC = decltype(zc_c)::value;
        constexpr bool FIRST_CHUNK = decltype(first_c)::value;      // chunk 0 of the layer: its tap 0 defines the accumulators
        const int buf = (chunk + buf0) & 1;
        const int next_chunk = chunk + 1 < nchunks ? chunk + 1 : 0;    // the last chunk re-fetches chunk 0 (unused):
                                                                       // keeps the DMA count per step constant
        // SPLIT: the pieces of chunk + 2 go out during chunk + 1: its producer must have published before the barrier at the
        // end of this chunk opens.  The flag is requested now and looked at there.
        unsigned flag_ahead = ~0u;
        const bool needs_flag = SPLIT && sync.epoch > 0 && chunk + 2 < nchunks && (chunk + 2) % SL == 0;
        if constexpr (SPLIT && CHUNKED) {     // (NS = 4: measured 1..3 % slower with the early look -- the extra load sits among wave 0's counted waits)
            if (needs_flag && wave == 0) flag_ahead = peek_chunk(chunk + 2);
        } else if constexpr (SPLIT) {
            flag_ahead = 0;
        }
        const unsigned x_addr = lds_base + buf * BUF + x_lane;
        const unsigned w_addr_c = CHUNKED ? w_addr + buf * BUF : w_addr;
        const int st0 = chunk * 9;

        auto step = [&](auto tap_c) {
            constexpr int TAP = decltype(tap_c)::value;
            constexpr int SHIFT = (TAP / 3 - 1) * 20 + (TAP % 3 - 1);
            constexpr int SHIFT_NEXT = ((TAP + 1) % 9 / 3 - 1) * 20 + ((TAP + 1) % 9 % 3 - 1);
            // DMA instructions issued after slab st: slab st is issued during step st - 3, the activation piece of a step
            // (taps 0..XP-1 only) after its filter pieces:
            //     2 * W + |{TAP-3, TAP-2, TAP-1} mod 9  intersected with  {0..XP-1}|
            constexpr int N = 2 * WI + ((TAP + 6) % 9 < XP) + ((TAP + 7) % 9 < XP) + ((TAP + 8) % 9 < XP);
            // ... after slab st + 1, seen from pair AP of step st (all of this step's DMA instructions come before AP)
            constexpr int N_NEXT = 2 * WI + ((TAP + 7) % 9 < XP) + ((TAP + 8) % 9 < XP) + (TAP < XP);
            constexpr int WOFF = CHUNKED ? TAP * SL * 2048 : (TAP % kRing) * S::SlabBytes;
            constexpr int WOFF_NEXT = CHUNKED ? (TAP + 1) % 9 * SL * 2048 : ((TAP + 1) % kRing) * S::SlabBytes;
            constexpr int XOFF = (kHalo + SHIFT) * 16;
            constexpr int XOFF_NEXT = (kHalo + SHIFT_NEXT) * 16;
            constexpr bool FIRST = TAP == 0, LAST = TAP == 8;
            constexpr int P = PIPE ? (TAP & 1) : 0;
            constexpr int PH = PIPE ? (TAP * NP) % BD : 0;        // where the rotation stands at the start of this step
#ifdef DGNN_TAP_STAMPS
            if (chunk == 1) tap_stamp[TAP] = __builtin_amdgcn_s_memtime();
#endif

            if constexpr (!PIPE) {
                if constexpr (!(ABL & 8)) wait_vmcnt<N>();
                read_a(std::integral_constant<int, 0>(), std::integral_constant<int, WOFF>(), w_addr);
            }
            if constexpr (S::StageF && FIRST) {
                __builtin_amdgcn_sched_barrier(0);
                stage_f_load(chunk + 1);          // (chunks past the last one: the next layer's first chunk, or padding nobody reads)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CHUNKED && FIRST) read_a(std::integral_constant<int, 0>(), std::integral_constant<int, WOFF>(), w_addr_c);
            if constexpr (!PIPE || FIRST) {
                static_for<(PIPE ? BD : BD - 1)>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    bf[I] = lds_read<XOFF + S::b_offset(I)>(x_addr);
                });
            }
            static_for<NP>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                if constexpr (PIPE) {
                    constexpr int YOUNGER = lds_younger_mfma_first<NP, BD, S::a_pair(), CHUNKED, (DIRECT_A ? 0 : W)>(I, FIRST, LAST);
                    if constexpr (I == 0 && DIRECT_A) {
                        // this step's A fragments were requested during the previous step (pairs w_pair(0..3)); younger than them are
                        // that step's activation piece and the skip rows it asked for at its end
                        constexpr int PT = (TAP + 8) % 9;                         // the previous tap
                        constexpr bool Z_PREV = ZC >= 0 && TAP >= 1 && PT <= 5 && ZC * 6 + PT < 12;
                        constexpr int N_A = FIRST ? 0 : (PT < XP ? 1 : 0) + (Z_PREV ? 2 * CT : 0);
                        if constexpr (OWNER_Z && Z_PREV) {
                            // only the owner of the previous tap's slot has skip rows among its younger operations (wave-uniform)
                            if ((ZC * 6 + PT) / RT == g) wait_a<N_A>();
                            else wait_a<N_A - 2 * CT>();
                        } else {
                            wait_a<N_A>();
                        }
                        wait_lgkm<YOUNGER>(bf[PH % BD]);
                    }
                    else if constexpr (I == 0) wait_first(std::integral_constant<int, YOUNGER>(), std::integral_constant<int, P>(), bf[PH % BD]);
                    else wait_lgkm<YOUNGER>(bf[(PH + I) % BD]);
                } else {
                    constexpr bool MORE = I + BD - 1 < NP && !(ABL & 4);
                    if constexpr (MORE) bf[(I + BD - 1) % BD] = lds_read<XOFF + S::b_offset(I + BD - 1)>(x_addr);
                    constexpr int YOUNGER = (ABL & 4) ? 0 : I + BD - 1 < NP ? BD - 1 : NP - 1 - I;
                    if constexpr (I == 0) wait_first(std::integral_constant<int, YOUNGER>(), std::integral_constant<int, 0>(), bf[0]);
                    else wait_lgkm<YOUNGER>(bf[I % BD]);
                }
                // pair I = 16-row tile I: its B fragment (16 rows x 32 channels) meets the wave's CT2 filter fragments (16 channels x 32).
                // A four-pass MFMA leaves two or three issue slots in its shadow: the pair's memory instructions go out between
                // its two halves, the LDS read that refills the B register behind its last MFMA.
                auto mfmas = [&](auto from_c, auto to_c) {
                    static_for<decltype(to_c)::value - decltype(from_c)::value>([&](auto cc) {
                        constexpr int C2 = decltype(from_c)::value + decltype(cc)::value;
                        if constexpr (ABL & 1) {
                            const half8& fa = af[P][C2 / 2][C2 % 2];     // (named first: an asm operand alone does not make a generic
                            const half8& fb = bf[(PH + I) % BD];         // lambda capture the array)
                            asm volatile("" ::"v"(fa), "v"(fb));
                        } else if constexpr (FIRST_CHUNK && TAP == 0) {
                            mfma_asm_init<S::acc_in_agpr(I / 2)>(acc[C2][I], af[P][C2 / 2][C2 % 2], bf[(PH + I) % BD]);
                        } else {
                            mfma_asm<S::acc_in_agpr(I / 2)>(acc[C2][I], af[P][C2 / 2][C2 % 2], bf[(PH + I) % BD]);     // volatile asm: stays in program order
                        }
                    });
                };
                mfmas(std::integral_constant<int, 0>(), std::integral_constant<int, CT2 / 2>());
                if constexpr (CHUNKED) {
                    // this wave's share of the NEXT chunk's activations and filter rows -> the other buffer
                    static_for<S::PPW>([&](auto jc) {
                        constexpr int J = decltype(jc)::value;
                        if constexpr (S::piece_tap(J) == TAP && S::piece_pair(J) == I) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(next_chunk, chunk + 1, buf ^ 1, J);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                } else if constexpr (!(ABL & 2)) {
                    // The A fragments of this step are in registers since the first wait (I == 0): the slab may be
                    // refilled from now on.  The step's DMA instructions are spread over its MFMAs so that the wave has
                    // no separate, MFMA-free "issue phase".
                    static_for<WI>([&](auto jc) {
                        constexpr int J = decltype(jc)::value;
                        if constexpr (I == S::w_pair(J % W) + J / W) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (DIRECT_A) {
                                // fragment J of the NEXT step into the other half of af: its last readers, the MFMAs of the
                                // previous step, were issued a k-step ago
                                const uint16_t* const base = (TAP == 8 && st0 + 9 >= nsteps_total) ? w_past_s : w_wave_s;
                                load_a_asm<J * 1024>(af[P ^ 1][J / 2][J % 2], base + (long)(st0 + TAP + 1) * step_halfs, lane_bytes);
                            } else {
                                issue_w_piece(st0 + TAP + kRing, TAP % kRing, J % W);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    if constexpr (I == S::x_pair() && TAP < XP) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_x(next_chunk, buf ^ 1, TAP);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                mfmas(std::integral_constant<int, CT2 / 2>(), std::integral_constant<int, CT2>());
                if constexpr (PIPE) {
                    if constexpr (I == S::a_pair() && CHUNKED) {
                        if constexpr (!LAST) read_a(std::integral_constant<int, P ^ 1>(), std::integral_constant<int, WOFF_NEXT>(), w_addr_c);
                    } else if constexpr (I == S::a_pair() && !DIRECT_A) {
                        if constexpr (!(ABL & 8)) wait_vmcnt<N_NEXT>();
                        read_a(std::integral_constant<int, P ^ 1>(), std::integral_constant<int, WOFF_NEXT>(), w_addr);
                    }
                    constexpr int T = I + BD;
                    if constexpr (T < NP) bf[(PH + T) % BD] = lds_read<XOFF + S::b_offset(T)>(x_addr);
                    else if constexpr (!LAST) bf[(PH + T) % BD] = lds_read<XOFF_NEXT + S::b_offset(T - NP)>(x_addr);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            // skip input folded into the accumulators: row tile 6 * ZC + t is requested at the end of tap t of chunk ZC and
            // added at the end of tap t + 1
            // WHEN a row tile's skip is added is part of the summation order, so it is fixed per BOARD row tile T = 0..11
            // (chunk T / 6, end of tap T % 6 + 1) whatever the work split: the wave that owns T (T / RT == g) fetches the
            // rows, every other wave adds zeros in that slot (x + 0 = x exactly; the accumulators never hold -0).
            if constexpr (ZC >= 0 && TAP >= 1 && TAP <= 6 && ZC * 6 + TAP - 1 < 12) {
                // the z rows requested at the end of the previous step are this wave's youngest memory operations but for
                // the DMA instructions this step has issued
                constexpr int YOUNGER_VM = CHUNKED ? S::pieces_in_tap(TAP) : WI + (TAP < XP);
                if constexpr (OWNER_Z) {
                    // wave-uniform branch: only the owner of this slot has rows in flight, and only the owner adds.  (Every other wave
                    // used to add zeros here -- exact, and 350 cycles per slot with its two fences: eleven of a wave's twelve slots in
                    // the row split.)  Inside the branch: the counted wait, the fences around the tile, VALU code; the accumulators
                    // are tied asm operands on both sides of the join.
                    if ((ZC * 6 + TAP - 1) / RT == g) {
                        if constexpr (CT == 1) wait_z<YOUNGER_VM>(zl[0][0], zl[0][1]);
                        else wait_z<YOUNGER_VM>(zl[0][0], zl[0][1], zl[CT - 1][0], zl[CT - 1][1]);
                        __builtin_amdgcn_sched_barrier(0);
                        add_z_unit(std::integral_constant<int, (ZC * 6 + TAP - 1) % RT>());
                    }
                } else {
                    if constexpr (CT == 1) wait_z<YOUNGER_VM>(zl[0][0], zl[0][1]);
                    else wait_z<YOUNGER_VM>(zl[0][0], zl[0][1], zl[CT - 1][0], zl[CT - 1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                    add_z_unit(std::integral_constant<int, (ZC * 6 + TAP - 1) % RT>());
                }
            }
            // extra memory operations only make the counted waits above more conservative (they require that many
            // more of the OLDER operations to have completed), never unsafe
            if constexpr (ZC >= 0 && TAP <= 5 && ZC * 6 + TAP < 12) {
                constexpr int T = ZC * 6 + TAP;
                if constexpr (OWNER_Z) { if (T / RT == g) load_z_unit(T % RT, true); }
                else load_z_unit(T % RT, RT == 12 || T / RT == g);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>());
        step(std::integral_constant<int, 1>());
        step(std::integral_constant<int, 2>());
        step(std::integral_constant<int, 3>());
        step(std::integral_constant<int, 4>());
        step(std::integral_constant<int, 5>());
        step(std::integral_constant<int, 6>());
        step(std::integral_constant<int, 7>());
        step(std::integral_constant<int, 8>());
#ifdef DGNN_TAP_STAMPS
        if (chunk == 1) tap_stamp[9] = __builtin_amdgcn_s_memtime();
#endif

        // chunk boundary: this wave's pieces of the next chunk landed at tap 8's wait (issued at taps 0..3, at
        // least 5 slabs ago); its reads of this chunk retired with the MFMAs above.  With seven pieces per wave (taps 0..6)
        // the last one is younger than that wait covers: only the filter pieces of the steps after it may stay in flight.
        if constexpr (CHUNKED) {
            // every piece of the next chunk this wave asked for has landed (tap 8 left no LDS read in flight); the barrier
            // below publishes all of them
            if constexpr (S::StageF) {
                if constexpr (S::FPW == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fstage[0]), "+v"(fstage[1]), "+v"(fstage[2]) :: "memory");
                else if constexpr (S::FPW == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fstage[0]), "+v"(fstage[1]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fstage[0]) :: "memory");
                stage_f_store(buf ^ 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the writes are in the LDS before this wave reaches the barrier
            } else {
                wait_vmcnt<0>();
            }
        } else if constexpr (XP > 4) {
            wait_vmcnt<WI * (8 - (XP - 1))>();
        }
        if constexpr (CHUNKED) {
        } else if constexpr (DIRECT_A) {
            // tap 8 asked for the next step's A fragments into af[1]; tap 0 of the next chunk looks in af[0].  They were requested
            // most of a k-step ago (and the barrier below waits for the slowest wave anyway)
            if constexpr (CT == 1) wait_vm_for<0>(af[1][0][0], af[1][0][1]);
            else wait_vm_for<0>(af[1][0][0], af[1][0][1], af[1][1][0], af[1][1][1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                af[0][ct][0] = af[1][ct][0];
                af[0][ct][1] = af[1][ct][1];
            }
        } else if constexpr (PIPE) {
            // the only LDS reads still out are the A fragments of the next step (tap 8 requests no B read-ahead); tap 8
            // left them in af[1], tap 0 of the next chunk looks in af[0]
            if constexpr (CT == 1) wait_lgkm_for<0>(af[1][0][0], af[1][0][1]);
            else wait_lgkm_for<0>(af[1][0][0], af[1][0][1], af[1][1][0], af[1][1][1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                af[0][ct][0] = af[1][ct][0];
                af[0][ct][1] = af[1][ct][1];
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (SPLIT) {
            if (needs_flag && tid == 0 && flag_ahead < sync.base + (unsigned)sync.epoch) await_chunk(chunk + 2);
        }
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef DGNN_TAP_STAMPS
        if (chunk == 1) tap_stamp[10] = __builtin_amdgcn_s_memtime();
#endif
    };
    int first_loop_chunk = 1;
    if constexpr (SKIP == 2) {          // the launcher guarantees nchunks >= 2
        run_chunk(0, std::integral_constant<int, 0>(), std::true_type());
        run_chunk(1, std::integral_constant<int, 1>(), std::false_type());
        first_loop_chunk = 2;
    } else {
        run_chunk(0, std::integral_constant<int, -1>(), std::true_type());
    }
    for (int chunk = first_loop_chunk; chunk < nchunks; ++chunk) run_chunk(chunk, std::integral_constant<int, -1>(), std::false_type());

    const unsigned long long t_epilogue = tracing ? __builtin_amdgcn_s_memtime() : 0;

    // SPLIT, a workgroup that does not produce channels 0..31 itself: if the one that does has ALREADY published this layer
    // (it runs ahead: its first NS input chunks are its own), the next layer's first chunk is requested at the END of the
    // epilogue and lands while the stores drain and the next prologue runs.  One look, no waiting, and the look itself (a round
    // trip to the L2) is requested here and read behind the stores.  Activation buffer 0 is free: nobody reads activations
    // after the barrier that ended the last chunk, and every wave's DMA into buffer 0 had landed before it.
    bool x0_next = false;
    unsigned flag_x0 = 0;
    const bool may_prefetch_x0 = SPLIT && nb != 0 && hand_over;
    auto prefetch_x0 = [&] {
        unsigned* const word = reinterpret_cast<unsigned*>(lds + S::LdsBytes);
        if (tid == 0) *word = flag_x0 >= sync.base + (unsigned)sync.epoch + 1u ? 1u : 0u;
        __syncthreads();
        x0_next = *word != 0;
        if (x0_next) {
            const uint16_t* const y_board = a.y + (long)b * cout * kRows + lane * 8;
#pragma unroll
            for (int j = 0; j < XP; ++j) issue_x_from(y_board, 0, buf_next, j);
        }
    };
    if constexpr (SPLIT) {
        if (may_prefetch_x0 && wave == 0) {
            flag_x0 = __hip_atomic_load(done_board, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (RS == 2) {          // both row halves of channel block 0
                const unsigned other = __hip_atomic_load(done_board + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag_x0 = other < flag_x0 ? other : flag_x0;
            }
        }
        if constexpr (!CHUNKED) {             // NS = 4: look and request now, in front of the epilogue (the longer epilogue covers it)
            if (may_prefetch_x0) prefetch_x0();
        }
    }

    // ---- epilogue ----
    // D[i][j] of a 16 x 16 tile: j = lane & 15 = row within the tile, i = 4 * (lane >> 4) + r = channel within the tile: the lane owns
    // 4 consecutive channels = 8 bytes of a k-slot row; v_permlane16_swap over the two tiles of a 32-channel tile makes that the
    // whole 16-byte row of k-slot sigma(lane >> 4) (see the skip input above; T21 of the CDNA guide): each store covers four
    // contiguous 256-byte runs.
    float bias[S::CT2][4];
#pragma unroll
    for (int c2 = 0; c2 < S::CT2; ++c2) {
        const half4 b4 = *reinterpret_cast<const __attribute__((address_space(1))) half4*>((__UINTPTR_TYPE__)(a.bias + co_base + c2 * 16 + h4 * 4));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bias[c2][jj] = (float)b4[jj];
    }

    const uint16_t* const zb = a.z + ep_base;
    uint16_t* const yb = a.y + ep_base;

    uint4 zq[2][CT];                                  // skip rows of two 16-row tiles in flight: [tile parity][32-channel tile]
    auto load_z = [&](int rt16) {
        if constexpr (SKIP == 1) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) zq[rt16 & 1][ct] = load_row<SPLIT>(zb + ((long)(ct * 4) * kRows + row_of(rt16)) * 8);
        }
    };
    load_z(0);
    load_z(1);
    if constexpr (ASM) {
        // the K loop's last MFMAs are a few instructions back: a fence per block of accumulator tiles before anything reads it
        static_for<RT>([&](auto rc) {
            fence_block(std::true_type(), rc, std::integral_constant<int, 0>());
            if constexpr (CT == 2) fence_block(std::true_type(), rc, std::integral_constant<int, CT - 1>());
        });
    }
#pragma unroll
    for (int rt = 0; rt < S::RT2; ++rt) {
        const int row = row_of(rt);
        const unsigned live = ((row % 20 != 19) && row < 380) ? 0xffffffffu : 0u;   // pad column / pad row -> zero
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            // skip values of the two channel tiles (4 channels each) after the exchange
            uint2 zx = make_uint2(0, 0), zy = make_uint2(0, 0);
            if constexpr (SKIP == 1) {
                const uint4 v = zq[rt & 1][ct];
                const auto s0 = __builtin_amdgcn_permlane16_swap(v.x, v.z, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(v.y, v.w, false, false);
                zx = make_uint2(s0[0], s1[0]);
                zy = make_uint2(s0[1], s1[1]);
            }
            uint2 out[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c2 = 2 * ct + kk;
                const half4 z4 = __builtin_bit_cast(half4, kk == 0 ? zx : zy);
                float t[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj += 2) {
                    // two fp32 fmas per instruction (v_pk_fma_f32: same rounding per element; no MFMA is running beside it here)
                    const floatx2 av = {acc[c2][rt][jj], acc[c2][rt][jj + 1]}, bv = {bias[c2][jj], bias[c2][jj + 1]}, sv = {a.alpha1, a.alpha1};
                    const floatx2 tv = __builtin_elementwise_fma(sv, av, bv);
                    t[jj] = tv[0];
                    t[jj + 1] = tv[1];
                }
                if constexpr (SKIP == 1) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) t[jj] = __builtin_fmaf(a.alpha2, (float)z4[jj], t[jj]);
                }
                // relu behind the conversion: the same bits as max(t, 0) in fp32 first (rounding is monotone, 0 is exact, a NaN gives 0 either way)
                unsigned y01 = cvt_pk_f16(t[0], t[1]), y23 = cvt_pk_f16(t[2], t[3]);
                if constexpr (RELU) { y01 = pk_relu_f16(y01); y23 = pk_relu_f16(y23); }
                out[kk] = make_uint2(y01 & live, y23 & live);
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(out[0].x, out[1].x, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(out[0].y, out[1].y, false, false);
            const uint4 unit = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            store_row<SPLIT>(yb + ((long)(ct * 4) * kRows + row) * 8, unit);
            // hand-over: channels 0..31 are the next layer's first input chunk; its place is activation buffer 0, which
            // nobody reads any more (the last chunk sits in buffer 1, and the barrier that ended it came after every
            // wave's last DMA into buffer 0 had landed)
            if (RS == 1 && ct == 0 && wn == 0 && nb == 0 && hand_over)
                *reinterpret_cast<uint4*>(lds + buf_next * BUF + slot_of_lane * kSlotBytes + (kHalo + row) * 16) = unit;
        }
        if (rt + 2 < S::RT2) load_z(rt + 2);
    }
    if (a.timing) {
        wait_vmcnt<0>();
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0 && wave < 8) {
            unsigned long long* out = a.timing + ((long)(nb * gridDim.x + b) * 8 + wave) * 4;
            out[0] = t_start; out[1] = t_loop; out[2] = t_epilogue; out[3] = t_end;
        }
    }
    if constexpr (SPLIT && CHUNKED) {
        if (may_prefetch_x0) prefetch_x0();
    }
    // the tail's surplus DMA (slabs past the last step) must not outlive the workgroup's LDS; pieces requested just now are
    // the next prologue's to wait for (they are this wave's youngest XP operations)
    if (CHUNKED && x0_next) wait_vmcnt<XP>();
    else wait_vmcnt<0>();
    if constexpr (SPLIT) {
        // every wave's write-through stores have left (vmcnt counts stores): one lane says so to the rest of the board
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_board + nb * RS + rh, sync.base + (unsigned)sync.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sync.trace != nullptr && tid == 0) {
            unsigned long long* out = sync.trace + (((long)sync.epoch * a.batch + b) * sync.groups + nb * RS + rh) * 4;
            out[0] = t_start; out[1] = t_loop; out[2] = t_epilogue | (x0_next ? 1ull << 63 : 0ull); out[3] = __builtin_amdgcn_s_memtime();
        }
#ifdef DGNN_TAP_STAMPS
        if (sync.trace != nullptr && lane == 0 && wave < 12) {      // behind the regular records: [layer][board][group][12 waves][12]
            unsigned long long* out = sync.trace + (long)gridDim.x * 0 + (long)a.batch * sync.groups * 4 * DGNN_TAP_STAMPS +
                                      ((((long)sync.epoch * a.batch + b) * sync.groups + nb * RS + rh) * 12 + wave) * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) out[k] = tap_stamp[k];
        }
#endif
    }
    return (x0_next ? kX0InFlight : 0) | (hand_over ? kHandedOver : 0) | (buf_next ? kStartBuffer : 0);
}


}  // namespace
}  // namespace dgnn
