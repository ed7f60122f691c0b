/*
 * dgnn_oracle.c -- CPU restatement of dream-go's neural-network leaf evaluation.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it, and only as the
 * checker / the reported CPU baseline.  The product path (libdgnn_hip.so) never
 * links, loads or calls anything in this directory.
 *
 * What it restates (paths relative to the reference checkout, kblomdahl/dream-go):
 *   src/libdg_nn/graph.rs:123-158                      forward() dataflow
 *   src/libdg_nn/layers/conv2d.rs:99-117,171-220       3x3 cross-correlation, pad 1, NHWC x, KRSC w
 *   src/libdg_cuda/cudnn/convolution_bias_activation.rs:21-41
 *                                                      y = act(a1*conv(x,w) + a2*z + bias)
 *   src/libdg_nn/layers/residual_block.rs:40-78        gated skip, bias pre-scaled by alpha
 *   src/libdg_nn/layers/policy_head.rs:43-103          conv -> dense*tau -> softmax
 *   src/libdg_nn/layers/value_head.rs:40-84            conv -> dense -> tanh
 *   src/libdg_nn/layers/dense.rs:79-96,108-135         1x1 "convolution" GEMM, stored [in][out]
 *   src/libdg_cuda/cudnn/scale_tensor.rs:43-60         in-place fp16 scale
 *   src/libdg_cuda/cudnn/softmax.rs:58-79              ACCURATE (max-subtracted), per instance
 *   src/libdg_cuda/cudnn/activation.rs:60-82           tanh
 *
 * The arithmetic of the reference lives in closed-source cuDNN v8, which is not
 * in the reference tree and cannot run here.  Numerics contract of this
 * restatement: every tensor the reference materialises is fp16 (layers/common.rs:61-98),
 * so every layer output is rounded to fp16 (round-to-nearest-even); inside a layer all
 * arithmetic is fp32 (cuDNN "Float" compute type, which the reference forces for the
 * heads and dense layers and which is a superset of the unobservable "true half" tower
 * mode, conv2d.rs:54).
 *
 * PINNING: the three layer-level known-answer tests of the reference
 * (conv2d.rs:252-291, dense.rs:243-275, dense.rs:277-299) are reproduced bit-for-bit
 * by tests/test_oracle_kat.py.  Whole-network policy/value: PARITY UNPINNED -- the
 * reference holds no golden vector for a whole forward pass and cannot be executed here.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -fopenmp; no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DGO_OK 0
#define DGO_BAD_ARGUMENT -4
#define DGO_OUT_OF_MEMORY -5

/* ---------------------------------------------------------------------------
 * fp16 <-> fp32, IEEE 754-2008 binary16, round-to-nearest-even.
 * Restates src/libdg_utils/types/fp16.rs:16-70 (llvm.convert.{to,from}.fp16).
 * ------------------------------------------------------------------------- */
float dgo_f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;

    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }

    float out;
    memcpy(&out, &bits, sizeof(out));
    return out;
}

uint16_t dgo_f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, sizeof(x));

    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;

    if (absx >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (absx > 0x7f800000u ? 0x200u | ((absx >> 13) & 0x3ffu) : 0));
    }
    if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (or exactly 2^-25, ties-to-even) rounds to zero */
        return (uint16_t)sign;
    }

    int32_t e = (int32_t)(absx >> 23) - 127; /* unbiased exponent */
    uint32_t m = (absx & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    uint32_t shift, half_bits;

    if (e < -14) { /* result is subnormal */
        shift = (uint32_t)(13 + (-14 - e));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t)(e + 15) << 10;
        m &= 0x7fffffu;
    }

    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);

    if (rem > halfway || (rem == halfway && (q & 1u))) {
        q += 1; /* may carry into the exponent, which is the correct result */
    }

    return (uint16_t)(sign | (half_bits + q));
}

void dgo_f16_to_f32_array(const uint16_t* src, float* dst, size_t n)
{
#pragma omp parallel for schedule(static) if (n > 65536)
    for (long i = 0; i < (long)n; ++i) dst[i] = dgo_f16_to_f32(src[i]);
}

void dgo_f32_to_f16_array(const float* src, uint16_t* dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) dst[i] = dgo_f32_to_f16(src[i]);
}

/* The fp16 value nearest to a DOUBLE (ties to even), returned as a float that converts exactly: the "exact arithmetic, same
 * rounding points" variant of the network (dgo_forward_ex, accumulate_f64) rounds every materialised tensor ONCE -- going through
 * float first would round twice. */
static float dgo_round_f64_to_f16_value(double x)
{
    if (!(x == x) || x == 0.0 || isinf(x)) return (float)x;
    int e;
    (void)frexp(fabs(x), &e);                    /* |x| = m * 2^e, m in [0.5, 1): floor(log2 |x|) = e - 1 */
    int exponent = e - 1;
    if (exponent < -14) exponent = -14;          /* subnormal range: fixed spacing 2^-24 */
    const double ulp = ldexp(1.0, exponent - 10);
    return (float)(nearbyint(x / ulp) * ulp);    /* power-of-two scaling is exact; nearbyint rounds ties to even */
}

/* ---------------------------------------------------------------------------
 * cudnnScaleTensor on an fp16 tensor: b <- fp16(alpha * b).
 * src/libdg_cuda/cudnn/scale_tensor.rs:43-60; used by residual_block.rs:72-74
 * (alpha * b2) and policy_head.rs:87-89 (tau * b_l).
 * ------------------------------------------------------------------------- */
void dgo_scale(uint16_t* b, size_t n, float alpha)
{
    for (size_t i = 0; i < n; ++i) b[i] = dgo_f32_to_f16(alpha * dgo_f16_to_f32(b[i]));
}

/* ---------------------------------------------------------------------------
 * cudnnTransformTensor as used by Dense::prepare (dense.rs:79-96,182-189):
 * the stored matrix is [in][out]; the filter cuDNN consumes is [out][in].
 * ------------------------------------------------------------------------- */
void dgo_transpose_in_out(const uint16_t* w_in_out, uint16_t* w_out_in, int num_in, int num_out)
{
    for (int i = 0; i < num_in; ++i)
        for (int o = 0; o < num_out; ++o)
            w_out_in[(size_t)o * num_in + i] = w_in_out[(size_t)i * num_out + o];
}

/* ---------------------------------------------------------------------------
 * cudnnConvolutionBiasActivationForward, 3x3 / pad 1 / stride 1 / cross-correlation.
 *
 *   y = act( alpha1 * (x * w) + alpha2 * z + bias )
 *
 * x, y, z : NHWC fp16  [batch][size][size][c]           layers/common.rs:61-67
 * w       : KRSC fp16  [cout][3][3][cin]                conv2d.rs:111-117
 * bias    : fp16 [cout]
 * z is NOT read when alpha2 == 0 (the reference aliases it to the uninitialised
 * output, conv2d.rs:188-190).
 * relu != 0 -> ReLU, else identity.
 * ------------------------------------------------------------------------- */
#define DGO_CO_BLOCK 64

/* The same layer with EXACT products and sums (double: a product of two fp16 values is exact, a sum of a few thousand of them
 * loses nothing that matters) and the same single rounding to fp16 per output: what both fp32 implementations -- this file's and
 * the HIP kernels' -- approximate.  Used only as the yardstick of tests/test_round4_gpu.py ("is the HIP path as near to the exact
 * result as the fp32 oracle is?"). */
static int dgo_conv3x3_f64(const uint16_t* x, int batch, int size, int cin,
                           const uint16_t* w, const uint16_t* bias, int cout,
                           float alpha1, float alpha2, const uint16_t* z, int relu,
                           uint16_t* y)
{
    const size_t num_pos = (size_t)batch * size * size;
    const int cout_pad = (cout + DGO_CO_BLOCK - 1) / DGO_CO_BLOCK * DGO_CO_BLOCK;
    float* xf = (float*)malloc(sizeof(float) * (num_pos * cin + 1));
    float* wf = (float*)calloc((size_t)9 * cin * cout_pad, sizeof(float));
    if (!xf || !wf) { free(xf); free(wf); return DGO_OUT_OF_MEMORY; }
    dgo_f16_to_f32_array(x, xf, num_pos * cin);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co)
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < cin; ++ci)
                wf[((size_t)tap * cin + ci) * cout_pad + co] = dgo_f16_to_f32(w[((size_t)co * 9 + tap) * cin + ci]);
    const long num_blocks = (cout + DGO_CO_BLOCK - 1) / DGO_CO_BLOCK;
    const long num_units = (long)batch * num_blocks;
#pragma omp parallel for schedule(dynamic, 1)
    for (long unit = 0; unit < num_units; ++unit) {
        const int b = (int)(unit / num_blocks);
        const int co0 = (int)(unit % num_blocks) * DGO_CO_BLOCK;
        const int co_end = (co0 + DGO_CO_BLOCK < cout) ? DGO_CO_BLOCK : cout - co0;
        for (int yy = 0; yy < size; ++yy) {
            for (int xx = 0; xx < size; ++xx) {
                const size_t out_pos = ((size_t)b * size + yy) * size + xx;
                double acc[DGO_CO_BLOCK];
                for (int j = 0; j < DGO_CO_BLOCK; ++j) acc[j] = 0.0;
                for (int tap = 0; tap < 9; ++tap) {
                    const int sy = yy + tap / 3 - 1;
                    const int sx = xx + tap % 3 - 1;
                    if (sy < 0 || sy >= size || sx < 0 || sx >= size) continue;
                    const float* xin = xf + (((size_t)b * size + sy) * size + sx) * cin;
                    const float* wt = wf + (size_t)tap * cin * cout_pad + co0;
                    for (int ci = 0; ci < cin; ++ci) {
                        const double v = (double)xin[ci];
                        const float* wr = wt + (size_t)ci * cout_pad;
                        for (int j = 0; j < DGO_CO_BLOCK; ++j) acc[j] += v * (double)wr[j];
                    }
                }
                for (int j = 0; j < co_end; ++j) {
                    const int co = co0 + j;
                    double t = (double)alpha1 * acc[j] + (double)dgo_f16_to_f32(bias[co]);
                    if (alpha2 != 0.0f) t += (double)alpha2 * (double)dgo_f16_to_f32(z[out_pos * cout + co]);
                    if (relu && !(t > 0.0)) t = 0.0;
                    y[out_pos * cout + co] = dgo_f32_to_f16(dgo_round_f64_to_f16_value(t));
                }
            }
        }
    }
    free(xf); free(wf);
    return DGO_OK;
}

static int dgo_dense_f64(const uint16_t* x, int batch, int num_in, const uint16_t* w_out_in, const uint16_t* bias, int num_out,
                         float alpha1, int relu, uint16_t* y)
{
#pragma omp parallel for schedule(static)
    for (int b = 0; b < batch; ++b) {
        const uint16_t* xb = x + (size_t)b * num_in;
        for (int o = 0; o < num_out; ++o) {
            const uint16_t* wr = w_out_in + (size_t)o * num_in;
            double acc = 0.0;
            for (int i = 0; i < num_in; ++i) acc += (double)dgo_f16_to_f32(xb[i]) * (double)dgo_f16_to_f32(wr[i]);
            double t = (double)alpha1 * acc + (double)dgo_f16_to_f32(bias[o]);
            if (relu && !(t > 0.0)) t = 0.0;
            y[(size_t)b * num_out + o] = dgo_f32_to_f16(dgo_round_f64_to_f16_value(t));
        }
    }
    return DGO_OK;
}

/* The fp32 convolution as PLAIN loops: the readable statement (the form rounds 1-5 ran).  For every output: k runs tap-major then
 * channel, off-board taps are skipped, one fused multiply-add per product, in order, starting from +0; then
 * fp16(act(alpha1 * acc + bias [+ alpha2 * z])).  dgo_conv3x3 below is the same arithmetic made fast;
 * tests/test_oracle_kat.py::test_fast_convolution_is_the_plain_one holds the two bit for bit against each other. */
int dgo_conv3x3_plain(const uint16_t* x, int batch, int size, int cin,
                      const uint16_t* w, const uint16_t* bias, int cout,
                      float alpha1, float alpha2, const uint16_t* z, int relu,
                      uint16_t* y)
{
    if (batch < 0 || size <= 0 || cin <= 0 || cout <= 0) return DGO_BAD_ARGUMENT;
    if (alpha2 != 0.0f && z == NULL) return DGO_BAD_ARGUMENT;
#pragma omp parallel for schedule(static)
    for (long pos = 0; pos < (long)batch * size * size; ++pos) {
        const int b = (int)(pos / (size * size)), yy = (int)(pos / size % size), xx = (int)(pos % size);
        for (int co = 0; co < cout; ++co) {
            float acc = 0.0f;
            for (int tap = 0; tap < 9; ++tap) {
                const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
                if (sy < 0 || sy >= size || sx < 0 || sx >= size) continue;
                const uint16_t* xin = x + (((size_t)b * size + sy) * size + sx) * cin;
                const uint16_t* wr = w + ((size_t)co * 9 + tap) * cin;
                for (int ci = 0; ci < cin; ++ci) acc = fmaf(dgo_f16_to_f32(xin[ci]), dgo_f16_to_f32(wr[ci]), acc);
            }
            float t = fmaf(alpha1, acc, dgo_f16_to_f32(bias[co]));
            if (alpha2 != 0.0f) t = fmaf(alpha2, dgo_f16_to_f32(z[(size_t)pos * cout + co]), t);
            if (relu && !(t > 0.0f)) t = 0.0f;
            y[(size_t)pos * cout + co] = dgo_f32_to_f16(t);
        }
    }
    return DGO_OK;
}

/* The fp32 convolution (what every parity test and bench.py's cpu_baseline run), fast.  Round 6: register-tiled -- 5 positions of a
 * board row x 16 output channels per micro-kernel (ten AVX2 accumulators), on a ZERO-PADDED copy of the input, the filter rows of
 * a 16-channel tile contiguous, hardware fp16 conversions (F16C: exact half -> float; float -> half to nearest even, the rounding
 * of dgo_f32_to_f16), a scratch buffer per calling thread instead of 121 MB of fresh pages per layer.  THE ARITHMETIC OF AN OUTPUT
 * IS UNCHANGED: k runs tap-major then channel, one fused multiply-add per product, in order, starting from +0 -- a tap that falls
 * off the board adds fma(0, w, acc) = acc (exact for finite w; acc is never -0: it starts at +0 and +0 + (-0) = +0) where the
 * plain loop skips it.  (The round 1-5 form streamed a unit's 590 KB of filter through the L2 for every position, rows 1 KB apart
 * in four cache sets: 36 evaluations per second on 16 cores.) */
#include <immintrin.h>
#define DGO_CO_TILE 16
#define DGO_POS_TILE 5

static void dgo_f16_to_f32_run(const uint16_t* src, float* dst, size_t n)
{
    size_t i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_ps(dst + i, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(src + i))));
    for (; i < n; ++i) dst[i] = dgo_f16_to_f32(src[i]);
}

static _Thread_local float* dgo_scratch = NULL;          /* the padded input of the calling thread's convolutions (grow-only) */
static _Thread_local size_t dgo_scratch_floats = 0;
static _Thread_local long dgo_scratch_layout[3] = {0, 0, 0};   /* (batch, size, cin) it was last zero-bordered for */

int dgo_conv3x3(const uint16_t* x, int batch, int size, int cin,
                const uint16_t* w, const uint16_t* bias, int cout,
                float alpha1, float alpha2, const uint16_t* z, int relu,
                uint16_t* y)
{
    if (batch < 0 || size <= 0 || cin <= 0 || cout <= 0) return DGO_BAD_ARGUMENT;
    if (alpha2 != 0.0f && z == NULL) return DGO_BAD_ARGUMENT;

    const int cout_pad = (cout + DGO_CO_TILE - 1) / DGO_CO_TILE * DGO_CO_TILE;
    const int tiles_x = (size + DGO_POS_TILE - 1) / DGO_POS_TILE;
    const int pw = tiles_x * DGO_POS_TILE + 2;          /* padded row: x = -1 .. tiles_x * 5 */
    const int ph = size + 2;
    const size_t plane = (size_t)ph * pw * cin;
    const size_t need = (size_t)batch * plane + 8;

    /* the border must be zero; the interior is rewritten by every call: a buffer that last served the same layout keeps its border */
    if (need > dgo_scratch_floats) {
        free(dgo_scratch);
        dgo_scratch = (float*)calloc(need, sizeof(float));
        dgo_scratch_floats = dgo_scratch ? need : 0;
        if (!dgo_scratch) return DGO_OUT_OF_MEMORY;
    } else if (dgo_scratch_layout[0] != batch || dgo_scratch_layout[1] != size || dgo_scratch_layout[2] != cin) {
        float* const base = dgo_scratch;
#pragma omp parallel for schedule(static)
        for (long b = 0; b < batch; ++b) memset(base + (size_t)b * plane, 0, plane * sizeof(float));
    }
    dgo_scratch_layout[0] = batch; dgo_scratch_layout[1] = size; dgo_scratch_layout[2] = cin;
    float* const xp = dgo_scratch;
    /* weights re-laid as [cout tile][tap][cin][16]: the rows a micro-kernel walks are CONTIGUOUS (a [tap][cin][cout_pad] order puts them
     * 1 KB apart at 256 channels -- four cache sets for the whole filter: every row then comes from the L3) */
    float* wf = (float*)calloc((size_t)9 * cin * cout_pad, sizeof(float));
    float* bf = (float*)calloc((size_t)cout_pad, sizeof(float));
    if (!wf || !bf) { free(wf); free(bf); return DGO_OUT_OF_MEMORY; }

#pragma omp parallel for schedule(static)
    for (long row = 0; row < (long)batch * size; ++row) {
        const int b = (int)(row / size), yy = (int)(row % size);
        dgo_f16_to_f32_run(x + ((size_t)b * size + yy) * size * cin, xp + (size_t)b * plane + ((size_t)(yy + 1) * pw + 1) * cin, (size_t)size * cin);
    }
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        bf[co] = dgo_f16_to_f32(bias[co]);
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < cin; ++ci)
                wf[(((size_t)(co / DGO_CO_TILE) * 9 + tap) * cin + ci) * DGO_CO_TILE + co % DGO_CO_TILE] =
                    dgo_f16_to_f32(w[((size_t)co * 9 + tap) * cin + ci]);
    }

    /* Work units = (board, PAIR of 16-channel tiles), the two tiles one after the other: the 9 * cin * 16 weights of a tile (147 KB at
     * cin = 256) stay in the core's cache while it walks the board, and a unit owns whole 64-byte lines of y (32 fp16 per position:
     * with one tile per unit two threads wrote the two halves of every line -- eight threads ran slower than one). */
    const long num_tiles = cout_pad / DGO_CO_TILE;
    const long num_pairs = (num_tiles + 1) / 2;
    const long num_units = (long)batch * num_pairs * 2;
    const __m256 va1 = _mm256_set1_ps(alpha1), va2 = _mm256_set1_ps(alpha2), zero = _mm256_setzero_ps();

#pragma omp parallel for schedule(dynamic, 2)
    for (long unit = 0; unit < num_units; ++unit) {
        const int b = (int)(unit / (num_pairs * 2));
        const long tile = unit % (num_pairs * 2);
        if (tile >= num_tiles) continue;
        const int co0 = (int)tile * DGO_CO_TILE;
        const int co_end = (co0 + DGO_CO_TILE < cout) ? DGO_CO_TILE : cout - co0;
        const float* xb = xp + (size_t)b * plane;
        const __m256 b0 = _mm256_loadu_ps(bf + co0), b1 = _mm256_loadu_ps(bf + co0 + 8);

        for (int yy = 0; yy < size; ++yy) {
            for (int xt = 0; xt < tiles_x; ++xt) {
                const int x0 = xt * DGO_POS_TILE;
                __m256 acc[DGO_POS_TILE][2];
                for (int p = 0; p < DGO_POS_TILE; ++p) { acc[p][0] = _mm256_setzero_ps(); acc[p][1] = _mm256_setzero_ps(); }

                for (int tap = 0; tap < 9; ++tap) {
                    /* padded coordinates: (yy + dy + 1, x0 + p + dx + 1) */
                    const float* xin = xb + ((size_t)(yy + tap / 3) * pw + (x0 + tap % 3)) * cin;
                    const float* wt = wf + ((size_t)(co0 / DGO_CO_TILE) * 9 + tap) * cin * DGO_CO_TILE;
                    for (int ci = 0; ci < cin; ++ci) {
                        const __m256 w0 = _mm256_loadu_ps(wt + (size_t)ci * DGO_CO_TILE);
                        const __m256 w1 = _mm256_loadu_ps(wt + (size_t)ci * DGO_CO_TILE + 8);
                        for (int p = 0; p < DGO_POS_TILE; ++p) {
                            const __m256 v = _mm256_broadcast_ss(xin + (size_t)p * cin + ci);
                            acc[p][0] = _mm256_fmadd_ps(v, w0, acc[p][0]);       /* fmaf(v, w, acc) per lane */
                            acc[p][1] = _mm256_fmadd_ps(v, w1, acc[p][1]);
                        }
                    }
                }

                for (int p = 0; p < DGO_POS_TILE; ++p) {
                    const int xx = x0 + p;
                    if (xx >= size) break;
                    const size_t out_pos = ((size_t)b * size + yy) * size + xx;
                    if (co_end == DGO_CO_TILE) {
                        /* t = fmaf(alpha1, acc, bias); [t = fmaf(alpha2, z, t);] relu: t > 0 ? t : +0 (a NaN gives 0, as !(t > 0) does); fp16 nearest even */
                        __m256 t0 = _mm256_fmadd_ps(va1, acc[p][0], b0), t1 = _mm256_fmadd_ps(va1, acc[p][1], b1);
                        if (alpha2 != 0.0f) {
                            const uint16_t* zr = z + out_pos * cout + co0;
                            t0 = _mm256_fmadd_ps(va2, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)zr)), t0);
                            t1 = _mm256_fmadd_ps(va2, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(zr + 8))), t1);
                        }
                        if (relu) {
                            t0 = _mm256_and_ps(t0, _mm256_cmp_ps(t0, zero, _CMP_GT_OQ));
                            t1 = _mm256_and_ps(t1, _mm256_cmp_ps(t1, zero, _CMP_GT_OQ));
                        }
                        uint16_t* yr = y + out_pos * cout + co0;
                        _mm_storeu_si128((__m128i*)yr, _mm256_cvtps_ph(t0, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
                        _mm_storeu_si128((__m128i*)(yr + 8), _mm256_cvtps_ph(t1, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
                        continue;
                    }
                    float a[DGO_CO_TILE];
                    _mm256_storeu_ps(a, acc[p][0]);
                    _mm256_storeu_ps(a + 8, acc[p][1]);
                    for (int j = 0; j < co_end; ++j) {
                        const int co = co0 + j;
                        float t = fmaf(alpha1, a[j], bf[co]);
                        if (alpha2 != 0.0f) t = fmaf(alpha2, dgo_f16_to_f32(z[out_pos * cout + co]), t);
                        if (relu && !(t > 0.0f)) t = 0.0f;
                        y[out_pos * cout + co] = dgo_f32_to_f16(t);
                    }
                }
            }
        }
    }

    free(wf); free(bf);
    return DGO_OK;
}

/* ---------------------------------------------------------------------------
 * Dense layer = 1x1 convolution over [batch][in][1][1], fp32 compute
 * (dense.rs:108-135,197-220):   y = act( alpha1 * (w_out_in . x) + bias )
 * w_out_in is the TRANSPOSED filter [out][in] (what cuDNN sees after
 * Dense::prepare).  relu != 0 -> ReLU (the Dense default, dense.rs:97-103).
 * ------------------------------------------------------------------------- */
int dgo_dense(const uint16_t* x, int batch, int num_in,
              const uint16_t* w_out_in, const uint16_t* bias, int num_out,
              float alpha1, int relu, uint16_t* y)
{
    if (batch < 0 || num_in <= 0 || num_out <= 0) return DGO_BAD_ARGUMENT;

    float* wf = (float*)malloc(sizeof(float) * (size_t)num_in * num_out);
    if (!wf) return DGO_OUT_OF_MEMORY;
    dgo_f16_to_f32_array(w_out_in, wf, (size_t)num_in * num_out);

#pragma omp parallel for schedule(static)
    for (int b = 0; b < batch; ++b) {
        const uint16_t* xb = x + (size_t)b * num_in;

        for (int o = 0; o < num_out; ++o) {
            const float* wr = wf + (size_t)o * num_in;
            float acc = 0.0f;
            for (int i = 0; i < num_in; ++i) acc = fmaf(dgo_f16_to_f32(xb[i]), wr[i], acc);

            float t = fmaf(alpha1, acc, dgo_f16_to_f32(bias[o]));
            if (relu && !(t > 0.0f)) t = 0.0f;
            y[(size_t)b * num_out + o] = dgo_f32_to_f16(t);
        }
    }

    free(wf);
    return DGO_OK;
}

/* cudnnSoftmaxForward(ACCURATE, MODE_INSTANCE) over n entries per instance,
 * fp16 in / fp16 out, fp32 inside (softmax.rs:58-79; policy_head.rs:98-100). */
void dgo_softmax(const uint16_t* x, int batch, int n, uint16_t* y)
{
    for (int b = 0; b < batch; ++b) {
        const uint16_t* xb = x + (size_t)b * n;
        uint16_t* yb = y + (size_t)b * n;

        float max_value = -INFINITY;
        for (int i = 0; i < n; ++i) {
            const float v = dgo_f16_to_f32(xb[i]);
            if (v > max_value) max_value = v;
        }

        float sum = 0.0f;
        for (int i = 0; i < n; ++i) sum += expf(dgo_f16_to_f32(xb[i]) - max_value);

        for (int i = 0; i < n; ++i)
            yb[i] = dgo_f32_to_f16(expf(dgo_f16_to_f32(xb[i]) - max_value) / sum);
    }
}

/* cudnnActivationForward(tanh), in place, fp16 (activation.rs:60-82; value_head.rs:81). */
void dgo_tanh(uint16_t* x, size_t n)
{
    for (size_t i = 0; i < n; ++i) x[i] = dgo_f32_to_f16(tanhf(dgo_f16_to_f32(x[i])));
}

/* ---------------------------------------------------------------------------
 * Whole network.  All weight pointers are the tensors exactly as stored in the
 * weights file (loader.rs:36-116): conv filters KRSC, dense matrices [in][out],
 * biases un-scaled.  The one-time device-side preparation of the reference
 * (alpha*b2, tau*b_l, dense transpose) is done here per call.
 * ------------------------------------------------------------------------- */
typedef struct {
    int num_blocks;       /* residual blocks found by name probing (graph.rs:76-96) */
    int num_channels;     /* num_channels:0, default 128 (layers/common.rs:22-36) */
    int num_samples;      /* num_samples:0, default 8   (layers/common.rs:38-50) */
    int num_features;     /* 32 (libdg_go/utils/features.rs:88-90) */
    int board_size;       /* 19 (conv2d.rs:50) */
    float softmax_temperature; /* 0.709888 (libdg_utils/config.rs:176-177) */

    const uint16_t* up_w;            /* 01_upsample/conv_1:0        [C][3][3][32] */
    const uint16_t* up_b;            /* 01_upsample/conv_1/offset:0 [C] */
    const uint16_t* const* res_w1;   /* {02+i}_residual/conv_1:0    [C][3][3][C] */
    const uint16_t* const* res_b1;
    const uint16_t* const* res_w2;
    const uint16_t* const* res_b2;
    const float* res_alpha;          /* {02+i}_residual/alpha:0, default 0.5 */
    const uint16_t* policy_conv_w;   /* {2+blocks}p_policy/conv_1:0   [S][3][3][C] */
    const uint16_t* policy_conv_b;
    const uint16_t* policy_linear_w; /* {2+blocks}p_policy/linear_1:0 [361*S][362] */
    const uint16_t* policy_linear_b; /* [362] */
    const uint16_t* value_conv_w;    /* {2+blocks}v_value/conv_1:0    [2][3][3][C] */
    const uint16_t* value_conv_b;
    const uint16_t* value_linear_w;  /* {2+blocks}v_value/linear_2:0  [722][1] */
    const uint16_t* value_linear_b;  /* [1] */
} dgo_network;

/* graph.rs:123-158.  features: [batch][361][32] fp16 (HWC, features.rs:52-60);
 * value_out: [batch] fp16; policy_out: [batch][362] fp16. */
/* dgo_forward_ex: the same pass, optionally handing out the two fp16 tensors in FRONT of the output activations -- the
 * temperature-scaled logits (input of the soft-max, policy_head.rs:95-100; [batch][362]) and the value head's dense output (input of
 * tanh, value_head.rs:80-81; [batch]) -- and optionally with exact (double) products and sums inside every layer, same fp16
 * rounding points (accumulate_f64 != 0: the yardstick both fp32 implementations are measured against). */
int dgo_forward_ex(const dgo_network* net, const uint16_t* features, int batch,
                   uint16_t* value_out, uint16_t* policy_out, uint16_t* logits_out, uint16_t* value_pre_out, int accumulate_f64);

int dgo_forward(const dgo_network* net, const uint16_t* features, int batch,
                uint16_t* value_out, uint16_t* policy_out)
{
    return dgo_forward_ex(net, features, batch, value_out, policy_out, NULL, NULL, 0);
}

static int conv_any(int f64, const uint16_t* x, int batch, int size, int cin, const uint16_t* w, const uint16_t* bias, int cout,
                    float alpha1, float alpha2, const uint16_t* z, int relu, uint16_t* y)
{
    return f64 ? dgo_conv3x3_f64(x, batch, size, cin, w, bias, cout, alpha1, alpha2, z, relu, y)
               : dgo_conv3x3(x, batch, size, cin, w, bias, cout, alpha1, alpha2, z, relu, y);
}

int dgo_forward_ex(const dgo_network* net, const uint16_t* features, int batch,
                   uint16_t* value_out, uint16_t* policy_out, uint16_t* logits_out, uint16_t* value_pre_out, int accumulate_f64)
{
    if (!net || !features || !value_out || !policy_out || batch <= 0) return DGO_BAD_ARGUMENT;
    const int f64 = accumulate_f64 != 0;

    const int size = net->board_size;
    const int num_pos = size * size;
    const int c = net->num_channels;
    const int s = net->num_samples;
    const size_t act_elems = (size_t)batch * num_pos * c;
    int status = DGO_OK;

    uint16_t* t0 = (uint16_t*)malloc(sizeof(uint16_t) * act_elems);
    uint16_t* t1 = (uint16_t*)malloc(sizeof(uint16_t) * act_elems);
    uint16_t* t2 = (uint16_t*)malloc(sizeof(uint16_t) * act_elems);
    uint16_t* bias_tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(c > 362 ? c : 362));
    uint16_t* policy_1 = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)batch * num_pos * s);
    uint16_t* policy_2 = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)batch * (num_pos + 1));
    uint16_t* policy_wt = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)num_pos * s * (num_pos + 1));
    uint16_t* value_1 = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)batch * num_pos * 2);

    if (!t0 || !t1 || !t2 || !bias_tmp || !policy_1 || !policy_2 || !policy_wt || !value_1) {
        status = DGO_OUT_OF_MEMORY;
        goto done;
    }

    /* up_block.rs:56: relu(conv(features, W_up) + b_up) */
    status = conv_any(f64, features, batch, size, net->num_features, net->up_w, net->up_b, c,
                         1.0f, 0.0f, NULL, 1, t0);
    if (status) goto done;

    /* residual_block.rs:65-78 */
    for (int i = 0; i < net->num_blocks; ++i) {
        const float alpha = net->res_alpha[i];

        status = conv_any(f64, t0, batch, size, c, net->res_w1[i], net->res_b1[i], c,
                             1.0f, 0.0f, NULL, 1, t1);
        if (status) goto done;

        memcpy(bias_tmp, net->res_b2[i], sizeof(uint16_t) * (size_t)c);
        dgo_scale(bias_tmp, (size_t)c, alpha); /* residual_block.rs:72-74 */

        status = conv_any(f64, t1, batch, size, c, net->res_w2[i], bias_tmp, c,
                             alpha, 1.0f - alpha, t0, 1, t2);
        if (status) goto done;

        uint16_t* swap = t0; t0 = t2; t2 = swap;
    }

    /* value_head.rs:66-84 */
    status = conv_any(f64, t0, batch, size, c, net->value_conv_w, net->value_conv_b, 2,
                         1.0f, 0.0f, NULL, 1, value_1);
    if (status) goto done;
    /* [722][1] -> [1][722] is the identity permutation */
    status = f64 ? dgo_dense_f64(value_1, batch, num_pos * 2, net->value_linear_w, net->value_linear_b, 1, 1.0f, 0, value_out)
                 : dgo_dense(value_1, batch, num_pos * 2, net->value_linear_w, net->value_linear_b, 1, 1.0f, 0, value_out);
    if (status) goto done;
    if (value_pre_out) memcpy(value_pre_out, value_out, sizeof(uint16_t) * (size_t)batch);
    if (f64) { for (int b = 0; b < batch; ++b) value_out[b] = dgo_f32_to_f16(dgo_round_f64_to_f16_value(tanh((double)dgo_f16_to_f32(value_out[b])))); }
    else dgo_tanh(value_out, (size_t)batch);

    /* policy_head.rs:78-103 */
    {
        const float tau = 1.0f / net->softmax_temperature; /* policy_head.rs:46 */
        const int num_out = num_pos + 1;

        status = conv_any(f64, t0, batch, size, c, net->policy_conv_w, net->policy_conv_b, s,
                             1.0f, 0.0f, NULL, 1, policy_1);
        if (status) goto done;

        dgo_transpose_in_out(net->policy_linear_w, policy_wt, num_pos * s, num_out);
        memcpy(bias_tmp, net->policy_linear_b, sizeof(uint16_t) * (size_t)num_out);
        dgo_scale(bias_tmp, (size_t)num_out, tau); /* policy_head.rs:87-89 */

        status = f64 ? dgo_dense_f64(policy_1, batch, num_pos * s, policy_wt, bias_tmp, num_out, tau, 0, policy_2)
                     : dgo_dense(policy_1, batch, num_pos * s, policy_wt, bias_tmp, num_out, tau, 0, policy_2);
        if (status) goto done;
        if (logits_out) memcpy(logits_out, policy_2, sizeof(uint16_t) * (size_t)batch * num_out);
        if (f64) {
            for (int b = 0; b < batch; ++b) {
                const uint16_t* xb = policy_2 + (size_t)b * num_out;
                double max_value = -INFINITY, sum = 0.0;
                for (int i = 0; i < num_out; ++i) { const double v = (double)dgo_f16_to_f32(xb[i]); if (v > max_value) max_value = v; }
                for (int i = 0; i < num_out; ++i) sum += exp((double)dgo_f16_to_f32(xb[i]) - max_value);
                for (int i = 0; i < num_out; ++i)
                    policy_out[(size_t)b * num_out + i] = dgo_f32_to_f16(dgo_round_f64_to_f16_value(exp((double)dgo_f16_to_f32(xb[i]) - max_value) / sum));
            }
        } else {
            dgo_softmax(policy_2, batch, num_out, policy_out);
        }
    }

done:
    free(t0); free(t1); free(t2); free(bias_tmp);
    free(policy_1); free(policy_2); free(policy_wt); free(value_1);
    return status;
}

int dgo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* For the timed CPU baseline: one thread per usable core (the OpenMP runtime may already have been initialised by another
 * library of the process with a different count).  The results do not depend on the number of threads. */
void dgo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
