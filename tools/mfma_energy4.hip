// Fourth energy probe (round 5, after the energy table of DESIGN.md 14.4 said four fifths of the dynamic energy is the multiplier array):
// does the array care in which ORDER it sees its operands, and on which side?  The bare v_mfma_f32_16x16x32_f16 loop of mfma_energy3
// mode 0 (operands in registers, tied inline asm, accumulators in AGPRs, one wave per SIMD; random fp16 filter fragments, half-zero
// activation fragments), 48 MFMAs per iteration over 4 filter fragments x 12 activation fragments, in four arrangements:
//   0  the shipped order: an activation fragment held for four MFMAs, the filter fragment changes with every MFMA; filter = src0
//   1  a filter fragment held for twelve MFMAs, the activation fragment changes with every MFMA; filter = src0
//   2  order 0 with the sides exchanged (activations = src0, filter = src1: the transposed product)
//   3  order 1 with the sides exchanged
//   4  order 0 with three quarters of the activations zero (how much of the array's draw is the data's)
//   6  no MFMA: sleeping waves
// hipcc --offload-arch=gfx950 -O3 tools/mfma_energy4.hip -o /tmp/mfma_energy4 ; tools/mfma_energy.sh <seconds> mfma_energy4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <initializer_list>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mfma_loop(float* out, int iters)
{
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    half8 a[4], b[12];
    for (int f = 0; f < 16; ++f)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(id * 131u + f * 17u + e);
            float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
            if (f >= 4 && ((h >> 16 & 1) || (MODE == 4 && (h >> 17 & 1)))) x = 0.0f;
            if (f < 4) a[f][e] = (_Float16)x; else b[f - 4][e] = (_Float16)x;
        }
    floatx4 acc[4][12];
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) acc[c][t][r] = 0.0f;
    if (MODE == 6) {
        for (int i = 0; i < iters * 4; ++i) asm volatile("s_sleep 8");
        out[id] = 0.0f;
        return;
    }
    constexpr bool kFilterHeld = MODE == 1 || MODE == 3, kSwapped = MODE == 2 || MODE == 3;
    for (int i = 0; i < iters; ++i) {
        if (!kFilterHeld) {
#pragma unroll
            for (int t = 0; t < 12; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (kSwapped) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c][t]) : "v"(b[t]), "v"(a[c]));
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c][t]) : "v"(a[c]), "v"(b[t]));
                }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    if (kSwapped) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c][t]) : "v"(b[t]), "v"(a[c]));
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c][t]) : "v"(a[c]), "v"(b[t]));
                }
        }
    }
    float s = 0.0f;
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) s += acc[c][t][r];
    out[id] = s;
}

template <int MODE>
static void run(float* out, double seconds, int blocks)
{
    const int iters = 4000, threads = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flops_per_launch = (double)blocks * (threads / 64) * iters * 48 * 16384.0;
    double done = 0.0, last = 0.0;
    while (done < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        done += ms * 1e-3;
        last = 50.0 * flops_per_launch / (ms * 1e-3) / 1e12;
    }
    if (MODE == 6) last = 0.0;
    std::printf("t=%.2f workgroups %d mode %d final: %.1f TFLOP/s (%.1f %% of the nominal 2500 x %d / 256)\n", (double)time(nullptr), blocks, MODE, last,
                last / 25.0 * 256.0 / blocks, blocks);
    std::fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 0.5;
    float* out;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    for (int blocks : {256, 128}) {
        run<6>(out, seconds, blocks);
        run<0>(out, seconds, blocks); run<1>(out, seconds, blocks); run<2>(out, seconds, blocks); run<3>(out, seconds, blocks); run<4>(out, seconds, blocks);
        run<0>(out, seconds, blocks);           // (the first arrangement again: the drift of a run)
    }
    return 0;
}
