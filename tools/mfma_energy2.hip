// Second pass of tools/mfma_energy.hip: the two MFMA shapes in the ORDER the board kernel issues them, one wave per SIMD, random
// filter operand, half-zero activation operand, 4 A + 12 B fragments in registers, the same FLOPs per iteration.
//   mode 0  32x32x16: per (k16, row tile): one B fragment feeds the two channel tiles (the wide split's K loop)
//   mode 1  16x16x32: per 16-row tile: one B fragment feeds four 16-channel tiles
//   mode 2  16x16x32, A stationary: per channel tile, the 12 row tiles in turn
//   mode 3  32x32x16, A stationary
// hipcc --offload-arch=gfx950 -O3 tools/mfma_energy2.hip -o /tmp/mfma_energy2 && /tmp/mfma_energy2 [seconds per mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters)
{
    half8 a[4], b[12];
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    for (int f = 0; f < 16; ++f)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(id * 131u + f * 17u + e);
            float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
            if (f >= 4 && (h >> 16 & 1)) x = 0.0f;
            if (f < 4) a[f][e] = (_Float16)x; else b[f - 4][e] = (_Float16)x;
        }
    float s = 0.0f;
    if (MODE == 1 || MODE == 2) {
        floatx4 acc[4][12];
        for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) acc[c][t][r] = 0.0f;
        for (int i = 0; i < iters; ++i) {
            if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 12; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[t], acc[c][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < 12; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[t], acc[c][t], 0, 0, 0);
            }
        }
        for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) s += acc[c][t][r];
    } else {
        floatx16 acc[2][6];
        for (int c = 0; c < 2; ++c) for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c * 2 + k], b[k * 6 + t], acc[c][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int t = 0; t < 6; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c * 2 + k], b[k * 6 + t], acc[c][t], 0, 0, 0);
            }
        }
        for (int c = 0; c < 2; ++c) for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) s += acc[c][t][r];
    }
    out[id] = s;
}

template <int MODE>
static void run(float* out, double seconds)
{
    const int blocks = 256, iters = 4000, threads = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flops_per_launch = (double)blocks * (threads / 64) * iters * 24 * 32768.0;
    double done = 0.0, last = 0.0;
    while (done < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        done += ms * 1e-3;
        last = 100.0 * flops_per_launch / (ms * 1e-3) / 1e12;
    }
    std::printf("mode %d final: %.1f TFLOP/s (%.1f %% of the nominal 2500)\n", MODE, last, last / 25.0);
    std::fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 0.5;
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int rep = 0; rep < 2; ++rep) { run<0>(out, seconds); run<1>(out, seconds); run<2>(out, seconds); run<3>(out, seconds); }
    return 0;
}
