// What the MFMA pipe delivers with nothing else going on: 8 waves per workgroup (2 per SIMD), one workgroup per CU, every wave
// issuing independent v_mfma_f32_32x32x16_f16 back to back (12 accumulator tiles, like the board kernel).  Prints the TFLOP/s from
// the wall clock -- the practical ceiling the convolution is measured against in DESIGN.md section 4.3 -- for three kinds of
// operands, because the part runs against a package power cap and a multiplier array that sees constants draws far less:
//   mode 0  one constant A and B fragment (round 1's figure)
//   mode 1  random fp16 operands: 2 A fragments and 12 B fragments per wave in registers, every MFMA a different pair
//   mode 2  as mode 1 with half of the B elements zero (activations behind a ReLU)
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak [seconds per mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters)
{
    floatx16 acc[12];
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    half8 a[2], b[12];
    const unsigned id = blockIdx.x * 512 + threadIdx.x;
    for (int f = 0; f < 14; ++f)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(id * 131u + f * 17u + e);
            _Float16 v = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));      // uniform in [-1, 1)
            if (MODE == 0) v = (_Float16)(threadIdx.x * 0.001f + e);
            if (MODE == 2 && f >= 2 && (h >> 16 & 1)) v = (_Float16)0.0f;
            if (f < 2) a[f][e] = v; else b[f - 2][e] = v;
        }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 12; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MODE == 0 ? a[0] : a[k & 1], MODE == 0 ? b[0] : b[(t + 5 * k) % 12], acc[t], 0, 0, 0);
    }
    float s = 0.0f;
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[id] = s;
}

template <int MODE>
static void run(float* out, double seconds)
{
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops_per_launch = (double)blocks * 8 * iters * 48 * 32768.0;
    double done = 0.0;
    int rep = 0;
    while (done < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        done += ms * 1e-3;
        const double tf = 100.0 * flops_per_launch / (ms * 1e-3) / 1e12;
        const double ns_per_mfma = ms * 1e6 / 100.0 / (2.0 * iters * 48);   // two waves per SIMD; a 32x32x16 MFMA occupies the pipe for 32 cycles
        if (rep++ % 4 == 0)
            std::printf("mode %d t=%5.2fs: %.1f TFLOP/s (%.1f %% of the nominal 2500) = %.2f GHz at 32 cycles per MFMA\n", MODE, done, tf, tf / 25.0,
                        32.0 / ns_per_mfma);
        std::fflush(stdout);
    }
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 0.5;
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0>(out, seconds);
    run<1>(out, seconds);
    run<2>(out, seconds);
    return 0;
}
