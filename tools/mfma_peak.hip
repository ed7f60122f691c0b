// What the MFMA pipe delivers with nothing else going on: 8 waves per workgroup (2 per SIMD), one workgroup per CU, every wave
// issuing independent v_mfma_f32_32x32x16_f16 back to back (12 accumulator tiles, like the board kernel).  Prints cycles per
// MFMA per SIMD from s_memtime and the TFLOP/s from the wall clock -- the practical ceiling the convolution is measured
// against in DESIGN.md section 4.3.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* cycles, int iters)
{
    floatx16 acc[12];
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 12; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 512 * sizeof(float));
    (void)hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0; for (auto c : h) mean += (double)c; mean /= blocks;
        const double mfma_per_simd = 2.0 * iters * 48;                       // two waves per SIMD
        const double flops = 10.0 * blocks * 8 * iters * 48 * 32768.0;
        (void)mean;
        const double ns_per_mfma = ms * 1e6 / 10.0 / mfma_per_simd;          // a 32x32x16 MFMA occupies the pipe for 32 cycles
        std::printf("rep %d: %.3f ms for 10 launches -> %.1f TFLOP/s (%.1f %% of the nominal 2500); %.2f ns per MFMA per SIMD = %.2f GHz at 32 cycles each\n",
                    rep, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 25.0, ns_per_mfma, 32.0 / ns_per_mfma);
    }
    return 0;
}
