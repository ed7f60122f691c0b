// Where the joules of the matrix pipe go (round 3): the bare v_mfma loop of tools/mfma_peak.hip with ONE thing varied per mode, the
// socket power and clock sampled beside it (tools/mfma_energy.sh).  Every mode does the same FLOPs per "group" (48 MFMAs of
// 32x32x16, or 96 of 16x16x32).
//   mode 0  32x32x16, random A, half-zero B (= mfma_peak mode 2: the tower's kind of data), 2 waves per SIMD
//   mode 1  the same with ONE wave per SIMD (the wide split's occupancy)
//   mode 2  half-zero operand as A, random as B (which side should the ReLU'd activations feed?)
//   mode 3  16x16x32, random A, half-zero B
//   mode 4  mode 1 with a duty cycle: after every 48 MFMAs the wave idles for ~25 % of the group's time (s_nop), like the tower's
//           cycles without an MFMA -- does the clock rise to make up for it?
//   mode 5  mode 1 with the idle time filled by independent VALU work (v_pk_fma_f16 on private registers)
//   mode 6  mode 1, both operands random (no zeros), for reference
//   mode 7  mode 1, small-magnitude weights (|a| < 2^-4: the exponent bits hardly toggle), half-zero B
// hipcc --offload-arch=gfx950 -O3 tools/mfma_energy.hip -o /tmp/mfma_energy && /tmp/mfma_energy [seconds per mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters)
{
    half8 a[2], b[12];
    const unsigned id = blockIdx.x * 512 + threadIdx.x;
    for (int f = 0; f < 14; ++f)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(id * 131u + f * 17u + e);
            float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);      // uniform in [-1, 1)
            const bool sparse_side = MODE == 2 ? f < 2 : f >= 2;
            if (MODE == 7 && f < 2) x *= 0.0625f;
            if (MODE != 6 && sparse_side && (h >> 16 & 1)) x = 0.0f;
            if (f < 2) a[f][e] = (_Float16)x; else b[f - 2][e] = (_Float16)x;
        }
    float s = 0.0f;
    if (MODE == 3) {
        floatx4 acc[48];
        for (int t = 0; t < 48; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < 48; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + t) & 1], b[(t + 5 * k) % 12], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 48; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    } else {
        floatx16 acc[12];
        for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        half2v p0 = {(_Float16)0.5f, (_Float16)0.25f}, p1 = {(_Float16)1.5f, (_Float16)0.75f}, p2 = p0, p3 = p1;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 12; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 1], b[(t + 5 * k) % 12], acc[t], 0, 0, 0);
            if (MODE == 4) {
                // 48 MFMAs x 32 cycles = 1536 cycles of pipe; idle ~ 400 cycles = 25 x s_nop 15 (16 cycles each)
#pragma unroll
                for (int n = 0; n < 25; ++n) asm volatile("s_nop 15");
            }
            if (MODE == 5) {
#pragma unroll
                for (int n = 0; n < 25; ++n)
                    asm volatile("v_pk_fma_f16 %0, %0, %1, %1\n v_pk_fma_f16 %2, %2, %3, %3\n v_pk_fma_f16 %0, %0, %3, %1\n v_pk_fma_f16 %2, %2, %1, %3"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            }
        }
        for (int t = 0; t < 12; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        s += (float)p0[0] + (float)p2[1];
    }
    out[id] = s;
}

template <int MODE>
static void run(float* out, double seconds)
{
    const int blocks = 256, iters = 2000;
    const int threads = (MODE == 0 || MODE == 2 || MODE == 3) ? 512 : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops_per_launch = (double)blocks * (threads / 64) * iters * 48 * 32768.0;
    double done = 0.0;
    int rep = 0;
    double last = 0.0;
    while (done < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        done += ms * 1e-3;
        last = 100.0 * flops_per_launch / (ms * 1e-3) / 1e12;
        if (rep++ % 4 == 0) { std::printf("mode %d t=%5.2fs: %.1f TFLOP/s (%.1f %% of the nominal 2500)\n", MODE, done, last, last / 25.0); std::fflush(stdout); }
    }
    std::printf("mode %d final: %.1f TFLOP/s\n", MODE, last);
    std::fflush(stdout);
}

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 0.5;
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0>(out, seconds); run<1>(out, seconds); run<2>(out, seconds); run<3>(out, seconds);
    run<4>(out, seconds); run<5>(out, seconds); run<6>(out, seconds); run<7>(out, seconds);
    return 0;
}
