// Third pass of the energy probes: the bare 16x16x32 MFMA loop of mfma_energy2 (mode 1: the tower kernel's
// order, a B fragment feeds four 16-channel tiles) with the real K loop's OTHER instructions added one ingredient at a time, so that
// "what the part charges for everything that is not the multiplier array" becomes a number per ingredient.  One wave per SIMD, four
// waves per workgroup, one workgroup per CU (60 KB of LDS), random fp16 filter operand, half-zero activation operand -- the data the
// tower sees.  Per iteration a wave issues 48 MFMAs (half a k-step of the shipped loop: 12 row tiles x 4 channel tiles) and, by mode:
//   0  nothing else: operands stay in registers (mfma_energy2 mode 1)
//   1  + the B fragments from LDS: 12 ds_read_b128 (one per four MFMAs), double buffered, another 1 KB window every iteration
//   2  + the A fragments from memory: 2 x 2 global_load_dwordx4 per iteration (DirectA: the filter slice streams from the L2)
//   3  + the activation chunk's LDS-DMA: one 1 KB global_load_lds piece per wave every third iteration (25.6 KB per 9 k-steps and
//        workgroup), from a buffer that lives in the L2 / Infinity Cache
//   4  1 + 2 + 3 together: the shipped K loop's memory instructions without its epilogue and skip fold
//   5  4 + the epilogue's share of VALU work: 16 v_fma_f32 per iteration
//   6  no MFMA at all: sleeping waves on the same CUs (what the clocked chip draws without the matrix pipe)
// Each mode runs with 256, 128 and 64 workgroups (one per CU).
// The script (tools/mfma_energy.sh <seconds> mfma_energy3) samples the socket's clock and power twice a second beside it.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_energy3.hip -o /tmp/mfma_energy3 && /tmp/mfma_energy3 [seconds per mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <initializer_list>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

__device__ unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kLdsBytes = 60 * 1024;       // what the wide split holds: one workgroup per CU
constexpr int kWindow = 12 * 1024;         // the 12 B fragments of an iteration: 12 KB of LDS, four windows

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mfma_loop(float* out, const half8* __restrict__ filter, const half8* __restrict__ acts, int iters)
{
    extern __shared__ char lds[];
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool kLdsB = MODE == 1 || MODE >= 4, kGlobalA = MODE == 2 || MODE >= 4, kDma = MODE == 3 || MODE >= 4, kValu = MODE == 5;
    // random operands: a in [-1, 1), b the same with half of the elements zero (behind a ReLU)
    half8 a[2][4], b[2][12];
    for (int f = 0; f < 16; ++f)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(id * 131u + f * 17u + e);
            float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
            if (f >= 4 && (h >> 16 & 1)) x = 0.0f;
            if (f < 4) a[0][f][e] = a[1][f][e] = (_Float16)x; else b[0][f - 4][e] = b[1][f - 4][e] = (_Float16)x;
        }
    // LDS holds four windows of such B data
    for (int i = threadIdx.x; i < kLdsBytes / 16; i += 256) {
        half8 v;
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(blockIdx.x * 7919u + i * 8u + e);
            float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
            v[e] = (h >> 16 & 1) ? (_Float16)0.0f : (_Float16)x;
        }
        reinterpret_cast<half8*>(lds)[i] = v;
    }
    __syncthreads();
    floatx4 acc[4][12];
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) acc[c][t][r] = 0.0f;
    floatx2 side[8];
    for (int i = 0; i < 8; ++i) { side[i][0] = (float)id * 1e-9f + i; side[i][1] = 1.0f; }
    const half8* fsrc = filter + (size_t)(blockIdx.x & 7) * 8192 + wave * 256 + lane;     // a slice per wave, 8 slices chip-wide hot in the L2
    const half8* xsrc = acts + (size_t)blockIdx.x * 4096 + wave * 64 + lane;
    // The MFMAs are inline asm with the accumulator tied to the destination and pinned to AGPRs, as in the tower kernels (as a builtin
    // the instruction is untied to hipcc, which then shuffles accumulators through v_accvgpr moves and measures those instead); the
    // loads are inline asm too, placed BEHIND the MFMAs of a row tile like the shipped loop places them (hipcc would issue them in
    // one clump in front of the MFMAs), and waited for once per iteration: they were requested a whole iteration earlier.
    // Two iterations per trip so that the double buffers are indexed by constants (a run-time index would put them in scratch).
    if (MODE == 6) {                                           // the clock and the rest of the chip without the matrix pipe: idle waves
        for (int i = 0; i < iters * 4; ++i) asm volatile("s_sleep 8");
        out[id] = 0.0f;
        return;
    }
#define DGNN_STEP(cur, nxt, i)                                                                                                        \
    {                                                                                                                                 \
        const unsigned lds_at = (unsigned)(size_t)(lds + (((i) + 1) & 3) * kWindow + lane * 16);                                      \
        const half8* g = fsrc + (size_t)(((i) * 4) & 127) * 64;                                                                       \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                                                \
        _Pragma("unroll") for (int t = 0; t < 12; ++t) {                                                                              \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c][t]) : "v"(a[kGlobalA ? cur : 0][c]), "v"(b[kLdsB ? cur : 0][t])); \
            if (kLdsB) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[nxt][t]) : "v"(lds_at), "n"(t * 1024));                 \
            if (kGlobalA && t % 3 == 0) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(a[nxt][t / 3]) : "v"(g), "n"((t / 3) * 1024)); \
            if (kValu && t < 8) { side[t][0] = side[t][0] * 1.0001f + side[(t + 1) & 7][1]; side[t][1] = side[t][1] * 0.9999f + side[t][0]; } \
        }                                                                                                                             \
        if (kDma && (i) % 3 == 0) {                                                                                                   \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + (size_t)(((i) / 3) & 31) * 256),  \
                                             (__attribute__((address_space(3))) void*)(lds + 4 * kWindow + wave * 1024), 16, 0, 0);   \
        }                                                                                                                             \
    }
    for (int i = 0; i < iters; i += 2) {
        DGNN_STEP(0, 1, i)
        DGNN_STEP(1, 0, i + 1)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
#undef DGNN_STEP
    float s = 0.0f;
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 12; ++t) for (int r = 0; r < 4; ++r) s += acc[c][t][r];
    for (int i = 0; i < 8; ++i) s += side[i][0] + side[i][1];
    if (kDma) s += (float)lds[4 * kWindow + threadIdx.x];
    out[id] = s;
}

__global__ void fill_random(_Float16* p, size_t n, int half_zero)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = hash32((unsigned)i * 2654435761u + 12345u);
        float x = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
        if (half_zero && (h >> 16 & 1)) x = 0.0f;
        p[i] = (_Float16)x;
    }
}

template <int MODE>
static void run(float* out, const half8* filter, const half8* acts, double seconds, int blocks)
{
    const int iters = 4000, threads = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)mfma_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    const double flops_per_launch = (double)blocks * (threads / 64) * iters * 48 * 16384.0;
    double done = 0.0, last = 0.0;
    while (done < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(blocks), dim3(threads), kLdsBytes, 0, out, filter, acts, iters);
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
    half8 *filter, *acts;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&filter, 8 * 8192 * 16 * 2 + 65536);      // 8 slices of 128 KB (+ slack)
    (void)hipMalloc(&acts, 256 * 4096 * 16 + 65536);          // 64 KB per workgroup
    // the operands a probe loads must look like the operands it keeps in registers: random filter values, half-zero activations
    // (a first version memset both buffers: with a CONSTANT filter operand the part ran mode 2 at its maximum clock and 1 180 W --
    // what the multiplier array draws depends on how much its inputs toggle, profiles/r05/mfma_energy3_first.txt)
    hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, 0, (_Float16*)filter, (size_t)(8 * 8192 * 16 * 2 + 65536) / 2, 0);
    hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, 0, (_Float16*)acts, (size_t)(256 * 4096 * 16 + 65536) / 2, 1);
    (void)hipDeviceSynchronize();
    // every CU busy: the package sits at its power cap and the clock is the answer; half / a quarter of the CUs: nothing is capped,
    // every mode runs at the same clock and the socket's watts per TFLOP/s separate the ingredients
    for (int blocks : {256, 128, 64}) {
        run<6>(out, filter, acts, seconds, blocks);
        run<0>(out, filter, acts, seconds, blocks); run<1>(out, filter, acts, seconds, blocks); run<2>(out, filter, acts, seconds, blocks);
        run<3>(out, filter, acts, seconds, blocks); run<4>(out, filter, acts, seconds, blocks); run<5>(out, filter, acts, seconds, blocks);
    }
    return 0;
}
