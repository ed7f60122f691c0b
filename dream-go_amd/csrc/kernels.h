// Launch wrappers of the gfx950 kernels (kernels.hip).  Host-callable, plain C++.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdlib>

// Environment switches.  The SHIPPED library reads eight: SOFTMAX_TEMPERATURE (the reference's own, config.rs:176-177), DGNN_CONV
// (v1 / board: which convolution kernel), DGNN_TOWER (0: one launch per layer), DGNN_TOWER_SPLIT (0: never split a board inside the tower
// launch), DGNN_BOARD_NS and DGNN_BOARD_SKIP (work split / where the skip input is added: the parity tests walk every variant),
// DGNN_HOST_SIMD (0: the scalar host loops), DGNN_RCCL_DISABLE (tests: as if librccl were not installed).  Everything else --
// switches of experiments that were measured and lost, measurement hooks (stamps, traces), A/B toggles -- is read by the
// DGNN_ABLATIONS build only (ablation_env), fault injection by the test and ablations builds (test_env).
static inline const char* ablation_env(const char* name)   // (static: the test library links host objects built with -DDGNN_TEST_ENTRIES beside kernel objects built without -- two bodies of one name, so no external linkage)
{
#ifdef DGNN_ABLATIONS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
static inline const char* test_env(const char* name)
{
#if defined(DGNN_ABLATIONS) || defined(DGNN_TEST_ENTRIES)
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace dgnn {

// LOGICAL devices (test build only): DGNN_FAKE_DEVICES=N makes the host side see N devices, logical device d being physical device
// d % (devices present).  Every device id that crosses the C ABI, keys a Network's per-device state or names a lane of a leaf pool is
// logical; the HIP runtime is only ever handed physical_device(d).  With one GPU in the box the single-process multi-device code --
// dgnn_pool_create_devices over eight lanes, dgnn_broadcast_weights_devices, per-device weight images and workspaces -- runs with N = 8
// (the reference's process model, src/libdg_mcts/predictors/nn.rs:64-67,87-89).  In the shipped library
// test_env() is nullptr and all three are the identity.
int fake_devices();                       // N, or 0: devices are what the runtime says
int physical_device(int logical);
hipError_t device_count(int* count);      // hipGetDeviceCount, or N
hipError_t set_device(int logical);       // hipSetDevice(physical_device(logical))

struct ConvLaunch {
    const uint16_t* x;     // input  [num_pos][cin]  fp16, NHWC with N*H*W flattened
    const uint16_t* w;     // filter [9 * cin / 32][cout_pad][32] fp16 (network.cpp: pack_conv)
    const uint16_t* bias;  // [cout_pad] fp16
    const uint16_t* z;     // skip   [num_pos][cout] fp16, read only when alpha2 != 0
    uint16_t* y;           // output [num_pos][cout] fp16
    int num_pos;           // batch * board * board
    int board;             // board edge length (19; the layer-level known-answer test uses 3)
    int cin;
    int cout;
    int cout_pad;
    float alpha1;
    float alpha2;
    int relu;
};

// y = act(alpha1 * conv3x3(x, w) + alpha2 * z + bias) on 19x19 boards; see kernels.hip.
hipError_t launch_conv3x3(const ConvLaunch& args, hipStream_t stream);

struct ConvTileLaunch {
    const uint16_t* x;          // input  [num_pos][cin] fp16
    const uint16_t* w;          // filter [cout / 256][9 * cin / 64][256][8 slots][8] fp16, slots XOR-swizzled (pack_filter_tile)
    const uint16_t* bias;       // [cout] fp16
    const uint16_t* z;          // skip [num_pos][cout] fp16, read only when alpha2 != 0
    uint16_t* y;                // output [num_pos][cout] fp16
    const uint16_t* zero_page;  // >= 128 bytes of zeros in device memory (source of off-board taps)
    unsigned long long* timing; // instrumentation (DGNN_CONV_ABLATE=8): [workgroup][wave][4] cycle sums, or nullptr
    int num_pos;
    int board;
    int cin;                    // multiple of 64
    int cout;                   // multiple of 256
    float alpha1;
    float alpha2;
    int relu;
};

// Tuned residual-tower convolution (conv_tile.hip).  fm selects the position tile (32 * fm positions per workgroup).
bool conv_tile_supported(int cin, int cout);
int conv_tile_pick_fm(int num_pos, int cout);
// variant 1: lock-step pipeline (conv3x3_tile_kernel); variants 2 / 3: two staggered wave groups
// (conv3x3_tile2_kernel with SCHED = 2: load phase | matrix phase, SCHED = 3: balanced half-steps).
hipError_t launch_conv3x3_tile(const ConvTileLaunch& args, int fm, int variant, hipStream_t stream);

constexpr int kPackedWords = 368;   // == DGNN_PACKED_WORDS (include/dgnn.h)
constexpr int kBoardRows = 400;   // padded slot-major layout: 20 rows x 20 columns per board, row 19 / column 19 zero

struct ConvBoardLaunch {
    const uint16_t* x;     // input  [batch][cin / 8][400][8] fp16 (padded slot-major, pads zero)
    const uint16_t* w;     // filter [9 * cin / 32 k-steps][cout / 16 tiles][64 lanes][8] fp16 (pack_filter_board)
                           //        + 3 k-steps of readable padding behind the last one
    const uint16_t* bias;  // [cout] fp16
    const uint16_t* z;     // skip, same layout as y, read only when alpha2 != 0
    uint16_t* y;           // output [batch][cout / 8][400][8] fp16; pad rows < 384 are written as zero, rows 384..399
                           //        are never written (must be zero from initialisation)
    unsigned long long* timing;   // instrumentation: [workgroup][wave][4] s_memtime stamps (start, loop, epilogue, end) or nullptr
    // Layer-to-layer hand-over inside tower_board_kernel (both zero for a launch per layer):
    const uint16_t* w_next;       // filter of the NEXT layer: its first three slabs are fetched by this layer's last three steps
                                  // (which otherwise read the padding) and its first input chunk -- this layer's channels
                                  // 0..31 -- is left in LDS by the epilogue
    int carry_in;                 // the previous layer did that for this one: no prologue loads
    int batch;
    int cin;               // multiple of 32
    int cout;              // multiple of 64
    float alpha1;
    float alpha2;
    int relu;
};

// The board kernels fold the skip input into the accumulators DURING the K loop as (alpha2 / alpha1) * z: alpha1 must not be 0 and
// the ratio must be a finite number (a gate of 0 takes the epilogue form, which only the per-layer launcher has).
inline bool skip_scale_foldable(float alpha1, float alpha2)
{
    if (alpha1 == 0.0f) return false;
    const float r = alpha2 / alpha1;
    return r == r && r <= 3.0e38f && r >= -3.0e38f;
}

// A/B switches of the board kernel, read from the environment ONCE per workspace (read_board_switches), never per launch:
// DGNN_BOARD_NS=1|2|4|8 (slices per workgroup), DGNN_BOARD_SKIP=1 (skip added in the epilogue), DGNN_CONV_ABLATE (tools/ablate.sh).
struct BoardSwitches {
    int ns = 0;              // 0 = the launcher's own rule
    int epilogue_skip = 0;
    int ablate = 0;
};
BoardSwitches read_board_switches();

// Residual-tower convolution on the padded slot-major layout, one board per workgroup (conv_board.hip).
bool conv_board_supported(int cin, int cout);
int conv_board_pick_ns(int batch, int cout, int forced_ns = 0);   // 32-channel slices per workgroup the launcher will use
hipError_t launch_conv3x3_board(const ConvBoardLaunch& args, const BoardSwitches& env, hipStream_t stream);
// The residual tower as ONE launch, a workgroup per board walking all `count` layers (descriptors in HBM; conv_1 / conv_2 of
// every block, relu, C -> C).  Only where a workgroup owns all channels of its board: conv_board_pick_ns(batch, C) == C / 32.
bool tower_board_supported(int batch, int channels, bool crowded = false);
hipError_t launch_tower_board(const ConvBoardLaunch* d_layers, int count, int batch, int channels, hipStream_t stream);
// The tower as one launch with a board SPLIT over C / (32 * ns) workgroups that hand their channel blocks to each other
// (per-board flags in `d_done`: `done_words` >= batch * C / (32 * ns) words, zero when first used and never reset -- `*base` is
// the launcher's running count they are compared against; `d_err`: one word, zero before the first launch, nonzero after a
// launch if a poll gave up; `d_tickets`: eight running counters the workgroups draw their work items from, `*ticket_base` their value
// before the launch).  tower_split_ns: the slices per workgroup the launcher wants for this batch, 0 = do not split (a launch is
// kept to batch * C / (32 * ns) <= num_cus workgroups: it may run with fewer compute units than that -- conv_board_tower.hip -- but
// should not need two rounds of them).
int tower_split_ns(int batch, int channels, int num_cus);
hipError_t launch_tower_split(const ConvBoardLaunch* d_layers, int count, int batch, int channels, int ns, unsigned* d_done,
                              size_t done_words, unsigned* d_err, unsigned* base, unsigned* d_tickets, unsigned* ticket_base, hipStream_t stream,
                              bool inject_missing = false);
// Layout changes at the two ends of the tower: [B][361][C] NHWC <-> [B][C / 8][400][8].  C a multiple of 8.
hipError_t launch_nhwc_to_board(const uint16_t* in, uint16_t* out, int batch, int channels, hipStream_t stream);
hipError_t launch_board_to_nhwc(const uint16_t* in, uint16_t* out, int batch, int channels, hipStream_t stream);
// Packed features [B][kPackedWords] -> the 32 input planes, either [B][4][400][8] (board_layout) or [B][361][32] NHWC.
hipError_t launch_packed_features(const uint32_t* packed, uint16_t* out, int batch, bool board_layout, hipStream_t stream);
// Compact board records [B][DGNN_BOARD_RECORD_BYTES] -> the planes (features.hip): always the packed words [B][kPackedWords], and --
// planes_board != nullptr -- the up-layer's input [B][4][400][8] as well.
hipError_t launch_board_records(const uint8_t* records, uint32_t* packed, uint16_t* planes_board, int batch, hipStream_t stream);

struct HeadConvLaunch {
    const uint16_t* x;          // tower output [batch][cin / 8][400][8] fp16 (padded slot-major)
    const uint16_t* w;          // [cin / 32][9 taps][NT tiles of 16 rows][64 lanes][8] fp16 (NT = 1 for S + 2 <= 16, else 2): rows 0..S-1 policy conv, S..S+1 value conv, rest 0
    const uint16_t* bias;       // [32] fp16, same row order
    uint16_t* policy_features;  // [batch][361][S] fp16
    uint16_t* value_features;   // [batch][361][2] fp16
    int batch;
    int cin;                    // multiple of 32
    int samples;                // S, S + 2 <= 32
};
// Both head convolutions in one pass over the tower output (heads.hip).
hipError_t launch_head_conv_board(const HeadConvLaunch& args, hipStream_t stream);

struct PolicyGemmLaunch {
    const uint16_t* features;   // [batch][k_in] fp16 (policy conv output, NHWC flattened: index S * pos + s)
    const uint16_t* wt;         // [384][k_pad] fp16: the dense matrix TRANSPOSED ([out][in]), zero padded
    const uint16_t* bias;       // [362] fp16, already multiplied by tau
    uint16_t* logits;           // [batch][ld_out] fp16
    int batch;
    int k_in;                   // 361 * S
    int k_pad;                  // k_in rounded up to a multiple of 64
    float tau;
    int n_out = 362;            // outputs written per board (wt holds ceil(n_out / 32) * 32 rows)
    int ld_out = 384;           // row stride of logits
    int relu = 0;               // Dense with a ReLU (dense.rs:108-116); the heads use the identity
};
hipError_t launch_policy_gemm(const PolicyGemmLaunch& args, hipStream_t stream);

struct SoftmaxValueLaunch {
    const uint16_t* logits;         // [batch][384] fp16
    const uint16_t* value_features; // [batch][722] fp16
    const uint16_t* value_w;        // [722] fp16
    const uint16_t* value_b;        // [1] fp16
    uint16_t* policy_out;           // [batch][362] fp16
    uint16_t* value_out;            // [batch] fp16
    uint16_t* value_pre_out;        // [batch] fp16 or nullptr: the value head's dense output BEFORE tanh (the fp16 tensor value_head.rs:80
                                    // materialises; read back by the parity tests, dgnn_workspace_download_pre)
    int batch;
};
hipError_t launch_softmax_value(const SoftmaxValueLaunch& args, hipStream_t stream);

struct HeadsLaunch {
    const uint16_t* policy_features;  // [batch][361 * samples] fp16  (policy conv output, NHWC flattened)
    const uint16_t* value_features;   // [batch][722] fp16
    const uint16_t* policy_w;         // [361 * samples][362] fp16, as stored ([in][out])
    const uint16_t* policy_b;         // [362] fp16, already multiplied by tau
    const uint16_t* value_w;          // [722] fp16
    const uint16_t* value_b;          // [1] fp16
    uint16_t* value_out;              // [batch] fp16
    uint16_t* policy_out;             // [batch][362] fp16
    int batch;
    int samples;
    float tau;
};

// policy = softmax(tau * (W_l . p) + b_l'), value = tanh(W_d . v + b_d); see kernels.hip.
hipError_t launch_heads(const HeadsLaunch& args, hipStream_t stream);

}  // namespace dgnn
