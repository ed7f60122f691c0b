// Device-side resource model, the forward pass and the C ABI (include/dgnn.h).
//
// Resource model (replaces src/libdg_nn/network.rs:54-160, graph.rs:30-113 and the allocator stack of
// src/libdg_cuda/allocator.rs:24-221):
//   Network      immutable packed weight image on the host + one copy per device, uploaded on first use
//   Workspace    (device, batch size) -> one HIP stream, a static activation arena in HBM (no per-layer malloc),
//                pinned host staging for features / value / policy; pooled per (device, batch size)
// Nothing here depends on a process-global current device: every entry point sets the device of the object it
// was handed (hipSetDevice is per-thread).
#include <hip/hip_runtime_api.h>

#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>

#include "dgnn_internal.h"
#include "kernels.h"

namespace dgnn {

static thread_local int g_last_hip_error = 0;

int fake_devices()
{
    static const int n = [] { const char* e = test_env("DGNN_FAKE_DEVICES"); const int v = e ? std::atoi(e) : 0; return v > 0 && v <= 64 ? v : 0; }();
    return n;
}
static int physical_count()
{
    static const int n = [] { int c = 0; return hipGetDeviceCount(&c) == hipSuccess ? c : 0; }();
    return n;
}
int physical_device(int logical) { return fake_devices() > 0 && physical_count() > 0 ? logical % physical_count() : logical; }
hipError_t device_count(int* count)
{
    if (fake_devices() > 0) { *count = fake_devices(); return physical_count() > 0 ? hipSuccess : hipErrorNoDevice; }
    return hipGetDeviceCount(count);
}
hipError_t set_device(int logical) { return hipSetDevice(physical_device(logical)); }

void set_last_hip_error(int code) { g_last_hip_error = code; }

#define DGNN_HIP(expr)                                      \
    do {                                                    \
        const hipError_t hip_status__ = (expr);             \
        if (hip_status__ != hipSuccess) {                   \
            set_last_hip_error((int)hip_status__);          \
            return DGNN_HIP_ERROR;                          \
        }                                                   \
    } while (0)

constexpr int kSplitQueues = 8;          // ticket counters of a split launch: one per block id mod 8 (conv_board_tower.hip)
constexpr int kSplitWordsPerBoard = 16;   // split-tower flags per board: up to 8 channel blocks x 2 row halves (kernels.h: launch_tower_split)

// The ORDER of tower launches on a physical device, where it matters (process-global per device; two networks on one GPU -- a match
// between two nets -- go through the same chain):
//   * split launches may run SIDE BY SIDE as long as the workgroups that can be waiting for a sibling -- at most 8 * (workgroups per
//     board - 1) per launch, one unfinished board per ticket queue (conv_board_tower.hip) -- add up to no more than half the compute
//     units (kSplitWaitersBudget): the other half is always free for the siblings they wait for, so every launch makes progress.  A
//     launch that would take the sum past that waits for the launches in front of it (events; rounds 2-5 chained EVERY split launch
//     behind the previous one, which left the 64 CUs that a 96-board launch does not use idle with two more such launches queued).
//     A board on 16 workgroups (120 possible waiters) still runs alone; boards on 2 or 4 (8 / 24) overlap -- but only with a launch
//     that leaves compute units idle (fewer workgroups than CUs): beside one that fills the chip a second launch only interleaves
//     with it.  Same box, three submissions alive, PCIe-inclusive (tools/r06_sweep_small.py, profiles/r06_final/sweep_small_*.jsonl;
//     chained -> side by side): 48 boards 63.0 -> 69.8 k, 80: 63.1 -> 79.1 k, 96: 72.8 -> 79.7 k, 112: 76.0 -> 78.5 k evals/s; 64 and 128
//     boards (256 workgroups a launch) 72.8 -> 71.6 k and 80.0 -> 78.7 k when overlapped all the same, hence the condition.
//   * whole-board towers are NOT ordered against split launches.  Round 6 built that (a split launch behind every whole-board tower
//     enqueued before it, whole-board towers behind the split launch in front of them) on the theory that a split launch taking half
//     of the CUs a 256-board tower has just released turns the next tower's one round of the chip into two, and measured nothing:
//     two 256-board towers and a 64-board split launch 7.42 against 7.38 ms per round, the pool with 48 games in flight 79.6-79.7 k
//     against 80.5-80.7 k boards/s WITHOUT the ordering (profiles/r06_final/order_ab.jsonl, few_games_in_the_pool_split_alive.txt).  Taken out.
// (Another PROCESS's launches are out of reach; they end, which is all the ticket argument needs -- and the flag waits' 50 ms limit is
// the net, resolve_split_failure.)
struct SplitChain {
    std::mutex mutex;
    struct Launch { hipEvent_t done; int waiters; int workgroups; bool live; };
    Launch recent[4] = {};                  // the most recent split launches (ring, `next` is the oldest): the event behind each, and how many
    int next = 0;                           // of its workgroups can be waiting for a sibling at any one time
    hipEvent_t done = nullptr;              // (recent[0].done: what get_device_state checks)
};

struct Workspace {
    Network* net = nullptr;
    DeviceState* dev = nullptr;
    int device = 0;
    int batch = 0;

    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_done = nullptr;           // event behind the outputs' copies: what dgnn_forward_wait waits for
    long long submit_ns = 0;                // when it was recorded
    double wait_ns = 0.0;                   // how long submissions of `wait_boards` boards took lately (submit -> outputs on the host)
    double wait_last = 0.0;                 // ... the most recent one (wait_submission)
    int wait_boards = 0;
    std::vector<hipEvent_t> tower_events;   // pairs, grown on demand by dgnn_forward_timed
    // measurement only (dgnn_forward_stage_times): events between the stages of a submission of board records, on this stream
    enum { kStages = 8 };                   // start | H2D | planes kernel | tower | head convolution | policy GEMM | soft-max + value | D2H
    hipEvent_t stage_ev[kStages] = {};
    bool stage_on = false;

    void* arena = nullptr;                  // one allocation, carved below
    uint16_t* d_features = nullptr;         // [batch][361][32]
    uint32_t* d_packed = nullptr;           // [batch][DGNN_PACKED_WORDS]: packed features (dgnn_forward_packed)
    uint8_t* d_records = nullptr;           // [batch][DGNN_BOARD_RECORD_BYTES]: compact board records (dgnn_forward_boards)
    uint16_t* d_act[3] = {nullptr, nullptr, nullptr};   // rotating; [batch][361][C] NHWC, or [batch][C/8][400][8] on the board path
    uint16_t* d_features_board = nullptr;   // board path: features in padded slot-major layout [batch][4][400][8]
    uint16_t* d_tower_nhwc = nullptr;       // board path: tower output converted back to NHWC for the head kernels
    uint16_t* d_policy_features = nullptr;  // [batch][361][S]
    uint16_t* d_value_features = nullptr;   // [batch][361][2]
    uint16_t* d_logits = nullptr;           // [batch][384] fp16 policy logits
    uint16_t* d_value = nullptr;            // [batch]
    uint16_t* d_value_pre = nullptr;        // [batch] the value head's dense output before tanh (parity tests: dgnn_workspace_download_pre)
    uint16_t* d_policy = nullptr;           // [batch][362]

    void* pinned = nullptr;                 // host staging, one allocation
    uint16_t* h_features = nullptr;
    uint32_t* h_packed = nullptr;
    uint8_t* h_records = nullptr;
    uint16_t* h_value = nullptr;
    uint16_t* h_policy = nullptr;

    int small_others = 0;                   // boards of OTHER submissions in flight on the device that are not whole-board towers (split launches this one may run beside)
    bool crowded_per_layer = false;         // ... at more than 128 channels: a batch too small for a workgroup per board runs one launch per layer while
                                            // whole-board towers of other submissions hold the device (note_submission_begins)
    bool crowded = false;                   // the submission being enqueued shares the device with enough boards of other submissions
                                            // that a workgroup per board is the better tower at any size (tower_mode, note_submission_begins)
    int counted_in_flight = 0;              // boards of this workspace's submission counted in DeviceState::boards_in_flight
    int counted_whole = 0;                  // ... and in DeviceState::whole_in_flight
    bool outputs_on_host = false;           // the most recent pass wrote value / policy into the pinned host buffers, not into d_value / d_policy
    bool host_io = false;                   // the submission being enqueued reads its records from / writes its outputs to the PINNED HOST buffers
                                            // themselves (zero copy: no H2D / D2H blit kernels around the compute, see kZeroCopy)
    int active = 0;                         // boards of the submission in flight / being enqueued (<= batch, the capacity)
    bool pending = false;                   // a submit without its wait
    bool use_board = false;                 // padded slot-major layout + one-board-per-workgroup kernel (conv_board.hip)
    bool use_tower = false;                 // the residual tower as ONE launch where a workgroup owns a whole board (DGNN_TOWER)
    ConvBoardLaunch* d_tower_layers = nullptr;   // 2 * blocks descriptors for tower_board_kernel, filled on first use
    unsigned long long* d_tower_timing = nullptr;   // measurement only (DGNN_TOWER_TIMING): per-layer s_memtime stamps of a whole-board tower
    int tower_timing_layers = 0;
    bool tower_layers_ready = false;
    bool up_in_tower = true;                // the up-layer is the first layer of the tower launch (DGNN_TOWER_UP=0: a launch of its own)
    int split_pref = 1;                     // DGNN_TOWER_SPLIT: 0 = never split a board inside the tower launch, 1 = where a workgroup
                                            // per board is not chosen, 2 = wherever a split is possible (A/B)
    bool split_failed = false;              // a hand-off poll gave up (not every workgroup was resident): per-layer launches for a while
    int split_failures = 0;                 // how often that has happened: after kSplitMaxFailures the split stays off
    int split_cooldown = 0;                 // passes still to run per layer before the split is tried again
    int inject_giveup = 0;                  // DGNN_SPLIT_INJECT_GIVEUP when the workspace was built (fault injection for the tests): 1 = the
                                            // launch finds the give-up word set; 2 = eight workgroups of the launch never start (the others
                                            // really wait, and really time out)
    bool injected = false;                  // ... and it has fired
    unsigned* d_split_sync = nullptr;       // [batch * kSplitWordsPerBoard] layers published per (board, channel block, row half), then the give-up word
    unsigned split_base = 0;                // running count the flags are compared against (launch_tower_split)
    unsigned* d_split_tickets = nullptr;    // eight words behind the flags: the queues the workgroups of a split launch draw their work items from (running counts)
    unsigned split_ticket_base = 0;         // ... their value before the next launch (the eight stay in step: grid / 8 tickets each per launch)
    unsigned* d_split_err = nullptr;        // the give-up word: the last four bytes of the value block
    unsigned* h_split_err = nullptr;        // pinned copy of that word: it arrives with the outputs (one copy for value, word and policy)
    bool split_pending = false;             // the submission in flight used tower_split_kernel
    int pending_form = 0;                   // ... and took this input form (what to repeat if it gave up)
    bool use_tile = true;                   // tuned tile kernel where the layer shape allows (DGNN_CONV=v1 disables)
    int tile_fm = 8;                        // position-tile height of the tile kernel for this batch size
    int tile_variant = 2;                   // 1 = lock-step, 2 = staggered load|matrix phases, 3 = staggered balanced half-steps
    BoardSwitches board_env;                // DGNN_BOARD_NS / DGNN_BOARD_SKIP / DGNN_CONV_ABLATE, read ONCE when the workspace is built

    ~Workspace();
};

struct DeviceState {
    int device = 0;
    int num_cus = 0;
    uint8_t* d_image = nullptr;
    bool image_valid = false;               // d_image holds net->image (false between a reserve and the copy that fills it)
    struct SplitChain* chain = nullptr;     // the device's chain of split-tower launches (process-global, see split_chain)
    std::atomic<int> boards_in_flight{0};   // boards of submissions between their submit and their wait, all workspaces of this device
    std::atomic<int> whole_in_flight{0};    // ... of those, the boards of submissions whose tower is one launch with a workgroup per board
    // passes enqueued on this device by the form their residual tower took, and the split towers that gave up (dgnn_network_tower_stats):
    // [0] one launch per layer, [1] a workgroup per board, [2] split tower, [3] give-ups
    std::atomic<unsigned long long> tower_passes[4] = {};
    std::map<int, std::vector<Workspace*>> idle;   // batch size -> pooled workspaces
    std::vector<Workspace*> all;                   // ownership

    ~DeviceState();
};

Workspace::~Workspace()
{
    (void)dgnn::set_device(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (hipEvent_t e : tower_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : stage_ev) if (e) (void)hipEventDestroy(e);
    if (ev_begin) (void)hipEventDestroy(ev_begin);
    if (ev_end) (void)hipEventDestroy(ev_end);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (arena) (void)hipFree(arena);
    if (d_tower_layers) (void)hipFree(d_tower_layers);
    if (d_tower_timing) (void)hipFree(d_tower_timing);
    if (pinned) (void)hipHostFree(pinned);
    if (stream) (void)hipStreamDestroy(stream);
}

DeviceState::~DeviceState()
{
    for (Workspace* ws : all) delete ws;
    (void)dgnn::set_device(device);
    if (d_image) (void)hipFree(d_image);
}

Network::~Network() { devices.clear(); }

SplitChain* split_chain(int device)
{
    static std::mutex table_mutex;
    static std::map<int, SplitChain*> table;     // never freed: a few bytes per device for the life of the process
    std::lock_guard<std::mutex> lock(table_mutex);
    SplitChain*& chain = table[device];
    if (!chain) {
        chain = new SplitChain();
        for (SplitChain::Launch& l : chain->recent)
            if (hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) l.done = nullptr;
        chain->done = chain->recent[0].done && chain->recent[1].done && chain->recent[2].done && chain->recent[3].done ? chain->recent[0].done : nullptr;
    }
    return chain;
}

namespace {

size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

// Lazily creates the per-device state: weight image upload (tensor.rs:123-141 done once per device).
int get_device_state(Network* net, int device, DeviceState** out, bool upload = true)
{
    auto it = net->devices.find(device);
    if (it != net->devices.end()) {
        DeviceState* state = it->second.get();
        if (upload && !state->image_valid) {          // reserved for a transfer that never filled it: never run on garbage
            DGNN_HIP(dgnn::set_device(device));
            DGNN_HIP(hipMemcpy(state->d_image, net->image.data(), net->image.size(), hipMemcpyHostToDevice));
            DGNN_HIP(hipDeviceSynchronize());         // (a copy from pageable memory may return with the DMA still on its way; the
                                                      // workspaces' streams are non-blocking and would not wait for it)
            state->image_valid = true;
        }
        *out = state;
        return DGNN_OK;
    }

    int count = 0;
    DGNN_HIP(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(device));

    std::unique_ptr<DeviceState> state(new (std::nothrow) DeviceState());
    if (!state) return DGNN_OUT_OF_MEMORY;
    state->device = device;
    DGNN_HIP(hipDeviceGetAttribute(&state->num_cus, hipDeviceAttributeMultiprocessorCount, physical_device(device)));
    state->chain = split_chain(physical_device(device));   // (logical devices of a test build share their GPU's chain)
    if (!state->chain->done) return DGNN_HIP_ERROR;
    DGNN_HIP(hipMalloc((void**)&state->d_image, net->image.size()));
    if (upload) {
        DGNN_HIP(hipMemcpy(state->d_image, net->image.data(), net->image.size(), hipMemcpyHostToDevice));
        DGNN_HIP(hipDeviceSynchronize());
        state->image_valid = true;
    }

    *out = state.get();
    net->devices[device] = std::move(state);
    net->frozen = true;
    return DGNN_OK;
}

// Kernel selection: the staggered tile kernel by default; DGNN_CONV=tile1 selects the lock-step tile kernel,
// DGNN_CONV=v1 forces the generic kernel, DGNN_CONV_FM=4|6|8
// overrides the position-tile height (both exist for A/B measurements and as a safety valve).
// DGNN_CONV=board forces the board kernel, DGNN_CONV=tile1|tile2|tile3|v1 rule it out; by default it is used whenever the
// channel count allows (a multiple of 64).
bool board_impl_from_env(int batch, int channels)
{
    if (!conv_board_supported(channels, channels)) return false;
    const char* impl = std::getenv("DGNN_CONV");
    if (impl && *impl) return std::strcmp(impl, "board") == 0;
    // Measured on MI355X (C = 256): the board kernel with 2 / 4 / 8 slices per workgroup beats the NHWC tile kernel at
    // every batch size (B = 8: 21 vs 37 us per residual convolution, B = 64: 28 vs 40, B = 128: 47 vs 56, B = 256: 84 vs 107)
    // and is the only tuned kernel for C that is not a multiple of 256.
    (void)batch;
    return true;
}

void conv_impl_from_env(int num_pos, int cout, bool* use_tile, int* fm, int* variant)
{
    const char* impl = std::getenv("DGNN_CONV");
    *use_tile = !(impl && std::strcmp(impl, "v1") == 0);
    *variant = 2;
    if (impl && std::strcmp(impl, "tile1") == 0) *variant = 1;
    if (impl && std::strcmp(impl, "tile3") == 0) *variant = 3;
    *fm = conv_tile_pick_fm(num_pos, cout >= 256 ? cout : 256);
    const char* f = ablation_env("DGNN_CONV_FM");
    if (f && (std::atoi(f) == 4 || std::atoi(f) == 6 || std::atoi(f) == 8)) *fm = std::atoi(f);
}

int create_workspace(Network* net, DeviceState* dev, int batch, Workspace** out)
{
    std::unique_ptr<Workspace> ws(new (std::nothrow) Workspace());
    if (!ws) return DGNN_OUT_OF_MEMORY;

    const Topology& t = net->topo;
    ws->net = net;
    ws->dev = dev;
    ws->device = dev->device;
    ws->batch = batch;
    ws->active = batch;

    conv_impl_from_env(batch * DGNN_NUM_POINTS, t.num_channels, &ws->use_tile, &ws->tile_fm, &ws->tile_variant);
    ws->use_board = board_impl_from_env(batch, t.num_channels) && t.up.w_board_offset != 0;
    {
        // one launch for the whole residual tower (tower_board_kernel) wherever a workgroup owns a whole board; measured on
        // MI355X, 20x256 batch 256: 85.3 us per convolution against 91.3 with one launch per layer.  DGNN_TOWER=0: per layer.
        const char* tower = std::getenv("DGNN_TOWER");
        ws->use_tower = ws->use_board && !(tower && std::atoi(tower) == 0) && t.num_blocks > 0;
        const char* up = ablation_env("DGNN_TOWER_UP");
        ws->up_in_tower = !(up && std::atoi(up) == 0);
        const char* split = std::getenv("DGNN_TOWER_SPLIT");
        if (split && *split) ws->split_pref = std::atoi(split);
        const char* inject = test_env("DGNN_SPLIT_INJECT_GIVEUP");
        ws->inject_giveup = inject ? std::atoi(inject) : 0;
    }

    ws->board_env = read_board_switches();
    if (ws->use_board) {
        if (t.head_w_board_offset == 0 && t.policy_conv.w_offset == 0) return DGNN_UNSUPPORTED;
    } else {
        // NHWC paths: the image must carry the generic layout (DGNN_CONV was different when the network was built)
        if (t.up.w_offset == 0 || t.policy_conv.w_offset == 0) return DGNN_UNSUPPORTED;
    }

    DGNN_HIP(dgnn::set_device(dev->device));
    DGNN_HIP(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));   // stream.rs:55
    DGNN_HIP(hipEventCreate(&ws->ev_begin));
    DGNN_HIP(hipEventCreate(&ws->ev_end));
    DGNN_HIP(hipEventCreateWithFlags(&ws->ev_done, hipEventBlockingSync | hipEventDisableTiming));

    const size_t num_pos = (size_t)batch * DGNN_NUM_POINTS;
    const size_t sz_features = align_up(num_pos * DGNN_NUM_FEATURES * 2, 256);
    // board path: 400 padded rows per board, + 1 KiB of slack (the last DMA piece of a k-slot reads rows 384..447)
    const size_t board_pos = (size_t)batch * kBoardRows;
    const size_t sz_act = ws->use_board ? align_up(board_pos * t.num_channels * 2 + 1024, 256)
                                        : align_up(num_pos * t.num_channels * 2, 256);
    const size_t sz_fboard = ws->use_board ? align_up(board_pos * DGNN_NUM_FEATURES * 2 + 1024, 256) : 0;
    const size_t sz_nhwc = ws->use_board ? align_up(num_pos * t.num_channels * 2, 256) : 0;
    const size_t sz_pf = align_up(num_pos * t.num_samples * 2, 256);
    const size_t sz_vf = align_up(num_pos * 2 * 2, 256);
    const size_t sz_value = align_up((size_t)batch * 2 + sizeof(unsigned), 256);   // + the split tower's give-up word in its last four bytes:
                                                                                   // it travels with the outputs (enqueue_outputs)
    const size_t sz_policy = align_up((size_t)batch * DGNN_NUM_MOVES * 2, 256);
    const size_t sz_logits = align_up((size_t)batch * 384 * 2, 256);
    const size_t sz_packed = align_up((size_t)batch * DGNN_PACKED_WORDS * 4, 256);
    const size_t sz_split = align_up((size_t)batch * kSplitWordsPerBoard * sizeof(unsigned) + kSplitQueues * sizeof(unsigned), 256);   // + the ticket counters
    const size_t sz_records = align_up((size_t)batch * DGNN_BOARD_RECORD_BYTES, 256);
    const size_t sz_vpre = align_up((size_t)batch * 2, 256);
    const size_t total = sz_features + 3 * sz_act + sz_pf + sz_vf + sz_value + sz_policy + sz_fboard + sz_nhwc + sz_logits + sz_packed + sz_split + sz_records + sz_vpre;

    DGNN_HIP(hipMalloc(&ws->arena, total));
    // board path: pad rows 384..399 of every k-slot are never written by the kernels and must read as zero
    if (ws->use_board) {
        DGNN_HIP(hipMemset(ws->arena, 0, total));
        // hipMemset is asynchronous to the host and runs on the NULL stream; the workspace's stream is non-blocking and does not wait
        // for it.  Without this the first pass could run beside (or ahead of) its own arena being zeroed: found in round 4 by two
        // processes starting on one GPU at the same moment -- the first forward of one of them differed from all its later ones
        // (tools/diag_two_procs.py: 1 438 policy entries).
        DGNN_HIP(hipDeviceSynchronize());
    }
    uint8_t* p = (uint8_t*)ws->arena;
    ws->d_features = (uint16_t*)p; p += sz_features;
    for (int i = 0; i < 3; ++i) { ws->d_act[i] = (uint16_t*)p; p += sz_act; }
    ws->d_policy_features = (uint16_t*)p; p += sz_pf;
    ws->d_value_features = (uint16_t*)p; p += sz_vf;
    ws->d_value = (uint16_t*)p; p += sz_value;
    ws->d_split_err = (unsigned*)(p - sizeof(unsigned));
    ws->d_policy = (uint16_t*)p; p += sz_policy;
    ws->d_logits = (uint16_t*)p; p += sz_logits;
    ws->d_packed = (uint32_t*)p; p += sz_packed;
    ws->d_split_sync = (unsigned*)p; p += sz_split;
    ws->d_split_tickets = ws->d_split_sync + (size_t)batch * kSplitWordsPerBoard;
    // (tests) the running counters of the split towers started where a job finds them after a long time: "flags,tickets" -- the flag base a
    // few launches below its refill at 0xf0000000 (two days of launches), the 32-bit ticket counts a few launches below their wrap
    if (const char* start = test_env("DGNN_SPLIT_COUNTERS_START")) {
        unsigned flags_at = 0, arrivals_at = 0;
        if (std::sscanf(start, "%u,%u", &flags_at, &arrivals_at) == 2) {
            ws->split_base = flags_at;
            ws->split_ticket_base = arrivals_at;
            const unsigned all[kSplitQueues] = {arrivals_at, arrivals_at, arrivals_at, arrivals_at, arrivals_at, arrivals_at, arrivals_at, arrivals_at};
            DGNN_HIP(hipMemcpy(ws->d_split_tickets, all, sizeof(all), hipMemcpyHostToDevice));
        }
    }
    ws->d_records = p; p += sz_records;
    ws->d_value_pre = (uint16_t*)p; p += sz_vpre;
    if (ws->use_board) {
        ws->d_features_board = (uint16_t*)p; p += sz_fboard;
        ws->d_tower_nhwc = (uint16_t*)p;
    }

    DGNN_HIP(hipHostMalloc(&ws->pinned, sz_features + sz_packed + sz_records + sz_value + sz_policy + 256, hipHostMallocCoherent));   // (kernel-written, host-read behind ev_done: fine-grained, spelled out)
    uint8_t* h = (uint8_t*)ws->pinned;
    ws->h_features = (uint16_t*)h; h += sz_features;
    ws->h_packed = (uint32_t*)h; h += sz_packed;
    ws->h_records = h; h += sz_records;
    ws->h_value = (uint16_t*)h; h += sz_value;
    ws->h_split_err = (unsigned*)(h - sizeof(unsigned));
    ws->h_policy = (uint16_t*)h; h += sz_policy;
    *ws->h_split_err = 0;

    *out = ws.release();
    return DGNN_OK;
}

// The tower as one launch: where a workgroup owns a whole board, and only with gates the in-loop skip can express
// (it accumulates (alpha2 / alpha1) * z: a gate of 0 needs the epilogue form, which only the per-layer launcher has).
// Two forms: a workgroup per board (tower_board_kernel), or -- batches too small for that to fill the chip -- a board split
// over 2 / 4 / 8 workgroups that hand their channel blocks to each other inside the launch (tower_split_kernel).
// Returns 0 = one launch per layer, -1 = a workgroup per board, n > 0 = split tower with n slices per workgroup.
extern const bool kSplitSizedForTheCrowd;
int tower_mode(const Workspace* ws)
{
    if (!(ws->use_board && ws->use_tower && ws->net->topo.gates_nonzero)) return 0;
    const int channels = ws->net->topo.num_channels;
    const bool whole = tower_board_supported(ws->active, channels, ws->crowded);
    if (ws->crowded_per_layer && !whole && ws->split_pref == 1) return 0;
    if (ws->split_pref != 0 && !ws->split_failed && (!whole || ws->split_pref == 2)) {
        // The split is sized for the boards that will SHARE the chip, not for this submission alone: tower_split_ns spreads a lone
        // batch over as many workgroups as fill the chip (64 boards of 256 channels: four per board, the fabric-bound two-slice split);
        // beside other split launches (SplitChain) fewer, larger workgroups per board do the same work with fewer bytes, and the
        // launches together fill the chip (round 6; capped at half the CUs in boards: the sizing of a 128-board batch).
        if (kSplitSizedForTheCrowd && ws->small_others > 0) {
            const int crowd = std::min(ws->active + ws->small_others, std::max(ws->active, ws->dev->num_cus / 2));
            const int ns = tower_split_ns(crowd, channels, ws->dev->num_cus);
            if (ns > 0 && ns != 32) return ns;
        }
        const int ns = tower_split_ns(ws->active, channels, ws->dev->num_cus);
        if (ns > 0) return ns;
    }
    return whole ? -1 : 0;
}

bool tower_selected(const Workspace* ws) { return tower_mode(ws) != 0; }

// Workgroups of split launches that may be waiting for a sibling at the same time (SplitChain): half the compute units; DGNN_SPLIT_BESIDE=0
// (test / ablations build): every split launch behind the previous one, as in rounds 2-5 (A/B).
int kSplitWaitersBudget(int num_cus)
{
    static const bool beside = [] { const char* e = test_env("DGNN_SPLIT_BESIDE"); return !(e && std::atoi(e) == 0); }();
    return beside ? num_cus / 2 : 0;
}

// Split towers sized for the boards that share the chip (tower_mode); DGNN_SPLIT_CROWD=0 (test / ablations build): for the submission alone (A/B).
const bool kSplitSizedForTheCrowd = [] { const char* e = test_env("DGNN_SPLIT_CROWD"); return !(e && std::atoi(e) == 0); }();

// Whole-board towers for small batches on a crowded device (round 5); DGNN_CROWDED=0 (test / ablations build): the size alone decides (A/B).
const bool kCrowdedTowers = [] { const char* e = test_env("DGNN_CROWDED"); return !(e && std::atoi(e) == 0); }();

// Enqueues graph.rs:134-152 on the workspace's stream.  `tower_pair` (may be -1) selects the event pair that
// brackets the residual convolutions.
enum InputForm { kInputFeatures = 0, kInputPacked = 1, kInputRecords = 2 };   // d_features / d_packed / d_records hold the batch

int enqueue_forward(Workspace* ws, int tower_pair, int input_form = kInputFeatures)
{
    const bool packed_input = input_form == kInputPacked;
    const Topology& t = ws->net->topo;
    const uint8_t* image = ws->dev->d_image;
    const int num_pos = ws->active * DGNN_NUM_POINTS;

    auto conv = [&](const ConvParams& c, const uint16_t* x, const uint16_t* z, uint16_t* y) -> hipError_t {
        if (ws->use_tile && c.w_tile_offset != 0 && conv_tile_supported(c.cin, c.cout)) {
            ConvTileLaunch a;
            a.x = x;
            a.w = (const uint16_t*)(image + c.w_tile_offset);
            a.bias = (const uint16_t*)(image + c.b_offset);
            a.z = c.alpha2 != 0.0f ? z : nullptr;
            a.y = y;
            a.zero_page = (const uint16_t*)image;
            a.timing = nullptr;
            a.num_pos = num_pos;
            a.board = 19;
            a.cin = c.cin;
            a.cout = c.cout;
            a.alpha1 = c.alpha1;
            a.alpha2 = c.alpha2;
            a.relu = 1;
            return launch_conv3x3_tile(a, ws->tile_fm, ws->tile_variant, ws->stream);
        }
        ConvLaunch a;
        a.x = x;
        a.w = (const uint16_t*)(image + c.w_offset);
        a.bias = (const uint16_t*)(image + c.b_offset);
        a.z = c.alpha2 != 0.0f ? z : nullptr;
        a.y = y;
        a.num_pos = num_pos;
        a.board = 19;   // conv2d.rs:50
        a.cin = c.cin;
        a.cout = c.cout;
        a.cout_pad = c.cout_pad;
        a.alpha1 = c.alpha1;
        a.alpha2 = c.alpha2;
        a.relu = 1;   // every 3x3 convolution of the network uses ReLU (conv2d.rs:119-125, value_head.rs:45)
        return launch_conv3x3(a, ws->stream);
    };

    auto conv_board = [&](const ConvParams& c, const uint16_t* x, const uint16_t* z, uint16_t* y) -> hipError_t {
        ConvBoardLaunch a;
        a.x = x;
        a.w = (const uint16_t*)(image + c.w_board_offset);
        a.bias = (const uint16_t*)(image + c.b_offset);
        a.z = c.alpha2 != 0.0f ? z : nullptr;
        a.y = y;
        a.timing = nullptr;
        a.w_next = nullptr;
        a.carry_in = 0;
        a.batch = ws->active;
        a.cin = c.cin;
        a.cout = c.cout;
        a.alpha1 = c.alpha1;
        a.alpha2 = c.alpha2;
        a.relu = 1;
        return launch_conv3x3_board(a, ws->board_env, ws->stream);
    };

    int cur = 0;
    const uint16_t* tower = nullptr;   // NHWC tower output for the heads
    auto stamp = [&](int stage) -> hipError_t { return ws->stage_on ? hipEventRecord(ws->stage_ev[stage], ws->stream) : hipSuccess; };
    // compact board records: the planes are built on the device (features.hip), straight into the up-layer's input on the board path
    if (input_form == kInputRecords) DGNN_HIP(launch_board_records(ws->host_io ? ws->h_records : ws->d_records, ws->d_packed, ws->use_board ? ws->d_features_board : nullptr, ws->active, ws->stream));
    DGNN_HIP(stamp(2));
    if ((packed_input || input_form == kInputRecords) && !ws->use_board) DGNN_HIP(launch_packed_features(ws->d_packed, ws->d_features, ws->active, false, ws->stream));
    if (ws->use_board) {
        // features.rs:52-60 HWC fp16, or the packed form: either way the up-layer reads the padded slot-major layout
        if (input_form == kInputRecords) {}
        else if (packed_input) DGNN_HIP(launch_packed_features(ws->d_packed, ws->d_features_board, ws->active, true, ws->stream));
        else DGNN_HIP(launch_nhwc_to_board(ws->d_features, ws->d_features_board, ws->active, DGNN_NUM_FEATURES, ws->stream));
        // the up-layer (C_in = 32: one chunk) is the tower launch's first layer where the tower is one launch: its output
        // channels 0..31 stay in LDS for the first residual convolution like any other layer's (DGNN_TOWER_UP=0: own launch)
        const bool up_in_tower = tower_selected(ws) && ws->up_in_tower;
        if (!up_in_tower) DGNN_HIP(conv_board(t.up, ws->d_features_board, nullptr, ws->d_act[cur]));
        if (tower_pair >= 0) DGNN_HIP(hipEventRecord(ws->tower_events[2 * tower_pair], ws->stream));
        if (tower_selected(ws)) {
            if (!ws->tower_layers_ready) {                     // the descriptors only hold addresses fixed for the workspace's life
                std::vector<ConvBoardLaunch> layers;
                int c = cur;
                auto describe = [&](const ConvParams& p, const uint16_t* x, const uint16_t* z, uint16_t* y) {
                    ConvBoardLaunch a;
                    a.x = x; a.w = (const uint16_t*)(image + p.w_board_offset); a.bias = (const uint16_t*)(image + p.b_offset);
                    a.z = p.alpha2 != 0.0f ? z : nullptr; a.y = y; a.timing = nullptr; a.batch = ws->batch; a.cin = p.cin; a.cout = p.cout;
                    a.alpha1 = p.alpha1; a.alpha2 = p.alpha2; a.relu = 1;
                    a.w_next = nullptr; a.carry_in = 0;
                    layers.push_back(a);
                };
                if (up_in_tower) describe(t.up, ws->d_features_board, nullptr, ws->d_act[c]);
                for (int i = 0; i < t.num_blocks; ++i) {
                    const int tmp = (c + 1) % 3, next = (c + 2) % 3;
                    describe(t.res1[i], ws->d_act[c], nullptr, ws->d_act[tmp]);
                    describe(t.res2[i], ws->d_act[tmp], ws->d_act[c], ws->d_act[next]);
                    c = next;
                }
                // measurement only (tools/tower_trace.py): DGNN_TOWER_TIMING=<file> gives every layer of a whole-board tower its own
                // block of s_memtime stamps [workgroup][wave][start, K loop, epilogue, end]; dumped after every launch
                static const char* const timing_path = ablation_env("DGNN_TOWER_TIMING");
                if (timing_path && *timing_path && tower_mode(ws) < 0) {
                    const size_t words = (size_t)layers.size() * ws->batch * 8 * 4;
                    if (!ws->d_tower_timing) DGNN_HIP(hipMalloc((void**)&ws->d_tower_timing, words * 8));
                    DGNN_HIP(hipMemset(ws->d_tower_timing, 0, words * 8));
                    for (size_t l = 0; l < layers.size(); ++l) layers[l].timing = ws->d_tower_timing + l * ws->batch * 8 * 4;
                    ws->tower_timing_layers = (int)layers.size();
                }
                const char* carry = ablation_env("DGNN_TOWER_CARRY");          // =0: no hand-over between the layers (A/B)
                if (!(carry && std::atoi(carry) == 0))
                    for (size_t l = 0; l + 1 < layers.size(); ++l) { layers[l].w_next = layers[l + 1].w; layers[l + 1].carry_in = 1; }
                if (!ws->d_tower_layers) DGNN_HIP(hipMalloc((void**)&ws->d_tower_layers, (layers.size() + 1) * sizeof(ConvBoardLaunch)));
                DGNN_HIP(hipMemcpy(ws->d_tower_layers, layers.data(), layers.size() * sizeof(ConvBoardLaunch), hipMemcpyHostToDevice));
                DGNN_HIP(hipDeviceSynchronize());      // (pageable source, NULL stream: the launch below is on a non-blocking stream)
                ws->tower_layers_ready = true;
            }
            const int layer_count = 2 * t.num_blocks + (up_in_tower ? 1 : 0);
            const int mode = tower_mode(ws);
            ws->dev->tower_passes[mode > 0 ? 2 : 1].fetch_add(1, std::memory_order_relaxed);
            if (mode > 0) {
                const int groups = mode == 32 ? t.num_channels / 32 * 2 : t.num_channels / 32 / mode;
                SplitChain* chain = ws->dev->chain;
                std::lock_guard<std::mutex> chained(chain->mutex);
                // newest to oldest: launches this one may run beside while the possible waiters stay within the budget; from the first
                // one that does not fit on, it waits (for that one and everything older)
                const int waiters = 8 * (groups - 1);
                int beside = waiters;
                bool wait_rest = false;
                for (int k = 1; k <= 4; ++k) {
                    SplitChain::Launch& l = chain->recent[(chain->next + 4 - k) & 3];
                    if (!l.live) continue;
                    if (!wait_rest && hipEventQuery(l.done) == hipSuccess) { l.live = false; continue; }
                    if (!wait_rest && l.workgroups < ws->dev->num_cus && beside + l.waiters <= kSplitWaitersBudget(ws->dev->num_cus)) { beside += l.waiters; continue; }
                    wait_rest = true;
                    DGNN_HIP(hipStreamWaitEvent(ws->stream, l.done, 0));
                }
                SplitChain::Launch& mine = chain->recent[chain->next];       // the oldest slot: waited for above if it was still live and did not fit...
                if (mine.live && !wait_rest) DGNN_HIP(hipStreamWaitEvent(ws->stream, mine.done, 0));   // ... or now: its event is about to be re-recorded
                chain->next = (chain->next + 1) & 3;
                bool missing = false;
                if (ws->inject_giveup != 0 && !ws->injected) {  // fault injection (tests)
                    ws->injected = true;
                    if (ws->inject_giveup == 1) DGNN_HIP(hipMemsetAsync(ws->d_split_err, 0xff, sizeof(unsigned), ws->stream));
                    else missing = true;
                }
                DGNN_HIP(launch_tower_split(ws->d_tower_layers, layer_count, ws->active, t.num_channels, mode, ws->d_split_sync,
                                            (size_t)ws->batch * kSplitWordsPerBoard, ws->d_split_err, &ws->split_base, ws->d_split_tickets,
                                            &ws->split_ticket_base, ws->stream, missing));
                DGNN_HIP(hipEventRecord(mine.done, ws->stream));
                mine.waiters = waiters;
                mine.workgroups = ws->active * groups;
                mine.live = true;
            } else {
                DGNN_HIP(launch_tower_board(ws->d_tower_layers, layer_count, ws->active, t.num_channels, ws->stream));
                if (ws->d_tower_timing) {
                    DGNN_HIP(hipStreamSynchronize(ws->stream));
                    const size_t words = (size_t)ws->tower_timing_layers * ws->batch * 8 * 4;
                    std::vector<unsigned long long> h(words);
                    DGNN_HIP(hipMemcpy(h.data(), ws->d_tower_timing, words * 8, hipMemcpyDeviceToHost));
                    if (FILE* f = std::fopen(ablation_env("DGNN_TOWER_TIMING"), "wb")) {
                        const int header[4] = {ws->tower_timing_layers, ws->batch, 8, 4};
                        std::fwrite(header, sizeof(int), 4, f);
                        std::fwrite(h.data(), 8, words, f);
                        std::fclose(f);
                    }
                }
            }
            for (int i = 0; i < t.num_blocks; ++i) cur = (cur + 2) % 3;
        } else {
            ws->dev->tower_passes[0].fetch_add(1, std::memory_order_relaxed);
            for (int i = 0; i < t.num_blocks; ++i) {
                const int tmp = (cur + 1) % 3, next = (cur + 2) % 3;
                DGNN_HIP(conv_board(t.res1[i], ws->d_act[cur], nullptr, ws->d_act[tmp]));
                DGNN_HIP(conv_board(t.res2[i], ws->d_act[tmp], ws->d_act[cur], ws->d_act[next]));
                cur = next;
            }
        }
        if (tower_pair >= 0) DGNN_HIP(hipEventRecord(ws->tower_events[2 * tower_pair + 1], ws->stream));
        DGNN_HIP(stamp(3));
        if (t.head_w_board_offset != 0) {
            HeadConvLaunch hc;                                                      // value_head.rs:76 + policy_head.rs:92
            hc.x = ws->d_act[cur];
            hc.w = (const uint16_t*)(image + t.head_w_board_offset);
            hc.bias = (const uint16_t*)(image + t.head_b_offset);
            hc.policy_features = ws->d_policy_features;
            hc.value_features = ws->d_value_features;
            hc.batch = ws->active;
            hc.cin = t.num_channels;
            hc.samples = t.num_samples;
            DGNN_HIP(launch_head_conv_board(hc, ws->stream));
        } else {
            DGNN_HIP(launch_board_to_nhwc(ws->d_act[cur], ws->d_tower_nhwc, ws->active, t.num_channels, ws->stream));
            tower = ws->d_tower_nhwc;
        }
    } else {
    DGNN_HIP(conv(t.up, ws->d_features, nullptr, ws->d_act[cur]));                 // up_block.rs:56

    if (tower_pair >= 0) DGNN_HIP(hipEventRecord(ws->tower_events[2 * tower_pair], ws->stream));
    for (int i = 0; i < t.num_blocks; ++i) {                                        // graph.rs:139-143
        const int tmp = (cur + 1) % 3, next = (cur + 2) % 3;
        DGNN_HIP(conv(t.res1[i], ws->d_act[cur], nullptr, ws->d_act[tmp]));        // residual_block.rs:76
        DGNN_HIP(conv(t.res2[i], ws->d_act[tmp], ws->d_act[cur], ws->d_act[next])); // residual_block.rs:77
        cur = next;
    }
    if (tower_pair >= 0) DGNN_HIP(hipEventRecord(ws->tower_events[2 * tower_pair + 1], ws->stream));
    tower = ws->d_act[cur];
    DGNN_HIP(stamp(3));
    }

    if (tower != nullptr) {
        DGNN_HIP(conv(t.value_conv, tower, nullptr, ws->d_value_features));        // value_head.rs:76
        DGNN_HIP(conv(t.policy_conv, tower, nullptr, ws->d_policy_features));      // policy_head.rs:92
    }

    DGNN_HIP(stamp(4));
    PolicyGemmLaunch pg;                                                           // policy_head.rs:95
    pg.features = ws->d_policy_features;
    pg.wt = (const uint16_t*)(image + t.policy_wt_offset);
    pg.bias = (const uint16_t*)(image + t.policy_b_offset);
    pg.logits = ws->d_logits;
    pg.batch = ws->active;
    pg.k_in = DGNN_NUM_POINTS * t.num_samples;
    pg.k_pad = t.policy_k_pad;
    pg.tau = 1.0f / t.softmax_temperature;
    DGNN_HIP(launch_policy_gemm(pg, ws->stream));
    DGNN_HIP(stamp(5));

    SoftmaxValueLaunch sv;                                                         // policy_head.rs:98-100, value_head.rs:80-81
    sv.logits = ws->d_logits;
    sv.value_features = ws->d_value_features;
    sv.value_w = (const uint16_t*)(image + t.value_w_offset);
    sv.value_b = (const uint16_t*)(image + t.value_b_offset);
    sv.policy_out = ws->host_io ? ws->h_policy : ws->d_policy;
    sv.value_out = ws->host_io ? ws->h_value : ws->d_value;
    ws->outputs_on_host = ws->host_io;      // where the outputs of the pass being enqueued land (dgnn_workspace_download)
    sv.value_pre_out = ws->d_value_pre;
    sv.batch = ws->active;
    DGNN_HIP(launch_softmax_value(sv, ws->stream));
    DGNN_HIP(stamp(6));
    return DGNN_OK;
}

// How a host thread waits for a submission.  hipStreamSynchronize SPINS on this runtime: one thread waiting for 3.3 ms passes
// burns 1.8 cores, and so does hipEventSynchronize on an event created with hipEventBlockingSync (1.0 core; tools/wait_probe.sh) --
// a leaf pool keeps two such waiters per device, more host CPU than the whole search.  So the waiter SLEEPS: for most of the time
// the last submissions of this size took, then in 30 us naps between hipEventQuery calls (0.03 cores per waiter; the late wake-up
// is hidden by the second batch in flight).  DGNN_WAIT=spin restores hipStreamSynchronize, DGNN_WAIT=event the blocking event.
// Zero copy (round 5): a submission's 896-byte records are read by the planes kernel straight from the pinned host buffer, and the
// soft-max kernel writes value and policy straight into the pinned host buffers -- 0.2 MB each way per 256 boards across PCIe by the
// kernels' own loads and stores instead of two blit kernels (22 + 19 us of the ~120 us a device batch spends outside its tower,
// profiles/r05/stage_times_20x256.jsonl), each of which also had to wait for a compute unit the other batch's tower holds.
// DGNN_ZEROCOPY=0 restores the copies (A/B).
const bool kZeroCopy = [] { const char* e = ablation_env("DGNN_ZEROCOPY"); return !(e && std::atoi(e) == 0); }();

enum WaitMode { kWaitPoll = 0, kWaitSpin = 1, kWaitEvent = 2 };
WaitMode wait_mode_from_env()
{
    const char* m = ablation_env("DGNN_WAIT");
    if (m && std::strcmp(m, "spin") == 0) return kWaitSpin;
    if (m && std::strcmp(m, "event") == 0) return kWaitEvent;
    return kWaitPoll;
}
const WaitMode kWaitMode = wait_mode_from_env();

long long now_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

void nap_ns(long long ns)
{
    if (ns <= 0) return;
    struct timespec ts = {(time_t)(ns / 1000000000ll), (long)(ns % 1000000000ll)};
    nanosleep(&ts, nullptr);
}

// A small batch alone on the device is best split over many workgroups (latency); the same batch beside other submissions is best run
// with a workgroup per board: the CUs it leaves are taken by the others' workgroups, and a whole-board workgroup moves the fewest bytes
// per board.  From 11 / 16 of the CU count in boards in flight on the device (this submission included; 176 on an MI355X) on, a workgroup per
// board: below that the split's extra workgroups find free CUs of their own (2 x 64 boards run better split, 3 x 64 the same either way).
// Up to 128 channels only: there a CU holds two boards' workgroups, whole-board launches of different submissions overlap, and the
// POOL gains what the sweep promised (9x128, 16 / 32 games in flight: + 11 / + 9 % moves per second).  At 256 channels a CU holds one;
// a small batch behind a full one gets the chip sooner as a split launch -- the CUs of the batch in front come free together -- and the
// pool LOSES 2.4 % at 48 games in flight although three equal submissions at once (the sweep) gain 14 %
// (profiles/r05_final/crowded_in_the_pool_*.txt).
//
// At MORE than 128 channels (round 6): a batch too small for a workgroup per board that finds at least a whole-board tower's worth of
// boards (129) of other submissions' WHOLE-BOARD towers on the device runs ONE LAUNCH PER LAYER, not a split tower.  Alone on the device the split tower
// is the fast form (latency: 25-30 % ahead of 41 launches); beside whole-board towers its workgroups wait for siblings
// that wait for compute units (they hold theirs meanwhile), while the workgroups of a per-layer launch wait for nothing and take CUs
// as the towers release them.  Measured in the pool, 20x256, 48 games in flight (13 % of the device batches below 129 boards; same
// box, interleaved, two runs each; profiles/r06_final/few_games_in_the_pool_split_alive.txt): split towers 79.6-79.7 k boards/s (80.5-80.7 k
// without the ordering that round 6 tried), per layer 82.0-82.2 k.  (Round 5's pool ran per layer too -- by accident: its split towers
// gave up behind the first whole-board tower, VERDICT r05 weak #2.  Now nothing gives up, and the form is chosen.)  DGNN_TOWER_SPLIT=2
// keeps the split tower wherever one is possible (tests of the split tower in a crowd; A/B).
const bool kCrowdedPerLayer = [] { const char* e = test_env("DGNN_CROWDED_PER_LAYER"); return !(e && std::atoi(e) == 0); }();
void note_submission_begins(Workspace* ws)       // in front of enqueue_forward: which tower this submission gets
{
    const int others = ws->dev->boards_in_flight.load(std::memory_order_relaxed);
    ws->crowded = kCrowdedTowers && ws->net->topo.num_channels <= 128 && others > 0 && others + ws->active >= ws->dev->num_cus * 11 / 16;
    ws->small_others = std::max(0, others - ws->dev->whole_in_flight.load(std::memory_order_relaxed));
    // (the boards of WHOLE-BOARD towers: small submissions among themselves are split launches that share the chip -- SplitChain)
    ws->crowded_per_layer = kCrowdedPerLayer && ws->net->topo.num_channels > 128 &&
                            ws->dev->whole_in_flight.load(std::memory_order_relaxed) >= ws->dev->num_cus / 2 + 1;
}
void note_submission_in_flight(Workspace* ws)    // with `pending = true`
{
    ws->counted_in_flight = ws->active;
    ws->dev->boards_in_flight.fetch_add(ws->active, std::memory_order_relaxed);
    ws->counted_whole = tower_mode(ws) < 0 ? ws->active : 0;
    if (ws->counted_whole > 0) ws->dev->whole_in_flight.fetch_add(ws->counted_whole, std::memory_order_relaxed);
}
struct SubmissionBegins {                        // a submit that fails half way leaves no tower choice behind for the resident entry points
    Workspace* ws;
    explicit SubmissionBegins(Workspace* w) : ws(w) { note_submission_begins(w); }
    ~SubmissionBegins() { if (!ws->pending) { ws->crowded = ws->crowded_per_layer = false; ws->small_others = 0; } }
};
void note_submission_ends(Workspace* ws)
{
    if (ws->counted_in_flight > 0) ws->dev->boards_in_flight.fetch_sub(ws->counted_in_flight, std::memory_order_relaxed);
    if (ws->counted_whole > 0) ws->dev->whole_in_flight.fetch_sub(ws->counted_whole, std::memory_order_relaxed);
    ws->counted_in_flight = ws->counted_whole = 0;
    ws->crowded = ws->crowded_per_layer = false;
    ws->small_others = 0;
}

int mark_submission(Workspace* ws)      // behind the last copy of a submission
{
    if (kWaitMode != kWaitSpin) DGNN_HIP(hipEventRecord(ws->ev_done, ws->stream));
    ws->submit_ns = now_ns();
    return DGNN_OK;
}

// value[active] and policy[active][362] to the pinned host buffers.  The two lie behind each other on both sides, so they travel as
// ONE copy (the value block whole, then the active policy rows): a device-to-host copy of this size is a blit kernel of ~5.6 us, and
// two of them were 5 % of a 9x128 batch-32 pass.
hipError_t enqueue_outputs(Workspace* ws)
{
    if (ws->host_io) {
        // the soft-max kernel wrote value and policy into the pinned host buffers itself; only a split tower's give-up word is
        // still on the device (its workgroups poll it: it cannot live across PCIe)
        if (tower_mode(ws) > 0) return hipMemcpyAsync(ws->h_split_err, ws->d_split_err, sizeof(unsigned), hipMemcpyDeviceToHost, ws->stream);
        return hipSuccess;
    }
    const size_t value_block = (size_t)((const uint8_t*)ws->d_policy - (const uint8_t*)ws->d_value);
    if ((size_t)((const uint8_t*)ws->h_policy - (const uint8_t*)ws->h_value) != value_block) return hipErrorInvalidValue;
    return hipMemcpyAsync(ws->h_value, ws->d_value, value_block + (size_t)ws->active * DGNN_NUM_MOVES * 2, hipMemcpyDeviceToHost, ws->stream);
}

int wait_submission(Workspace* ws)
{
    if (kWaitMode == kWaitSpin) { DGNN_HIP(hipStreamSynchronize(ws->stream)); return DGNN_OK; }
    if (kWaitMode == kWaitEvent) { DGNN_HIP(hipEventSynchronize(ws->ev_done)); return DGNN_OK; }
    // the long sleep first: 85 % of what is left of the expected duration, scaled down for a smaller batch, at most 50 ms
    bool napped = false;
    if (ws->wait_boards > 0 && ws->wait_ns > 0.0) {
        double expected = ws->wait_ns * (ws->active < ws->wait_boards ? (double)ws->active / ws->wait_boards : 1.0);
        const long long left = (long long)(0.85 * expected) - (now_ns() - ws->submit_ns);
        if (left > 0) { nap_ns(left < 50000000ll ? left : 50000000ll); napped = true; }
    }
    int polls = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(ws->ev_done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { set_last_hip_error((int)q); return DGNN_HIP_ERROR; }
        nap_ns(30000);
        ++polls;
    }
    const double took = (double)(now_ns() - ws->submit_ns);
    const double scaled = ws->wait_ns * (ws->wait_boards > 0 && ws->active < ws->wait_boards ? (double)ws->active / ws->wait_boards : 1.0);
    if (napped && polls == 0) {
        // done before the long sleep ended: `took` is the sleep, not the pass.  Averaging it in would keep an over-estimate alive for
        // dozens of passes (a pass can never look shorter than 85 % of the estimate): forget it, the next passes are polled from
        // their start and measure themselves.
        ws->wait_ns = 0.0;
        ws->wait_last = 0.0;
    } else if (ws->wait_ns <= 0.0 || took > 2.0 * scaled) {
        // no estimate yet, or a pass more than twice as long as expected: ONE such sample sends nobody to sleep -- the first pass of
        // a workspace carries kernel loads and the first touch of its staging buffers (60 ms seen), and a synchronous caller then
        // slept 85 % of that (a batch sweep showed 14 ms per 3.4 ms pass).  Two in a row are believed, at the shorter of the two.
        if (ws->wait_last > 0.0) {
            ws->wait_ns = took < ws->wait_last ? took : ws->wait_last;
            ws->wait_boards = ws->active;
            ws->wait_last = 0.0;
        } else {
            ws->wait_last = took;
        }
    } else {
        ws->wait_ns = ws->wait_boards == ws->active ? 0.75 * ws->wait_ns + 0.25 * took : took;
        ws->wait_boards = ws->active;
        ws->wait_last = 0.0;
    }
    return DGNN_OK;
}

// After enqueue_forward: fetch the split tower's give-up word with the outputs.
int enqueue_split_check(Workspace* ws, int input_form)
{
    const int mode = tower_mode(ws);
    ws->split_pending = mode > 0;
    ws->pending_form = input_form;
    return DGNN_OK;                         // (the word itself arrives with the outputs: enqueue_outputs)
}

// After the stream has drained: a split tower in which a hand-off was not published within 50 ms (wait_published, the net under the
// tickets of conv_board_tower.hip: a sibling never got a compute unit -- something that does not end holds the device -- or the launch
// lost workgroups) gives up instead of hanging; its outputs are garbage.  The pass is repeated with one launch per layer, the next
// kSplitCooldown passes of this workspace run per layer too, then the split is tried again; after kSplitMaxFailures give-ups it stays
// off until dgnn_network_synchronize re-arms it.  Towers of OTHER workspaces of this process in front of a split launch are not such
// a case (round 5 treated them as one and lost the split inside the leaf pool): its workgroups draw their work in the order in which
// they start, and wait for each other for as long as those towers run -- milliseconds.
constexpr int kSplitCooldown = 64, kSplitMaxFailures = 3;

int note_split_failure(Workspace* ws)
{
    *ws->h_split_err = 0;
    ws->split_failed = true;
    ++ws->split_failures;
    ws->dev->tower_passes[3].fetch_add(1, std::memory_order_relaxed);
    ws->split_cooldown = kSplitCooldown;
    std::fprintf(stderr, "dgnn: tower_split_kernel gave up: a hand-off between its workgroups took more than 50 ms on device %d (a workgroup of the launch "
                         "never ran); the pass is repeated with one launch per layer, and so are the next passes %s\n", ws->device,
                 ws->split_failures >= kSplitMaxFailures ? "until dgnn_network_synchronize" : "of this workspace for a while");
    DGNN_HIP(hipMemsetAsync(ws->d_split_err, 0, sizeof(unsigned), ws->stream));   // the word is sticky on the device
    DGNN_HIP(hipMemsetAsync(ws->d_split_tickets, 0, kSplitQueues * sizeof(unsigned), ws->stream)); // (a launch that was short of workgroups left the queues out of step)
    ws->split_ticket_base = 0;
    return DGNN_OK;
}

void rearm_split(Workspace* ws, bool forget)
{
    if (forget) ws->split_failures = 0;
    if (ws->split_failures < kSplitMaxFailures) { ws->split_failed = false; ws->split_cooldown = 0; }
}

int resolve_split_failure(Workspace* ws)
{
    if (!ws->split_pending) {
        if (ws->split_failed && ws->split_cooldown > 0 && --ws->split_cooldown == 0) rearm_split(ws, false);
        return DGNN_OK;
    }
    ws->split_pending = false;
    if (*ws->h_split_err == 0) return DGNN_OK;
    int status = note_split_failure(ws);
    if (status != DGNN_OK) return status;
    status = enqueue_forward(ws, -1, ws->pending_form);
    if (status != DGNN_OK) return status;
    DGNN_HIP(enqueue_outputs(ws));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return DGNN_OK;
}

// The resident / timed entry points (measurement harness) have no outputs to repeat: after their stream has drained they only
// look at the give-up word, switch the workspace to per-layer launches and say so -- a timed run that computed garbage must not
// be reported as a fast one.  Returns DGNN_OK, or DGNN_BUSY when the passes just timed gave up (run them again).
int check_split_after_sync(Workspace* ws)
{
    if (tower_mode(ws) <= 0) return DGNN_OK;
    unsigned err = 0;
    DGNN_HIP(hipMemcpy(&err, ws->d_split_err, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (err == 0) return DGNN_OK;
    const int status = note_split_failure(ws);
    if (status != DGNN_OK) return status;
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return DGNN_BUSY;
}

float env_softmax_temperature()
{
    // config.rs:176-177: get_env("SOFTMAX_TEMPERATURE").unwrap_or(0.709888)
    const char* s = std::getenv("SOFTMAX_TEMPERATURE");
    if (s && *s) {
        char* end = nullptr;
        const float v = std::strtof(s, &end);
        if (end != s && *end == '\0' && v > 0.0f) return v;
    }
    return 0.709888f;
}

int finish_network(std::unique_ptr<Network>& net, dgnn_network** out)
{
    net->topo.softmax_temperature = env_softmax_temperature();
    const int status = build_network(*net);
    if (status != DGNN_OK) return status;
    *out = reinterpret_cast<dgnn_network*>(net.release());
    return DGNN_OK;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

// dgnn_layer_conv3x3 on the board path: NHWC host buffers -> device -> padded slot-major layout -> board kernel ->
// NHWC -> host.  Exercises the layout kernels too.
static int layer_conv3x3_board(const uint16_t* x, int batch, int cin, const uint16_t* w_krsc, const uint16_t* bias, int cout,
                               float alpha1, float alpha2, const uint16_t* z, int relu, uint16_t* y)
{
    const size_t num_pos = (size_t)batch * DGNN_NUM_POINTS;
    const size_t board_pos = (size_t)batch * kBoardRows;
    const size_t sz_xn = align_up(num_pos * cin * 2, 256), sz_yn = align_up(num_pos * cout * 2, 256);
    const size_t sz_xb = align_up(board_pos * cin * 2 + 1024, 256), sz_yb = align_up(board_pos * cout * 2 + 1024, 256);
    const size_t sz_w = align_up(board_filter_elements(cout, cin) * 2, 256), sz_b = align_up((size_t)cout * 2, 256);

    std::vector<uint16_t> packed(board_filter_elements(cout, cin), 0);
    pack_filter_board(w_krsc, cout, cin, packed.data());

    uint8_t* d = nullptr;
    const size_t total = sz_xn + sz_yn + sz_xb + 2 * sz_yb + sz_w + sz_b;
    DGNN_HIP(hipMalloc((void**)&d, total));
    uint16_t* d_xn = (uint16_t*)d;
    uint16_t* d_yn = (uint16_t*)(d + sz_xn);
    uint16_t* d_xb = (uint16_t*)(d + sz_xn + sz_yn);
    uint16_t* d_yb = (uint16_t*)(d + sz_xn + sz_yn + sz_xb);
    uint16_t* d_zb = (uint16_t*)(d + sz_xn + sz_yn + sz_xb + sz_yb);
    uint16_t* d_w = (uint16_t*)(d + sz_xn + sz_yn + sz_xb + 2 * sz_yb);
    uint16_t* d_b = (uint16_t*)(d + sz_xn + sz_yn + sz_xb + 2 * sz_yb + sz_w);

    int status = DGNN_OK;
    hipError_t err = hipMemset(d, 0, total);
    if (err == hipSuccess) err = hipMemcpy(d_xn, x, num_pos * cin * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = launch_nhwc_to_board(d_xn, d_xb, batch, cin, nullptr);
    if (err == hipSuccess && alpha2 != 0.0f) {
        err = hipMemcpy(d_yn, z, num_pos * cout * 2, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = launch_nhwc_to_board(d_yn, d_zb, batch, cout, nullptr);
        if (err == hipSuccess) err = hipDeviceSynchronize();
    }
    if (err == hipSuccess) err = hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_b, bias, (size_t)cout * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        ConvBoardLaunch a;
        a.x = d_xb; a.w = d_w; a.bias = d_b; a.z = alpha2 != 0.0f ? d_zb : nullptr; a.y = d_yb;
        a.batch = batch; a.cin = cin; a.cout = cout; a.alpha1 = alpha1; a.alpha2 = alpha2; a.relu = relu;
        a.timing = nullptr; a.w_next = nullptr; a.carry_in = 0;
        // instrumentation: DGNN_CONV_TIMING=1 prints where the workgroups spend their cycles (s_memtime stamps)
        const char* ab = ablation_env("DGNN_CONV_TIMING");
        const size_t num_wg = (size_t)batch * (cout / 256);
        unsigned long long* d_timing = nullptr;
        if (ab && std::atoi(ab) == 1 && hipMalloc((void**)&d_timing, num_wg * 8 * 4 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(d_timing, 0, num_wg * 8 * 4 * sizeof(unsigned long long));    // variants with fewer than 8 waves leave slots empty
            a.timing = d_timing;
        }
        const int reps = a.timing ? 3 : 1;   // the last repetition is reported (caches warm)
        const BoardSwitches env = read_board_switches();   // per call: the parity tests switch variants between calls
        for (int r = 0; r < reps && err == hipSuccess; ++r) err = launch_conv3x3_board(a, env, nullptr);
        if (err == hipSuccess && d_timing && hipDeviceSynchronize() == hipSuccess) {
            std::vector<unsigned long long> t(num_wg * 8 * 4);
            if (hipMemcpy(t.data(), d_timing, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double pro = 0, loop = 0, epi = 0;
                unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
                size_t waves = 0;
                for (size_t i = 0; i < num_wg * 8; ++i) {
                    if (t[i * 4] == 0) continue;
                    ++waves;
                    pro += (double)(t[i * 4 + 1] - t[i * 4 + 0]);
                    loop += (double)(t[i * 4 + 2] - t[i * 4 + 1]);
                    epi += (double)(t[i * 4 + 3] - t[i * 4 + 2]);
                    if (t[i * 4] < first) first = t[i * 4];
                    if (t[i * 4 + 3] > last) last = t[i * 4 + 3];
                    if (t[i * 4 + 3] < first_end) first_end = t[i * 4 + 3];
                }
                const double n = (double)(waves ? waves : 1);
                std::fprintf(stderr, "[dgnn timing] board kernel, %zu workgroups: mean cycles per wave  prologue %.0f  k-loop %.0f (%.0f per k-step)  "
                             "epilogue %.0f | first start -> last end %llu, first end at %llu (s_memtime ticks)\n",
                             num_wg, pro / n, loop / n, loop / n / (9.0 * cin / 32), epi / n, last - first, first_end - first);
            }
        }
        if (d_timing) (void)hipFree(d_timing);
    }
    if (err == hipSuccess) err = launch_board_to_nhwc(d_yb, d_yn, batch, cout, nullptr);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(y, d_yn, num_pos * cout * 2, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { set_last_hip_error((int)err); status = DGNN_HIP_ERROR; }
    (void)hipFree(d);
    return status;
}

static Network* N(dgnn_network* p) { return reinterpret_cast<Network*>(p); }
static const Network* N(const dgnn_network* p) { return reinterpret_cast<const Network*>(p); }
static Workspace* W(dgnn_workspace* p) { return reinterpret_cast<Workspace*>(p); }
static const Workspace* W(const dgnn_workspace* p) { return reinterpret_cast<const Workspace*>(p); }

extern "C" {

const char* dgnn_status_string(int status)
{
    switch (status) {
        case DGNN_OK: return "success";
        case DGNN_MISSING_WEIGHTS: return "missing weights";
        case DGNN_MALFORMED_WEIGHTS: return "malformed weights";
        case DGNN_HIP_ERROR: return "HIP runtime error";
        case DGNN_BAD_ARGUMENT: return "bad argument";
        case DGNN_OUT_OF_MEMORY: return "out of host memory";
        case DGNN_UNSUPPORTED: return "unsupported network architecture";
        case DGNN_COMM_ERROR: return "RCCL error";
        case DGNN_BUSY: return "busy: workspaces are checked out, or a timed pass has to be repeated";
        default: return "unknown status";
    }
}

int dgnn_last_hip_error(void) { return g_last_hip_error; }

const char* dgnn_version(void)
{
#ifdef DGNN_ABLATIONS
    return "dgnn-hip 0.5 (gfx950; board kernel mfma_f32_16x16x32_f16; +ablations: tile kernels, pair split, ablated board kernels, measurement hooks; +test entries)";
#elif defined(DGNN_TEST_ENTRIES)
    return "dgnn-hip 0.5 (gfx950; board kernel mfma_f32_16x16x32_f16; +test entries of include/dgnn_test.h)";
#else
    return "dgnn-hip 0.5 (gfx950; board kernel mfma_f32_16x16x32_f16)";
#endif
}

int dgnn_device_count(int* count)
{
    if (!count) return DGNN_BAD_ARGUMENT;
    *count = 0;
    DGNN_HIP(dgnn::device_count(count));
    return DGNN_OK;
}

int dgnn_device_pci_bus_id(int device, char* out, size_t capacity)
{
    if (!out || capacity < 16) return DGNN_BAD_ARGUMENT;
    int count = 0;
    DGNN_HIP(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(hipDeviceGetPCIBusId(out, (int)capacity, dgnn::physical_device(device)));
    return DGNN_OK;
}

int dgnn_network_load(const char* json_path, dgnn_network** out)
{
    if (!json_path || !out) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<Network> net(new (std::nothrow) Network());
    if (!net) return DGNN_OUT_OF_MEMORY;
    const int status = load_weights_file(json_path, net->tensors);
    if (status != DGNN_OK) return status;
    return finish_network(net, out);
}

int dgnn_network_from_json(const char* json_text, size_t json_len, dgnn_network** out)
{
    if (!json_text || !out) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<Network> net(new (std::nothrow) Network());
    if (!net) return DGNN_OUT_OF_MEMORY;
    const int status = parse_weights(json_text, json_len, net->tensors);
    if (status != DGNN_OK) return status;
    return finish_network(net, out);
}

int dgnn_network_load_default(const char* exe_path, dgnn_network** out)
{
    if (!out) return DGNN_BAD_ARGUMENT;
    *out = nullptr;

    std::vector<std::string> paths;
    if (exe_path && *exe_path) {   // network.rs:96-101: the executable's path with extension "json"
        std::string p(exe_path);
        const size_t slash = p.find_last_of('/');
        const size_t dot = p.find_last_of('.');
        if (dot != std::string::npos && (slash == std::string::npos || dot > slash + 1)) p.erase(dot);
        paths.push_back(p + ".json");
    } else {
        paths.push_back("dream_go.json");
    }
    paths.push_back("dream_go.json");
    paths.push_back("models/dream_go.json");
    paths.push_back("/usr/share/dreamgo/dream_go.json");
    paths.push_back("/usr/share/dream_go/dream_go.json");

    for (const std::string& path : paths) {
        const int status = dgnn_network_load(path.c_str(), out);
        if (status == DGNN_MISSING_WEIGHTS) continue;   // network.rs:112-113
        return status;                                   // success, or the reference's panic! (:114-116)
    }
    return DGNN_MISSING_WEIGHTS;
}

// A workspace holds a pointer to its network (its weights, its pool): destroying the network under a checked-out workspace would
// leave dgnn_workspace_release locking a freed mutex.  The reference cannot express that (WorkspaceGuard borrows the Network,
// network.rs:54-79); a C caller -- or a garbage collector that finalises two objects in an order of its own choosing -- can.  So
// the destruction is deferred to the release of the last checked-out workspace.
void dgnn_network_destroy(dgnn_network* net)
{
    if (!net) return;
    Network* n = N(net);
    {
        std::lock_guard<std::mutex> lock(n->mutex);
        if (n->checked_out > 0) { n->destroy_pending = true; return; }
    }
    delete n;
}

int dgnn_network_info(const dgnn_network* net, int* num_blocks, int* num_channels, int* num_samples)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    if (num_blocks) *num_blocks = N(net)->topo.num_blocks;
    if (num_channels) *num_channels = N(net)->topo.num_channels;
    if (num_samples) *num_samples = N(net)->topo.num_samples;
    return DGNN_OK;
}

int dgnn_network_set_softmax_temperature(dgnn_network* net, float temperature)
{
    if (!net || !(temperature > 0.0f)) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    if (n->frozen) return DGNN_BAD_ARGUMENT;   // tau * b_l is baked into the image that is already on a device
    n->topo.softmax_temperature = temperature;
    return build_network(*n);
}

int dgnn_network_get_workspace(dgnn_network* net, int device, int batch_size, dgnn_workspace** out)
{
    if (!net || !out || batch_size <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);

    DeviceState* dev = nullptr;
    int status = get_device_state(n, device, &dev);
    if (status != DGNN_OK) return status;

    std::vector<Workspace*>& idle = dev->idle[batch_size];
    if (!idle.empty()) {   // network.rs:137-140: candidates.try_pop()
        *out = reinterpret_cast<dgnn_workspace*>(idle.back());
        idle.pop_back();
        ++n->checked_out;
        return DGNN_OK;
    }

    Workspace* ws = nullptr;
    status = create_workspace(n, dev, batch_size, &ws);
    if (status != DGNN_OK) { delete ws; return status; }
    dev->all.push_back(ws);
    ++n->checked_out;
    *out = reinterpret_cast<dgnn_workspace*>(ws);
    return DGNN_OK;
}

void dgnn_workspace_release(dgnn_workspace* handle)
{
    if (!handle) return;
    Workspace* ws = W(handle);
    if (ws->pending) {   // never hand back a workspace with work in flight
        (void)dgnn::set_device(ws->device);
        (void)hipStreamSynchronize(ws->stream);
        ws->pending = false;
        note_submission_ends(ws);
    }
    Network* net = ws->net;
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(net->mutex);
        ws->dev->idle[ws->batch].push_back(ws);
        last = --net->checked_out == 0 && net->destroy_pending;
    }
    if (last) delete net;      // dgnn_network_destroy was called while this workspace was out
}

int dgnn_network_synchronize(dgnn_network* net)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);

    int status = DGNN_OK;
    for (auto& entry : n->devices) {
        DeviceState* dev = entry.second.get();
        if (dgnn::set_device(dev->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            status = DGNN_HIP_ERROR;   // "this should be allowed to fail", network.rs:150-153: keep going

        for (Workspace* ws : dev->all) rearm_split(ws, /*forget=*/true);     // a tower that gave up may try its split again
        // self.workspaces.clear(): drop everything that is pooled; checked-out workspaces stay valid
        for (auto& bucket : dev->idle) {
            for (Workspace* ws : bucket.second) {
                for (size_t i = 0; i < dev->all.size(); ++i)
                    if (dev->all[i] == ws) { dev->all.erase(dev->all.begin() + i); break; }
                delete ws;
            }
            bucket.second.clear();
        }
    }
    return status;
}

int dgnn_network_tower_stats(dgnn_network* net, int device, uint64_t* counts)
{
    if (!net || !counts) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    for (int k = 0; k < 4; ++k) counts[k] = 0;
    for (auto& entry : n->devices) {
        if (device >= 0 && entry.first != device) continue;
        for (int k = 0; k < 4; ++k) counts[k] += entry.second->tower_passes[k].load(std::memory_order_relaxed);
    }
    return DGNN_OK;
}

// ---------------------------------------------------------------------------------------------- hot path

int dgnn_forward_submit(dgnn_workspace* handle, const uint16_t* features, size_t num_features)
{
    if (!handle || !features) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (num_features != (size_t)ws->batch * DGNN_FEATURE_SIZE) return DGNN_BAD_ARGUMENT;   // graph.rs:124-125
    return dgnn_forward_submit_n(handle, features, ws->batch);
}

int dgnn_forward_submit_n(dgnn_workspace* handle, const uint16_t* features, int count)
{
    if (!handle || !features) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (count <= 0 || count > ws->batch) return DGNN_BAD_ARGUMENT;
    if (ws->pending) return DGNN_BAD_ARGUMENT;
    ws->active = count;
    SubmissionBegins begins(ws);
    const size_t num_features = (size_t)count * DGNN_FEATURE_SIZE;

    DGNN_HIP(dgnn::set_device(ws->device));
    const size_t in_bytes = num_features * sizeof(uint16_t);
    if (features != ws->h_features) std::memcpy(ws->h_features, features, in_bytes);
    DGNN_HIP(hipMemcpyAsync(ws->d_features, ws->h_features, in_bytes, hipMemcpyHostToDevice, ws->stream));   // graph.rs:130-131
    ws->host_io = kZeroCopy;                // (the planes go by copy: 23 KB per board are read more than once; the outputs are written in place)

    int status = enqueue_forward(ws, -1);
    if (status != DGNN_OK) return status;
    status = enqueue_split_check(ws, kInputFeatures);
    if (status != DGNN_OK) return status;

    DGNN_HIP(enqueue_outputs(ws));
    { const int marked = mark_submission(ws); if (marked != DGNN_OK) return marked; }
    ws->pending = true;
    note_submission_in_flight(ws);
    return DGNN_OK;
}

int dgnn_forward_wait(dgnn_workspace* handle, uint16_t* value_out, uint16_t* policy_out)
{
    if (!handle || !value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (!ws->pending) return DGNN_BAD_ARGUMENT;

    DGNN_HIP(dgnn::set_device(ws->device));
    ws->pending = false;
    struct Ends { Workspace* w; ~Ends() { note_submission_ends(w); } } ends{ws};         // (on every way out: the device's count of boards in flight)
    { const int waited = wait_submission(ws); if (waited != DGNN_OK) return waited; }   // graph.rs:155-156 (to_vec + synchronize)
    const int split_status = resolve_split_failure(ws);
    if (split_status != DGNN_OK) { ws->active = ws->batch; return split_status; }
    if (value_out != ws->h_value) std::memcpy(value_out, ws->h_value, (size_t)ws->active * 2);
    if (policy_out != ws->h_policy) std::memcpy(policy_out, ws->h_policy, (size_t)ws->active * DGNN_NUM_MOVES * 2);
    ws->active = ws->batch;
    return DGNN_OK;
}

int dgnn_forward_n(dgnn_workspace* ws, const uint16_t* features, int count, uint16_t* value_out, uint16_t* policy_out)
{
    if (!value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    const int status = dgnn_forward_submit_n(ws, features, count);
    if (status != DGNN_OK) return status;
    return dgnn_forward_wait(ws, value_out, policy_out);
}

int dgnn_forward(dgnn_workspace* ws, const uint16_t* features, size_t num_features,
                 uint16_t* value_out, uint16_t* policy_out)
{
    if (!value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    const int status = dgnn_forward_submit(ws, features, num_features);
    if (status != DGNN_OK) return status;
    return dgnn_forward_wait(ws, value_out, policy_out);
}

// Packed features: 361 plane masks + the two komi constants per board (1 472 bytes instead of 23 104), expanded on the
// device by the kernel that also produces the up-layer's input layout.  Same outputs, bit for bit, as dgnn_forward on the
// fp16 features the words describe.
int dgnn_forward_submit_packed(dgnn_workspace* handle, const uint32_t* packed, int count)
{
    if (!handle || !packed) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (count <= 0 || count > ws->batch) return DGNN_BAD_ARGUMENT;
    if (ws->pending) return DGNN_BAD_ARGUMENT;
    ws->active = count;
    SubmissionBegins begins(ws);
    const size_t in_bytes = (size_t)count * DGNN_PACKED_WORDS * sizeof(uint32_t);

    DGNN_HIP(dgnn::set_device(ws->device));
    if (packed != ws->h_packed) std::memcpy(ws->h_packed, packed, in_bytes);
    DGNN_HIP(hipMemcpyAsync(ws->d_packed, ws->h_packed, in_bytes, hipMemcpyHostToDevice, ws->stream));
    ws->host_io = kZeroCopy;

    int status = enqueue_forward(ws, -1, kInputPacked);
    if (status != DGNN_OK) return status;
    status = enqueue_split_check(ws, kInputPacked);
    if (status != DGNN_OK) return status;

    DGNN_HIP(enqueue_outputs(ws));
    { const int marked = mark_submission(ws); if (marked != DGNN_OK) return marked; }
    ws->pending = true;
    note_submission_in_flight(ws);
    return DGNN_OK;
}

int dgnn_forward_packed(dgnn_workspace* ws, const uint32_t* packed, int count, uint16_t* value_out, uint16_t* policy_out)
{
    if (!value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    const int status = dgnn_forward_submit_packed(ws, packed, count);
    if (status != DGNN_OK) return status;
    return dgnn_forward_wait(ws, value_out, policy_out);
}

// Compact board records: 896 bytes of board state per position; the feature planes are built on the device (features.hip).
int dgnn_forward_submit_boards(dgnn_workspace* handle, const uint8_t* records, int count)
{
    if (!handle || !records) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (count <= 0 || count > ws->batch) return DGNN_BAD_ARGUMENT;
    if (ws->pending) return DGNN_BAD_ARGUMENT;
    ws->active = count;
    SubmissionBegins begins(ws);
    const size_t in_bytes = (size_t)count * DGNN_BOARD_RECORD_BYTES;

    DGNN_HIP(dgnn::set_device(ws->device));
    if (records != ws->h_records) std::memcpy(ws->h_records, records, in_bytes);
    if (ws->stage_on) DGNN_HIP(hipEventRecord(ws->stage_ev[0], ws->stream));
    ws->host_io = kZeroCopy;
    if (!ws->host_io) DGNN_HIP(hipMemcpyAsync(ws->d_records, ws->h_records, in_bytes, hipMemcpyHostToDevice, ws->stream));
    if (ws->stage_on) DGNN_HIP(hipEventRecord(ws->stage_ev[1], ws->stream));

    int status = enqueue_forward(ws, -1, kInputRecords);
    if (status != DGNN_OK) return status;
    status = enqueue_split_check(ws, kInputRecords);
    if (status != DGNN_OK) return status;

    DGNN_HIP(enqueue_outputs(ws));
    if (ws->stage_on) DGNN_HIP(hipEventRecord(ws->stage_ev[7], ws->stream));
    { const int marked = mark_submission(ws); if (marked != DGNN_OK) return marked; }
    ws->pending = true;
    note_submission_in_flight(ws);
    return DGNN_OK;
}

int dgnn_forward_boards(dgnn_workspace* ws, const uint8_t* records, int count, uint16_t* value_out, uint16_t* policy_out)
{
    if (!value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    const int status = dgnn_forward_submit_boards(ws, records, count);
    if (status != DGNN_OK) return status;
    return dgnn_forward_wait(ws, value_out, policy_out);
}

int dgnn_workspace_records_buffer(dgnn_workspace* ws, uint8_t** h_records)
{
    if (!ws || !h_records) return DGNN_BAD_ARGUMENT;
    *h_records = W(ws)->h_records;
    return DGNN_OK;
}

int dgnn_board_records_to_packed_device(int device, const uint8_t* records, int count, uint32_t* packed)
{
    if (!records || !packed || count <= 0) return DGNN_BAD_ARGUMENT;
    int devices = 0;
    DGNN_HIP(dgnn::device_count(&devices));
    if (device < 0 || device >= devices) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(device));
    const size_t sz_r = align_up((size_t)count * DGNN_BOARD_RECORD_BYTES, 256), sz_p = (size_t)count * DGNN_PACKED_WORDS * 4;
    uint8_t* d = nullptr;
    DGNN_HIP(hipMalloc((void**)&d, sz_r + sz_p));
    int status = DGNN_OK;
    hipError_t err = hipMemcpy(d, records, (size_t)count * DGNN_BOARD_RECORD_BYTES, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = launch_board_records(d, (uint32_t*)(d + sz_r), nullptr, count, nullptr);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(packed, d + sz_r, sz_p, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { set_last_hip_error((int)err); status = DGNN_HIP_ERROR; }
    (void)hipFree(d);
    return status;
}

int dgnn_workspace_packed_buffer(dgnn_workspace* ws, uint32_t** h_packed)
{
    if (!ws || !h_packed) return DGNN_BAD_ARGUMENT;
    *h_packed = W(ws)->h_packed;
    return DGNN_OK;
}

int dgnn_features_pack(const uint16_t* features, int count, uint32_t* packed)
{
    if (!features || !packed || count <= 0) return DGNN_BAD_ARGUMENT;
    for (int b = 0; b < count; ++b) {
        const uint16_t* f = features + (size_t)b * DGNN_FEATURE_SIZE;
        uint32_t* out = packed + (size_t)b * DGNN_PACKED_WORDS;
        uint16_t v[2] = {0, 0};
        for (int pos = 0; pos < DGNN_NUM_POINTS; ++pos) {
            uint32_t mask = 0;
            for (int c = 0; c < DGNN_NUM_FEATURES; ++c) {
                const uint16_t h = f[pos * DGNN_NUM_FEATURES + c];
                if ((h & 0x7fff) == 0) continue;                      // +0 / -0
                if (c < 2) {                                          // one constant per board (features.rs:236-239)
                    if (v[c] == 0) v[c] = h;
                    else if (v[c] != h) return DGNN_BAD_ARGUMENT;
                } else if (h != 0x3c00) return DGNN_BAD_ARGUMENT;     // every other plane is 0 / 1
                mask |= 1u << c;
            }
            out[pos] = mask;
        }
        out[DGNN_NUM_POINTS] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        for (int i = DGNN_NUM_POINTS + 1; i < DGNN_PACKED_WORDS; ++i) out[i] = 0;
    }
    return DGNN_OK;
}

int dgnn_features_unpack(const uint32_t* packed, int count, uint16_t* features)
{
    if (!features || !packed || count <= 0) return DGNN_BAD_ARGUMENT;
    for (int b = 0; b < count; ++b) {
        const uint32_t* in = packed + (size_t)b * DGNN_PACKED_WORDS;
        uint16_t* f = features + (size_t)b * DGNN_FEATURE_SIZE;
        const uint16_t v[2] = {(uint16_t)(in[DGNN_NUM_POINTS] & 0xffff), (uint16_t)(in[DGNN_NUM_POINTS] >> 16)};
        for (int pos = 0; pos < DGNN_NUM_POINTS; ++pos)
            for (int c = 0; c < DGNN_NUM_FEATURES; ++c)
                f[pos * DGNN_NUM_FEATURES + c] = !((in[pos] >> c) & 1u) ? 0 : c < 2 ? v[c] : 0x3c00;
    }
    return DGNN_OK;
}

// ---------------------------------------------------------------------------------------------- harness

int dgnn_workspace_batch_size(const dgnn_workspace* ws, int* batch_size, int* device)
{
    if (!ws) return DGNN_BAD_ARGUMENT;
    if (batch_size) *batch_size = W(ws)->batch;
    if (device) *device = W(ws)->device;
    return DGNN_OK;
}

int dgnn_workspace_device_buffers(dgnn_workspace* ws, void** d_features, void** d_value, void** d_policy)
{
    if (!ws) return DGNN_BAD_ARGUMENT;
    if (d_features) *d_features = W(ws)->d_features;
    if (d_value) *d_value = W(ws)->d_value;
    if (d_policy) *d_policy = W(ws)->d_policy;
    return DGNN_OK;
}

int dgnn_workspace_host_buffers(dgnn_workspace* ws, uint16_t** h_features, uint16_t** h_value, uint16_t** h_policy)
{
    if (!ws) return DGNN_BAD_ARGUMENT;
    if (h_features) *h_features = W(ws)->h_features;
    if (h_value) *h_value = W(ws)->h_value;
    if (h_policy) *h_policy = W(ws)->h_policy;
    return DGNN_OK;
}

const char* dgnn_workspace_conv_kernel(const dgnn_workspace* handle)
{
    if (!handle) return "";
    const Workspace* ws = W(handle);
    const Topology& t = ws->net->topo;
    if (tower_mode(ws) > 0) return "tower_split_kernel";
    if (tower_selected(ws)) return "tower_board_kernel";
    if (ws->use_board) return "conv3x3_board_kernel";
    if (ws->use_tile && conv_tile_supported(t.num_channels, t.num_channels)) {
        static const char* const names[3][3] = {
            {"conv3x3_tile_kernel fm=4", "conv3x3_tile_kernel fm=6", "conv3x3_tile_kernel fm=8"},
            {"conv3x3_tile2_kernel fm=4", "conv3x3_tile2_kernel fm=6", "conv3x3_tile2_kernel fm=8"},
            {"conv3x3_tile2_kernel(balanced) fm=4", "conv3x3_tile2_kernel(balanced) fm=6", "conv3x3_tile2_kernel(balanced) fm=8"}};
        return names[ws->tile_variant - 1][(ws->tile_fm - 4) / 2];
    }
    return "conv3x3_kernel";
}

int dgnn_workspace_stream(dgnn_workspace* ws, void** hip_stream)
{
    if (!ws || !hip_stream) return DGNN_BAD_ARGUMENT;
    *hip_stream = (void*)W(ws)->stream;
    return DGNN_OK;
}

int dgnn_workspace_upload(dgnn_workspace* handle, const uint16_t* features, size_t num_features)
{
    if (!handle || !features) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    if (num_features != (size_t)ws->batch * DGNN_FEATURE_SIZE) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(ws->device));
    DGNN_HIP(hipMemcpyAsync(ws->d_features, features, num_features * 2, hipMemcpyHostToDevice, ws->stream));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return DGNN_OK;
}

int dgnn_workspace_download(dgnn_workspace* handle, uint16_t* value_out, uint16_t* policy_out)
{
    if (!handle || !value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    DGNN_HIP(dgnn::set_device(ws->device));
    if (ws->outputs_on_host) {
        // the last pass was a dgnn_forward* with zero-copy outputs: its soft-max kernel wrote into the pinned host buffers and left
        // d_value / d_policy as an EARLIER resident pass had them (this call used to hand those out)
        DGNN_HIP(hipStreamSynchronize(ws->stream));
        std::memcpy(value_out, ws->h_value, (size_t)ws->batch * 2);
        std::memcpy(policy_out, ws->h_policy, (size_t)ws->batch * DGNN_NUM_MOVES * 2);
        return DGNN_OK;
    }
    DGNN_HIP(hipMemcpyAsync(value_out, ws->d_value, (size_t)ws->batch * 2, hipMemcpyDeviceToHost, ws->stream));
    DGNN_HIP(hipMemcpyAsync(policy_out, ws->d_policy, (size_t)ws->batch * DGNN_NUM_MOVES * 2, hipMemcpyDeviceToHost, ws->stream));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return DGNN_OK;
}

// Test-only (include/dgnn_test.h): the two fp16 tensors in front of the output activations, as the LAST pass on this workspace left them.
int dgnn_workspace_download_pre(dgnn_workspace* handle, uint16_t* logits_out, uint16_t* value_pre_out)
{
    if (!handle || !logits_out || !value_pre_out) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    DGNN_HIP(dgnn::set_device(ws->device));
    DGNN_HIP(hipMemcpy2DAsync(logits_out, DGNN_NUM_MOVES * 2, ws->d_logits, 384 * 2, DGNN_NUM_MOVES * 2, (size_t)ws->batch, hipMemcpyDeviceToHost,
                              ws->stream));
    DGNN_HIP(hipMemcpyAsync(value_pre_out, ws->d_value_pre, (size_t)ws->batch * 2, hipMemcpyDeviceToHost, ws->stream));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return DGNN_OK;
}

int dgnn_forward_resident(dgnn_workspace* handle)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    DGNN_HIP(dgnn::set_device(ws->device));
    ws->host_io = false;                    // resident inputs, outputs left in HBM (dgnn_workspace_download)
    return enqueue_forward(ws, -1);
}

int dgnn_workspace_sync(dgnn_workspace* handle)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    DGNN_HIP(dgnn::set_device(ws->device));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    return check_split_after_sync(ws);      // DGNN_BUSY: the passes since the last sync gave up on the split tower; run them again
}

int dgnn_forward_timed(dgnn_workspace* handle, int steps, float* total_ms, float* tower_ms, int* tower_launches)
{
    if (!handle || steps <= 0) return DGNN_BAD_ARGUMENT;
    Workspace* ws = W(handle);
    DGNN_HIP(dgnn::set_device(ws->device));

    while (ws->tower_events.size() < (size_t)2 * steps) {
        hipEvent_t e = nullptr;
        DGNN_HIP(hipEventCreate(&e));
        ws->tower_events.push_back(e);
    }

    DGNN_HIP(hipEventRecord(ws->ev_begin, ws->stream));
    for (int i = 0; i < steps; ++i) {
        ws->host_io = false;
        const int status = enqueue_forward(ws, i);
        if (status != DGNN_OK) return status;
    }
    DGNN_HIP(hipEventRecord(ws->ev_end, ws->stream));
    DGNN_HIP(hipStreamSynchronize(ws->stream));
    { const int gave_up = check_split_after_sync(ws); if (gave_up != DGNN_OK) return gave_up; }

    float ms = 0.0f;
    DGNN_HIP(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
    if (total_ms) *total_ms = ms;

    float tower = 0.0f;
    for (int i = 0; i < steps; ++i) {
        float t = 0.0f;
        DGNN_HIP(hipEventElapsedTime(&t, ws->tower_events[2 * i], ws->tower_events[2 * i + 1]));
        tower += t;
    }
    if (tower_ms) *tower_ms = tower;
    const bool one_launch = tower_selected(ws);
    if (tower_launches) *tower_launches = one_launch ? steps : 2 * ws->net->topo.num_blocks * steps;
    return DGNN_OK;
}

// PCIe-inclusive rate as a leaf pool drives a device: two workspaces, one submission always queued behind the running one
// (two batches alive per device, predictors/nn.rs:64-67), host buffers in, host buffers out, no interpreter in the loop.
int dgnn_forward_pipelined_timed_n(dgnn_workspace* const* ws, int in_flight, int form, const void* input, int count, int steps, double* seconds)
{
    if (!ws || in_flight < 1 || in_flight > 8 || !input || count <= 0 || steps <= 0 || !seconds || form < kInputFeatures || form > kInputRecords) return DGNN_BAD_ARGUMENT;
    for (int i = 0; i < in_flight; ++i) {
        if (!ws[i] || count > W(ws[i])->batch) return DGNN_BAD_ARGUMENT;
        for (int j = 0; j < i; ++j) if (ws[i] == ws[j]) return DGNN_BAD_ARGUMENT;
    }
    std::vector<uint16_t> value((size_t)count), policy((size_t)count * DGNN_NUM_MOVES);
    auto submit = [&](dgnn_workspace* w) {
        return form == kInputFeatures ? dgnn_forward_submit_n(w, static_cast<const uint16_t*>(input), count)
             : form == kInputPacked   ? dgnn_forward_submit_packed(w, static_cast<const uint32_t*>(input), count)
                                      : dgnn_forward_submit_boards(w, static_cast<const uint8_t*>(input), count);
    };
    int status = DGNN_OK;                                        // warm: every workspace has run once before the clock starts
    for (int i = 0; i < in_flight && status == DGNN_OK; ++i) status = submit(ws[i]);
    for (int i = 0; i < in_flight && status == DGNN_OK; ++i) status = dgnn_forward_wait(ws[i], value.data(), policy.data());
    if (status != DGNN_OK) return status;
    const long long t0 = now_ns();
    int submitted = 0, waited = 0;
    while (waited < steps && status == DGNN_OK) {
        while (submitted < steps && submitted - waited < in_flight && status == DGNN_OK) status = submit(ws[submitted++ % in_flight]);
        if (status == DGNN_OK) status = dgnn_forward_wait(ws[waited++ % in_flight], value.data(), policy.data());
    }
    *seconds = (double)(now_ns() - t0) * 1e-9;
    return status;
}

// Where the wall time of ONE device batch goes: `steps` submissions of `count` board records with HIP events
// between the stages on the workspace's own stream, `in_flight` workspaces kept busy like a leaf pool keeps batches alive.
//   stage_us[7]: mean microseconds between consecutive stamps -- H2D of the records | planes kernel | tower | head convolution |
//                policy GEMM | soft-max + value | D2H of the outputs.  A stage that has to WAIT for compute units another
//                submission's tower holds shows here as long (a whole-board tower workgroup owns its CU's whole register file).
//   host_us[4]:  mean microseconds of (0) the submit call itself, (1) submit call -> answers in the caller's hands, (2) the device's
//                span first stamp -> last stamp, (3) what (1) has on top of (2): enqueue latency and the host noticing the end.
int dgnn_forward_stage_times(dgnn_workspace* const* ws, int in_flight, const uint8_t* records, int count, int steps, float* stage_us, double* host_us)
{
    if (!ws || in_flight < 1 || in_flight > 8 || !records || count <= 0 || steps <= 0 || !stage_us || !host_us) return DGNN_BAD_ARGUMENT;
    for (int i = 0; i < in_flight; ++i) {
        if (!ws[i] || count > W(ws[i])->batch) return DGNN_BAD_ARGUMENT;
        for (int j = 0; j < i; ++j) if (ws[i] == ws[j]) return DGNN_BAD_ARGUMENT;
    }
    std::vector<uint16_t> value((size_t)count), policy((size_t)count * DGNN_NUM_MOVES);
    int status = DGNN_OK;
    for (int i = 0; i < in_flight && status == DGNN_OK; ++i) {
        Workspace* w = W(ws[i]);
        DGNN_HIP(dgnn::set_device(w->device));
        for (hipEvent_t& e : w->stage_ev) if (!e) DGNN_HIP(hipEventCreate(&e));
        status = dgnn_forward_boards(ws[i], records, count, value.data(), policy.data());      // warm, unstamped
        w->stage_on = true;
    }
    double stage_sum[Workspace::kStages - 1] = {}, host_sum[4] = {};
    long long submitted_at[8] = {}, submit_took[8] = {};
    int submitted = 0, waited = 0;
    auto off = [&] { for (int i = 0; i < in_flight; ++i) W(ws[i])->stage_on = false; };
    while (waited < steps + in_flight && status == DGNN_OK) {                                  // (the first round of submissions is not counted)
        while (submitted < steps + in_flight && submitted - waited < in_flight && status == DGNN_OK) {
            const int k = submitted++ % in_flight;
            submitted_at[k] = now_ns();
            status = dgnn_forward_submit_boards(ws[k], records, count);
            submit_took[k] = now_ns() - submitted_at[k];
        }
        if (status != DGNN_OK) break;
        const int k = waited % in_flight;
        status = dgnn_forward_wait(ws[k], value.data(), policy.data());
        const long long back = now_ns();
        if (status != DGNN_OK) break;
        if (waited++ < in_flight) continue;
        Workspace* w = W(ws[k]);
        float span = 0.0f;
        for (int s = 0; s + 1 < Workspace::kStages; ++s) {
            float ms = 0.0f;
            const hipError_t e = hipEventElapsedTime(&ms, w->stage_ev[s], w->stage_ev[s + 1]);
            if (e != hipSuccess) { off(); set_last_hip_error((int)e); return DGNN_HIP_ERROR; }
            stage_sum[s] += 1e3 * ms;
            span += 1e3f * ms;
        }
        host_sum[0] += 1e-3 * (double)submit_took[k];
        host_sum[1] += 1e-3 * (double)(back - submitted_at[k]);
        host_sum[2] += span;
        host_sum[3] += 1e-3 * (double)(back - submitted_at[k]) - span;
    }
    off();
    if (status != DGNN_OK) return status;
    for (int s = 0; s + 1 < Workspace::kStages; ++s) stage_us[s] = (float)(stage_sum[s] / steps);
    for (int i = 0; i < 4; ++i) host_us[i] = host_sum[i] / steps;
    return DGNN_OK;
}

int dgnn_forward_pipelined_timed(dgnn_workspace* wa, dgnn_workspace* wb, int form, const void* input, int count, int steps, double* seconds)
{
    dgnn_workspace* const both[2] = {wa, wb};
    return dgnn_forward_pipelined_timed_n(both, 2, form, input, count, steps, seconds);
}

// ---------------------------------------------------------------------------------------------- layer level

int dgnn_layer_conv3x3(int device, const uint16_t* x, int batch, int board_size, int cin,
                       const uint16_t* w_krsc, const uint16_t* bias, int cout,
                       float alpha1, float alpha2, const uint16_t* z, int relu, uint16_t* y)
{
    if (!x || !w_krsc || !bias || !y || batch <= 0 || board_size <= 0 || cout <= 0) return DGNN_BAD_ARGUMENT;
    if (cin <= 0 || cin % 32 != 0) return DGNN_UNSUPPORTED;
    if (alpha2 != 0.0f && !z) return DGNN_BAD_ARGUMENT;

    int count = 0;
    DGNN_HIP(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(device));

    const int cout_pad = (cout + 15) / 16 * 16;
    const size_t num_pos = (size_t)batch * board_size * board_size;
    bool use_tile = true;
    int fm = 8, variant = 2;
    conv_impl_from_env((int)num_pos, cout, &use_tile, &fm, &variant);
    use_tile = use_tile && conv_tile_supported(cin, cout);
    {
        const char* impl = std::getenv("DGNN_CONV");
        if (impl && std::strcmp(impl, "board") == 0 && board_size == 19 && conv_board_supported(cin, cout))
            return layer_conv3x3_board(x, batch, cin, w_krsc, bias, cout, alpha1, alpha2, z, relu, y);
    }

    const size_t sz_x = align_up(num_pos * cin * 2, 256);
    const size_t sz_y = align_up(num_pos * cout * 2, 256);
    const size_t sz_w = align_up((size_t)9 * cin * cout_pad * 2, 256);
    const size_t sz_b = align_up((size_t)cout_pad * 2, 256);

    std::vector<uint16_t> packed((size_t)9 * cin * cout_pad, 0), bias_pad(cout_pad, 0);
    if (use_tile) pack_filter_tile(w_krsc, cout, cin, packed.data());
    else pack_filter(w_krsc, cout, cin, packed.data());
    std::memcpy(bias_pad.data(), bias, (size_t)cout * 2);

    uint8_t* d = nullptr;
    DGNN_HIP(hipMalloc((void**)&d, kZeroPageBytes + sz_x + 2 * sz_y + sz_w + sz_b));
    uint16_t* d_zero = (uint16_t*)d;
    uint16_t* d_x = (uint16_t*)(d + kZeroPageBytes);
    uint16_t* d_y = (uint16_t*)(d + kZeroPageBytes + sz_x);
    uint16_t* d_z = (uint16_t*)(d + kZeroPageBytes + sz_x + sz_y);
    uint16_t* d_w = (uint16_t*)(d + kZeroPageBytes + sz_x + 2 * sz_y);
    uint16_t* d_b = (uint16_t*)(d + kZeroPageBytes + sz_x + 2 * sz_y + sz_w);

    int status = DGNN_OK;
    hipError_t err = hipMemset(d_zero, 0, kZeroPageBytes);
    if (err == hipSuccess) err = hipMemcpy(d_x, x, num_pos * cin * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_b, bias_pad.data(), bias_pad.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess && alpha2 != 0.0f) err = hipMemcpy(d_z, z, num_pos * cout * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess && use_tile) {
        ConvTileLaunch a;
        a.x = d_x; a.w = d_w; a.bias = d_b; a.z = alpha2 != 0.0f ? d_z : nullptr; a.y = d_y; a.zero_page = d_zero;
        a.num_pos = (int)num_pos; a.board = board_size; a.cin = cin; a.cout = cout;
        a.alpha1 = alpha1; a.alpha2 = alpha2; a.relu = relu;
        a.timing = nullptr;
        // instrumentation: DGNN_CONV_ABLATE=8 makes the staggered kernel accumulate s_memtime per pipeline segment
        const char* ab = ablation_env("DGNN_CONV_ABLATE");
        const bool timing = ab && std::atoi(ab) == 8 && fm == 6 && variant >= 2;
        const size_t num_wg = ((num_pos + 32 * fm - 1) / (32 * fm)) * (cout / 256);
        unsigned long long* d_timing = nullptr;
        if (timing && hipMalloc((void**)&d_timing, num_wg * 8 * 4 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(d_timing, 0, num_wg * 8 * 4 * sizeof(unsigned long long));
            a.timing = d_timing;
        }
        err = launch_conv3x3_tile(a, fm, variant, nullptr);
        if (err == hipSuccess && d_timing) {
            std::vector<unsigned long long> t(num_wg * 8 * 4);
            if (hipDeviceSynchronize() == hipSuccess &&
                hipMemcpy(t.data(), d_timing, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double sum[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                for (size_t wg = 0; wg < num_wg; ++wg)
                    for (int w = 0; w < 8; ++w)
                        for (int i = 0; i < 4; ++i) sum[w >> 2][i] += (double)t[(wg * 8 + w) * 4 + i];
                const double n = (double)num_wg * 4 * (9.0 * cin / 64);
                std::fprintf(stderr, "[dgnn timing] variant %d: cycles per k-step per wave  group0: phase1 %.0f barrier1 %.0f phase2 %.0f barrier2 %.0f | "
                             "group1: phase1 %.0f barrier1 %.0f phase2 %.0f barrier2 %.0f\n", variant,
                             sum[0][0] / n, sum[0][1] / n, sum[0][2] / n, sum[0][3] / n,
                             sum[1][0] / n, sum[1][1] / n, sum[1][2] / n, sum[1][3] / n);
            }
        }
        if (d_timing) (void)hipFree(d_timing);
    } else if (err == hipSuccess) {
        ConvLaunch a;
        a.x = d_x; a.w = d_w; a.bias = d_b; a.z = alpha2 != 0.0f ? d_z : nullptr; a.y = d_y;
        a.num_pos = (int)num_pos; a.board = board_size; a.cin = cin; a.cout = cout; a.cout_pad = cout_pad;
        a.alpha1 = alpha1; a.alpha2 = alpha2; a.relu = relu;
        err = launch_conv3x3(a, nullptr);
    }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(y, d_y, num_pos * cout * 2, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { set_last_hip_error((int)err); status = DGNN_HIP_ERROR; }
    (void)hipFree(d);
    return status;
}

int dgnn_layer_dense(int device, const uint16_t* x, int batch, int k_in, const uint16_t* w_in_out, const uint16_t* bias,
                     int n_out, float alpha, int relu, uint16_t* y)
{
    if (!x || !w_in_out || !bias || !y || batch <= 0 || k_in <= 0 || n_out <= 0) return DGNN_BAD_ARGUMENT;
    int count = 0;
    DGNN_HIP(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(device));

    // what build_network does for linear_1: the stored [in][out] matrix transposed to [out][in], zero padded (dense.rs:182-189)
    const int k_pad = (k_in + 63) / 64 * 64;
    const int rows = (n_out + 31) / 32 * 32;
    std::vector<uint16_t> wt((size_t)rows * k_pad, 0), bias_pad((size_t)rows, 0);
    pack_dense_transposed(w_in_out, k_in, n_out, k_pad, wt.data());
    std::memcpy(bias_pad.data(), bias, (size_t)n_out * 2);

    const size_t sz_x = align_up((size_t)batch * k_in * 2 + 64, 256), sz_w = align_up(wt.size() * 2, 256);
    const size_t sz_b = align_up(bias_pad.size() * 2, 256), sz_y = align_up((size_t)batch * n_out * 2, 256);
    uint8_t* d = nullptr;
    DGNN_HIP(hipMalloc((void**)&d, sz_x + sz_w + sz_b + sz_y));
    int status = DGNN_OK;
    hipError_t err = hipMemset(d, 0, sz_x);
    if (err == hipSuccess) err = hipMemcpy(d, x, (size_t)batch * k_in * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + sz_x, wt.data(), wt.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + sz_x + sz_w, bias_pad.data(), bias_pad.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        PolicyGemmLaunch pg;
        pg.features = (const uint16_t*)d;
        pg.wt = (const uint16_t*)(d + sz_x);
        pg.bias = (const uint16_t*)(d + sz_x + sz_w);
        pg.logits = (uint16_t*)(d + sz_x + sz_w + sz_b);
        pg.batch = batch; pg.k_in = k_in; pg.k_pad = k_pad; pg.tau = alpha;
        pg.n_out = n_out; pg.ld_out = n_out; pg.relu = relu;
        err = launch_policy_gemm(pg, nullptr);
    }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(y, d + sz_x + sz_w + sz_b, (size_t)batch * n_out * 2, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { set_last_hip_error((int)err); status = DGNN_HIP_ERROR; }
    (void)hipFree(d);
    return status;
}

int dgnn_layer_heads(int device, const uint16_t* logits, const uint16_t* value_features, int batch, const uint16_t* value_w,
                     const uint16_t* value_b, uint16_t* policy_out, uint16_t* value_out)
{
    if (!logits || !value_features || !value_w || !value_b || !policy_out || !value_out || batch <= 0) return DGNN_BAD_ARGUMENT;
    int count = 0;
    DGNN_HIP(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP(dgnn::set_device(device));
    const size_t sz_l = align_up((size_t)batch * 384 * 2, 256), sz_vf = align_up((size_t)batch * 722 * 2, 256);
    const size_t sz_w = align_up(722 * 2, 256), sz_b = 256, sz_p = align_up((size_t)batch * DGNN_NUM_MOVES * 2, 256), sz_v = align_up((size_t)batch * 2, 256);
    std::vector<uint16_t> padded((size_t)batch * 384, 0);
    for (int b = 0; b < batch; ++b) std::memcpy(padded.data() + (size_t)b * 384, logits + (size_t)b * DGNN_NUM_MOVES, DGNN_NUM_MOVES * 2);
    uint8_t* d = nullptr;
    DGNN_HIP(hipMalloc((void**)&d, sz_l + sz_vf + sz_w + sz_b + sz_p + sz_v));
    int status = DGNN_OK;
    hipError_t err = hipMemcpy(d, padded.data(), padded.size() * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + sz_l, value_features, (size_t)batch * 722 * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + sz_l + sz_vf, value_w, 722 * 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + sz_l + sz_vf + sz_w, value_b, 2, hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        SoftmaxValueLaunch sv;
        sv.logits = (const uint16_t*)d;
        sv.value_features = (const uint16_t*)(d + sz_l);
        sv.value_w = (const uint16_t*)(d + sz_l + sz_vf);
        sv.value_b = (const uint16_t*)(d + sz_l + sz_vf + sz_w);
        sv.policy_out = (uint16_t*)(d + sz_l + sz_vf + sz_w + sz_b);
        sv.value_out = (uint16_t*)(d + sz_l + sz_vf + sz_w + sz_b + sz_p);
        sv.value_pre_out = nullptr;
        sv.batch = batch;
        err = launch_softmax_value(sv, nullptr);
    }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemcpy(policy_out, d + sz_l + sz_vf + sz_w + sz_b, (size_t)batch * DGNN_NUM_MOVES * 2, hipMemcpyDeviceToHost);
    if (err == hipSuccess) err = hipMemcpy(value_out, d + sz_l + sz_vf + sz_w + sz_b + sz_p, (size_t)batch * 2, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { set_last_hip_error((int)err); status = DGNN_HIP_ERROR; }
    (void)hipFree(d);
    return status;
}

// ---------------------------------------------------------------------------------------------- multi-GPU

int dgnn_network_weight_image(dgnn_network* net, int device, void** d_image, size_t* num_bytes)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    DeviceState* dev = nullptr;
    const int status = get_device_state(n, device, &dev);
    if (status != DGNN_OK) return status;
    if (d_image) *d_image = dev->d_image;
    if (num_bytes) *num_bytes = n->image.size();
    return DGNN_OK;
}

int dgnn_network_weight_image_reserve(dgnn_network* net, int device, void** d_image, size_t* num_bytes)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    // A receiver is about to be OVERWRITTEN: not while a workspace is checked out (a forward may be reading the range), and from
    // now until dgnn_network_weight_image_commit the range counts as empty -- also on a device that already held a valid image: a
    // failed or partial transfer then ends in a fresh upload from the host instead of kernels on a torn image
    if (n->checked_out > 0) return DGNN_BUSY;
    DeviceState* dev = nullptr;
    const int status = get_device_state(n, device, &dev, /*upload=*/false);
    if (status != DGNN_OK) return status;
    dev->image_valid = false;
    if (d_image) *d_image = dev->d_image;
    if (num_bytes) *num_bytes = n->image.size();
    return DGNN_OK;
}

// The range dgnn_network_weight_image_reserve handed out has been filled with this network's image by the caller (a transfer that
// is known to have succeeded): from now on it is what the device's kernels read.
int dgnn_network_weight_image_commit(dgnn_network* net, int device)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    auto it = n->devices.find(device);
    if (it == n->devices.end()) return DGNN_BAD_ARGUMENT;
    it->second->image_valid = true;
    return DGNN_OK;
}

int dgnn_network_weight_checksum(const dgnn_network* net, uint64_t* checksum)
{
    if (!net || !checksum) return DGNN_BAD_ARGUMENT;
    uint64_t h = 1469598103934665603ull;   // FNV-1a
    for (uint8_t byte : N(net)->image) { h ^= byte; h *= 1099511628211ull; }
    *checksum = h;
    return DGNN_OK;
}

int dgnn_network_set_weight_image(dgnn_network* net, int device, const void* src, size_t num_bytes, int src_on_device)
{
    if (!net || !src) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    if (num_bytes != n->image.size()) return DGNN_BAD_ARGUMENT;
    // a forward that is in flight reads the image and the launch descriptors this call replaces: refuse while any workspace is
    // checked out (a leaf pool holds one per batch it is evaluating: pause the pool first, dgnn_pool_pause)
    if (n->checked_out > 0) return DGNN_BUSY;

    // the new image on the host first: its header is checked against this network (same architecture, same packing) BEFORE
    // anything is overwritten -- neither the host image nor any device copy -- and it carries the launch scalars (gates,
    // temperature) the kernels are started with
    std::vector<uint8_t> incoming;
    const uint8_t* host = (const uint8_t*)src;
    if (src_on_device) {
        incoming.resize(num_bytes);
        DGNN_HIP(dgnn::set_device(device));
        DGNN_HIP(hipDeviceSynchronize());   // e.g. the collective that wrote it
        DGNN_HIP(hipMemcpy(incoming.data(), src, num_bytes, hipMemcpyDeviceToHost));
        host = incoming.data();
    }
    int status = adopt_image(*n, host, num_bytes);   // net->image, alpha per block, tau; refuses a foreign image, changing nothing
    if (status != DGNN_OK) return status;

    DeviceState* dev = nullptr;                      // the device the image sits on gets its copy from there
    status = get_device_state(n, device, &dev, /*upload=*/false);
    if (status != DGNN_OK) return status;
    for (auto& entry : n->devices) {                 // every device copy (devices first touched later upload net->image, the new one)
        DeviceState* d = entry.second.get();
        DGNN_HIP(dgnn::set_device(d->device));
        DGNN_HIP(hipDeviceSynchronize());            // (nothing is checked out, so nothing is in flight; belt and braces)
        d->image_valid = false;
        if (d == dev && src_on_device) {
            if ((const void*)d->d_image != src) DGNN_HIP(hipMemcpy(d->d_image, src, num_bytes, hipMemcpyDeviceToDevice));
        } else {
            DGNN_HIP(hipMemcpy(d->d_image, n->image.data(), num_bytes, hipMemcpyHostToDevice));
        }
        DGNN_HIP(hipDeviceSynchronize());            // the copy has LANDED before a (non-blocking) workspace stream may read the image
        d->image_valid = true;
        for (Workspace* ws : d->all) ws->tower_layers_ready = false;   // the cached descriptors hold the old gates
    }
    return DGNN_OK;
}

// What must agree between the ranks of a broadcast BEFORE a byte of the image moves: the image size and a hash of the
// architecture and packing (blocks, channels, samples, filter layouts) -- not of the weights, gates or temperature, which are
// what the broadcast is for.  descriptor[0] = bytes, descriptor[1] = the hash.
int dgnn_network_image_descriptor(const dgnn_network* net, uint64_t* descriptor)
{
    if (!net || !descriptor) return DGNN_BAD_ARGUMENT;
    const Topology& t = N(net)->topo;
    descriptor[0] = (uint64_t)N(net)->image.size();
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)t.num_blocks, (uint64_t)t.num_channels, (uint64_t)t.num_samples, (uint64_t)t.layouts}) { h ^= v; h *= 1099511628211ull; }
    descriptor[1] = h;
    return DGNN_OK;
}

int dgnn_network_weight_checksum_device(dgnn_network* net, int device, uint64_t* checksum)
{
    if (!net || !checksum) return DGNN_BAD_ARGUMENT;
    Network* n = N(net);
    std::lock_guard<std::mutex> lock(n->mutex);
    DeviceState* dev = nullptr;
    const int status = get_device_state(n, device, &dev);
    if (status != DGNN_OK) return status;
    std::vector<uint8_t> copy(n->image.size());
    DGNN_HIP(dgnn::set_device(device));
    DGNN_HIP(hipMemcpy(copy.data(), dev->d_image, copy.size(), hipMemcpyDeviceToHost));
    uint64_t h = 1469598103934665603ull;
    for (uint8_t byte : copy) { h ^= byte; h *= 1099511628211ull; }
    *checksum = h;
    return DGNN_OK;
}

int dgnn_network_weight_image_host(const dgnn_network* net, const void** image, size_t* num_bytes)
{
    if (!net) return DGNN_BAD_ARGUMENT;
    if (image) *image = N(net)->image.data();
    if (num_bytes) *num_bytes = N(net)->image.size();
    return DGNN_OK;
}

}  // extern "C"
