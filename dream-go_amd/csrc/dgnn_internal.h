// Internal declarations shared by the host side of libdgnn_hip.so.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "tsan_mutex.h"
#include "../../include/dgnn_test.h"   // dgnn.h + dgnn_host.h + the test-only entry points (exported by the test build only)

namespace dgnn {

// map[i] = packed index of `transform` applied to point i (pass maps to pass); batcher.cpp
const uint16_t* symmetry_map(int transform);

// ----------------------------------------------------------------------------------------------
// fp16 helpers (host).  Same rounding as llvm.convert.to.fp16 (src/libdg_utils/types/fp16.rs:53-70).
// ----------------------------------------------------------------------------------------------
float half_to_float(uint16_t h);
uint16_t float_to_half(float f);

// ----------------------------------------------------------------------------------------------
// Weights file (src/libdg_nn/loader.rs, src/libdg_nn/tensor.rs)
// ----------------------------------------------------------------------------------------------
enum class DataType { Int8, Int32, Half, Float };

struct HostTensor {
    DataType dtype = DataType::Float;   // Tensor::default(), tensor.rs:50-60
    std::vector<uint8_t> bytes;         // raw little-endian payload as stored
    float scale = 1.0f;                 // parsed, never applied (tensor.rs:89-91)

    size_t element_size() const;
    size_t num_elements() const { return bytes.size() / element_size(); }
    float as_f32() const;               // tensor.rs:106-112
    int32_t as_i32() const;             // tensor.rs:114-120
};

using TensorMap = std::map<std::string, HostTensor>;

// base85 (RFC 1924 alphabet), src/libdg_utils/b85.rs:101-132.  Returns false on an invalid character.
bool b85_decode(const char* text, size_t len, std::vector<uint8_t>& out);

// loader::load_aux, loader.rs:36-100.  Returns a DGNN_* status.
int parse_weights(const char* text, size_t len, TensorMap& out);
int load_weights_file(const char* path, TensorMap& out);

// ----------------------------------------------------------------------------------------------
// Topology + packed weight image
// ----------------------------------------------------------------------------------------------
struct ConvParams {
    // one 3x3 convolution as the kernels consume it
    size_t w_offset = 0;     // byte offset of the filter in the image, layout [9 * cin / 32][cout_pad][32] fp16
    size_t w_tile_offset = 0; // byte offset of the filter in tile layout (pack_filter_tile), 0 = not available
    size_t w_board_offset = 0; // byte offset of the filter in board layout (pack_filter_board), 0 = not available
    size_t b_offset = 0;     // byte offset of the bias, [cout_pad] fp16 (already scaled where the reference scales)
    int cin = 0;
    int cout = 0;
    int cout_pad = 0;        // cout rounded up to a multiple of 16
    float alpha1 = 1.0f;     // scale of the convolution result
    float alpha2 = 0.0f;     // scale of the skip input (0: skip is not read)
};

// filter layouts an image may carry (Topology::layouts, network.cpp: pack_layouts_from_env)
constexpr unsigned kPackGeneric = 1, kPackTile = 2, kPackBoard = 4;

struct Topology {
    unsigned layouts = 0;        // kPack* bits of the image
    int num_blocks = 0;
    int num_channels = 0;
    int num_samples = 0;
    float softmax_temperature = 0.709888f;
    bool gates_nonzero = true;   // every residual gate can be folded in during the K loop (skip_scale_foldable): what the one-launch tower needs

    ConvParams up;
    std::vector<ConvParams> res1, res2;
    ConvParams policy_conv, value_conv;
    size_t policy_b_offset = 0;  // [362] fp16, tau * b_l rounded to fp16
    size_t value_w_offset = 0;   // [722] fp16
    size_t value_b_offset = 0;   // [1] fp16
    size_t policy_wt_offset = 0; // [384][policy_k_pad] fp16: linear_1 transposed to [out][in] (dense.rs:182-189), zero padded
    int policy_k_pad = 0;        // 361 * S rounded up to a multiple of 64
    size_t head_w_board_offset = 0; // both head convolutions as one 32-row filter in board fragment order, 0 = not available
    size_t head_b_offset = 0;       // [32] fp16: policy conv offsets, then the two value conv offsets, then zeros
};

struct DeviceState;   // per-device weight copy + workspace pool (network.cpp)
struct Workspace;     // workspace.hip

struct Network {
    TensorMap tensors;             // as loaded; kept so that the temperature can be re-baked
    Topology topo;
    std::vector<uint8_t> image;    // packed host weight image
    bool frozen = false;           // set once the first workspace exists
    int checked_out = 0;           // workspaces handed out and not yet released (under `mutex`): weights are not replaced meanwhile
    bool destroy_pending = false;  // dgnn_network_destroy came while workspaces were checked out: the last release deletes the network

    std::mutex mutex;
    MutexLife mutex_life{&mutex};
    std::map<int, std::unique_ptr<DeviceState>> devices;

    ~Network();
};

// Filter [cout][3][3][cin] (KRSC, conv2d.rs:111-117) -> [9 * cin / 32][cout_pad][32] (zero padded rows):
// k-step = tap * (cin / 32) + chunk; one k-step is one contiguous [cout_pad][32] slab.  dst must be zeroed.
void pack_filter(const uint16_t* krsc, int cout, int cin, uint16_t* dst);

// Filter KRSC -> the LDS image order of conv3x3_tile_kernel (conv_tile.hip):
// [cout / 256][9 * cin / 64][256 rows][8 slots][8 halfs], k-step = chunk * 9 + tap, and the 16-byte slot s of row r
// holds channels chunk * 64 + (s ^ ((r >> 1) & 7)) * 8 .. + 8.  Requires cout % 256 == 0 and cin % 64 == 0.
void pack_filter_tile(const uint16_t* krsc, int cout, int cin, uint16_t* dst);

// Filter KRSC -> the per-wave MFMA fragment stream of conv3x3_board_kernel (conv_board.hip):
// [9 * cin / 32 k-steps][cout / 32 slices][2 k16][64 lanes][8 halfs], k-step = chunk * 9 + tap, lane l holds
// w[co = slice * 32 + (l & 31)][tap][ci = chunk * 32 + k16 * 16 + (l >> 5) * 8 .. + 8].  Requires cout % 64 == 0 and
// cin % 32 == 0.  dst holds board_filter_elements(cout, cin) halfs: the kernel reads (never uses) up to 9 k-steps (one chunk) past the end.
void pack_filter_board(const uint16_t* krsc, int cout, int cin, uint16_t* dst);
size_t board_filter_elements(int cout, int cin);

constexpr size_t kZeroPageBytes = 256;   // the image starts with a page of zeros (source of off-board taps)

// Builds topo + image from tensors.  Returns a DGNN_* status.
int build_network(Network& net);

// Weight refresh: `image` (num_bytes == net.image.size()) was produced by build_network of a network with the same
// architecture and packing -- checked against the header it carries -- possibly with other gates / temperature / weights.
// Copies it over net.image and takes the launch scalars (alpha per block, tau) from its header.
int adopt_image(Network& net, const uint8_t* image, size_t num_bytes);

// dense.rs:182-189: a dense matrix stored [k_in][n_out] -> [n_out][k_pad] (dst zeroed by the caller, rows >= n_out stay zero)
void pack_dense_transposed(const uint16_t* w_in_out, int k_in, int n_out, int k_pad, uint16_t* dst);

// The network behind a search: Predictor::predict (src/libdg_mcts/predictor.rs:92) on fp16 feature planes, or -- `boards` set --
// on compact board records (the planes are then built on the device, dgnn_forward_boards) and the game's `record` callback is
// used instead of `features`.
struct Predictor {
    dgnn_predict_fn features = nullptr;
    dgnn_predict_boards_fn boards = nullptr;
    void* ctx = nullptr;

    bool takes_records() const { return boards != nullptr; }
    size_t item_bytes() const { return takes_records() ? (size_t)DGNN_BOARD_RECORD_BYTES : (size_t)DGNN_FEATURE_SIZE * sizeof(uint16_t); }
    // one position of `board` into slot `index` of `buffer` (item_bytes() per slot)
    void encode(const dgnn_game_ops* ops, void* board, int to_move, int symmetry, uint8_t* buffer, size_t index) const
    {
        if (takes_records()) ops->record(board, to_move, symmetry, buffer + index * DGNN_BOARD_RECORD_BYTES);
        else ops->features(board, to_move, symmetry, reinterpret_cast<uint16_t*>(buffer) + index * DGNN_FEATURE_SIZE);
    }
    int predict(const uint8_t* buffer, int n, uint16_t* value, uint16_t* policy) const
    {
        return takes_records() ? boards(ctx, buffer, n, value, policy) : features(ctx, reinterpret_cast<const uint16_t*>(buffer), n, value, policy);
    }
    // completed weight refreshes of the pool behind this predictor (dgnn_pool_generation); 0 for any other predictor (search.cpp)
    uint64_t generation() const;
    // the pool behind this predictor lets requests FOLLOW a position in flight: which evaluation a search sees is then a matter of timing
    // across games, like a shared table -- games with a table of their own (reproducible games) must not run on it
    bool coalesces() const;
    bool usable_with(const dgnn_game_ops* ops) const { return takes_records() ? ops->record != nullptr : (features != nullptr && ops->features != nullptr); }
};

bool pool_coalesces(const dgnn_pool* pool);      // batcher.cpp
int search_run(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, const Predictor& predictor, dgnn_cache* cache, int num_rollouts,
               int num_threads, int leaves_per_thread, uint64_t seed, uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out);
int full_forward(const dgnn_game_ops* ops, void* board, int to_move, const Predictor& predictor, dgnn_cache* cache, float* value_out, float* policy_out);
int selfplay_go(const Predictor& predictor, dgnn_cache* shared_cache, const dgnn_selfplay_params* params, dgnn_selfplay_stats* stats,
                int* moves_out, char* sgf_out, size_t sgf_capacity);

// search.cpp: pool/policy_helper.rs:28-134 on the game callbacks
void create_initial_policy(const dgnn_game_ops* ops, void* board, int to_move, float* policy, int* indices);
void add_valid_candidates(float* dst, const uint16_t* src_f16, const int* indices, int transform);
void normalize_policy(float* policy, float sum_to);

// Host SIMD: the AVX2 / F16C forms of the hot host loops exist on x86-64 only and are used when the CPU has them; everywhere else --
// and with DGNN_HOST_SIMD=0 in the environment (read once; the tests compare the two) -- the scalar forms run.  Same bits either way.
#if defined(__x86_64__)
#define DGNN_X86_SIMD 1
#else
#define DGNN_X86_SIMD 0
#endif
bool host_simd(bool need_f16c = false);

// thread-local HIP error detail
void set_last_hip_error(int code);


// Test hook (include/dgnn_test.h: dgnn_test_fail_push_after): >= 0 counts down with every board pushed into a batcher's queue; the
// push that finds it at 0 fails with DGNN_OUT_OF_MEMORY, as an allocation that failed would.  -1 (always, outside the tests): off.
extern std::atomic<int> g_fail_push_countdown;

// A per-thread free list of blocks of ONE size, for the two objects the search allocates per rollout that are too large for
// glibc's per-thread cache (it stops at 1 KB): a tree node (1.8 KB) and the leaf's copy of the board (3 KB) went through the
// arena lock and the bins on every rollout -- a seventh of the host's time with 256 games in flight.  A block freed on another
// thread than the one that allocated it simply changes lists; a list keeps at most `Keep` blocks and returns the rest (games run
// as fibers that migrate between host threads and pool workers free what search threads allocated, so a consumer thread's list
// fills to Keep and stays there: Keep is what ONE search of a game turns over, not more -- ADVICE r04).  A thread that has ended
// took its list with it: a block freed there later goes back to the heap.  The sanitizer builds allocate directly, so that ASan
// sees every node's lifetime; tests/test_tree_host.py runs the pooled path across threads in the shipped build.
template <size_t Bytes, int Keep>
struct BlockPool {
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
    static void* get() { return ::operator new(Bytes, std::nothrow); }
    static void put(void* p) { ::operator delete(p); }
#else
    struct Free { Free* next; };
    struct List {
        Free* head = nullptr;
        int count = 0;
        bool dead = false;
        ~List() { while (head) { Free* next = head->next; ::operator delete(head); head = next; } count = 0; dead = true; }   // (after it: pass-through)
    };
    static List& list() { static thread_local List l; return l; }
    // (never inlined: games run as fibers that may resume on another host thread, and a caller that contained a switch point could
    // otherwise keep the address of its OLD thread's list across it -- the sanitizer builds, which bypass the pool, would not see it)
    __attribute__((noinline)) static void* get()
    {
        List& l = list();
        if (!l.head) return ::operator new(Bytes, std::nothrow);
        Free* f = l.head;
        l.head = f->next;
        --l.count;
        return f;
    }
    __attribute__((noinline)) static void put(void* p)
    {
        if (!p) return;
        List& l = list();
        if (l.dead || l.count >= Keep) { ::operator delete(p); return; }
        Free* f = static_cast<Free*>(p);
        f->next = l.head;
        l.head = f;
        ++l.count;
    }
#endif
};

}  // namespace dgnn
