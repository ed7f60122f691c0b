// Scope row f-2 (second part): the search driver around the tree -- what dream-go's worker pool does for one search
// (src/libdg_mcts/pool/worker_thread.rs:50-176, pool/event.rs:45-59, pool/policy_helper.rs:28-134), with the game behind
// callbacks and the neural network behind the Predictor::predict signature.
//
//   one rollout   clone the starting board, tree::probe (stones placed on the clone), pick a random symmetry, ask the game
//                 for the features of the leaf, predict, rebuild the prior (legal candidates 0 / others -inf, symmetric
//                 duplicates eliminated, network policy de-transformed and renormalised), tree::insert with the win rate
//   termination   time_control::is_done over a RolloutLimit (time_control/mod.rs:86-101, rollout_limit.rs:35-44): the limit, or
//                 EARLY-C -- the runner-up cannot catch up in the rollouts that remain; like the reference the check
//                 happens before a probe, so with T threads up to T - 1 rollouts more may land
//   cache         an optional transposition table is asked before the features are built (pool/event.rs:50-51) and filled
//                 after a prediction (worker_thread.rs:111)
//   determinism   every random choice comes from a seeded generator (the reference uses thread_rng() everywhere,
//                 SURVEY.md section 7): with ONE search thread the visit counts are a pure function of (seed, network
//                 outputs); with several threads they are not, exactly as in the reference (virtual loss + timing).
//
// Also here: full_forward (the 8-symmetry root evaluation, lib.rs:83-130), dirichlet noise (dirichlet.rs:40-70), and a
// rule-free synthetic game with a synthetic self-play loop for throughput measurements of the host + device pipeline
// without Go rules.  The synthetic game is NOT Go: every empty point is a legal move, nothing is ever captured, features
// are pseudo-random bits of the position hash.  The real game is go.cpp.
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "dgnn_internal.h"

namespace dgnn {
namespace {

struct Rng {      // splitmix64
    uint64_t state;
    uint64_t next() { uint64_t z = (state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace

namespace {
struct CandidateRows {          // eight candidate bits -> eight prior entries (0 for a candidate, -inf otherwise)
    float row[256][8];
    CandidateRows() { for (int b = 0; b < 256; ++b) for (int j = 0; j < 8; ++j) row[b][j] = (b >> j) & 1 ? 0.0f : -std::numeric_limits<float>::infinity(); }
};
const CandidateRows kCandidateRows;
}  // namespace

// pool/policy_helper.rs:28-76: legal candidates 0, everything else -inf; `indices` maps every point to the smallest index
// among its images under the board's own symmetries (and the duplicates are set to -inf)
void create_initial_policy(const dgnn_game_ops* ops, void* board, int to_move, float* policy /*368*/, int* indices /*362*/)
{
    const float minus_inf = -std::numeric_limits<float>::infinity();
    if (ops->candidate_bits) {
        uint64_t bits[6];
        ops->candidate_bits(board, to_move, bits);
        const uint8_t* bytes = reinterpret_cast<const uint8_t*>(bits);        // little endian: byte k holds moves 8 k .. 8 k + 7
        for (int k = 0; k < 46; ++k) std::memcpy(policy + 8 * k, kCandidateRows.row[bytes[k]], 8 * sizeof(float));
        for (int i = DGNN_NUM_MOVES; i < 368; ++i) policy[i] = minus_inf;
    } else {
        unsigned char candidate[DGNN_NUM_MOVES];
        ops->candidates(board, to_move, candidate);
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) policy[i] = candidate[i] ? 0.0f : minus_inf;
        for (int i = DGNN_NUM_MOVES; i < 368; ++i) policy[i] = minus_inf;
    }
    const uint16_t* maps[8];
    int num_symmetric = 0;
    for (int t = 1; t < 8; ++t) if (ops->is_symmetric && ops->is_symmetric(board, t)) maps[num_symmetric++] = symmetry_map(t);
    for (int i = 0; i <= DGNN_NUM_POINTS; ++i) indices[i] = i;
    if (num_symmetric == 0) return;                                           // almost every position after the first few stones
    for (int i = 0; i < DGNN_NUM_POINTS; ++i) {
        int target = i;
        for (int k = 0; k < num_symmetric; ++k) target = std::min(target, (int)maps[k][i]);
        indices[i] = target;
        if (target != i) policy[i] = minus_inf;
    }
}

namespace {

#if DGNN_X86_SIMD
__attribute__((target("avx2,f16c")))
void halfs_to_floats_f16c(const uint16_t* src, float* dst, int n)            // n a multiple of 8
{
    for (int i = 0; i < n; i += 8) _mm256_storeu_ps(dst + i, _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i))));
}
#endif
const bool kHaveF16c = host_simd(true);

// exact conversions either way: the hardware instruction and half_to_float give the same bits
void halfs_to_floats(const uint16_t* src, float* dst, int n)
{
    const int vector = kHaveF16c ? n & ~7 : 0;
#if DGNN_X86_SIMD
    if (vector) halfs_to_floats_f16c(src, dst, vector);
#endif
    for (int i = vector; i < n; ++i) dst[i] = half_to_float(src[i]);
}

// asm::sum_finite_f32 (src/libdg_mcts/asm/sum_finite.rs:23-58) in the order of its AVX2 form, which is the one an x86 host runs:
// eight lanes, lane l sums the finite elements l, l + 8, ...; then ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)).  n a multiple of 8.
float sum_finite_lanes_scalar(const float* v, int n)
{
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i += 8)
        for (int l = 0; l < 8; ++l) lane[l] += std::isfinite(v[i + l]) ? v[i + l] : 0.0f;
    return ((lane[0] + lane[1]) + (lane[2] + lane[3])) + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
}
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
float sum_finite_lanes_avx2(const float* v, int n)
{
    const __m256 sign = _mm256_set1_ps(-0.0f), inf = _mm256_set1_ps(std::numeric_limits<float>::infinity());
    __m256 acc = _mm256_setzero_ps();
    for (int i = 0; i < n; i += 8) {
        const __m256 x = _mm256_loadu_ps(v + i);
        const __m256 is_inf = _mm256_cmp_ps(_mm256_andnot_ps(sign, x), inf, _CMP_EQ_OQ);
        const __m256 is_number = _mm256_cmp_ps(x, x, _CMP_EQ_OQ);
        acc = _mm256_add_ps(acc, _mm256_andnot_ps(is_inf, _mm256_and_ps(is_number, x)));
    }
    float lane[8];
    _mm256_storeu_ps(lane, acc);
    return ((lane[0] + lane[1]) + (lane[2] + lane[3])) + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
}
#endif
const bool kHaveAvx2 = host_simd();

}  // namespace

// dgnn_internal.h: the one switch of the AVX2 / F16C host paths (DGNN_HOST_SIMD=0 forces the scalar forms)
bool host_simd(bool need_f16c)
{
#if DGNN_X86_SIMD
    static const bool off = [] { const char* e = std::getenv("DGNN_HOST_SIMD"); return e && std::atoi(e) == 0; }();
    if (off) return false;
    return __builtin_cpu_supports("avx2") && (!need_f16c || __builtin_cpu_supports("f16c"));
#else
    (void)need_f16c;
    return false;
#endif
}

float sum_finite(const float* v, int n)
{
#if DGNN_X86_SIMD
    if (kHaveAvx2) return sum_finite_lanes_avx2(v, n);
#endif
    return sum_finite_lanes_scalar(v, n);
}

// policy_helper.rs:87-103: de-transform the network's policy to the identity orientation and add it
void add_valid_candidates(float* dst, const uint16_t* src_f16, const int* indices, int transform)
{
    float src[368];
    halfs_to_floats(src_f16, src, DGNN_NUM_MOVES);
    const uint16_t* inverse = symmetry_map(dgnn_symmetry_inverse(transform));
    dst[DGNN_NUM_POINTS] += src[DGNN_NUM_POINTS];
    for (int i = 0; i < DGNN_NUM_POINTS; ++i) dst[indices[inverse[i]]] += src[i];
}

// policy_helper.rs:112-134 over asm::sum_finite_f32 and asm::normalize_finite_f32 (asm/normalize_finite.rs:51-64): the sum in
// the lane order of the reference's AVX2 form, the scaling as a multiplication by the reciprocal -- the EXACT reciprocal of its
// portable path; its AVX2 path uses the 12-bit _mm_rcp_ps estimate, whose bits differ between CPU vendors.  policy holds 368 floats.
void normalize_policy(float* policy, float sum_to)
{
    const float sum = sum_finite(policy, 368);
    if (sum < 1e-6f) {
        int finite = 0;
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) finite += std::isfinite(policy[i]);
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) if (std::isfinite(policy[i])) policy[i] = sum_to / (float)finite;
    } else {
        const float recip = 1.0f / (sum / sum_to);
        for (int i = 0; i < 368; ++i) policy[i] *= recip;                     // -inf stays -inf
    }
}

namespace {

void rebuild_prior(const dgnn_game_ops* ops, void* board, int to_move, int transform, const uint16_t* policy_f16, float* prior /*368*/)
{
    int indices[DGNN_NUM_MOVES];
    create_initial_policy(ops, board, to_move, prior, indices);
    add_valid_candidates(prior, policy_f16, indices, transform);
    normalize_policy(prior, 1.0f);
}

struct SearchShared {
    dgnn_tree_node* root;
    const dgnn_game_ops* ops;
    void* board;
    Predictor predictor;
    dgnn_cache* cache = nullptr;        // transposition table (needs ops->hash), may be NULL
    int limit;
    int leaves_per_thread = 1;
    std::atomic<int> status{DGNN_OK};
    std::atomic<uint64_t> evals{0}, conflicts{0}, cache_hits{0};
};

// One worker.  The reference's worker has ONE leaf in flight (probe, predict, insert); with leaves_per_thread = K this one
// collects up to K leaves first -- every probe leaves its virtual loss behind, so the next probe goes elsewhere, exactly as
// if K threads had probed one after the other -- and sends them to the predictor in ONE call.  K = 1 is the reference's
// loop.  On a host with few cores this replaces K threads (and K wake-ups per round trip) by one, and it keeps a search
// with one thread deterministic while K leaves are on the device.
void search_worker(SearchShared* s, uint64_t seed)
{
    struct Leaf { void* board; dgnn_trace* trace; int to_move; int transform; uint64_t key; uint64_t generation; };
    const int capacity = std::max(1, s->leaves_per_thread);
    Rng rng{seed};
    std::vector<uint16_t> policy((size_t)capacity * DGNN_NUM_MOVES), value((size_t)capacity);
    std::vector<uint64_t> encoded(((size_t)capacity * s->predictor.item_bytes() + 7) / 8);      // fp16 planes or board records, 8-byte aligned
    uint8_t* const features = reinterpret_cast<uint8_t*>(encoded.data());
    std::vector<Leaf> leaves;
    leaves.reserve((size_t)capacity);
    float prior[368];
    dgnn_board_ops place_ops = {s->ops->place};
    const bool cached = s->cache && s->ops->hash;
    int idle_conflicts = 0;

    auto finish = [&](const Leaf& leaf, uint16_t v, const uint16_t* p) {         // worker_thread.rs:101-113
        rebuild_prior(s->ops, leaf.board, leaf.to_move, leaf.transform, p, prior);
        dgnn_tree_insert(leaf.trace, leaf.to_move, 0.5f * half_to_float(v) + 0.5f, prior);   // predictor.rs:50-52
        dgnn_trace_free(leaf.trace);
        s->ops->destroy(leaf.board);
    };

    bool over = false;
    while (!over && s->status.load(std::memory_order_relaxed) == DGNN_OK) {
        leaves.clear();
        bool conflict = false;
        while ((int)leaves.size() < capacity) {
            // before EVERY probe, as the reference's worker does (worker_thread.rs:149, time_control/mod.rs:86-101); the leaves
            // this worker still holds are not in total_count yet, so they are counted against the limit here
            if (dgnn_rollout_limit_is_done(s->root, s->limit) == 1) { over = true; break; }
            int total = 0;
            dgnn_tree_node_info(s->root, nullptr, &total, nullptr, nullptr, nullptr);
            if (total + (int)leaves.size() >= s->limit) break;
            void* board = s->ops->clone(s->board);
            dgnn_trace* trace = nullptr;
            const int r = dgnn_tree_probe(s->root, &place_ops, board, &trace);
            if (r != DGNN_PROBE_FOUND) {
                s->ops->destroy(board);
                if (r == DGNN_PROBE_CONFLICT) { conflict = true; s->conflicts.fetch_add(1, std::memory_order_relaxed); }   // :161-163
                else over = true;                                             // NoResult: the search is over (:164-166)
                break;
            }
            Leaf leaf{board, trace, 0, 0, 0, 0};
            int last_color = 0;
            dgnn_trace_get(trace, dgnn_trace_len(trace) - 1, nullptr, &last_color, nullptr);
            leaf.to_move = 1 - last_color;                                    // event.rs:48-49
            leaf.transform = (int)(rng.next() % 8);                           // event.rs:47
            leaf.key = cached ? s->ops->hash(board) : 0;
            leaf.generation = cached ? dgnn_cache_generation(s->cache) : 0;   // before the fetch: the answer to this miss belongs to it
            uint16_t v = 0;
            if (cached && dgnn_cache_fetch(s->cache, leaf.key, leaf.to_move, leaf.transform, &v, policy.data()) == 1) {   // event.rs:50-51
                s->cache_hits.fetch_add(1, std::memory_order_relaxed);
                finish(leaf, v, policy.data());
                continue;
            }
            s->predictor.encode(s->ops, board, leaf.to_move, leaf.transform, features, leaves.size());
            leaves.push_back(leaf);
        }

        if (leaves.empty()) {
            if (conflict) {
                // somebody else is expanding what we want: let them -- yield first, then sleep for about the time a leaf
                // evaluation takes, so that the threads of a narrow tree (late in a game) do not spin on cores others need
                if (++idle_conflicts > 8) std::this_thread::sleep_for(std::chrono::microseconds(50));
                else if (idle_conflicts > 2) std::this_thread::yield();
            }
            continue;
        }
        idle_conflicts = 0;

        const int n = (int)leaves.size();
        const int status = s->predictor.predict(features, n, value.data(), policy.data());
        if (status != DGNN_OK) {
            for (const Leaf& leaf : leaves) {
                dgnn_tree_undo(leaf.trace, 1);
                dgnn_trace_free(leaf.trace);
                s->ops->destroy(leaf.board);
            }
            s->status.store(status);
            break;
        }
        s->evals.fetch_add((uint64_t)n, std::memory_order_relaxed);
        for (int i = 0; i < n; ++i) {
            const uint16_t* p = policy.data() + (size_t)i * DGNN_NUM_MOVES;
            if (cached) dgnn_cache_insert_at(s->cache, leaves[i].generation, leaves[i].key, leaves[i].to_move, leaves[i].transform, value[i], p);    // worker_thread.rs:111
            finish(leaves[i], value[i], p);
        }
    }
}

// ---- the rule-free synthetic game -------------------------------------------------------------------------------

struct SyntheticBoard {
    uint64_t hash;
    unsigned char stones[DGNN_NUM_POINTS];
    int moves;
};

void* syn_clone(void* b) { return new SyntheticBoard(*static_cast<SyntheticBoard*>(b)); }
void syn_destroy(void* b) { delete static_cast<SyntheticBoard*>(b); }
void syn_place(void* b, int color, int index)
{
    SyntheticBoard* s = static_cast<SyntheticBoard*>(b);
    if (index >= 0 && index < DGNN_NUM_POINTS) s->stones[index] = (unsigned char)(color + 1);
    Rng mix{s->hash ^ (0x100000001b3ull * (uint64_t)(index * 2 + color + 1))};
    s->hash = mix.next();
    ++s->moves;
}
void syn_features(void* b, int to_move, int transform, uint16_t* out)
{
    const SyntheticBoard* s = static_cast<SyntheticBoard*>(b);
    Rng rng{s->hash ^ ((uint64_t)to_move << 32) ^ ((uint64_t)transform << 40)};
    for (int i = 0; i < DGNN_FEATURE_SIZE; i += 16) {                        // Bernoulli(~0.19) bits, like batch_sizes.rs:44-52
        const uint64_t a = rng.next(), c = rng.next(), d = rng.next();
        const uint64_t bits = a & c & (d | (d >> 7));                        // p = 1/4 * ~3/4
        for (int j = 0; j < 16; ++j) out[i + j] = ((bits >> (j * 3)) & 1u) ? 0x3c00 : 0;
    }
}
void syn_candidates(void* b, int, unsigned char* out)
{
    const SyntheticBoard* s = static_cast<SyntheticBoard*>(b);
    for (int i = 0; i < DGNN_NUM_POINTS; ++i) out[i] = s->stones[i] == 0;
    out[DGNN_NUM_POINTS] = 1;
}

uint64_t syn_hash(void* b) { return static_cast<SyntheticBoard*>(b)->hash; }

const dgnn_game_ops kSyntheticOps = {syn_clone, syn_destroy, syn_place, syn_features, syn_candidates, nullptr, syn_hash};

// ---- predictors ---------------------------------------------------------------------------------------------------

int pool_predict(void* ctx, const uint16_t* features, int n, uint16_t* value, uint16_t* policy)
{
    return dgnn_pool_predict(static_cast<dgnn_pool*>(ctx), features, n, value, policy);
}
int pool_predict_boards(void* ctx, const uint8_t* records, int n, uint16_t* value, uint16_t* policy)
{
    return dgnn_pool_predict_boards(static_cast<dgnn_pool*>(ctx), records, n, value, policy);
}

// A deterministic CPU stand-in for the network (the role of the reference's Random / FakePredictor,
// src/libdg_mcts/predictors/{random,fake}.rs): policy and value are a hash of the feature bits.
int hash_predict(void*, const uint16_t* features, int n, uint16_t* value, uint16_t* policy)
{
    for (int b = 0; b < n; ++b) {
        uint64_t h = 1469598103934665603ull;
        const uint16_t* f = features + (size_t)b * DGNN_FEATURE_SIZE;
        for (int i = 0; i < DGNN_FEATURE_SIZE; ++i) { h ^= f[i]; h *= 1099511628211ull; }
        Rng rng{h};
        float p[DGNN_NUM_MOVES], sum = 0.0f;
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) { const double u = rng.uniform(); p[i] = (float)(u * u * u); sum += p[i]; }
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) policy[(size_t)b * DGNN_NUM_MOVES + i] = float_to_half(p[i] / sum);
        value[b] = float_to_half((float)(2.0 * rng.uniform() - 1.0) * 0.5f);
    }
    return DGNN_OK;
}

// The same stand-in on board records, with a stand-in for the DEVICE behind it: `ctx` (may be NULL) points to a dgnn_dry_device, and
// a call then takes base_us + per_board_us * n of wall time, one call at a time -- like batches queueing on one GPU -- while the
// calling thread sleeps.  `bench.py --dry` measures the host side of self-play against it (rollouts per second and core).
int hash_predict_boards(void* ctx, const uint8_t* records, int n, uint16_t* value, uint16_t* policy)
{
    for (int b = 0; b < n; ++b) {
        uint64_t h = 1469598103934665603ull, words[DGNN_BOARD_RECORD_BYTES / 8];
        std::memcpy(words, records + (size_t)b * DGNN_BOARD_RECORD_BYTES, sizeof(words));
        for (uint64_t w : words) { h ^= w; h *= 1099511628211ull; h ^= h >> 29; }
        uint16_t* p = policy + (size_t)b * DGNN_NUM_MOVES;
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) {                           // fp16 in [2^-10, 2^-8): sums to about 1
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            p[i] = (uint16_t)(0x1400u + ((h >> 40) & 0x7ffu));
        }
        value[b] = (uint16_t)(((h >> 20) & 0x8000u) | 0x2800u | ((h >> 8) & 0x3ffu));   // +-[0.03, 0.06)
    }
    if (ctx) {
        dgnn_dry_device* device = static_cast<dgnn_dry_device*>(ctx);
        static std::mutex one_at_a_time;
        std::lock_guard<std::mutex> lock(one_at_a_time);
        std::this_thread::sleep_for(std::chrono::nanoseconds((long long)(1000.0 * (device->base_us + device->per_board_us * n))));
    }
    return DGNN_OK;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" {

dgnn_predict_boards_fn dgnn_hash_boards_predictor(void) { return hash_predict_boards; }

const dgnn_game_ops* dgnn_synthetic_game_ops(void) { return &kSyntheticOps; }

void* dgnn_synthetic_board_new(uint64_t seed)
{
    SyntheticBoard* b = new (std::nothrow) SyntheticBoard();
    if (!b) return nullptr;
    std::memset(b->stones, 0, sizeof(b->stones));
    b->hash = seed;
    b->moves = 0;
    return b;
}

void dgnn_synthetic_board_free(void* board) { syn_destroy(board); }

dgnn_predict_fn dgnn_hash_predictor(void) { return hash_predict; }
dgnn_predict_fn dgnn_pool_predictor(void) { return pool_predict; }

uint64_t dgnn::Predictor::generation() const
{
    const bool pool = takes_records() ? boards == pool_predict_boards : features == pool_predict;
    return pool ? dgnn_pool_generation(static_cast<const dgnn_pool*>(ctx)) : 0;
}

// 1 when the predictor is a leaf pool that coalesces positions in flight (dgnn_pool_set_coalescing)
bool dgnn::Predictor::coalesces() const
{
    const bool pool = takes_records() ? boards == pool_predict_boards : features == pool_predict;
    return pool && pool_coalesces(static_cast<const dgnn_pool*>(ctx));
}

dgnn_predict_boards_fn dgnn_pool_boards_predictor(void) { return pool_predict_boards; }

int dgnn_search_run(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                    int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out, uint64_t* conflicts_out)
{
    return dgnn_search_run_cached(root, ops, board, predict, predict_ctx, nullptr, num_rollouts, num_threads, seed, evals_out,
                                  conflicts_out, nullptr);
}

// lib.rs:83-130: value and policy of a position as the average over the eight symmetries; cached symmetries are not evaluated
// again, the missing ones go to the predictor as ONE batch and are cached.
int dgnn_full_forward(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_fn predict, void* predict_ctx,
                      dgnn_cache* cache, float* value_out, float* policy_out /* 368 */)
{
    Predictor predictor;
    predictor.features = predict;
    predictor.ctx = predict_ctx;
    return full_forward(ops, board, to_move, predictor, cache, value_out, policy_out);
}

int dgnn_full_forward_boards(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_boards_fn predict, void* predict_ctx,
                             dgnn_cache* cache, float* value_out, float* policy_out /* 368 */)
{
    if (!predict) return DGNN_BAD_ARGUMENT;
    Predictor predictor;
    predictor.boards = predict;
    predictor.ctx = predict_ctx;
    return full_forward(ops, board, to_move, predictor, cache, value_out, policy_out);
}

}  // extern "C"

int dgnn::full_forward(const dgnn_game_ops* ops, void* board, int to_move, const Predictor& predictor, dgnn_cache* cache, float* value_out,
                       float* policy_out /* 368 */)
{
    if (!ops || !ops->candidates || !predictor.usable_with(ops) || !value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    float initial[368];
    int indices[DGNN_NUM_MOVES];
    create_initial_policy(ops, board, to_move, initial, indices);
    for (int i = 0; i < 368; ++i) policy_out[i] = initial[i];
    float value = 0.0f;
    const bool cached = cache && ops->hash;
    const uint64_t key = cached ? ops->hash(board) : 0;

    std::vector<uint16_t> values(8), policies(8 * DGNN_NUM_MOVES);
    std::vector<uint64_t> encoded;
    int missing[8], num_missing = 0;
    auto accumulate = [&](int t, uint16_t v, const uint16_t* p) {
        float one[368];
        for (int i = 0; i < 368; ++i) one[i] = initial[i];
        add_valid_candidates(one, p, indices, t);
        normalize_policy(one, 0.125f);
        value += (0.5f * half_to_float(v) + 0.5f) * 0.125f;
        for (int i = 0; i < DGNN_NUM_MOVES; ++i) policy_out[i] += one[i];
    };
    const uint64_t generation = cached ? dgnn_cache_generation(cache) : 0;
    for (int t = 0; t < 8; ++t) {
        uint16_t v = 0, p[DGNN_NUM_MOVES];
        if (cached && dgnn_cache_fetch(cache, key, to_move, t, &v, p) == 1) { accumulate(t, v, p); continue; }
        if (encoded.empty()) encoded.resize((8 * predictor.item_bytes() + 7) / 8);
        predictor.encode(ops, board, to_move, t, reinterpret_cast<uint8_t*>(encoded.data()), (size_t)num_missing);
        missing[num_missing++] = t;
    }
    if (num_missing > 0) {
        const int status = predictor.predict(reinterpret_cast<const uint8_t*>(encoded.data()), num_missing, values.data(), policies.data());
        if (status != DGNN_OK) return status;
        for (int k = 0; k < num_missing; ++k) {
            accumulate(missing[k], values[k], policies.data() + (size_t)k * DGNN_NUM_MOVES);
            if (cached) dgnn_cache_insert_at(cache, generation, key, to_move, missing[k], values[k], policies.data() + (size_t)k * DGNN_NUM_MOVES);
        }
    }
    *value_out = value;
    return num_missing;       // >= 0: how many symmetries had to be evaluated
}

extern "C" {

// dirichlet::add_ex (dirichlet.rs:40-70): x <- (1 - beta) x + beta Dirichlet(shape) over the finite entries; gamma variates by
// Marsaglia-Tsang with the shape < 1 boost, from a seeded generator instead of thread_rng().
void dgnn_dirichlet_add(float* x, int n, float shape, float beta, uint64_t seed)
{
    if (!x || n <= 0 || !(shape > 0.0f) || !(shape < 1.0f)) return;
    Rng rng{seed};
    auto normal = [&]() {
        const double u1 = std::max(rng.uniform(), 1e-300), u2 = rng.uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    };
    auto gamma = [&](double k) {                       // k < 1: gamma(k + 1) * u^(1 / k)
        const double d = k + 1.0 - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
        for (;;) {
            double xn, v;
            do { xn = normal(); v = 1.0 + c * xn; } while (v <= 0.0);
            v = v * v * v;
            const double u = rng.uniform();
            if (u < 1.0 - 0.0331 * xn * xn * xn * xn || std::log(std::max(u, 1e-300)) < 0.5 * xn * xn + d * (1.0 - v + std::log(v)))
                return d * v * std::pow(std::max(rng.uniform(), 1e-300), 1.0 / k);
        }
    };
    std::vector<double> g((size_t)n, 0.0);
    double g_sum = 0.0;
    int count = 0;
    do {
        g_sum = 0.0; count = 0;
        for (int i = 0; i < n; ++i) if (std::isfinite(x[i])) { g[i] = gamma(shape); g_sum += g[i]; ++count; }
    } while (count > 0 && !(g_sum > std::numeric_limits<double>::min()));
    for (int i = 0; i < n; ++i) if (std::isfinite(x[i])) x[i] = (1.0f - beta) * x[i] + beta * (float)(g[i] / g_sum);
}

int dgnn_search_run_cached(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out,
                           uint64_t* conflicts_out, uint64_t* cache_hits_out)
{
    return dgnn_search_run_batched(root, ops, board, predict, predict_ctx, cache, num_rollouts, num_threads, 1, seed, evals_out,
                                   conflicts_out, cache_hits_out);
}

int dgnn_search_run_batched(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                            dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                            uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out)
{
    Predictor predictor;
    predictor.features = predict;
    predictor.ctx = predict_ctx;
    return search_run(root, ops, board, predictor, cache, num_rollouts, num_threads, leaves_per_thread, seed, evals_out, conflicts_out, cache_hits_out);
}

int dgnn_search_run_boards(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_boards_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                           uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out)
{
    if (!predict) return DGNN_BAD_ARGUMENT;
    Predictor predictor;
    predictor.boards = predict;
    predictor.ctx = predict_ctx;
    return search_run(root, ops, board, predictor, cache, num_rollouts, num_threads, leaves_per_thread, seed, evals_out, conflicts_out, cache_hits_out);
}

}  // extern "C"

int dgnn::search_run(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, const Predictor& predictor, dgnn_cache* cache, int num_rollouts,
                     int num_threads, int leaves_per_thread, uint64_t seed, uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out)
{
    if (!root || !ops || !ops->clone || !ops->destroy || !ops->place || !ops->candidates || !predictor.usable_with(ops) ||
        num_rollouts <= 0 || num_threads <= 0 || leaves_per_thread <= 0 || leaves_per_thread > 64)
        return DGNN_BAD_ARGUMENT;
    SearchShared shared;
    shared.root = root;
    shared.ops = ops;
    shared.board = board;
    shared.predictor = predictor;
    shared.cache = cache;
    shared.limit = num_rollouts;
    shared.leaves_per_thread = leaves_per_thread;
    Rng seeds{seed};
    if (num_threads == 1) {
        search_worker(&shared, seeds.next());
    } else {
        std::vector<std::thread> threads;
        for (int t = 0; t < num_threads; ++t) threads.emplace_back(search_worker, &shared, seeds.next());
        for (std::thread& t : threads) t.join();
    }
    if (evals_out) *evals_out = shared.evals.load();
    if (conflicts_out) *conflicts_out = shared.conflicts.load();
    if (cache_hits_out) *cache_hits_out = shared.cache_hits.load();
    return shared.status.load();
}

extern "C" {

// Synthetic self-play: `num_games` games in parallel (one thread each + `threads_per_search` search threads per move),
// `num_moves` moves per game, `num_rollouts` rollouts per move on a fresh tree, every network call through `predict`.
int dgnn_selfplay_synthetic(dgnn_predict_fn predict, void* predict_ctx, int num_games, int num_moves, int num_rollouts,
                            int threads_per_search, uint64_t seed, dgnn_selfplay_stats* stats)
{
    if (!predict || num_games <= 0 || num_moves <= 0 || num_rollouts <= 0 || threads_per_search <= 0 || !stats)
        return DGNN_BAD_ARGUMENT;
    std::atomic<uint64_t> evals{0}, conflicts{0}, moves{0};
    std::atomic<int> status{DGNN_OK};
    const auto t0 = std::chrono::steady_clock::now();

    auto game = [&](int g) {
        Rng rng{seed + 0x9e3779b97f4a7c15ull * (uint64_t)(g + 1)};
        void* board = dgnn_synthetic_board_new(rng.next());
        int to_move = DGNN_COLOR_BLACK;
        std::vector<uint16_t> features(DGNN_FEATURE_SIZE), policy(DGNN_NUM_MOVES);
        float prior[368];
        for (int m = 0; m < num_moves && status.load() == DGNN_OK; ++m) {
            // the root's own evaluation (the reference averages 8 symmetries here, lib.rs:83-130; one is enough for a
            // throughput harness)
            uint16_t value = 0;
            syn_features(board, to_move, 0, features.data());
            int st = predict(predict_ctx, features.data(), 1, &value, policy.data());
            if (st != DGNN_OK) { status.store(st); break; }
            rebuild_prior(&kSyntheticOps, board, to_move, 0, policy.data(), prior);
            dgnn_tree_node* root = dgnn_tree_node_new(to_move, 0.5f * half_to_float(value) + 0.5f, prior);
            uint64_t e = 0, c = 0;
            st = dgnn_search_run(root, &kSyntheticOps, board, predict, predict_ctx, num_rollouts, threads_per_search, rng.next(), &e, &c);
            evals.fetch_add(e + 1);
            conflicts.fetch_add(c);
            if (st != DGNN_OK) { status.store(st); dgnn_tree_node_free(root); break; }
            float best_value = 0.0f;
            int best = DGNN_NUM_POINTS;
            dgnn_tree_node_best(root, m < 8 ? 0.8f : 0.0f, rng.uniform(), &best_value, &best);   // lib.rs:188-192, TEMPERATURE 0.8
            dgnn_tree_node_free(root);
            syn_place(board, to_move, best);
            to_move = 1 - to_move;
            moves.fetch_add(1);
        }
        dgnn_synthetic_board_free(board);
    };
    std::vector<std::thread> threads;
    for (int g = 0; g < num_games; ++g) threads.emplace_back(game, g);
    for (std::thread& t : threads) t.join();

    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->games = (uint64_t)num_games;
    stats->moves = moves.load();
    stats->evals = evals.load();
    stats->conflicts = conflicts.load();
    stats->cache_hits = 0;
    stats->games_ended = 0;
    stats->ex_it_searches = 0;
    return status.load();
}

}  // extern "C"
