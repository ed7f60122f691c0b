// Scope row f-3: the Go board, the ladder reader and the V1 feature planes of dream-go's libdg_go, as a C++ library behind
// the C ABI -- what produces the INPUT of the hot path.
//
//   rules        src/libdg_go/board_fast.rs:200-470 (Tromp-Taylor placement: opponent blocks without liberties are
//                removed, one's own are NOT -- "place" never checks legality), is_valid (:213-241, no suicide),
//                get_n_liberty (:169-173), get_n_liberty_if (:484-541, including what it does and does not count)
//   super-ko     src/libdg_go/board.rs:120-145: only on points that have held a stone, against the 16 most recent position
//                hashes (small_set.rs:17-47; the set starts out zero-filled, so the empty-board hash is always "seen")
//   ladders      src/libdg_go/utils/ladder.rs:34-170 (capture and escape readers, neighbour order E, N, W, S as in
//                iter/adjacent_iter.rs:41-43)
//   features     src/libdg_go/utils/features.rs:99-251: 32 planes, HWC, fp16
//   candidates   src/libdg_mcts/options.rs:44-48 (pass or a valid move), symmetry::is_symmetric (symmetry.rs:136-144)
// Blocks and liberty counts are recomputed by flood fill after every placement (361 points); the reference maintains them
// incrementally in packed words, with the same observable counts.  Zobrist numbers are this build's own (splitmix64 of a
// fixed seed): only equality of hashes is observable.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "dgnn_internal.h"

namespace dgnn {
namespace {

constexpr int kN = 19;
constexpr int kPoints = 361;
constexpr int kPass = 361;
constexpr int kEmpty = 0;     // stones: 0 empty, 1 black, 2 white  (colour c in the ABI = stone c + 1)

const int kDx[4] = {1, 0, -1, 0};                  // adjacent_iter.rs:42-43
const int kDy[4] = {0, -1, 0, 1};

struct Zobrist {
    uint64_t table[2][kPoints];
    Zobrist()
    {
        uint64_t s = 0x5d6ea60d6ea60ull;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < kPoints; ++i) {
                uint64_t z = (s += 0x9e3779b97f4a7c15ull);
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
                table[c][i] = z ^ (z >> 31);
            }
    }
};
const Zobrist& zobrist() { static const Zobrist z; return z; }

inline int neighbour(int p, int d)
{
    const int x = p % kN + kDx[d], y = p / kN + kDy[d];
    return (x < 0 || x >= kN || y < 0 || y >= kN) ? -1 : y * kN + x;
}

struct Fast {                         // BoardFast
    unsigned char stone[kPoints];
    unsigned char visited[kPoints];
    short head[kPoints];              // block id = smallest point of the block (any unique id would do)
    short libs[kPoints];              // liberties of the block, stored at every stone

    Fast() { std::memset(stone, 0, sizeof(stone)); std::memset(visited, 0, sizeof(visited)); std::memset(head, 0, sizeof(head)); std::memset(libs, 0, sizeof(libs)); }

    void rebuild()
    {
        bool done[kPoints] = {false};
        short lib_stamp[kPoints];                       // liberty already counted for block number lib_stamp[r]
        std::memset(lib_stamp, 0xff, sizeof(lib_stamp));
        short stack[kPoints];
        short members[kPoints];
        for (int p = 0; p < kPoints; ++p) {
            if (stone[p] == kEmpty || done[p]) continue;
            int top = 0, count = 0, nlibs = 0;
            stack[top++] = (short)p;
            done[p] = true;
            while (top > 0) {
                const int q = stack[--top];
                members[count++] = (short)q;
                for (int d = 0; d < 4; ++d) {
                    const int r = neighbour(q, d);
                    if (r < 0) continue;
                    if (stone[r] == kEmpty) { if (lib_stamp[r] != (short)p) { lib_stamp[r] = (short)p; ++nlibs; } }
                    else if (stone[r] == stone[p] && !done[r]) { done[r] = true; stack[top++] = (short)r; }
                }
            }
            for (int i = 0; i < count; ++i) { head[members[i]] = (short)p; libs[members[i]] = (short)nlibs; }
        }
    }

    int n_liberty(int p) const { return libs[p]; }                            // board_fast.rs:169-173
    bool has_n_liberty(int p, int n) const { return libs[p] >= n; }

    bool is_valid(int color, int p) const                                     // board_fast.rs:213-241
    {
        if (stone[p] != kEmpty) return false;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            if (stone[q] == kEmpty) return true;
            if ((stone[q] == color + 1) == has_n_liberty(q, 2)) return true;
        }
        return false;
    }

    // zobrist adjustment of a placement, without placing (board_fast.rs:392-409)
    uint64_t place_if(int color, int p) const
    {
        uint64_t adjust = zobrist().table[color][p];
        int seen[4] = {-1, -1, -1, -1};
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0 || stone[q] != 2 - color || has_n_liberty(q, 2)) continue;
            const int h = head[q];
            if (std::find(seen, seen + 4, h) != seen + 4) continue;
            seen[d] = h;
            for (int r = 0; r < kPoints; ++r) if (stone[r] == 2 - color && head[r] == h) adjust ^= zobrist().table[1 - color][r];
        }
        return adjust;
    }

    // board_fast.rs:420-463: no legality check; opponent blocks left without liberties are captured, one's own are not
    uint64_t place(int color, int p)
    {
        uint64_t hash = zobrist().table[color][p];
        stone[p] = (unsigned char)(color + 1);
        visited[p] = 1;
        rebuild();
        bool captured = false;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0 || stone[q] != 2 - color || libs[q] != 0) continue;
            const int h = head[q];
            for (int r = 0; r < kPoints; ++r)
                if (stone[r] == 2 - color && head[r] == h) { stone[r] = kEmpty; hash ^= zobrist().table[1 - color][r]; }
            captured = true;
        }
        if (captured) rebuild();
        return hash;
    }

    int n_liberty_if(int color, int p) const                                  // board_fast.rs:484-541
    {
        bool seen[kPoints] = {false};
        int num = 0;
        seen[p] = true;
        int captured[4] = {-1, -1, -1, -1}, connected[4] = {-1, -1, -1, -1};
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            if (stone[q] == color + 1) {
                if (std::find(connected, connected + 4, (int)head[q]) == connected + 4) connected[d] = head[q];
            } else if (stone[q] == 2 - color) {
                if (libs[q] == 1) { captured[d] = head[q]; seen[q] = true; ++num; }
            } else {
                seen[q] = true;
                ++num;
            }
        }
        for (int c = 0; c < 4; ++c) {
            if (connected[c] < 0) continue;
            for (int s = 0; s < kPoints; ++s) {
                if (stone[s] != color + 1 || head[s] != connected[c]) continue;
                for (int d = 0; d < 4; ++d) {
                    const int a = neighbour(s, d);
                    if (a < 0 || seen[a]) continue;
                    seen[a] = true;
                    if (stone[a] == kEmpty) ++num;
                    else if (std::find(captured, captured + 4, (int)head[a]) != captured + 4 && stone[a] == 2 - color) ++num;
                }
            }
        }
        return num;
    }

    int a_liberty(int p) const                                                // board_fast.rs:181-191
    {
        for (int s = 0; s < kPoints; ++s) {
            if (stone[s] != stone[p] || head[s] != head[p]) continue;
            for (int d = 0; d < 4; ++d) { const int q = neighbour(s, d); if (q >= 0 && stone[q] == kEmpty) return q; }
        }
        return -1;
    }
};

// ---- ladders (utils/ladder.rs) ----

bool can_escape_with_capture(const Fast& b, int color, int p)                 // ladder.rs:34-42
{
    for (int s = 0; s < kPoints; ++s) {
        if (b.stone[s] != b.stone[p] || b.head[s] != b.head[p]) continue;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(s, d);
            if (q >= 0 && b.stone[q] == 2 - color && !b.has_n_liberty(q, 2)) return true;
        }
    }
    return false;
}

// A stone takes exactly one liberty from each adjacent block, so the placement leaves an opponent block in atari only if
// one with exactly two liberties is adjacent: without one the reader answers "no" before it copies the board.
bool may_start_ladder(const Fast& b, int color, int p)
{
    for (int d = 0; d < 4; ++d) {
        const int q = neighbour(p, d);
        if (q >= 0 && b.stone[q] == 2 - color && b.libs[q] == 2) return true;
    }
    return false;
}

bool ladder_capture_impl(Fast b, int color, int p, int depth);

bool ladder_capture(const Fast& b, int color, int p, int depth = 0)           // ladder.rs:53-122
{
    if (depth > 200 || !may_start_ladder(b, color, p)) return false;          // (the reference recurses unbounded)
    return ladder_capture_impl(b, color, p, depth);
}

bool ladder_capture_impl(Fast b, int color, int p, int depth)
{
    b.place(color, p);
    const int opponent = 1 - color;
    int extend = -1;
    for (int d = 0; d < 4 && extend < 0; ++d) {
        const int q = neighbour(p, d);
        if (q < 0 || b.stone[q] != opponent + 1) continue;
        if (!b.has_n_liberty(q, 2) && !can_escape_with_capture(b, opponent, q)) {
            const int lib = b.a_liberty(q);
            if (lib >= 0 && b.is_valid(opponent, lib)) extend = lib;
        }
    }
    if (extend < 0) return false;
    b.place(opponent, extend);
    if (!b.has_n_liberty(extend, 2)) return true;
    if (b.has_n_liberty(extend, 3)) return false;
    for (int d = 0; d < 4; ++d) {                                             // did the extension put one of my stones in atari?
        const int q = neighbour(extend, d);
        if (q >= 0 && b.stone[q] == color + 1 && !b.has_n_liberty(q, 2)) return false;
    }
    for (int d = 0; d < 4; ++d) {
        const int q = neighbour(extend, d);
        if (q >= 0 && b.is_valid(color, q) && ladder_capture(b, color, q, depth + 1)) return true;
    }
    return false;
}

bool ladder_escape(const Fast& self, int color, int p)                        // ladder.rs:145-170
{
    bool connected_to_one = false;
    for (int d = 0; d < 4; ++d) {
        const int q = neighbour(p, d);
        if (q >= 0 && self.stone[q] == color + 1 && !self.has_n_liberty(q, 2)) { connected_to_one = true; break; }
    }
    if (!connected_to_one) return false;
    Fast b = self;
    b.place(color, p);
    if (b.n_liberty(p) != 2) return false;
    for (int d = 0; d < 4; ++d) {
        const int q = neighbour(p, d);
        if (q < 0) continue;
        if (b.is_valid(1 - color, q) && ladder_capture(b, 1 - color, q)) return false;
    }
    return true;
}

}  // namespace

struct GoBoard {                       // Board, board.rs:28-50
    Fast inner;
    int history[8];                    // most recent first; -1 = nothing
    uint64_t hash = 0;
    uint64_t hash_history[16];         // small_set.rs:17-47
    int hash_count = 0;
    float komi;
    int count = 0;
    int last_played = -1;

    explicit GoBoard(float k) : komi(k)
    {
        for (int& h : history) h = -1;
        std::memset(hash_history, 0, sizeof(hash_history));
    }

    bool is_ko(int color, int p) const                                        // board.rs:120-130
    {
        if (!inner.visited[p]) return false;
        const uint64_t next = hash ^ inner.place_if(color, p);
        for (uint64_t h : hash_history) if (h == next) return true;
        return false;
    }
    bool is_valid(int color, int p) const { return inner.is_valid(color, p) && !is_ko(color, p); }   // board.rs:140-142

    void place(int color, int p)                                              // board.rs:152-176
    {
        hash ^= inner.place(color, p);
        last_played = color;
        ++count;
        for (int i = 7; i > 0; --i) history[i] = history[i - 1];
        history[0] = p;
        hash_history[hash_count] = hash;
        hash_count = (hash_count + 1) % 16;
    }
};

namespace {

// features::V1::get_features::<HWC, f16> (features.rs:99-251)
void features_v1(const GoBoard& b, int to_move, int symmetry, uint16_t* out)
{
    const uint16_t one = 0x3c00;
    std::memset(out, 0, DGNN_FEATURE_SIZE * sizeof(uint16_t));
    auto at = [&](int plane, int point) -> uint16_t& { return out[32 * dgnn_symmetry_apply(symmetry, point) + plane]; };
    const int opponent = 1 - to_move;

    for (int i = 0; i < 2; ++i)                                               // planes 3, 4: the two most recent moves
        if (b.history[i] >= 0 && b.history[i] < kPoints) at(3 + i, b.history[i]) = one;

    for (int p = 0; p < kPoints; ++p) {                                       // liberties
        if (b.inner.stone[p] != kEmpty) {
            const int start = b.inner.stone[p] == to_move + 1 ? 5 : 17;
            const int n = std::min(b.inner.n_liberty(p), 6);
            for (int i = 0; i < n; ++i) at(start + i, p) = one;
        } else {
            if (b.inner.is_valid(to_move, p)) {
                const int n = std::min(b.inner.n_liberty_if(to_move, p), 6);
                for (int i = 0; i < n; ++i) at(11 + i, p) = one;
            }
            if (b.inner.is_valid(opponent, p)) {
                const int n = std::min(b.inner.n_liberty_if(opponent, p), 6);
                for (int i = 0; i < n; ++i) at(23 + i, p) = one;
            }
        }
    }

    bool any_ko = false;                                                      // vertex properties
    for (int p = 0; p < kPoints; ++p) {
        if (b.inner.stone[p] != kEmpty || !b.inner.is_valid(to_move, p)) continue;
        if (b.is_ko(to_move, p)) { any_ko = true; at(29, p) = one; }
        if (ladder_capture(b.inner, to_move, p)) at(30, p) = one;
        if (ladder_escape(b.inner, to_move, p)) at(31, p) = one;
    }

    const float komi = std::max(0.0f, std::min(1.0f, 0.5f + (0.5f * b.komi) / 7.5f));   // global properties
    const uint16_t c_komi = float_to_half(komi);
    for (int p = 0; p < kPoints; ++p) {
        at(0, p) = to_move == 0 ? c_komi : 0;
        at(1, p) = to_move == 1 ? c_komi : 0;
        at(2, p) = any_ko ? one : 0;
    }
}

// ---- game callbacks for the search driver ----
void* go_clone(void* b) { return new GoBoard(*static_cast<GoBoard*>(b)); }
void go_destroy(void* b) { delete static_cast<GoBoard*>(b); }
void go_place(void* b, int color, int index) { if (index >= 0 && index < kPoints) static_cast<GoBoard*>(b)->place(color, index); }
void go_features(void* b, int to_move, int symmetry, uint16_t* out) { features_v1(*static_cast<GoBoard*>(b), to_move, symmetry, out); }
void go_candidates(void* b, int to_move, unsigned char* out)                  // options.rs:44-48
{
    const GoBoard* g = static_cast<GoBoard*>(b);
    for (int p = 0; p < kPoints; ++p) out[p] = g->is_valid(to_move, p);
    out[kPass] = 1;
}
int go_is_symmetric(void* b, int symmetry)                                    // symmetry.rs:136-144
{
    const GoBoard* g = static_cast<GoBoard*>(b);
    for (int p = 0; p < kPoints; ++p) if (g->inner.stone[p] != g->inner.stone[dgnn_symmetry_apply(symmetry, p)]) return 0;
    return 1;
}
uint64_t go_hash(void* b) { return static_cast<GoBoard*>(b)->hash; }      // BoardTuple, predictors/nn.rs:31-45
const dgnn_game_ops kGoOps = {go_clone, go_destroy, go_place, go_features, go_candidates, go_is_symmetric, go_hash};

struct Rng2 {
    uint64_t state;
    uint64_t next() { uint64_t z = (state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" {

dgnn_go_board* dgnn_go_board_new(float komi) { return reinterpret_cast<dgnn_go_board*>(new (std::nothrow) GoBoard(komi)); }
dgnn_go_board* dgnn_go_board_clone(const dgnn_go_board* b) { return b ? reinterpret_cast<dgnn_go_board*>(new (std::nothrow) GoBoard(*reinterpret_cast<const GoBoard*>(b))) : nullptr; }
void dgnn_go_board_free(dgnn_go_board* b) { delete reinterpret_cast<GoBoard*>(b); }

static bool on_board(int x, int y) { return x >= 0 && x < kN && y >= 0 && y < kN; }
#define GO(b) reinterpret_cast<GoBoard*>(b)
#define CGO(b) reinterpret_cast<const GoBoard*>(b)

int dgnn_go_place(dgnn_go_board* b, int color, int x, int y)
{
    if (!b || (color != 0 && color != 1) || !on_board(x, y)) return DGNN_BAD_ARGUMENT;
    GO(b)->place(color, y * kN + x);
    return DGNN_OK;
}
int dgnn_go_at(const dgnn_go_board* b, int x, int y) { return (!b || !on_board(x, y)) ? -2 : (int)CGO(b)->inner.stone[y * kN + x] - 1; }
int dgnn_go_is_valid(const dgnn_go_board* b, int color, int x, int y) { return (b && on_board(x, y) && (color == 0 || color == 1)) ? CGO(b)->is_valid(color, y * kN + x) : 0; }
int dgnn_go_to_move(const dgnn_go_board* b) { return (!b || CGO(b)->last_played < 0) ? 0 : 1 - CGO(b)->last_played; }   // board.rs:100-105
int dgnn_go_count(const dgnn_go_board* b) { return b ? CGO(b)->count : 0; }
uint64_t dgnn_go_hash(const dgnn_go_board* b) { return b ? CGO(b)->hash : 0; }
int dgnn_go_n_liberty(const dgnn_go_board* b, int x, int y) { return (b && on_board(x, y) && CGO(b)->inner.stone[y * kN + x]) ? CGO(b)->inner.n_liberty(y * kN + x) : -1; }
int dgnn_go_n_liberty_if(const dgnn_go_board* b, int color, int x, int y)
{
    if (!b || !on_board(x, y) || !CGO(b)->inner.is_valid(color, y * kN + x)) return -1;
    return CGO(b)->inner.n_liberty_if(color, y * kN + x);
}
int dgnn_go_is_ladder_capture(const dgnn_go_board* b, int color, int x, int y)
{
    if (!b || !on_board(x, y) || !CGO(b)->inner.is_valid(color, y * kN + x)) return -1;
    return ladder_capture(CGO(b)->inner, color, y * kN + x);
}
int dgnn_go_is_ladder_escape(const dgnn_go_board* b, int color, int x, int y)
{
    if (!b || !on_board(x, y) || !CGO(b)->inner.is_valid(color, y * kN + x)) return -1;
    return ladder_escape(CGO(b)->inner, color, y * kN + x);
}
int dgnn_go_features(const dgnn_go_board* b, int to_move, int symmetry, uint16_t* out)
{
    if (!b || !out || (to_move != 0 && to_move != 1) || symmetry < 0 || symmetry >= 8) return DGNN_BAD_ARGUMENT;
    features_v1(*CGO(b), to_move, symmetry, out);
    return DGNN_OK;
}
int dgnn_go_is_symmetric(const dgnn_go_board* b, int symmetry) { return (b && symmetry >= 0 && symmetry < 8) ? go_is_symmetric((void*)b, symmetry) : 0; }

const dgnn_game_ops* dgnn_go_game_ops(void) { return &kGoOps; }

// Self-play on the real board, following self_play_one / Player::predict / predict (src/libdg_mcts/self_play.rs:423-459,
// 331-368, lib.rs:144-200): random komi, per-player trees re-used across moves (Node::forward), win-rate driven rollout
// budget max(0.1, 4w(1-w)) * num_rollouts, 8-symmetry root evaluation through the transposition cache, dirichlet noise on
// the root prior, temperature 0.8 for the first eight stones, every network call through `predict`.
// What is missing: scoring (is_scorable / Benson / ScoringSearch), so passing is disqualified at every move and a game is
// `num_moves` stones long (or ends when nothing is legal); resignation; expert iteration.
// moves_out (num_games * num_moves packed indices, -1 = not played) and sgf_out (one "(;GM[1]FF[4]SZ[19]KM[..]...)" record
// per game, newline separated, cut off at sgf_capacity) may be NULL.
int dgnn_selfplay_go(dgnn_predict_fn predict, void* predict_ctx, dgnn_cache* cache, int num_games, int num_moves, int num_rollouts,
                     int threads_per_search, uint64_t seed, dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity)
{
    if (!predict || num_games <= 0 || num_moves <= 0 || num_rollouts <= 0 || threads_per_search <= 0 || !stats) return DGNN_BAD_ARGUMENT;
    std::atomic<uint64_t> evals{0}, conflicts{0}, moves{0}, hits{0};
    std::atomic<int> status{DGNN_OK};
    std::vector<std::string> records((size_t)num_games);
    const auto t0 = std::chrono::steady_clock::now();

    auto game = [&](int g) {
        Rng2 rng{seed + 0x9e3779b97f4a7c15ull * (uint64_t)(g + 1)};
        const double kv = rng.uniform();                                       // get_random_komi, lib.rs:210-224
        const float komi = kv < 0.4 ? 7.5f : kv < 0.8 ? 6.5f : kv < 0.9 ? 0.5f : (float)((int)(rng.next() % 16) - 8) + 0.5f;
        GoBoard board(komi);
        dgnn_tree_node* roots[2] = {nullptr, nullptr};                         // Player::root of black / white
        float winrate[2] = {0.5f, 0.5f};                                       // MovingAverage::new(0.5, 0.2)
        char header[64];
        std::snprintf(header, sizeof(header), "(;GM[1]FF[4]SZ[19]RU[Chinese]KM[%.1f]", (double)komi);   // game_result.rs:33,40 without DT / RE
        std::string& sgf = records[(size_t)g];
        sgf = header;
        if (moves_out) for (int m = 0; m < num_moves; ++m) moves_out[(size_t)g * num_moves + m] = -1;

        for (int m = 0; m < num_moves && m < 722 && status.load() == DGNN_OK; ++m) {
            const int me = m & 1;                                              // black starts, no passes: colours alternate
            const float w = winrate[me];
            const float scale = std::max(0.1f, 4.0f * w * (1.0f - w));        // Player::num_rollout, self_play.rs:234-241
            const int limit = std::max(2, (int)(scale * (float)num_rollouts));

            float value = 0.0f, prior[368];
            int st = dgnn_full_forward(&kGoOps, &board, me, predict, predict_ctx, cache, &value, prior);   // lib.rs:154-159
            if (st < 0) { status.store(st); break; }
            evals.fetch_add((uint64_t)st);
            prior[kPass] = -std::numeric_limits<float>::infinity();            // allow_pass == false (no scoring yet)
            bool any = false;
            for (int i = 0; i < kPoints; ++i) any = any || std::isfinite(prior[i]);
            if (!any) break;
            normalize_policy(prior, 1.0f);
            dgnn_dirichlet_add(prior, DGNN_NUM_MOVES, 0.03f, 0.25f, rng.next());   // lib.rs:163-165

            dgnn_tree_node* root = roots[me];                                  // lib.rs:170-186: re-use the tree or start one
            roots[me] = nullptr;
            if (root) dgnn_tree_node_set_prior(root, prior);
            else root = dgnn_tree_node_new(me, value, prior);
            dgnn_tree_node_disqualify(root, kPass);                            // self_play.rs:253-256
            uint64_t e = 0, c = 0, h = 0;
            st = dgnn_search_run_cached(root, &kGoOps, &board, predict, predict_ctx, cache, limit, threads_per_search, rng.next(), &e, &c, &h);
            evals.fetch_add(e);
            conflicts.fetch_add(c);
            hits.fetch_add(h);
            if (st != DGNN_OK) { status.store(st); dgnn_tree_node_free(root); break; }

            float best_value = 0.0f;
            int best = kPass;
            dgnn_tree_node_best(root, board.count < 8 ? 0.8f : 0.0f, rng.uniform(), &best_value, &best);   // lib.rs:188-192
            if (best == kPass || !board.is_valid(me, best)) { dgnn_tree_node_free(root); break; }

            float softmax[DGNN_NUM_MOVES];                                      // Played::from_mcts -> the f-4 record
            dgnn_tree_node_softmax(root, softmax);
            int total = 0;
            dgnn_tree_node_info(root, nullptr, &total, nullptr, nullptr, nullptr);
            char move[2048];
            dgnn_sgf_played(move, sizeof(move), me, best, total, softmax, 1, best_value, -1, "");
            sgf += move;

            winrate[me] -= 0.2f * (winrate[me] - best_value);                  // MovingAverage::update, self_play.rs:56-58
            roots[me] = dgnn_tree_node_forward(root, best);                    // self_play.rs:366, 447
            if (roots[1 - me]) roots[1 - me] = dgnn_tree_node_forward(roots[1 - me], best);
            board.place(me, best);
            if (moves_out) moves_out[(size_t)g * num_moves + m] = best;
            moves.fetch_add(1);
        }
        sgf += ")";
        dgnn_tree_node_free(roots[0]);
        dgnn_tree_node_free(roots[1]);
    };
    std::vector<std::thread> threads;
    for (int g = 0; g < num_games; ++g) threads.emplace_back(game, g);
    for (std::thread& t : threads) t.join();
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->games = (uint64_t)num_games;
    stats->moves = moves.load();
    stats->evals = evals.load();
    stats->conflicts = conflicts.load();
    stats->cache_hits = hits.load();
    if (sgf_out && sgf_capacity > 0) {
        size_t used = 0;
        for (const std::string& r : records) {
            if (used + r.size() + 2 > sgf_capacity) break;
            std::memcpy(sgf_out + used, r.data(), r.size());
            used += r.size();
            sgf_out[used++] = '\n';
        }
        sgf_out[used] = '\0';
    }
    return status.load();
}

}  // extern "C"
