// Scope row f-3: the Go board, the ladder reader, the V1 feature planes, Benson / scoring of dream-go's libdg_go, and the
// self-play game loop of libdg_mcts on top of them, as a C++ library behind the C ABI -- what produces the INPUT of the hot
// path and turns its outputs into game records.
//
//   rules        src/libdg_go/board_fast.rs:200-470 (Tromp-Taylor placement: opponent blocks without liberties are
//                removed, one's own are NOT -- "place" never checks legality), is_valid (:213-241, no suicide),
//                get_n_liberty (:169-173), get_n_liberty_if (:484-541, including what it does and does not count)
//   super-ko     src/libdg_go/board.rs:120-145: only on points that have held a stone, against the 16 most recent position
//                hashes (small_set.rs:17-47; the set starts out zero-filled, so the empty-board hash is always "seen")
//   ladders      src/libdg_go/utils/ladder.rs:34-170 (capture and escape readers, neighbour order E, N, W, S as in
//                iter/adjacent_iter.rs:41-43)
//   features     src/libdg_go/utils/features.rs:99-251: 32 planes, HWC, fp16
//   candidates   src/libdg_mcts/options.rs:44-214 (StandardSearch: pass or a valid move; ScoringSearch: no pass, no Benson
//                eye, no own simple eye), symmetry::is_symmetric (symmetry.rs:136-144)
//   scoring      src/libdg_go/utils/benson.rs:55-333, utils/score.rs:105-283, src/libdg_mcts/game_result.rs:26-93
//   self-play    src/libdg_mcts/self_play.rs:216-505, lib.rs:144-224
// Blocks (head id + circular stone list) and liberty counts are kept incrementally by place(), like the reference's packed
// words, with the same observable counts; Fast::consistent() is the from-scratch flood fill they are tested against.
// Zobrist numbers are this build's own (splitmix64 of a fixed seed): only equality of hashes is observable; the positions
// of the reference's three real-game tests reproduce its expected hashes under its own table (tests/golden/).
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <ctime>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "dgnn_internal.h"
#include "fiber.h"

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

struct Neighbours {
    short table[kPoints][4];
    Neighbours()
    {
        for (int p = 0; p < kPoints; ++p)
            for (int d = 0; d < 4; ++d) {
                const int x = p % kN + kDx[d], y = p / kN + kDy[d];
                table[p][d] = (short)((x < 0 || x >= kN || y < 0 || y >= kN) ? -1 : y * kN + x);
            }
    }
};
const Neighbours kNeighbours;
inline int neighbour(int p, int d) { return kNeighbours.table[p][d]; }

struct Fast {                         // BoardFast
    unsigned char stone[kPoints];
    unsigned char visited[kPoints];
    short head[kPoints];              // block id = one of the block's points (any unique id would do)
    short libs[kPoints];              // liberties of the block, valid at the block's head stone only
    short next[kPoints];              // circular list of the block's stones (the reference's next_link, board_fast.rs:37-60)

    Fast() { std::memset(stone, 0, sizeof(stone)); std::memset(visited, 0, sizeof(visited)); std::memset(head, 0, sizeof(head)); std::memset(libs, 0, sizeof(libs)); std::memset(next, 0, sizeof(next)); }

    // From-scratch flood fill of blocks and liberties: the checker of the incremental bookkeeping in place()
    // (dgnn_go_check, tests/test_go_host.py), not used while playing.
    bool consistent() const
    {
        bool done[kPoints] = {false};
        short lib_stamp[kPoints];                       // liberty already counted for block number lib_stamp[r]
        std::memset(lib_stamp, 0xff, sizeof(lib_stamp));
        short stack[kPoints];
        for (int p = 0; p < kPoints; ++p) {
            if (stone[p] == kEmpty || done[p]) continue;
            int top = 0, count = 0, nlibs = 0;
            stack[top++] = (short)p;
            done[p] = true;
            while (top > 0) {
                const int q = stack[--top];
                ++count;
                if (head[q] != head[p]) return false;                           // one block, one id
                for (int d = 0; d < 4; ++d) {
                    const int r = neighbour(q, d);
                    if (r < 0) continue;
                    if (stone[r] == kEmpty) { if (lib_stamp[r] != (short)p) { lib_stamp[r] = (short)p; ++nlibs; } }
                    else if (stone[r] == stone[p] && !done[r]) { done[r] = true; stack[top++] = (short)r; }
                }
            }
            const int h = head[p];
            if (h < 0 || h >= kPoints || stone[h] != stone[p] || head[h] != h || libs[h] != nlibs) return false;
            int listed = 0, r = p;                                              // the circular list holds exactly the block
            do { if (stone[r] != stone[p] || head[r] != h || ++listed > count) return false; r = next[r]; } while (r != p);
            if (listed != count) return false;
        }
        for (int p = 0; p < kPoints; ++p)                                       // the shortcuts of n_liberty_if against its general path
            if (stone[p] == kEmpty)
                for (int c = 0; c < 2; ++c) if (n_liberty_if_impl(c, p, true) != n_liberty_if_impl(c, p, false)) return false;
        return true;
    }

    int n_liberty(int p) const { return libs[head[p]]; }                      // board_fast.rs:169-173
    bool has_n_liberty(int p, int n) const { return libs[head[p]] >= n; }

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
            int r = q;
            do { adjust ^= zobrist().table[1 - color][r]; r = next[r]; } while (r != q);
        }
        return adjust;
    }

    // board_fast.rs:420-463: no legality check; opponent blocks left without liberties are captured, one's own are not.
    // Incremental like the reference's: the stone takes a liberty from each adjacent opponent block, blocks at zero are
    // lifted (every lifted stone is a new liberty of each of its own-colour neighbours' blocks), adjacent own blocks are
    // spliced into one list, and only the merged block's liberties are counted again.
    uint64_t place(int color, int p)
    {
        static thread_local unsigned int stamp[kPoints];
        static thread_local unsigned int serial = 0;
        const unsigned char mine = (unsigned char)(color + 1), theirs = (unsigned char)(2 - color);
        uint64_t hash = zobrist().table[color][p];
        stone[p] = mine;
        visited[p] = 1;
        head[p] = (short)p;
        next[p] = (short)p;
        libs[p] = 0;

        int seen[4], n_seen = 0;
        for (int d = 0; d < 4; ++d) {                                          // one liberty less for every adjacent opponent block
            const int q = neighbour(p, d);
            if (q < 0 || stone[q] != theirs) continue;
            const int h = head[q];
            if (std::find(seen, seen + n_seen, h) != seen + n_seen) continue;
            seen[n_seen++] = h;
            if (--libs[h] > 0) continue;
            int r = q;                                                         // captured
            do {
                stone[r] = kEmpty;
                hash ^= zobrist().table[1 - color][r];
                int near[4], n_near = 0;
                for (int e = 0; e < 4; ++e) {
                    const int a = neighbour(r, e);
                    if (a < 0 || stone[a] != mine) continue;
                    const int ha = head[a];
                    if (std::find(near, near + n_near, ha) != near + n_near) continue;
                    near[n_near++] = ha;
                    ++libs[ha];
                }
                r = next[r];
            } while (r != q);
        }

        for (int d = 0; d < 4; ++d) {                                          // splice the adjacent own blocks into p's list
            const int q = neighbour(p, d);
            if (q < 0 || stone[q] != mine || head[q] == p) continue;
            int r = q;
            do { head[r] = (short)p; r = next[r]; } while (r != q);
            std::swap(next[p], next[q]);
        }

        if (++serial == 0) { std::memset(stamp, 0, sizeof(stamp)); serial = 1; }
        int nlibs = 0, r = p;                                                  // liberties of the merged block
        do {
            for (int d = 0; d < 4; ++d) {
                const int a = neighbour(r, d);
                if (a >= 0 && stone[a] == kEmpty && stamp[a] != serial) { stamp[a] = serial; ++nlibs; }
            }
            r = next[r];
        } while (r != p);
        libs[p] = (short)nlibs;
        return hash;
    }

    int n_liberty_if(int color, int p) const { return n_liberty_if_impl(color, p, true); }   // board_fast.rs:484-541

    int n_liberty_if_impl(int color, int p, bool shortcuts) const
    {
        // what touches p: empty points, own blocks it would join, opponent blocks it would capture
        int empties[4], n_empty = 0, own[4], own_at[4], n_own = 0;
        bool captures = false;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            if (stone[q] == kEmpty) empties[n_empty++] = q;
            else if (stone[q] == color + 1) {
                const int h = head[q];
                if (std::find(own, own + n_own, h) == own + n_own) { own[n_own] = h; own_at[n_own++] = q; }
            } else if (libs[head[q]] == 1) captures = true;
        }
        if (shortcuts && !captures && n_own == 0) return n_empty;             // a lone stone
        if (shortcuts && !captures && n_own == 1) {
            // one block joined, nothing captured: its liberties without p, plus p's empty neighbours it does not touch yet
            int num = libs[own[0]] - 1;
            for (int i = 0; i < n_empty; ++i) {
                bool touches = false;
                for (int d = 0; d < 4 && !touches; ++d) {
                    const int a = neighbour(empties[i], d);
                    touches = a >= 0 && a != p && stone[a] == color + 1 && head[a] == own[0];
                }
                if (!touches) ++num;
            }
            return num;
        }

        // the general case: the union of everything, counted once with a per-call stamp
        static thread_local unsigned int stamp[kPoints];                      // "seen in call number serial"
        static thread_local unsigned int serial = 0;
        if (++serial == 0) { std::memset(stamp, 0, sizeof(stamp)); serial = 1; }
        int num = 0;
        stamp[p] = serial;
        int captured[4] = {-1, -1, -1, -1};
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            if (stone[q] == 2 - color) {
                if (libs[head[q]] == 1) { captured[d] = head[q]; stamp[q] = serial; ++num; }
            } else if (stone[q] == kEmpty) {
                stamp[q] = serial;
                ++num;
            }
        }
        for (int c = 0; c < n_own; ++c) {
            int s = own_at[c];
            do {
                for (int d = 0; d < 4; ++d) {
                    const int a = neighbour(s, d);
                    if (a < 0 || stamp[a] == serial) continue;
                    stamp[a] = serial;
                    if (stone[a] == kEmpty) ++num;
                    else if (stone[a] == 2 - color && std::find(captured, captured + 4, (int)head[a]) != captured + 4) ++num;
                }
                s = next[s];
            } while (s != own_at[c]);
        }
        return num;
    }

    int a_liberty(int p) const                                                // board_fast.rs:181-191
    {
        int s = p;
        do {
            for (int d = 0; d < 4; ++d) { const int q = neighbour(s, d); if (q >= 0 && stone[q] == kEmpty) return q; }
            s = next[s];
        } while (s != p);
        return -1;
    }
};

// ---- ladders (utils/ladder.rs) ----

bool can_escape_with_capture(const Fast& b, int color, int p)                 // ladder.rs:34-42
{
    int s = p;
    do {
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(s, d);
            if (q >= 0 && b.stone[q] == 2 - color && !b.has_n_liberty(q, 2)) return true;
        }
        s = b.next[s];
    } while (s != p);
    return false;
}

// A stone takes exactly one liberty from each adjacent block, so the placement leaves an opponent block in atari only if
// one with exactly two liberties is adjacent: without one the reader answers "no" before it copies the board.
bool may_start_ladder(const Fast& b, int color, int p)
{
    for (int d = 0; d < 4; ++d) {
        const int q = neighbour(p, d);
        if (q >= 0 && b.stone[q] == 2 - color && b.n_liberty(q) == 2) return true;
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
    const uint16_t* map = symmetry_map(symmetry);
    auto at = [&](int plane, int point) -> uint16_t& { return out[32 * map[point] + plane]; };
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


// ---- the compact board record (dgnn_forward_boards): what the DEVICE needs to build the 32 planes itself ----
//
// features_v1 above costs 10-45 us of host time per leaf and ships 23 KB; the record costs well under a microsecond and 896
// bytes.  It holds what the host keeps incrementally anyway -- stones and block ids -- plus the three planes whose definition
// needs history or look-ahead (super-ko against the hash ring, the two ladder readers), restricted to the few points where they
// can be set at all.  Liberty counts, liberties-after-move, validity, the move history planes and the constant planes are
// computed from it on the device (csrc/features.hip), bit-identical to features_v1 (tests/test_features_gpu.py), and on the
// host by record_to_packed below (the statement of the same algorithm the CPU tests check against features_v1).
struct BoardRecord {
    uint16_t cell[368];            // bits 0-1: 0 empty / 1 black / 2 white; bits 2-10: block id (a point of the block); 361..367 zero
    uint32_t ko[12];               // bit p: p is empty, not suicide for to_move, and playing there repeats a position (board.rs:120-130)
    uint32_t ladder_capture[12];   // bit p: ... and it captures in a ladder (utils/ladder.rs:124-136)
    uint32_t ladder_escape[12];    // bit p: ... and it escapes a ladder (utils/ladder.rs:138-170)
    uint8_t to_move, symmetry;
    uint16_t komi;                 // fp16 bits of clamp(0.5 + 0.5 * komi / 7.5, 0, 1) (features.rs:236-239)
    int16_t last[2];               // the two most recent moves, -1 = none / pass
    uint8_t reserved[8];
};
static_assert(sizeof(BoardRecord) == DGNN_BOARD_RECORD_BYTES, "BoardRecord layout");

void board_record(const GoBoard& b, int to_move, int symmetry, BoardRecord* out);    // (below, behind the bit-set scans it uses)

// The planes from a record alone: the host statement of csrc/features.hip (same steps, same order), producing the packed
// form of dgnn_forward_packed (361 plane masks + the komi word).  Used by the CPU tests (against features_v1 on every position of
// the committed games) and by leaf pools whose lanes are predictor CALLBACKS (dry runs, the oracle as the predictor); a pool over
// devices never calls it.
void record_to_packed(const BoardRecord& r, uint32_t* out /* DGNN_PACKED_WORDS */)
{
    struct Set { uint32_t w[12]; };
    static thread_local Set nb[kPoints], st[kPoints];      // per block id: every neighbour of its stones / its stones
    unsigned char stone[kPoints];
    short head[kPoints], libs[kPoints];
    Set empty;
    std::memset(&empty, 0, sizeof(empty));
    for (int p = 0; p < kPoints; ++p) {
        stone[p] = (unsigned char)(r.cell[p] & 3);
        head[p] = (short)(r.cell[p] >> 2);
        if (stone[p] == kEmpty) empty.w[p >> 5] |= 1u << (p & 31);
        else if (head[p] == p) { std::memset(&nb[p], 0, sizeof(Set)); std::memset(&st[p], 0, sizeof(Set)); }
    }
    for (int p = 0; p < kPoints; ++p) {
        if (stone[p] == kEmpty) continue;
        const int h = head[p];
        st[h].w[p >> 5] |= 1u << (p & 31);
        for (int d = 0; d < 4; ++d) { const int q = neighbour(p, d); if (q >= 0) nb[h].w[q >> 5] |= 1u << (q & 31); }
    }
    for (int p = 0; p < kPoints; ++p) {
        if (stone[p] == kEmpty || head[p] != p) continue;
        int n = 0;
        for (int i = 0; i < 12; ++i) n += __builtin_popcount(nb[p].w[i] & empty.w[i]);
        libs[p] = (short)n;
    }
    auto valid = [&](int color, int p) {
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            if (stone[q] == kEmpty) return true;
            if ((stone[q] == color + 1) == (libs[head[q]] >= 2)) return true;
        }
        return false;
    };
    auto liberties_if = [&](int color, int p) {
        Set reach, count_in = empty;                         // reach: N(p) + the neighbourhoods of the own blocks p joins
        std::memset(&reach, 0, sizeof(reach));
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p, d);
            if (q < 0) continue;
            reach.w[q >> 5] |= 1u << (q & 31);
            if (stone[q] == color + 1) for (int i = 0; i < 12; ++i) reach.w[i] |= nb[head[q]].w[i];
            else if (stone[q] == 2 - color && libs[head[q]] == 1) for (int i = 0; i < 12; ++i) count_in.w[i] |= st[head[q]].w[i];   // lifted
        }
        reach.w[p >> 5] &= ~(1u << (p & 31));
        int n = 0;
        for (int i = 0; i < 12; ++i) n += __builtin_popcount(reach.w[i] & count_in.w[i]);
        return n;
    };

    const int to_move = r.to_move, opponent = 1 - to_move;
    const uint16_t* map = symmetry_map(r.symmetry);
    uint32_t word[kPoints];
    bool any_ko = false;
    for (int p = 0; p < kPoints; ++p) {
        uint32_t m = 0;
        if (stone[p] != kEmpty) {
            const int start = stone[p] == to_move + 1 ? 5 : 17;
            const int n = std::min((int)libs[head[p]], 6);
            m |= ((1u << n) - 1u) << start;
        } else {
            const bool ok = valid(to_move, p);
            if (ok) m |= ((1u << std::min(liberties_if(to_move, p), 6)) - 1u) << 11;
            if (valid(opponent, p)) m |= ((1u << std::min(liberties_if(opponent, p), 6)) - 1u) << 23;
            const uint32_t bit = 1u << (p & 31);
            if (ok && (r.ko[p >> 5] & bit)) { m |= 1u << 29; any_ko = true; }
            if (ok && (r.ladder_capture[p >> 5] & bit)) m |= 1u << 30;
            if (ok && (r.ladder_escape[p >> 5] & bit)) m |= 1u << 31;
        }
        if (r.last[0] == p) m |= 1u << 3;
        if (r.last[1] == p) m |= 1u << 4;
        word[p] = m;
    }
    const bool komi_set = (r.komi & 0x7fff) != 0;
    const uint32_t global = (komi_set ? 1u << to_move : 0u) | (any_ko ? 4u : 0u);
    for (int p = 0; p < kPoints; ++p) out[map[p]] = word[p] | global;
    out[kPoints] = komi_set ? (uint32_t)r.komi << (16 * to_move) : 0u;
    for (int i = kPoints + 1; i < DGNN_PACKED_WORDS; ++i) out[i] = 0;
}

// ---- Benson's unconditional life, Tromp-Taylor score, stone status (utils/benson.rs, utils/score.rs) ----
enum : unsigned char { kStatusNone = 0, kStatusBlock = 1, kStatusRegion = 2 };      // benson.rs:48-53

struct Bits { uint64_t w[6]; void set(int i) { w[i >> 6] |= 1ull << (i & 63); } bool get(int i) const { return (w[i >> 6] >> (i & 63)) & 1; } };

// (This flood-fill form follows the reference step by step; since round 3 it is the CHECKER of the bit-board form below, which
// is what the search calls: dgnn_go_benson_flood, tests/test_go_host.py.)
// Benson::new (benson.rs:62-78) over AllBlocksImpl / AllRegionsImpl (:231-253, :300-333).  The reference's variant of the
// algorithm, kept as it is: a region is every point reachable from an EMPTY point without crossing a stone of `color`
// (so it holds the opponent's stones too); it is vital to a block when EVERY one of its points -- stones included -- is
// adjacent to the block (is_vital :207-209 over adjacencies_of, board_fast.rs:154-159).
void benson_flood(const Fast& b, int color, unsigned char* status)
{
    const unsigned char mine = (unsigned char)(color + 1);
    struct Group { short first, count; bool live; };
    short block_points[kPoints], region_points[kPoints], queue[kPoints];
    short region_id[kPoints];
    Group blocks[kPoints], regions[kPoints];
    Bits block_adjacent[kPoints / 2 + 2];                                     // only used for index < num_blocks (<= 181 blocks of one colour)
    int num_blocks = 0, num_regions = 0, used = 0;
    short block_of[kPoints];
    std::memset(status, kStatusNone, kPoints);
    std::memset(block_of, 0xff, sizeof(block_of));
    std::memset(region_id, 0xff, sizeof(region_id));

    for (int p = 0; p < kPoints; ++p) {                                       // blocks of `color` with everything next to them
        if (b.stone[p] != mine || block_of[p] >= 0) continue;
        Group& g = blocks[num_blocks];
        g = {(short)used, 0, true};
        Bits& adjacent = block_adjacent[num_blocks];
        std::memset(&adjacent, 0, sizeof(adjacent));
        int head = 0, tail = 0;
        queue[tail++] = (short)p;
        block_of[p] = (short)num_blocks;
        while (head < tail) {
            const int q = queue[head++];
            block_points[used++] = (short)q;
            for (int d = 0; d < 4; ++d) {
                const int r = neighbour(q, d);
                if (r < 0) continue;
                adjacent.set(r);
                if (b.stone[r] == mine && block_of[r] < 0) { block_of[r] = (short)num_blocks; queue[tail++] = (short)r; }
            }
        }
        g.count = (short)(used - g.first);
        ++num_blocks;
    }

    used = 0;
    for (int p = 0; p < kPoints; ++p) {                                       // regions enclosed by `color`
        if (b.stone[p] != kEmpty || region_id[p] >= 0) continue;
        Group& g = regions[num_regions];
        g = {(short)used, 0, true};
        bool has_neighbour = false;
        int head = 0, tail = 0;
        queue[tail++] = (short)p;
        region_id[p] = (short)num_regions;
        while (head < tail) {
            const int q = queue[head++];
            region_points[used++] = (short)q;
            for (int d = 0; d < 4; ++d) {
                const int r = neighbour(q, d);
                if (r < 0) continue;
                if (b.stone[r] == mine) has_neighbour = true;
                else if (region_id[r] < 0) { region_id[r] = (short)num_regions; queue[tail++] = (short)r; }
            }
        }
        g.count = (short)(used - g.first);
        if (!has_neighbour) { used = g.first; for (int i = 0; i < g.count; ++i) region_id[region_points[g.first + i]] = (short)kPoints; continue; }   // benson.rs:324-328
        ++num_regions;
    }
    if (num_blocks == 0 || num_regions == 0) return;

    // vital[r] = the blocks region r is vital to: only blocks next to its first point can qualify (at most four)
    short vital_first[kPoints + 1], vital_store[4 * kPoints];
    struct { short* data; int count; void push_back(short v) { data[count++] = v; } int size() const { return count; } } vital{vital_store, 0};
    for (int r = 0; r < num_regions; ++r) {
        vital_first[r] = (short)vital.size();
        const int p0 = region_points[regions[r].first];
        short seen[4]; int n_seen = 0;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p0, d);
            if (q < 0 || b.stone[q] != mine) continue;
            const short k = block_of[q];
            if (std::find(seen, seen + n_seen, k) != seen + n_seen) continue;
            seen[n_seen++] = k;
            bool all = true;
            for (int i = 0; i < regions[r].count && all; ++i) all = block_adjacent[k].get(region_points[regions[r].first + i]);
            if (all) vital.push_back(k);
        }
    }
    vital_first[num_regions] = (short)vital.size();

    auto mark = [&](const short* points, const Group& g, unsigned char value) { for (int i = 0; i < g.count; ++i) status[points[g.first + i]] = value; };
    for (int k = 0; k < num_blocks; ++k) mark(block_points, blocks[k], kStatusBlock);
    for (int r = 0; r < num_regions; ++r) mark(region_points, regions[r], kStatusRegion);
    for (int r = 0; r < num_regions; ++r)                                    // remove_non_vital_regions, :136-152
        if (vital_first[r] == vital_first[r + 1]) { regions[r].live = false; mark(region_points, regions[r], kStatusNone); }

    short healthy[kPoints];
    for (bool changed = true; changed;) {                                     // :73-75
        changed = false;
        std::fill(healthy, healthy + num_blocks, (short)0);                   // remove_non_alive_blocks, :95-112
        for (int r = 0; r < num_regions; ++r)
            if (regions[r].live) for (int i = vital_first[r]; i < vital_first[r + 1]; ++i) ++healthy[vital_store[i]];
        for (int k = 0; k < num_blocks; ++k)
            if (blocks[k].live && healthy[k] < 2) { blocks[k].live = false; mark(block_points, blocks[k], kStatusNone); changed = true; }
        for (int r = 0; r < num_regions; ++r) {                               // remove_non_surrounded_regions, :116-132
            if (!regions[r].live) continue;
            bool surrounded = true;
            for (int i = 0; i < regions[r].count && surrounded; ++i) {
                const int q = region_points[regions[r].first + i];
                for (int d = 0; d < 4; ++d) {
                    const int n = neighbour(q, d);
                    if (n >= 0 && b.stone[n] == mine && status[n] != kStatusBlock) { surrounded = false; break; }
                }
            }
            if (!surrounded) { regions[r].live = false; mark(region_points, regions[r], kStatusNone); changed = true; }
        }
    }
}

// ---- bit boards: 361 bits, point p = 19 y + x in bit p ----
struct BB {
    uint64_t w[6];
    bool any() const { return (w[0] | w[1] | w[2] | w[3] | w[4] | w[5]) != 0; }
    bool get(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    void set(int i) { w[i >> 6] |= 1ull << (i & 63); }
    int lowest() const { for (int i = 0; i < 6; ++i) if (w[i]) return i * 64 + __builtin_ctzll(w[i]); return -1; }
    bool operator==(const BB& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
inline BB operator&(const BB& a, const BB& b) { BB r; for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] & b.w[i]; return r; }
inline BB operator|(const BB& a, const BB& b) { BB r; for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] | b.w[i]; return r; }
inline BB andnot(const BB& a, const BB& b) { BB r; for (int i = 0; i < 6; ++i) r.w[i] = a.w[i] & ~b.w[i]; return r; }   // a & ~b
template <int K> inline BB shift_up(const BB& a)        // towards higher points
{
    BB r;
    r.w[0] = a.w[0] << K;
    for (int i = 1; i < 6; ++i) r.w[i] = (a.w[i] << K) | (a.w[i - 1] >> (64 - K));
    return r;
}
template <int K> inline BB shift_down(const BB& a)
{
    BB r;
    for (int i = 0; i < 5; ++i) r.w[i] = (a.w[i] >> K) | (a.w[i + 1] << (64 - K));
    r.w[5] = a.w[5] >> K;
    return r;
}
struct BoardMasks {
    BB board, not_col0, not_col18;
    BoardMasks()
    {
        std::memset(this, 0, sizeof(*this));
        for (int p = 0; p < kPoints; ++p) { board.set(p); if (p % kN != 0) not_col0.set(p); if (p % kN != kN - 1) not_col18.set(p); }
    }
};
const BoardMasks kMasks;
// the four neighbours of every point of x (x itself not included unless it neighbours another of its points)
inline BB neighbours_of(const BB& x)
{
    return (shift_up<1>(x & kMasks.not_col18) | shift_down<1>(x & kMasks.not_col0) | shift_up<kN>(x) | shift_down<kN>(x)) & kMasks.board;
}
inline BB dilate(const BB& x) { return x | neighbours_of(x); }
inline BB flood(BB seed, const BB& within)                 // everything of `within` connected to the seed
{
    for (;;) {
        const BB next = dilate(seed) & within;
        if (next == seed) return seed;
        seed = next;
    }
}
struct StoneSets { BB stones[2], empty, visited; };
static_assert(offsetof(Fast, visited) == offsetof(Fast, stone) + kPoints && offsetof(Fast, head) >= offsetof(Fast, visited) + kPoints &&
              offsetof(Fast, libs) == offsetof(Fast, head) + 2 * kPoints && offsetof(Fast, next) == offsetof(Fast, libs) + 2 * kPoints,
              "the vector loads below read up to 31 bytes past an array's last point: they must stay inside the struct");
void stone_sets_scalar(const Fast& b, StoneSets* s)
{
    for (int i = 0; i < 6; ++i) {
        uint64_t black = 0, white = 0, visited = 0;
        const int first = i * 64, last = std::min(first + 64, kPoints);
        for (int p = first; p < last; ++p) {
            black |= (uint64_t)(b.stone[p] == 1) << (p - first);
            white |= (uint64_t)(b.stone[p] == 2) << (p - first);
            visited |= (uint64_t)(b.visited[p] != 0) << (p - first);
        }
        s->stones[0].w[i] = black;
        s->stones[1].w[i] = white;
        s->visited.w[i] = visited;
    }
}
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
void stone_sets_avx2(const Fast& b, StoneSets* s)
{
    const __m256i one = _mm256_set1_epi8(1), two = _mm256_set1_epi8(2), zero = _mm256_setzero_si256();
    uint32_t black[12], white[12], visited[12];
    for (int k = 0; k < 12; ++k) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b.stone + 32 * k));
        const __m256i u = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b.visited + 32 * k));
        black[k] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, one));
        white[k] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, two));
        visited[k] = ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(u, zero));
    }
    for (int i = 0; i < 6; ++i) {
        s->stones[0].w[i] = ((uint64_t)black[2 * i] | (uint64_t)black[2 * i + 1] << 32) & kMasks.board.w[i];
        s->stones[1].w[i] = ((uint64_t)white[2 * i] | (uint64_t)white[2 * i + 1] << 32) & kMasks.board.w[i];
        s->visited.w[i] = ((uint64_t)visited[2 * i] | (uint64_t)visited[2 * i + 1] << 32) & kMasks.board.w[i];
    }
}
#endif
const bool kGoAvx2 = host_simd();
inline StoneSets stone_sets(const Fast& b)
{
    StoneSets s;
#if DGNN_X86_SIMD
    if (kGoAvx2) stone_sets_avx2(b, &s); else
#endif
    stone_sets_scalar(b, &s);
    s.empty = andnot(kMasks.board, s.stones[0] | s.stones[1]);
    return s;
}
// the heads of the blocks with fewer than `limit` liberties, as a bit set (limit 2, the weak blocks: rarely any)
BB weak_heads_scalar(const Fast& b, int limit)
{
    BB r;
    std::memset(&r, 0, sizeof(r));
    for (int p = 0; p < kPoints; ++p) if (b.stone[p] != kEmpty && b.head[p] == p && b.libs[p] < limit) r.set(p);
    return r;
}
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
BB weak_heads_avx2(const Fast& b, int limit)
{
    BB r;
    std::memset(&r, 0, sizeof(r));
    const __m256i step = _mm256_set1_epi16(16), two = _mm256_set1_epi16((short)limit), zero = _mm256_setzero_si256();
    __m256i index = _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    for (int k = 0; k < 23; ++k, index = _mm256_add_epi16(index, step)) {
        const __m256i head = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b.head + 16 * k));
        const __m256i libs = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(b.libs + 16 * k));
        const __m256i stone = _mm256_cvtepu8_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b.stone + 16 * k)));
        const __m256i hit = _mm256_andnot_si256(_mm256_cmpeq_epi16(stone, zero), _mm256_and_si256(_mm256_cmpeq_epi16(head, index), _mm256_cmpgt_epi16(two, libs)));
        unsigned mask = (unsigned)_mm256_movemask_epi8(hit);
        while (mask) {
            const int p = 16 * k + (__builtin_ctz(mask) >> 1);
            if (p < kPoints) r.set(p);
            mask &= mask - 1;
            mask &= mask - 1;      // two mask bits per 16-bit lane
        }
    }
    return r;
}
#endif
inline BB weak_heads(const Fast& b, int limit = 2)
{
#if DGNN_X86_SIMD
    if (kGoAvx2) return weak_heads_avx2(b, limit);
#endif
    return weak_heads_scalar(b, limit);
}
// stone | block id << 2 per point (BoardRecord::cell); points 361..367 of the array stay zero
void record_cells_scalar(const Fast& f, uint16_t* cell)
{
    for (int p = 0; p < kPoints; ++p) cell[p] = (uint16_t)(f.stone[p] | (f.stone[p] != kEmpty ? f.head[p] << 2 : 0));
}
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
void record_cells_avx2(const Fast& f, uint16_t* cell)
{
    const __m256i zero = _mm256_setzero_si256();
    int p = 0;
    for (; p + 16 <= kPoints; p += 16) {
        const __m256i stone = _mm256_cvtepu8_epi16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(f.stone + p)));
        const __m256i head = _mm256_slli_epi16(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(f.head + p)), 2);
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(cell + p), _mm256_or_si256(stone, _mm256_andnot_si256(_mm256_cmpeq_epi16(stone, zero), head)));
    }
    for (; p < kPoints; ++p) cell[p] = (uint16_t)(f.stone[p] | (f.stone[p] != kEmpty ? f.head[p] << 2 : 0));
}
#endif

void board_record(const GoBoard& b, int to_move, int symmetry, BoardRecord* out)
{
    const Fast& f = b.inner;
    std::memset(out, 0, sizeof(*out));
#if DGNN_X86_SIMD
    if (kGoAvx2) record_cells_avx2(f, out->cell); else
#endif
    record_cells_scalar(f, out->cell);
    out->to_move = (uint8_t)to_move;
    out->symmetry = (uint8_t)symmetry;
    out->komi = float_to_half(std::max(0.0f, std::min(1.0f, 0.5f + (0.5f * b.komi) / 7.5f)));
    for (int i = 0; i < 2; ++i) out->last[i] = (int16_t)((b.history[i] >= 0 && b.history[i] < kPoints) ? b.history[i] : -1);

    auto set = [](uint32_t* mask, int p) { mask[p >> 5] |= 1u << (p & 31); };
    // super-ko can only strike on a point that has held a stone (board.rs:121): the empty ones among them, from the bit sets
    const StoneSets sets = stone_sets(f);
    for (BB ko = sets.visited & sets.empty; ko.any();) {
        const int p = ko.lowest();
        ko.w[p >> 6] &= ~(1ull << (p & 63));
        if (f.is_valid(to_move, p) && b.is_ko(to_move, p)) set(out->ko, p);
    }
    // the ladder readers answer "no" unless an opponent block with exactly two liberties (capture, may_start_ladder) or an own
    // block with fewer than two (escape, ladder.rs:146-151) is adjacent: only the liberties of such blocks are ever read out
    const unsigned char mine = (unsigned char)(to_move + 1), theirs = (unsigned char)(2 - to_move);
    for (BB heads = weak_heads(f, 3); heads.any();) {                         // the blocks with at most two liberties, lowest head first
        const int h = heads.lowest();
        heads.w[h >> 6] &= ~(1ull << (h & 63));
        const bool capture = f.stone[h] == theirs && f.libs[h] == 2, escape = f.stone[h] == mine && f.libs[h] < 2;
        if (!capture && !escape) continue;
        int s = h;
        do {
            for (int d = 0; d < 4; ++d) {
                const int q = neighbour(s, d);
                if (q < 0 || f.stone[q] != kEmpty) continue;
                if (capture && !((out->ladder_capture[q >> 5] >> (q & 31)) & 1u) && f.is_valid(to_move, q) && ladder_capture(f, to_move, q)) set(out->ladder_capture, q);
                if (escape && !((out->ladder_escape[q >> 5] >> (q & 31)) & 1u) && f.is_valid(to_move, q) && ladder_escape(f, to_move, q)) set(out->ladder_escape, q);
            }
            s = f.next[s];
        } while (s != h);
    }
}

inline BB block_stones(const Fast& b, int q) { BB r; std::memset(&r, 0, sizeof(r)); int s = q; do { r.set(s); s = b.next[s]; } while (s != q); return r; }

// Benson's unconditional life (utils/benson.rs:62-190), same result as benson_flood above, on bit boards.
// A region can only be vital to a block when EVERY one of its points is adjacent to that block (is_vital, benson.rs:207-209), and
// regions without a vital block are dropped before the iteration starts (:136-152).  So a region that holds a point with no
// stone of `color` next to it never matters: flooding from those "open" points through the non-`color` points removes, in a few
// bit-parallel steps, every such region -- on most positions all of them, and the answer is "nothing is alive" without a single
// per-point loop.  What is left are the few small enclosed regions; they are walked in the reference's order (a region is found
// from its lowest EMPTY point, and only blocks next to that point can be its vital blocks, :300-333).  The removal iteration
// (:73-132) is monotone, so its fixed point does not depend on the order of removals.
void benson_sets(const Fast& b, const StoneSets& sets, int color, BB* alive, BB* eyes)
{
    std::memset(alive, 0, sizeof(*alive));
    std::memset(eyes, 0, sizeof(*eyes));
    const BB& mine = sets.stones[color];
    if (!mine.any()) return;
    const BB others = andnot(kMasks.board, mine);                       // what a region is made of: empty points and opponent stones
    const BB open = andnot(others, neighbours_of(mine));
    const BB enclosed = open.any() ? andnot(others, flood(open, others)) : others;
    BB seeds = enclosed & sets.empty;
    if (!seeds.any()) return;

    struct Region { BB points, border; short vital[4]; int num_vital; bool live; };
    struct Block { short id; BB stones; bool live; int healthy; };
    static thread_local Region regions[192];
    static thread_local Block blocks[192];
    int num_regions = 0, num_blocks = 0;
    while (seeds.any() && num_regions < 192) {
        const int p0 = seeds.lowest();
        BB start;
        std::memset(&start, 0, sizeof(start));
        start.set(p0);
        Region& r = regions[num_regions];
        r.points = flood(start, enclosed);
        seeds = andnot(seeds, r.points);
        r.border = neighbours_of(r.points) & mine;
        r.num_vital = 0;
        short seen[4]; int n_seen = 0;
        for (int d = 0; d < 4; ++d) {
            const int q = neighbour(p0, d);
            if (q < 0 || b.stone[q] != color + 1) continue;
            const short id = b.head[q];
            if (std::find(seen, seen + n_seen, id) != seen + n_seen) continue;
            seen[n_seen++] = id;
            int k = 0;
            while (k < num_blocks && blocks[k].id != id) ++k;
            if (k == num_blocks) {
                if (num_blocks == 192) continue;
                blocks[num_blocks++] = Block{id, block_stones(b, q), true, 0};
            }
            if (!andnot(r.points, dilate(blocks[k].stones)).any()) r.vital[r.num_vital++] = (short)k;    // is_vital, benson.rs:207-209
        }
        r.live = r.num_vital > 0;                                       // remove_non_vital_regions, :136-152
        if (r.live) ++num_regions;
    }
    if (num_regions == 0) return;

    for (bool changed = true; changed;) {                               // :73-75
        changed = false;
        for (int k = 0; k < num_blocks; ++k) blocks[k].healthy = 0;     // remove_non_alive_blocks, :95-112
        for (int r = 0; r < num_regions; ++r) if (regions[r].live) for (int i = 0; i < regions[r].num_vital; ++i) ++blocks[regions[r].vital[i]].healthy;
        BB living;
        std::memset(&living, 0, sizeof(living));
        for (int k = 0; k < num_blocks; ++k) {
            if (blocks[k].live && blocks[k].healthy < 2) { blocks[k].live = false; changed = true; }
            if (blocks[k].live) living = living | blocks[k].stones;
        }
        for (int r = 0; r < num_regions; ++r)                           // remove_non_surrounded_regions, :116-132
            if (regions[r].live && andnot(regions[r].border, living).any()) { regions[r].live = false; changed = true; }
    }
    for (int k = 0; k < num_blocks; ++k) if (blocks[k].live) *alive = *alive | blocks[k].stones;
    for (int r = 0; r < num_regions; ++r) if (regions[r].live) *eyes = *eyes | regions[r].points;
}

void benson(const Fast& b, int color, unsigned char* status)
{
    BB alive, eyes;
    benson_sets(b, stone_sets(b), color, &alive, &eyes);
    for (int p = 0; p < kPoints; ++p) status[p] = alive.get(p) ? kStatusBlock : eyes.get(p) ? kStatusRegion : kStatusNone;
}

bool is_scorable(const Fast& b)                                               // score.rs:105-117
{
    const StoneSets sets = stone_sets(b);
    BB alive[2], eyes[2];
    benson_sets(b, sets, 0, &alive[0], &eyes[0]);
    benson_sets(b, sets, 1, &alive[1], &eyes[1]);
    const BB both = eyes[0] | eyes[1];
    return !andnot(sets.empty, both).any() && !andnot(sets.stones[0], alive[0] | eyes[1]).any() && !andnot(sets.stones[1], alive[1] | eyes[0]).any();
}

// get_territory_distance (score.rs:248-283): 0 on a stone of `color`, the distance through empty points elsewhere, 0xff if unreachable
void territory_distance(const Fast& b, int color, unsigned char* distance)
{
    short queue[kPoints];
    int head = 0, tail = 0;
    std::memset(distance, 0xff, kPoints);
    for (int p = 0; p < kPoints; ++p) if (b.stone[p] == color + 1) { distance[p] = 0; queue[tail++] = (short)p; }
    while (head < tail) {                                                      // unit steps: breadth first gives what the relaxation converges to
        const int q = queue[head++];
        for (int d = 0; d < 4; ++d) {
            const int r = neighbour(q, d);
            if (r >= 0 && b.stone[r] == kEmpty && distance[r] == 0xff) { distance[r] = (unsigned char)(distance[q] + 1); queue[tail++] = (short)r; }
        }
    }
}

void tt_score(const Fast& b, int* black_out, int* white_out)                  // get_tt_score, score.rs:220-239
{
    unsigned char black[kPoints], white[kPoints];
    territory_distance(b, 0, black);
    territory_distance(b, 1, white);
    int nb = 0, nw = 0;
    for (int p = 0; p < kPoints; ++p) {
        if (black[p] == 0) ++nb;
        else if (white[p] == 0) ++nw;
        else if (white[p] == 0xff) ++nb;
        else if (black[p] == 0xff) ++nw;
    }
    *black_out = nb;
    *white_out = nw;
}

// get_stone_status(self, finished = self) (score.rs:146-186) as one bit mask per point
void stone_status(const Fast& b, unsigned char* out)
{
    unsigned char black[kPoints], white[kPoints], distance_black[kPoints], distance_white[kPoints];
    benson(b, 0, black);
    benson(b, 1, white);
    Fast cleaned = b;                                                          // clear_board, :197-211
    for (int p = 0; p < kPoints; ++p)
        if ((b.stone[p] == 2 && white[p] != kStatusBlock) || (b.stone[p] == 1 && black[p] != kStatusBlock)) cleaned.stone[p] = kEmpty;
    territory_distance(cleaned, 0, distance_black);
    territory_distance(cleaned, 1, distance_white);
    for (int p = 0; p < kPoints; ++p) {
        unsigned char m = 0;
        if (b.stone[p] == 2) m = white[p] == kStatusBlock ? (DGNN_STONE_ALIVE | DGNN_TERRITORY_WHITE) : black[p] == kStatusRegion ? (DGNN_STONE_DEAD | DGNN_TERRITORY_BLACK) : (DGNN_STONE_SEKI | DGNN_TERRITORY_WHITE);
        else if (b.stone[p] == 1) m = black[p] == kStatusBlock ? (DGNN_STONE_ALIVE | DGNN_TERRITORY_BLACK) : white[p] == kStatusRegion ? (DGNN_STONE_DEAD | DGNN_TERRITORY_WHITE) : (DGNN_STONE_SEKI | DGNN_TERRITORY_BLACK);
        else if (distance_black[p] != 0xff && distance_white[p] == 0xff) m = DGNN_TERRITORY_BLACK;
        else if (distance_white[p] != 0xff && distance_black[p] == 0xff) m = DGNN_TERRITORY_WHITE;
        out[p] = m;
    }
}

std::string sgf_vertex(int p) { return std::string(1, (char)('a' + p % kN)) + (char)('a' + p / kN); }      // CGoban::to_sgf, sgf.rs:36-43

std::string winner_sgf(const GoBoard& g, const unsigned char* status)         // get_winner_as_sgf, game_result.rs:82-93
{
    float black = 0.0f, white = g.komi;
    for (int p = 0; p < kPoints; ++p) { if (status[p] & DGNN_TERRITORY_BLACK) black += 1.0f; if (status[p] & DGNN_TERRITORY_WHITE) white += 1.0f; }
    char text[32];
    if (black > white) std::snprintf(text, sizeof(text), "B+%.1f", (double)(black - white));
    else if (white > black) std::snprintf(text, sizeof(text), "W+%.1f", (double)(white - black));
    else std::snprintf(text, sizeof(text), "0");
    return text;
}

std::string territory_sgf(const unsigned char* status)                        // get_territory_as_sgf, game_result.rs:53-72
{
    std::string black, white;
    for (int p = 0; p < kPoints; ++p) {
        if (status[p] & DGNN_TERRITORY_WHITE) white += "[" + sgf_vertex(p) + "]";
        else if (status[p] & DGNN_TERRITORY_BLACK) black += "[" + sgf_vertex(p) + "]";
    }
    return (black.empty() ? "" : "TB") + black + (white.empty() ? "" : "TW") + white;
}

// is_eye (options.rs:192-213): an empty point with enough of one's own stones around it
bool is_simple_eye(const Fast& b, int color, int p)
{
    const int x = p % kN, y = p / kN;
    auto filled = [&](int dx, int dy) { const int u = x + dx, v = y + dy; return u >= 0 && u < kN && v >= 0 && v < kN && b.stone[v * kN + u] == color + 1; };
    const int cross = filled(1, 0) + filled(-1, 0) + filled(0, 1) + filled(0, -1);
    const int diagonal = filled(1, 1) + filled(1, -1) + filled(-1, 1) + filled(-1, -1);
    const bool edge_x = x == 0 || x == 18, edge_y = y == 0 || y == 18;
    if (edge_x && edge_y) return cross >= 2 && diagonal >= 1;
    if (edge_x || edge_y) return cross >= 3 && diagonal >= 2;
    return cross >= 4 && diagonal >= 3;
}

// ---- game callbacks for the search driver ----
// (every leaf of a search plays on its own copy of the board, pool/event.rs:44-46: dgnn_internal.h BlockPool; GoBoard is trivially
// destructible, a copy is its bytes)
using GoBoardPool = BlockPool<sizeof(GoBoard), 512>;
static_assert(std::is_trivially_destructible<GoBoard>::value && std::is_trivially_copyable<GoBoard>::value, "go_clone copies bytes");
void* go_clone(void* b)
{
    void* copy = GoBoardPool::get();
    if (copy) std::memcpy(copy, b, sizeof(GoBoard));
    return copy;
}
void go_destroy(void* b) { GoBoardPool::put(b); }
void go_place(void* b, int color, int index) { if (index >= 0 && index < kPoints) static_cast<GoBoard*>(b)->place(color, index); }
void go_features(void* b, int to_move, int symmetry, uint16_t* out) { features_v1(*static_cast<GoBoard*>(b), to_move, symmetry, out); }
// Board::is_valid for every point at once (board_fast.rs:213-241 + board.rs:120-145): an empty point next to an empty point, to
// an own block with at least two liberties or to an opponent block with fewer than two; minus the super-ko points, which can only
// be points that have held a stone.
BB valid_moves(const GoBoard& g, const StoneSets& sets, int to_move)
{
    const Fast& f = g.inner;
    BB weak[2];                                                          // stones of blocks with fewer than two liberties
    std::memset(weak, 0, sizeof(weak));
    for (BB heads = weak_heads(f); heads.any();) {
        const int p = heads.lowest();
        heads.w[p >> 6] &= ~(1ull << (p & 63));
        weak[f.stone[p] - 1] = weak[f.stone[p] - 1] | block_stones(f, p);
    }
    const BB support = sets.empty | andnot(sets.stones[to_move], weak[to_move]) | weak[1 - to_move];
    BB valid = sets.empty & neighbours_of(support);
    for (BB ko = valid & sets.visited; ko.any();) {                      // super-ko: only on points that have held a stone
        const int p = ko.lowest();
        ko.w[p >> 6] &= ~(1ull << (p & 63));
        if (g.is_ko(to_move, p)) valid.w[p >> 6] &= ~(1ull << (p & 63));
    }
    return valid;
}
void write_candidates(const BB& set, unsigned char* out)
{
    for (int i = 0; i < 6; ++i) {
        const int first = i * 64, last = std::min(first + 64, kPoints);
        for (int p = first; p < last; ++p) out[p] = (unsigned char)((set.w[i] >> (p - first)) & 1);
    }
}
void go_candidates(void* b, int to_move, unsigned char* out)                  // options.rs:44-48
{
    const GoBoard* g = static_cast<GoBoard*>(b);
    write_candidates(valid_moves(*g, stone_sets(g->inner), to_move), out);
    out[kPass] = 1;
}
void go_candidate_bits(void* b, int to_move, uint64_t* out)
{
    const GoBoard* g = static_cast<GoBoard*>(b);
    const BB valid = valid_moves(*g, stone_sets(g->inner), to_move);
    std::memcpy(out, valid.w, sizeof(valid.w));
    out[kPass >> 6] |= 1ull << (kPass & 63);
}
int go_is_symmetric(void* b, int symmetry)                                    // symmetry.rs:136-144
{
    const GoBoard* g = static_cast<GoBoard*>(b);
    const uint16_t* map = symmetry_map(symmetry);
    for (int p = 0; p < kPoints; ++p) if (g->inner.stone[p] != g->inner.stone[map[p]]) return 0;
    return 1;
}
uint64_t go_hash(void* b) { return static_cast<GoBoard*>(b)->hash; }      // BoardTuple, predictors/nn.rs:31-45
void go_record(void* b, int to_move, int symmetry, uint8_t* out) { board_record(*static_cast<GoBoard*>(b), to_move, symmetry, reinterpret_cast<BoardRecord*>(out)); }
BB scoring_set(const GoBoard* g, int to_move)                                // ScoringPolicyChecker, options.rs:110-139
{
    const StoneSets sets = stone_sets(g->inner);
    BB alive, eyes[2];
    benson_sets(g->inner, sets, 0, &alive, &eyes[0]);
    benson_sets(g->inner, sets, 1, &alive, &eyes[1]);
    BB candidates = andnot(valid_moves(*g, sets, to_move), eyes[0] | eyes[1]);
    // one's own simple eye (options.rs:192-213) needs every on-board cross neighbour to be one's own stone: look only there
    BB closed = andnot(candidates, neighbours_of(andnot(kMasks.board, sets.stones[to_move])));
    while (closed.any()) {
        const int p = closed.lowest();
        closed.w[p >> 6] &= ~(1ull << (p & 63));
        if (is_simple_eye(g->inner, to_move, p)) candidates.w[p >> 6] &= ~(1ull << (p & 63));
    }
    return candidates;
}
void go_candidates_scoring(void* b, int to_move, unsigned char* out)
{
    write_candidates(scoring_set(static_cast<GoBoard*>(b), to_move), out);
    out[kPass] = 0;
}
void go_candidate_bits_scoring(void* b, int to_move, uint64_t* out)
{
    const BB set = scoring_set(static_cast<GoBoard*>(b), to_move);
    std::memcpy(out, set.w, sizeof(set.w));
}
const dgnn_game_ops kGoOps = {go_clone, go_destroy, go_place, go_features, go_candidates, go_is_symmetric, go_hash, go_record, go_candidate_bits};
const dgnn_game_ops kGoScoringOps = {go_clone, go_destroy, go_place, go_features, go_candidates_scoring, go_is_symmetric, go_hash, go_record, go_candidate_bits_scoring};

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
int dgnn_go_board_record(const dgnn_go_board* b, int to_move, int symmetry, uint8_t* out)
{
    if (!b || !out || (to_move != 0 && to_move != 1) || symmetry < 0 || symmetry >= 8) return DGNN_BAD_ARGUMENT;
    BoardRecord r;                                       // (the caller's buffer need not be aligned)
    board_record(*CGO(b), to_move, symmetry, &r);
    std::memcpy(out, &r, sizeof(r));
    return DGNN_OK;
}
int dgnn_board_records_to_packed(const uint8_t* records, int count, uint32_t* packed)
{
    if (!records || !packed || count <= 0) return DGNN_BAD_ARGUMENT;
    for (int i = 0; i < count; ++i) {
        BoardRecord r;
        std::memcpy(&r, records + (size_t)i * DGNN_BOARD_RECORD_BYTES, sizeof(r));
        if (r.to_move > 1 || r.symmetry > 7) return DGNN_BAD_ARGUMENT;
        for (int p = 0; p < kPoints; ++p) if ((r.cell[p] & 3) == 3 || (r.cell[p] >> 2) >= kPoints) return DGNN_BAD_ARGUMENT;
        record_to_packed(r, packed + (size_t)i * DGNN_PACKED_WORDS);
    }
    return DGNN_OK;
}
int dgnn_go_is_symmetric(const dgnn_go_board* b, int symmetry) { return (b && symmetry >= 0 && symmetry < 8) ? go_is_symmetric((void*)b, symmetry) : 0; }

const dgnn_game_ops* dgnn_go_game_ops(void) { return &kGoOps; }
const dgnn_game_ops* dgnn_go_scoring_ops(void) { return &kGoScoringOps; }

int dgnn_go_benson(const dgnn_go_board* b, int color, unsigned char* status_out)
{
    if (!b || !status_out || (color != 0 && color != 1)) return DGNN_BAD_ARGUMENT;
    benson(CGO(b)->inner, color, status_out);
    return DGNN_OK;
}
int dgnn_go_benson_flood(const dgnn_go_board* b, int color, unsigned char* status_out)
{
    if (!b || !status_out || (color != 0 && color != 1)) return DGNN_BAD_ARGUMENT;
    benson_flood(CGO(b)->inner, color, status_out);
    return DGNN_OK;
}
int dgnn_go_candidates(const dgnn_go_board* b, int to_move, int scoring, unsigned char* out)
{
    if (!b || !out || (to_move != 0 && to_move != 1)) return DGNN_BAD_ARGUMENT;
    (scoring ? go_candidates_scoring : go_candidates)((void*)b, to_move, out);
    return DGNN_OK;
}
int dgnn_go_check(const dgnn_go_board* b) { return b && CGO(b)->inner.consistent() ? 1 : 0; }
int dgnn_go_is_scorable(const dgnn_go_board* b) { return b ? (is_scorable(CGO(b)->inner) ? 1 : 0) : 0; }
int dgnn_go_score(const dgnn_go_board* b, int* black_out, int* white_out)     // Score::get_score, score.rs:130-136
{
    if (!b || !black_out || !white_out) return DGNN_BAD_ARGUMENT;
    *black_out = *white_out = 0;
    if (CGO(b)->hash != 0) tt_score(CGO(b)->inner, black_out, white_out);
    return DGNN_OK;
}
int dgnn_go_stone_status(const dgnn_go_board* b, unsigned char* status_out)
{
    if (!b || !status_out) return DGNN_BAD_ARGUMENT;
    stone_status(CGO(b)->inner, status_out);
    return DGNN_OK;
}
int dgnn_go_result_sgf(const dgnn_go_board* b, char* winner_out, size_t winner_capacity, char* territory_out, size_t territory_capacity)
{
    if (!b) return DGNN_BAD_ARGUMENT;
    unsigned char status[kPoints];
    stone_status(CGO(b)->inner, status);
    const std::string winner = winner_sgf(*CGO(b), status), territory = territory_sgf(status);
    if ((winner_out && winner.size() + 1 > winner_capacity) || (territory_out && territory.size() + 1 > territory_capacity)) return DGNN_BAD_ARGUMENT;
    if (winner_out) std::memcpy(winner_out, winner.c_str(), winner.size() + 1);
    if (territory_out) std::memcpy(territory_out, territory.c_str(), territory.size() + 1);
    return DGNN_OK;
}

// Self-play on the real board, following self_play_one / Player::predict / predict_aux / predict
// (src/libdg_mcts/self_play.rs:423-459, 331-420, 243-274, lib.rs:144-200):
//   * random komi (lib.rs:210-224); two players, each with a tree re-used across moves (Node::forward) and a moving average
//     of the win rate that sets the rollout budget max(0.1, 4w(1-w)) * num_rollouts (self_play.rs:234-241);
//   * while the board is not scorable (Score::is_scorable) the search is a ScoringSearch: no pass, nothing inside a Benson
//     eye, deterministic (no dirichlet noise, temperature 0); once it is scorable a StandardSearch with noise may pass;
//   * a search with nothing to play passes (self_play.rs:347-350); two passes on a scorable board end the game;
//   * a budget of one rollout or less plays from the 8-symmetry policy alone (self_play.rs:369-399);
//   * the record is GameResult::Ended (game_result.rs:35-41): header, RE[], the moves, TB / TW.
// `max_moves` caps the stones per game below the reference's 722 (0 = 722) so a measurement can be bounded; a game cut
// off there is recorded like one that reached 722.  num_ex_it_rollouts > 0 turns expert iteration on (--ex-it,
// self_play.rs:277-319).  Not built: resignation (the reference's self-play has none either, only policy_play), the
// pretty-printed variation the reference stores as C[] with every searched move.
// moves_out (num_games * max_moves packed indices of the stones in order, -1 = not played) and sgf_out (one record per game,
// newline separated, cut off at sgf_capacity) may be NULL.
// Sharding: the games of this call are the global game ids first_game, first_game + game_stride, ...;
// a game's random stream is a function of (seed, global id) only, so `world` ranks calling with first_game = rank and
// game_stride = world play the games one rank would.
// Concurrency (self_play, self_play.rs:478-495): min(num_parallel, num_games) game threads, each taking the next unplayed game
// until num_games are done, so the device sees a steady number of games instead of a tail (num_parallel <= 0: all at once).
int dgnn_selfplay_go(dgnn_predict_fn predict, void* predict_ctx, dgnn_cache* shared_cache, const dgnn_selfplay_params* params,
                     dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity)
{
    if (!predict) return DGNN_BAD_ARGUMENT;
    Predictor predictor;
    predictor.features = predict;
    predictor.ctx = predict_ctx;
    return selfplay_go(predictor, shared_cache, params, stats, moves_out, sgf_out, sgf_capacity);
}

// The same with the network taking compact board records (dgnn_forward_boards): the feature planes are built on the device.
int dgnn_selfplay_go_boards(dgnn_predict_boards_fn predict, void* predict_ctx, dgnn_cache* shared_cache, const dgnn_selfplay_params* params,
                            dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity)
{
    if (!predict) return DGNN_BAD_ARGUMENT;
    Predictor predictor;
    predictor.boards = predict;
    predictor.ctx = predict_ctx;
    return selfplay_go(predictor, shared_cache, params, stats, moves_out, sgf_out, sgf_capacity);
}

}  // extern "C"

int dgnn::selfplay_go(const Predictor& predictor, dgnn_cache* shared_cache, const dgnn_selfplay_params* params, dgnn_selfplay_stats* stats,
                      int* moves_out, char* sgf_out, size_t sgf_capacity)
{
    if (!params || !stats) return DGNN_BAD_ARGUMENT;
    const int num_games = params->num_games, first_game = params->first_game, game_stride = params->game_stride > 0 ? params->game_stride : 1;
    const int num_rollouts = params->num_rollouts, num_ex_it_rollouts = params->num_ex_it_rollouts;
    const int threads_per_search = params->threads_per_search > 0 ? params->threads_per_search : 1;
    const int leaves_per_thread = params->leaves_per_thread > 0 ? params->leaves_per_thread : 1;
    const uint64_t seed = params->seed;
    const int private_cache = params->cache_entries_per_game;
    int num_parallel = params->num_parallel, max_moves = params->max_moves;
    if (num_games <= 0 || first_game < 0 || max_moves < 0 || num_rollouts <= 0 || leaves_per_thread > 64) return DGNN_BAD_ARGUMENT;
    if (num_parallel <= 0 || num_parallel > num_games) num_parallel = num_games;
    if (max_moves == 0 || max_moves > 722) max_moves = 722;
    // a table per game promises games that are a function of their seed; a pool that coalesces positions in flight hands one game the
    // evaluation another game asked for (other last moves, other ko state behind the same stones): refused, not silently mixed
    if (private_cache > 0 && predictor.coalesces()) return DGNN_BAD_ARGUMENT;
    std::atomic<uint64_t> evals{0}, conflicts{0}, moves{0}, hits{0}, ended{0}, ex_it_searches{0}, started{0}, completed{0}, stones_completed{0};
    std::atomic<int> status{DGNN_OK};
    std::atomic<bool> stop{false};                                              // set at the end of a measurement window (params->window)
    std::vector<std::string> records(sgf_out ? (size_t)num_games : 0);          // kept only for a caller who asked for them all at the end
    std::mutex on_game_mutex;                                                   // (params->on_game: one call at a time)
    const int opening_games = params->opening_games > 0 ? std::min(params->opening_games, num_games) : num_games;
    const auto t0 = std::chrono::steady_clock::now();

    auto game = [&](int g) {
        Rng2 rng{seed + 0x9e3779b97f4a7c15ull * (uint64_t)(first_game + g * game_stride + 1)};
        const double kv = rng.uniform();                                       // get_random_komi, lib.rs:210-224
        const float komi = kv < 0.4 ? 7.5f : kv < 0.8 ? 6.5f : kv < 0.9 ? 0.5f : (float)((int)(rng.next() % 16) - 8) + 0.5f;
        GoBoard board(komi);
        // The transposition table is keyed like the reference's (zobrist hash + colour, predictors/nn.rs:31-45) although the
        // feature planes also hold the last moves and the ko state: WHICH of two such positions fills an entry first depends
        // on timing when games share the table.  A table of the game's own keeps the game a function of the seed.
        dgnn_cache* cache = shared_cache;
        if (private_cache > 0 && dgnn_cache_create((size_t)private_cache, 1, &cache) != DGNN_OK) { status.store(DGNN_OUT_OF_MEMORY); return; }
        dgnn_tree_node* roots[2] = {nullptr, nullptr};                         // Player::root of black / white
        float winrate[2] = {0.5f, 0.5f};                                       // MovingAverage::new(0.5, 0.2)
        std::string sgf;
        int pass_count = 0;
        bool finished = false;
        if (moves_out) for (int m = 0; m < max_moves; ++m) moves_out[(size_t)g * max_moves + m] = -1;
        int first_turn = 0;
        started.fetch_add(1);
        if (params->opening && params->opening_len > 0 && g < opening_games) {  // start from a given position (alternating colours, black first)
            for (int m = 0; m < params->opening_len && m < max_moves; ++m) {
                const int p = params->opening[(size_t)g * params->opening_len + m];
                if (p < 0 || p >= kPoints || !board.is_valid(m & 1, p)) break;
                if (moves_out) moves_out[(size_t)g * max_moves + board.count] = p;
                char text[16];
                dgnn_sgf_played(text, sizeof(text), m & 1, p, 0, nullptr, 0, 0.0f, -1, "");
                sgf += text;
                board.place(m & 1, p);
                ++first_turn;
            }
        }

        // one turn per iteration, black first; the turn cap only guards against two players passing for ever on a board
        // that never becomes scorable (the reference would spin there)
        uint64_t weights_seen = predictor.generation();
        for (int turn = first_turn; board.count < max_moves && turn < 4 * 722 && status.load() == DGNN_OK && !stop.load(std::memory_order_relaxed); ++turn) {
            const int me = turn & 1;
            // a weight refresh of the pool (dgnn_pool_pause ... resume) cannot reach a table that is the game's own: the game
            // drops it itself, before the first search that could be served the replaced network's outputs
            if (private_cache > 0) {
                const uint64_t now = predictor.generation();
                if (now != weights_seen) { dgnn_cache_clear(cache); weights_seen = now; }
            }
            const bool allow_pass = is_scorable(board.inner);                  // self_play.rs:437
            const dgnn_game_ops* ops = allow_pass ? &kGoOps : &kGoScoringOps;   // predict_aux, self_play.rs:243-274
            const bool deterministic = !allow_pass;                            // ScoringSearch::deterministic, options.rs:160-162
            const float w = winrate[me];
            const int limit = (int)(std::max(0.1f, 4.0f * w * (1.0f - w)) * (float)num_rollouts);   // Player::num_rollout

            // predict (lib.rs:144-200) behind predict_aux (self_play.rs:243-274): the root ensemble, noise unless deterministic,
            // the starting tree re-used with its prior replaced, the search, the best move.  Returns the tree (the caller owns
            // it) or NULL when there is nothing to search (*value_out is then not finite) or on an error (status is set).
            auto search = [&](int rollouts, dgnn_tree_node* start, float* value_out, int* index_out) -> dgnn_tree_node* {
                float value = 0.0f, prior[368];
                *value_out = std::numeric_limits<float>::quiet_NaN();
                *index_out = kPass;
                int st = full_forward(ops, &board, me, predictor, cache, &value, prior);   // lib.rs:154-159
                if (st < 0) { status.store(st); dgnn_tree_node_free(start); return nullptr; }
                evals.fetch_add((uint64_t)st);
                bool any = false;
                for (int i = 0; i < DGNN_NUM_MOVES; ++i) any = any || std::isfinite(prior[i]);
                if (!any) { dgnn_tree_node_free(start); return nullptr; }
                if (!deterministic) dgnn_dirichlet_add(prior, DGNN_NUM_MOVES, 0.03f, 0.25f, rng.next());   // lib.rs:163-165
                dgnn_tree_node* root = start;
                if (root) { if (!allow_pass) dgnn_tree_node_disqualify(root, kPass); dgnn_tree_node_set_prior(root, prior); }   // lib.rs:170-181
                else root = dgnn_tree_node_new(me, value, prior);
                uint64_t e = 0, c = 0, h = 0;
                st = search_run(root, ops, &board, predictor, cache, rollouts, threads_per_search, leaves_per_thread, rng.next(), &e, &c, &h);
                evals.fetch_add(e);
                conflicts.fetch_add(c);
                hits.fetch_add(h);
                if (st != DGNN_OK) { status.store(st); dgnn_tree_node_free(root); return nullptr; }
                dgnn_tree_node_best(root, (!deterministic && board.count < 8) ? 0.8f : 0.0f, rng.uniform(), value_out, index_out);   // lib.rs:188-192
                return root;
            };
            // Played::from_mcts (self_play.rs:120-137): C[] = tree::to_pretty, TR[] = the best move of the tree's prior, TV[] = the
            // tree's size, P[] = its visit distribution.
            // one move record as a string, sized by what dgnn_sgf_played asks for (a fixed buffer that is too small used to
            // drop the C[] text -- or the whole move -- silently; the reference always writes both)
            auto played_text = [&](int color, int point, int rollouts, const float* softmax, int has_value, float value, int prior_index,
                                   const char* explain) {
                std::string text(dgnn_sgf_played(nullptr, 0, color, point, rollouts, softmax, has_value, value, prior_index, explain), '\0');
                dgnn_sgf_played(&text[0], text.size(), color, point, rollouts, softmax, has_value, value, prior_index, explain);
                text.resize(text.size() - 1);       // (the NUL)
                return text;
            };
            auto record_mcts = [&](std::string& out, int point, float value, const dgnn_tree_node* tree) {
                float softmax[DGNN_NUM_MOVES], tree_prior[DGNN_NUM_MOVES + 6];
                dgnn_tree_node_softmax(tree, softmax);
                dgnn_tree_node_prior(tree, tree_prior);
                for (int i = DGNN_NUM_MOVES; i < DGNN_NUM_MOVES + 6; ++i) tree_prior[i] = -std::numeric_limits<float>::infinity();
                int prior_index = dgnn_argmax_f32(tree_prior, DGNN_NUM_MOVES + 6);
                if (prior_index < 0) prior_index = kPass;                      // Node::prior, tree.rs:1285-1289
                int total = 0;
                dgnn_tree_node_info(tree, nullptr, &total, nullptr, nullptr, nullptr);
                std::string explain(dgnn_tree_node_to_pretty(tree, nullptr, 0), '\0');        // tree::to_pretty: eleven lines, each with its
                dgnn_tree_node_to_pretty(tree, &explain[0], explain.size());                   // principal variation -- as long as the tree is deep
                out = played_text(me, point, total, softmax, 1, value, prior_index, explain.c_str());
            };
            // Player::ex_it (self_play.rs:277-305) behind is_good_candidate (:315-319): on an UNDECIDED position (the value of the move just
            // chosen within [-0.8, 0.8]; the random number is only drawn then, as in the reference's && chain) one move in twenty gets a
            // fresh, full-size search whose visit distribution and value replace the record's; the move that was chosen stays
            auto expert_iteration = [&](std::string& out, int point, float chosen_value) -> bool {
                if (num_ex_it_rollouts <= 0 || !(chosen_value >= -0.80f && chosen_value <= 0.80f) || rng.uniform() >= 0.05) return false;
                float value;
                int ignored;
                dgnn_tree_node* tree = search(num_ex_it_rollouts, nullptr, &value, &ignored);
                if (!tree) return false;
                if (std::isfinite(value)) record_mcts(out, point, value, tree);
                dgnn_tree_node_free(tree);
                ex_it_searches.fetch_add(1);
                return std::isfinite(value);
            };

            int played = kPass;
            std::string move;
            if (limit > 1) {                                                   // Player::predict, self_play.rs:340-368
                dgnn_tree_node* start = roots[me];
                roots[me] = nullptr;
                float best_value;
                dgnn_tree_node* root = search(limit, start, &best_value, &played);
                if (status.load() != DGNN_OK) break;
                if (!root || !std::isfinite(best_value) || played < 0 || (played != kPass && !board.is_valid(me, played))) {
                    played = kPass;                                            // nothing to play: pass (self_play.rs:347-350)
                    dgnn_tree_node_free(root);
                    move = played_text(me, kPass, 0, nullptr, 0, 0.0f, -1, "");
                } else {
                    if (!expert_iteration(move, played, best_value)) record_mcts(move, played, best_value, root);
                    winrate[me] -= 0.2f * (winrate[me] - best_value);          // MovingAverage::update, self_play.rs:56-58
                    roots[me] = dgnn_tree_node_forward(root, played);
                }
            } else {                                                           // policy play, self_play.rs:369-405
                float value = 0.0f, prior[368];
                const int st = full_forward(ops, &board, me, predictor, cache, &value, prior);
                if (st < 0) { status.store(st); break; }
                evals.fetch_add((uint64_t)st);
                if (!allow_pass) prior[kPass] = -std::numeric_limits<float>::infinity();
                double scaled[DGNN_NUM_MOVES];
                for (int i = 0; i < DGNN_NUM_MOVES; ++i) scaled[i] = (double)prior[i];
                if (dgnn_choose_f64(scaled, DGNN_NUM_MOVES, 0.5, 1.0 / 0.8, rng.uniform(), &played, nullptr) != 1) played = kPass;
                if (played < 0 || (played != kPass && !board.is_valid(me, played))) played = kPass;
                if (!expert_iteration(move, played, value)) move = played_text(me, played, 1, prior, 1, value, -1, "");
                if (status.load() != DGNN_OK) break;
                winrate[me] -= 0.2f * (winrate[me] - value);
                if (roots[me]) roots[me] = dgnn_tree_node_forward(roots[me], played);
            }
            sgf += move;

            if (played == kPass) {                                             // self_play.rs:441-450
                if (++pass_count >= 2 && is_scorable(board.inner)) { finished = true; break; }
            } else {
                pass_count = 0;
                if (moves_out) moves_out[(size_t)g * max_moves + board.count] = played;
                board.place(me, played);
                moves.fetch_add(1);
            }
            if (roots[1 - me]) roots[1 - me] = dgnn_tree_node_forward(roots[1 - me], played);   // players[1].forward, self_play.rs:453
        }
        if (finished) ended.fetch_add(1);
        const bool whole = (finished || board.count >= max_moves) && status.load() == DGNN_OK;   // a whole game: two passes on a scorable board, or the stone cap
        if (whole) {
            stones_completed.fetch_add((uint64_t)board.count);                  // (before `completed`: a window edge between the two counts a game without its stones at worst)
            completed.fetch_add(1);
        }
        dgnn_tree_node_free(roots[0]);
        dgnn_tree_node_free(roots[1]);
        if (private_cache > 0) dgnn_cache_destroy(cache);

        unsigned char point_status[kPoints];                                   // GameResult::Ended, game_result.rs:35-41
        stone_status(board.inner, point_status);
        char header[160], date[40];
        const std::time_t now = std::time(nullptr);
        std::tm utc;
        gmtime_r(&now, &utc);
        std::strftime(date, sizeof(date), "%Y-%m-%dT%H:%M:%S+0000", &utc);
        std::snprintf(header, sizeof(header), "(;GM[1]FF[4]DT[%s]SZ[19]RU[Chinese]KM[%.1f]RE[%s]", date, (double)komi, winner_sgf(board, point_status).c_str());
        // the record goes where the reference's goes -- over the channel to whoever prints it (self_play.rs:488-492) -- and is let go of:
        // a job that replaces its games for hours must not hold a megabyte and a half for each that ever ended
        std::string record = header + sgf + territory_sgf(point_status) + ")";
        if (params->on_game) {
            std::lock_guard<std::mutex> lock(on_game_mutex);
            params->on_game(params->on_game_ctx, first_game + g * game_stride, record.data(), record.size(), whole ? 1 : 0);
        }
        if (sgf_out) records[(size_t)g] = std::move(record);
    };
    std::atomic<int> next_game{0};
    auto play = [&](int) { for (int g; !stop.load(std::memory_order_relaxed) && (g = next_game.fetch_add(1)) < num_games && status.load() == DGNN_OK;) game(g); };
    // A measurement window over a job that is already running (games per hour on the driver's clock): a watcher thread
    // notes the job's counters `ramp_seconds` after the start and again `seconds` later, then stops the job -- every game finishes the
    // move it is searching and leaves.  With num_parallel games in flight and num_games larger than the window can consume, games that
    // end inside the window are replaced at once (self_play.rs:478-495), so the window sees a steady number of games.
    dgnn_selfplay_window* window = params->window;
    std::thread watcher;
    std::mutex watch_mutex;
    MutexLife watch_mutex_life{&watch_mutex};
    std::condition_variable watch_cv;
    bool job_over = false;
    if (window && window->seconds > 0.0) {
        watcher = std::thread([&] {
            struct Snapshot { uint64_t moves, evals, hits, conflicts, started, completed, ended, stones, ex_it; std::chrono::steady_clock::time_point at; };
            auto snapshot = [&] { return Snapshot{moves.load(), evals.load(), hits.load(), conflicts.load(), started.load(), completed.load(), ended.load(), stones_completed.load(), ex_it_searches.load(), std::chrono::steady_clock::now()}; };
            auto sleep_until = [&](double seconds_after_start) {              // false: the job ended before that
                std::unique_lock<std::mutex> lock(watch_mutex);
                const long long left_us = (long long)(1e6 * (seconds_after_start - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()));
                return !wait_for_us(watch_cv, lock, left_us > 0 ? left_us : 0, [&] { return job_over; });
            };
            sleep_until(std::max(0.0, window->ramp_seconds));
            if (window->on_edge) window->on_edge(window->ctx, 0);
            const Snapshot a = snapshot();
            sleep_until(std::max(0.0, window->ramp_seconds) + window->seconds);
            const Snapshot b = snapshot();
            if (window->on_edge) window->on_edge(window->ctx, 1);
            stop.store(true);
            window->measured_seconds = std::chrono::duration<double>(b.at - a.at).count();
            window->moves = b.moves - a.moves;
            window->evals = b.evals - a.evals;
            window->cache_hits = b.hits - a.hits;
            window->conflicts = b.conflicts - a.conflicts;
            window->games_started = b.started - a.started;
            window->games_completed = b.completed - a.completed;
            window->games_ended = b.ended - a.ended;
            window->stones_completed = b.stones - a.stones;
            window->ex_it_searches = b.ex_it - a.ex_it;
        });
    }
    if (params->host_threads > 0 && fibers_available()) {
        run_fibers(num_parallel, params->host_threads, play);             // num_parallel games in flight on host_threads OS threads
    } else {
        std::vector<std::thread> threads;
        for (int t = 0; t < num_parallel; ++t) threads.emplace_back(play, t);
        for (std::thread& t : threads) t.join();
    }
    if (watcher.joinable()) {
        { std::lock_guard<std::mutex> lock(watch_mutex); job_over = true; }
        watch_cv.notify_all();
        watcher.join();
    }
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->games = window && window->seconds > 0.0 ? started.load() : (uint64_t)num_games;
    stats->moves = moves.load();
    stats->evals = evals.load();
    stats->conflicts = conflicts.load();
    stats->cache_hits = hits.load();
    stats->games_ended = ended.load();
    stats->ex_it_searches = ex_it_searches.load();
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
