// Scope row f-2 (first part): the Monte-Carlo search tree of dream-go's libdg_mcts as a C++ library behind the C ABI.
//
//   Node, Child           src/libdg_mcts/tree.rs:259-360,1027-1095.  Children are SPARSE: a node holds its prior (368 floats) and
//                         a slot per child that has ever been selected or disqualified -- the reference's "small" children
//                         (tree.rs:633-770, SMALL_SIZE 8) taken to its end: a child without a slot has the values of
//                         Child::with_value (tree.rs:276-285), so nothing observable changes and a node is 1.8 KB instead of
//                         10.7 KB (one node per rollout; 256 trees of dense nodes were 2 GB and every select a cache miss)
//   select                tree.rs:1329-1384: child values, first-play urgency (tree.rs:180-254), PUCT (tree.rs:41-178), argmax
//                         (asm/argmax.rs:24-78, including which index wins a tie), "expanding" flag, virtual loss
//   probe / undo / insert tree.rs:1397-1510;  UCT::update (incremental mean and sum of squares) tree.rs:125-160
//   best / softmax        tree.rs:1262-1313, compare_children tree.rs:1524-1560, normal_lcb_m libdg_utils/lcb.rs:28-37,
//                         choose libdg_mcts/choose.rs:23-106
//   constants             VLOSS_CNT 32, UCT_EXP / FPU_REDUCE / CRITICAL_VALUE interpolation tables and get_intp_value,
//                         libdg_utils/config.rs:181-195,289-305
// The tree knows nothing about Go: the board is a pair of callbacks (place a stone while descending), legality is encoded
// by the caller in the prior (-inf for illegal moves) exactly as the reference's policy_helper does.  Concurrency follows
// the reference: atomic counters, one spin lock per node around the value update, the "expanding" flag against double
// expansion; several threads may probe / insert into one tree.
#include <algorithm>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "dgnn_internal.h"

namespace dgnn {
namespace {

constexpr int kMoves = DGNN_NUM_MOVES;     // 362
constexpr int kPadded = 368;               // tree.rs:1045: padded to a multiple of 8 for the SIMD loops
constexpr int kVlossCnt = 32;              // config.rs:187
constexpr int kMinLcbVisits = 80;          // tree.rs:34

struct Point2 { int x; float y; };
const Point2 kFpuReduce[] = {{0, 0.631571f}, {800, 0.431547f}, {1600, 0.656083f}, {3200, 0.429231f}, {6400, 0.514494f}};
const Point2 kUctExp[] = {{0, 0.77392f}, {800, 1.05439f}, {1600, 1.22798f}, {3200, 0.813532f}, {6400, 0.764326f}};
const Point2 kCriticalValue[] = {{0, 1.91753f}, {800, 1.86478f}, {1600, 1.86943f}, {3200, 2.20033f}, {6400, 1.78053f}};

float intp_value(const Point2* points, int n, int x)   // config.rs:289-305
{
    int i = 0;
    while (i < n && points[i].x < x) ++i;
    if (i == n) return points[n - 1].y;
    const Point2& x0 = points[i == 0 ? 0 : i - 1];
    const Point2& x1 = points[i];
    const float a = x0.x >= x1.x ? 0.5f : (float)(x - x0.x) / (float)(x1.x - x0.x);
    return (1.0f - a) * x0.y + a * x1.y;
}

// asm/argmax.rs:24-78: groups of 8; within the running maximum the LAST group that holds a maximal element wins and
// inside it the first such element; None (-1) when the winner is not finite.
int argmax_f32_scalar(const float* v, int n)
{
    float so_far = -std::numeric_limits<float>::infinity();
    int index = 0;
    for (int g = 0; g < n / 8; ++g) {
        float m = v[8 * g];
        for (int j = 1; j < 8; ++j) m = std::max(m, v[8 * g + j]);
        so_far = std::max(so_far, m);
        for (int j = 0; j < 8; ++j)
            if (v[8 * g + j] == so_far) { index = 8 * g + j; break; }
    }
    return std::isfinite(v[index]) ? index : -1;
}

// The same answer with 256-bit vectors (the reference's is AVX2 as well): the running maximum ends as the global maximum
// M, and the index ends in the last group that holds an element equal to M, at its first such element.
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
int argmax_f32_avx2(const float* v, int n)
{
    const int groups = n / 8;
    if (groups == 0) return std::isfinite(v[0]) ? 0 : -1;
    __m256 best = _mm256_loadu_ps(v);
    for (int g = 1; g < groups; ++g) best = _mm256_max_ps(best, _mm256_loadu_ps(v + 8 * g));
    __m128 m4 = _mm_max_ps(_mm256_castps256_ps128(best), _mm256_extractf128_ps(best, 1));
    m4 = _mm_max_ps(m4, _mm_movehl_ps(m4, m4));
    m4 = _mm_max_ss(m4, _mm_shuffle_ps(m4, m4, 1));
    const float top = _mm_cvtss_f32(m4);
    if (!std::isfinite(top)) return (top != top) ? argmax_f32_scalar(v, n) : -1;          // NaN: take the scalar walk; +-inf: None
    const __m256 vtop = _mm256_set1_ps(top);
    for (int g = groups - 1; g >= 0; --g) {
        const int mask = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(v + 8 * g), vtop, _CMP_EQ_OQ));
        if (mask) return 8 * g + __builtin_ctz((unsigned)mask);
    }
    return argmax_f32_scalar(v, n);
}
#endif

const bool kHaveAvx2 = host_simd();
int argmax_f32(const float* v, int n)
{
#if DGNN_X86_SIMD
    if (kHaveAvx2) return argmax_f32_avx2(v, n);
#endif
    return argmax_f32_scalar(v, n);
}

struct SpinLock {
    std::atomic_flag flag = ATOMIC_FLAG_INIT;
    void lock() { while (flag.test_and_set(std::memory_order_acquire)) {} }
    void unlock() { flag.clear(std::memory_order_release); }
};

}  // namespace

struct TreeNode;

// One child that has been touched: selected by a probe (set_expanding creates the slot), or disqualified.  Slots never move and
// never go away while the node lives, so a trace may keep a pointer to one.
struct ChildSlot {
    std::atomic<int> count{0};
    std::atomic<int> vcount{0};
    // the running mean and sum of squares (UCT::update, tree.rs:125-160): written under the node's lock, read by every select() of
    // every search thread without it, as in the reference (tree.rs:63-115 reads the arrays plainly).  Relaxed atomics: the same
    // loads and stores on x86, and a race detector has nothing to report
    std::atomic<float> value{0.0f};
    std::atomic<float> value_s{0.0f};
    std::atomic<TreeNode*> ptr{nullptr};
    uint16_t index = 0;
    std::atomic<unsigned char> expanding{0};
};

struct TreeNode {
    static constexpr int kInline = 8;          // SMALL_SIZE, tree.rs:633: the slots that live inside the node
    static constexpr int kChunks = 5;          // further slots in chunks of 16, 32, 64, 128, 256 (8 + 496 >= 362), allocated on demand
    static constexpr int kMapFrom = 16;        // from this many slots on a node keeps an index -> slot map (a root: hundreds of children)

    SpinLock lock;                             // the value update of a child (UCT::update) and the creation of a slot
    int to_move;
    float initial_value;
    int pass_count = 0;
    std::atomic<int> total_count{0};
    std::atomic<int> vtotal_count{0};
    std::atomic<int> num_slots{0};             // slots 0 .. num_slots-1 are initialised and visible (release / acquire)
    std::atomic<std::atomic<uint16_t>*> slot_of{nullptr};   // [362]: slot number + 1, 0 = no slot
    std::atomic<ChildSlot*> chunk[kChunks];
    ChildSlot first[kInline];
    float prior[kPadded];

    TreeNode(int color, float v, const float* p) : to_move(color), initial_value(v)
    {
        // one node per rollout: 1.5 KB of prior and nothing else to write (the node is not visible to any other thread before the
        // constructor returns -- dgnn_tree_insert publishes it with a CAS)
        for (auto& c : chunk) c.store(nullptr, std::memory_order_relaxed);
        std::memcpy(prior, p, kMoves * sizeof(float));
        for (int i = kMoves; i < kPadded; ++i) prior[i] = -std::numeric_limits<float>::infinity();   // tree.rs:1075-1076
    }

    ~TreeNode()
    {
        const int n = num_slots.load(std::memory_order_relaxed);
        for (int k = 0; k < n; ++k) delete slot(k)->ptr.load(std::memory_order_relaxed);
        for (auto& c : chunk) delete[] c.load(std::memory_order_relaxed);
        delete[] slot_of.load(std::memory_order_relaxed);
    }

    // (one node per rollout, freed by the thousand when a root moves on: dgnn_internal.h BlockPool)
    static void* operator new(size_t, const std::nothrow_t&) noexcept;
    static void operator delete(void* p) noexcept;

    // slot number -> its place: 0..7 inside the node, 8..23 chunk 0, 24..55 chunk 1, 56..119, 120..247, 248..503
    static int chunk_of(int k) { return 31 - __builtin_clz((unsigned)(k / kInline + 1)) - 1; }
    static int chunk_base(int j) { return kInline * ((2 << j) - 1); }
    static int chunk_size(int j) { return kInline * (2 << j); }
    ChildSlot* slot(int k)
    {
        if (k < kInline) return &first[k];
        const int j = chunk_of(k);
        return chunk[j].load(std::memory_order_acquire) + (k - chunk_base(j));
    }
    const ChildSlot* slot(int k) const { return const_cast<TreeNode*>(this)->slot(k); }

    // every visible slot, in slot order: the eight inside the node, then chunk after chunk (no per-slot address arithmetic)
    template <typename F>
    void for_each_slot(F&& f) const
    {
        const int n = num_slots.load(std::memory_order_acquire);
        for (int k = 0; k < n && k < kInline; ++k) f(first[k]);
        for (int j = 0, base = kInline; base < n; base += chunk_size(j), ++j) {
            const ChildSlot* c = chunk[j].load(std::memory_order_acquire);
            const int m = std::min(n - base, chunk_size(j));
            for (int k = 0; k < m; ++k) f(c[k]);
        }
    }

    // the slot of child `index`, or NULL: such a child is Child::with_value(initial_value) -- no visits, no virtual loss, not
    // expanding, no sub-tree (tree.rs:901-927)
    ChildSlot* find(int index)
    {
        if (const std::atomic<uint16_t>* map = slot_of.load(std::memory_order_acquire)) {
            const int k = map[index].load(std::memory_order_acquire);
            return k ? slot(k - 1) : nullptr;
        }
        const int n = num_slots.load(std::memory_order_acquire);
        if (n >= kMapFrom) {
            // another search thread added slots between the two loads above.  The map is published BEFORE the slot count that needs it
            // (find_or_create), so it is visible now; scanning on would run past chunk 0's sixteen slots
            const std::atomic<uint16_t>* map = slot_of.load(std::memory_order_acquire);
            const int k = map[index].load(std::memory_order_acquire);
            return k ? slot(k - 1) : nullptr;
        }
        for (int k = 0; k < n && k < kInline; ++k)                    // fewer than kMapFrom slots: the inline ones and at most 7 of chunk 0
            if (first[k].index == index) return &first[k];
        if (n > kInline) {
            ChildSlot* c = chunk[0].load(std::memory_order_acquire);
            for (int k = 0; k < n - kInline; ++k)
                if (c[k].index == index) return &c[k];
        }
        return nullptr;
    }
    const ChildSlot* find(int index) const { return const_cast<TreeNode*>(this)->find(index); }

    // tree.rs:929-995 (with_mut: a child that is written to gets a slot).  NULL only when memory is exhausted.
    ChildSlot* find_or_create(int index)
    {
        if (ChildSlot* s = find(index)) return s;
        lock.lock();
        ChildSlot* s = find(index);                      // (another search thread may have created it meanwhile)
        if (!s) {
            const int n = num_slots.load(std::memory_order_relaxed);
            if (n >= kInline) {
                const int j = chunk_of(n);
                if (!chunk[j].load(std::memory_order_relaxed)) {
                    ChildSlot* fresh = new (std::nothrow) ChildSlot[chunk_size(j)];
                    if (!fresh) { lock.unlock(); return nullptr; }
                    chunk[j].store(fresh, std::memory_order_release);
                }
            }
            std::atomic<uint16_t>* map = slot_of.load(std::memory_order_relaxed);
            if (!map && n + 1 >= kMapFrom) {
                map = new (std::nothrow) std::atomic<uint16_t>[kMoves];
                if (!map) { lock.unlock(); return nullptr; }            // (find() scans no further than chunk 0 without the map)
                for (int i = 0; i < kMoves; ++i) map[i].store(0, std::memory_order_relaxed);
                for (int k = 0; k < n; ++k) map[slot(k)->index].store((uint16_t)(k + 1), std::memory_order_relaxed);
            }
            s = slot(n);
            s->index = (uint16_t)index;
            s->value.store(initial_value, std::memory_order_relaxed);
            // order of publication: the slot's fields, its map entry, the map itself, THEN the count -- a reader that sees
            // num_slots >= kMapFrom thereby sees the map, and a reader that finds the slot through the map sees its fields
            if (map) map[index].store((uint16_t)(n + 1), std::memory_order_release);
            if (map && !slot_of.load(std::memory_order_relaxed)) slot_of.store(map, std::memory_order_release);
            num_slots.store(n + 1, std::memory_order_release);
        }
        lock.unlock();
        return s;
    }
};

using TreeNodePool = BlockPool<sizeof(TreeNode), 2048>;
void* TreeNode::operator new(size_t, const std::nothrow_t&) noexcept { return TreeNodePool::get(); }
void TreeNode::operator delete(void* p) noexcept { TreeNodePool::put(p); }

// A node's children by move index, for the walks that happen once per move (best, softmax, to_pretty): NULL = no slot.
struct ChildView {
    const TreeNode* node;
    const ChildSlot* at[kMoves];
    explicit ChildView(const TreeNode* n) : node(n)
    {
        std::memset(at, 0, sizeof(at));
        n->for_each_slot([&](const ChildSlot& s) { at[s.index] = &s; });
    }
    int count(int i) const { return at[i] ? at[i]->count.load(std::memory_order_relaxed) : 0; }
    float value(int i) const { return at[i] ? at[i]->value.load(std::memory_order_relaxed) : node->initial_value; }
    float value_s(int i) const { return at[i] ? at[i]->value_s.load(std::memory_order_relaxed) : 0.0f; }
    const TreeNode* ptr(int i) const { return at[i] ? at[i]->ptr.load(std::memory_order_acquire) : nullptr; }
};

struct TraceEntry { TreeNode* node; ChildSlot* slot; int color; int index; };
struct Trace { std::vector<TraceEntry> entries; };

namespace {

// FPU::apply + UCT::get (tree.rs:180-254, 86-115): value[i] <- fpu(value[i]) + prior[i] * bonus(total[i]).  A child without a
// slot has value = the node's initial value and no visits, so the whole array starts as fpu(initial) + prior[i] * bonus(0) -- the
// same operations on the same operands as the per-child form -- and the few children with a slot are then computed on their own.
struct PuctTerms { bool apply_fpu; float fpu_reduce, uct_exp_sqrt_n; };

void puct_scalar(const TreeNode* node, const PuctTerms& t, float* value)
{
    float v0 = node->initial_value;
    if (t.apply_fpu) v0 = std::max(v0 - t.fpu_reduce, 0.0f);
    for (int i = 0; i < kPadded; ++i) value[i] = v0 + node->prior[i] * t.uct_exp_sqrt_n;
    node->for_each_slot([&](const ChildSlot& s) {
        const int c = s.count.load(std::memory_order_relaxed) + s.vcount.load(std::memory_order_relaxed);
        float v = s.value.load(std::memory_order_relaxed);
        if (t.apply_fpu && c == 0) v = std::max(v - t.fpu_reduce, 0.0f);
        const float bonus = c == 0 ? t.uct_exp_sqrt_n : t.uct_exp_sqrt_n / (float)(1 + c);
        value[s.index] = v + node->prior[s.index] * bonus;
    });
}

// The reference's own formulation (tree.rs:63-115: cmpeq / div / mul / add on 8 lanes, no fused multiply-add), so both
// variants give the same bits; the children with a slot go through the same instructions, one lane at a time (max_ss, div_ss:
// what a lane of the reference's vectors computes, including which operand a NaN leaves behind).  The counters are read as plain
// loads, racing with other search threads exactly as the reference's do.
#if DGNN_X86_SIMD
__attribute__((target("avx2")))
void puct_avx2(const TreeNode* node, const PuctTerms& t, float* value)
{
    const __m256 u = _mm256_set1_ps(t.uct_exp_sqrt_n);
    __m256 v0 = _mm256_set1_ps(node->initial_value);
    if (t.apply_fpu) v0 = _mm256_max_ps(_mm256_sub_ps(v0, _mm256_set1_ps(t.fpu_reduce)), _mm256_setzero_ps());
    for (int i = 0; i < kPadded; i += 8)
        _mm256_storeu_ps(value + i, _mm256_add_ps(v0, _mm256_mul_ps(_mm256_loadu_ps(node->prior + i), u)));
    const __m128 u1 = _mm_set_ss(t.uct_exp_sqrt_n), fpu1 = _mm_set_ss(t.fpu_reduce), zero1 = _mm_setzero_ps();
    const bool apply_fpu = t.apply_fpu;
    node->for_each_slot([&](const ChildSlot& s) __attribute__((target("avx2"))) {
        const int c = s.count.load(std::memory_order_relaxed) + s.vcount.load(std::memory_order_relaxed);
        __m128 v = _mm_set_ss(s.value.load(std::memory_order_relaxed));
        if (apply_fpu && c == 0) v = _mm_max_ss(_mm_sub_ss(v, fpu1), zero1);
        const __m128 bonus = c == 0 ? u1 : _mm_div_ss(u1, _mm_cvtsi32_ss(zero1, 1 + c));
        value[s.index] = _mm_cvtss_f32(_mm_add_ss(v, _mm_mul_ss(_mm_set_ss(node->prior[s.index]), bonus)));
    });
}
#endif

// tree.rs:1329-1384.  Returns 1 found (index, slot, value filled), 0 no result, -1 conflict, -2 out of memory.
int select(TreeNode* node, bool apply_fpu, int* index_out, ChildSlot** slot_out, float* value_out)
{
    float value[kPadded];
    const int n = node->total_count.load(std::memory_order_relaxed) + node->vtotal_count.load(std::memory_order_relaxed);
    PuctTerms t;
    t.apply_fpu = apply_fpu;
    t.fpu_reduce = apply_fpu ? intp_value(kFpuReduce, 5, n) : 0.0f;          // tree.rs:1342-1348
    const float sqrt_n = std::sqrt((float)(1 + n));                          // tree.rs:86-115
    t.uct_exp_sqrt_n = intp_value(kUctExp, 5, n) * sqrt_n;
#if DGNN_X86_SIMD
    if (kHaveAvx2) puct_avx2(node, t, value);
    else
#endif
    puct_scalar(node, t, value);
    for (int i = kMoves; i < kPadded; ++i) value[i] = -std::numeric_limits<float>::infinity();

    const int max_i = argmax_f32(value, kPadded);
    if (max_i < 0) return 0;
    ChildSlot* slot = node->find_or_create(max_i);
    if (!slot) return -2;
    const bool was_expanding = slot->expanding.exchange(1, std::memory_order_acq_rel) != 0;   // set_expanding, tree.rs:445-449
    if (was_expanding && slot->ptr.load(std::memory_order_acquire) == nullptr) return -1;
    slot->vcount.fetch_add(kVlossCnt, std::memory_order_relaxed);
    node->vtotal_count.fetch_add(kVlossCnt, std::memory_order_relaxed);
    *index_out = max_i;
    *slot_out = slot;
    *value_out = value[max_i];
    return 1;
}

void undo(const Trace& trace, bool undo_expanding)                          // tree.rs:1397-1409
{
    for (const TraceEntry& e : trace.entries) {
        e.node->vtotal_count.fetch_sub(kVlossCnt, std::memory_order_relaxed);
        e.slot->vcount.fetch_sub(kVlossCnt, std::memory_order_relaxed);
        if (undo_expanding && e.slot->ptr.load(std::memory_order_acquire) == nullptr)
            e.slot->expanding.store(0, std::memory_order_release);
    }
}

float value_std(float value_s, int count)                                    // tree.rs:351-358
{
    return std::sqrt(value_s / ((float)count + 1e-5f));
}

float normal_lcb_m(float p_hat, float p_std, int n, int m)                  // lcb.rs:28-37
{
    if (n <= 0) return 0.0f;
    return p_hat - intp_value(kCriticalValue, 5, m) * p_std / std::sqrt((float)n);
}

// compare_children, tree.rs:1524-1560: < 0 when a is the worse candidate
int compare_children(const ChildView& view, int a, int b)
{
    const TreeNode* node = view.node;
    const int ac = view.count(a), bc = view.count(b);
    if (ac >= kMinLcbVisits && bc >= kMinLcbVisits) {
        const int m = node->total_count.load(std::memory_order_relaxed);
        const float al = normal_lcb_m(view.value(a), value_std(view.value_s(a), ac), ac, m);
        const float bl = normal_lcb_m(view.value(b), value_std(view.value_s(b), bc), bc, m);
        if (al != bl) return al < bl ? -1 : 1;
    }
    if (ac != bc) return ac < bc ? -1 : 1;
    if (node->prior[a] != node->prior[b]) return node->prior[a] < node->prior[b] ? -1 : 1;
    if (view.value(a) != view.value(b)) return view.value(a) < view.value(b) ? -1 : 1;
    return 0;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" {

float dgnn_config_uct_exp(int visits) { return intp_value(kUctExp, 5, visits); }
float dgnn_config_fpu_reduce(int visits) { return intp_value(kFpuReduce, 5, visits); }
float dgnn_config_lcb_critical_value(int visits) { return intp_value(kCriticalValue, 5, visits); }
int dgnn_config_vloss_cnt(void) { return kVlossCnt; }

float dgnn_intp_value(const int* xs, const float* ys, int n, int x)
{
    if (!xs || !ys || n <= 0 || n > 64) return std::numeric_limits<float>::quiet_NaN();
    Point2 pts[64];
    for (int i = 0; i < n; ++i) { pts[i].x = xs[i]; pts[i].y = ys[i]; }
    return intp_value(pts, n, x);
}

int dgnn_argmax_f32(const float* values, int n)
{
    if (!values || n <= 0 || n % 8 != 0) return -2;
    return argmax_f32(values, n);
}

dgnn_tree_node* dgnn_tree_node_new(int to_move, float value, const float* prior)
{
    if (!prior) return nullptr;
    return reinterpret_cast<dgnn_tree_node*>(new (std::nothrow) TreeNode(to_move, value, prior));
}

void dgnn_tree_node_free(dgnn_tree_node* node) { delete reinterpret_cast<TreeNode*>(node); }

int dgnn_tree_probe(dgnn_tree_node* root, const dgnn_board_ops* ops, void* board, dgnn_trace** out)
{
    if (!root || !out) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    Trace* trace = new (std::nothrow) Trace();
    if (!trace) return DGNN_OUT_OF_MEMORY;
    trace->entries.reserve(16);
    TreeNode* current = reinterpret_cast<TreeNode*>(root);

    for (;;) {                                                               // tree.rs:1421-1470
        int next_child = 0;
        float next_value = 0.0f;
        ChildSlot* slot = nullptr;
        const int r = select(current, !trace->entries.empty(), &next_child, &slot, &next_value);
        if (r == -2) { undo(*trace, false); delete trace; return DGNN_OUT_OF_MEMORY; }
        if (r < 0) { undo(*trace, false); delete trace; return DGNN_PROBE_CONFLICT; }   // (select's own -1, not the ABI's value)
        if (r == 0) { delete trace; return DGNN_PROBE_NO_RESULT; }            // (the reference leaks the partial trace's virtual loss here too)
        trace->entries.push_back({current, slot, current->to_move, next_child});
        if (next_child != kMoves - 1) {
            if (ops && ops->place) ops->place(board, current->to_move, next_child);
        } else if (current->pass_count >= 1) {
            break;                                                           // at least two consecutive passes
        }
        TreeNode* child = slot->ptr.load(std::memory_order_acquire);
        if (!child) break;
        current = child;
    }
    *out = reinterpret_cast<dgnn_trace*>(trace);
    return DGNN_PROBE_FOUND;
}

int dgnn_trace_len(const dgnn_trace* trace) { return trace ? (int)reinterpret_cast<const Trace*>(trace)->entries.size() : 0; }

int dgnn_trace_get(const dgnn_trace* trace, int i, dgnn_tree_node** node, int* color, int* index)
{
    if (!trace) return DGNN_BAD_ARGUMENT;
    const Trace* t = reinterpret_cast<const Trace*>(trace);
    if (i < 0 || i >= (int)t->entries.size()) return DGNN_BAD_ARGUMENT;
    if (node) *node = reinterpret_cast<dgnn_tree_node*>(t->entries[i].node);
    if (color) *color = t->entries[i].color;
    if (index) *index = t->entries[i].index;
    return DGNN_OK;
}

void dgnn_trace_free(dgnn_trace* trace) { delete reinterpret_cast<Trace*>(trace); }

int dgnn_tree_undo(dgnn_trace* trace, int undo_expanding)
{
    if (!trace) return DGNN_BAD_ARGUMENT;
    undo(*reinterpret_cast<Trace*>(trace), undo_expanding != 0);
    return DGNN_OK;
}

int dgnn_tree_insert(const dgnn_trace* handle, int color, float value, const float* prior)
{
    if (!handle || !prior) return DGNN_BAD_ARGUMENT;
    const Trace* trace = reinterpret_cast<const Trace*>(handle);
    if (!trace->entries.empty()) {                                           // tree.rs:1482-1507
        const TraceEntry& last = trace->entries.back();
        TreeNode* next = new (std::nothrow) TreeNode(color, value, prior);
        if (!next) return DGNN_OUT_OF_MEMORY;
        if (last.index == kMoves - 1) next->pass_count = last.node->pass_count + 1;
        TreeNode* expected = nullptr;
        if (!last.slot->ptr.compare_exchange_strong(expected, next, std::memory_order_acq_rel))
            delete next;                                                     // double expansion behind two passes
    }
    for (const TraceEntry& e : trace->entries) {                              // UCT::update, tree.rs:125-160
        TreeNode* node = e.node;
        const float v = color == node->to_move ? value : 1.0f - value;
        node->total_count.fetch_add(1, std::memory_order_relaxed);
        node->vtotal_count.fetch_sub(kVlossCnt, std::memory_order_relaxed);
        ChildSlot* slot = e.slot;
        node->lock.lock();
        const float prev_value = slot->value.load(std::memory_order_relaxed);
        const float prev_value_s = slot->value_s.load(std::memory_order_relaxed);
        const int prev_count = slot->count.fetch_add(1, std::memory_order_relaxed);
        const float next_value = prev_value + (v - prev_value) / (float)(prev_count + 1);
        slot->value.store(next_value, std::memory_order_relaxed);
        slot->value_s.store(prev_value_s + (v - prev_value) * (v - next_value), std::memory_order_relaxed);
        slot->vcount.fetch_sub(kVlossCnt, std::memory_order_relaxed);
        node->lock.unlock();
    }
    return DGNN_OK;
}

int dgnn_tree_node_child(const dgnn_tree_node* handle, int index, int* count, int* vcount, float* value, float* value_std_out,
                         int* has_child)
{
    if (!handle || index < 0 || index >= kMoves) return DGNN_BAD_ARGUMENT;
    const TreeNode* node = reinterpret_cast<const TreeNode*>(handle);
    const ChildSlot* slot = node->find(index);                                // none: Child::with_value, tree.rs:276-285
    const int visits = slot ? slot->count.load() : 0;
    if (count) *count = visits;
    if (vcount) *vcount = slot ? slot->vcount.load() : 0;
    if (value) *value = slot ? slot->value.load(std::memory_order_relaxed) : node->initial_value;
    if (value_std_out) *value_std_out = value_std(slot ? slot->value_s.load(std::memory_order_relaxed) : 0.0f, visits);
    if (has_child) *has_child = slot && slot->ptr.load() != nullptr;
    return DGNN_OK;
}

int dgnn_tree_node_info(const dgnn_tree_node* handle, int* to_move, int* total_count, int* vtotal_count, int* pass_count,
                        float* initial_value)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    const TreeNode* node = reinterpret_cast<const TreeNode*>(handle);
    if (to_move) *to_move = node->to_move;
    if (total_count) *total_count = node->total_count.load();
    if (vtotal_count) *vtotal_count = node->vtotal_count.load();
    if (pass_count) *pass_count = node->pass_count;
    if (initial_value) *initial_value = node->initial_value;
    return DGNN_OK;
}

int dgnn_tree_node_prior(const dgnn_tree_node* handle, float* prior_out)
{
    if (!handle || !prior_out) return DGNN_BAD_ARGUMENT;
    std::memcpy(prior_out, reinterpret_cast<const TreeNode*>(handle)->prior, kMoves * sizeof(float));
    return DGNN_OK;
}

// Node::forward (tree.rs:1225-1250): the sub-tree below `index` becomes the root, everything else is freed.  A pass that was
// never expanded yields a fresh node that only records the pass; any other unexpanded child yields NULL.
dgnn_tree_node* dgnn_tree_node_forward(dgnn_tree_node* handle, int index)
{
    if (!handle || index < 0 || index >= kMoves) return nullptr;
    TreeNode* node = reinterpret_cast<TreeNode*>(handle);
    ChildSlot* slot = node->find(index);
    TreeNode* next = slot ? slot->ptr.exchange(nullptr) : nullptr;
    if (!next && index == kMoves - 1) {
        float prior[kMoves];
        for (float& v : prior) v = 0.0f;
        next = new (std::nothrow) TreeNode(1 - node->to_move, 0.5f, prior);
        if (next) next->pass_count = node->pass_count + 1;
    }
    delete node;
    return reinterpret_cast<dgnn_tree_node*>(next);
}

// min_promote_rollouts (time_control/mod.rs:48-73): the lead of the most visited child over the runner-up
int dgnn_tree_node_min_promote(const dgnn_tree_node* handle)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    const TreeNode* node = reinterpret_cast<const TreeNode*>(handle);
    // the largest count and the largest among the OTHER children (a child without a slot counts 0, and so does the floor)
    int count_1 = 0, count_2 = 0;
    node->for_each_slot([&](const ChildSlot& s) {
        const int c = s.count.load(std::memory_order_relaxed);
        if (c > count_1) { count_2 = count_1; count_1 = c; }
        else if (c > count_2) count_2 = c;
    });
    return count_1 > count_2 ? count_1 - count_2 : 0;
}

// time_control::is_done over RolloutLimit (time_control/mod.rs:86-101, rollout_limit.rs:35-44): the limit is reached, or --
// EARLY-C -- the runner-up cannot catch up with the most visited child in the rollouts that remain.
int dgnn_rollout_limit_is_done(const dgnn_tree_node* handle, int limit)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    const int total = reinterpret_cast<const TreeNode*>(handle)->total_count.load(std::memory_order_relaxed);
    if (total == 0) return 0;
    if (total >= limit) return 1;
    return dgnn_tree_node_min_promote(handle) > limit - total ? 1 : 0;
}

int dgnn_tree_node_set_prior(dgnn_tree_node* handle, const float* prior)     // lib.rs:176-183: a re-used tree gets the new root prior
{
    if (!handle || !prior) return DGNN_BAD_ARGUMENT;
    std::memcpy(reinterpret_cast<TreeNode*>(handle)->prior, prior, kMoves * sizeof(float));
    return DGNN_OK;
}

int dgnn_tree_node_disqualify(dgnn_tree_node* handle, int index)             // tree.rs:1315-1320
{
    if (!handle || index < 0 || index >= kMoves) return DGNN_BAD_ARGUMENT;
    TreeNode* node = reinterpret_cast<TreeNode*>(handle);
    ChildSlot* slot = node->find_or_create(index);
    if (!slot) return DGNN_OUT_OF_MEMORY;
    slot->value.store(-std::numeric_limits<float>::infinity(), std::memory_order_relaxed);
    slot->count.store(0);
    return DGNN_OK;
}

int dgnn_tree_node_softmax(const dgnn_tree_node* handle, float* out)         // tree.rs:1293-1307
{
    if (!handle || !out) return DGNN_BAD_ARGUMENT;
    const ChildView view(reinterpret_cast<const TreeNode*>(handle));
    float total = 0.0f;
    for (int i = 0; i < kMoves; ++i) total += (float)view.count(i);
    for (int i = 0; i < kMoves; ++i) {
        const int c = view.count(i);
        out[i] = c != 0 ? (float)c / total : 0.0f;
    }
    return DGNN_OK;
}

// choose, libdg_mcts/choose.rs:23-106
extern "C++" {
template <typename T>
static int choose_impl(const T* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out)
{
    if (!items || n <= 0 || !index_out) return DGNN_BAD_ARGUMENT;
    double total = 0.0;
    for (int i = 0; i < n; ++i) if (std::isfinite((double)items[i])) total += (double)items[i];     // choose.rs:69-72
    // percentile: walk the items from the largest down until (1 - cutoff) of the total is covered
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a] < items[b]; });
    const double max_value = total * (1.0 - cutoff_percentile);
    double so_far = 0.0;
    T threshold = 0;
    bool found = false;
    for (int k = n - 1; k >= 0; --k) {
        so_far += (double)items[order[k]];
        if (so_far >= max_value) { threshold = items[order[k]]; found = true; break; }
    }
    if (!found) return 0;
    const double kept_total = so_far;
    std::vector<double> cum(n, std::numeric_limits<double>::quiet_NaN());
    double cum_total = 0.0;
    for (int i = 0; i < n; ++i)
        if (items[i] >= threshold) { cum_total += std::pow((double)items[i] / kept_total, temperature); cum[i] = cum_total; }
    const double target = at * cum_total;
    for (int i = 0; i < n; ++i)
        if (cum[i] >= target) { *index_out = i; if (cum_out) *cum_out = cum[i]; return 1; }
    return 0;
}
}  // extern "C++"

int dgnn_choose(const int* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out)
{
    return choose_impl(items, n, cutoff_percentile, temperature, at, index_out, cum_out);
}
int dgnn_choose_f64(const double* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out)
{
    return choose_impl(items, n, cutoff_percentile, temperature, at, index_out, cum_out);
}

// tree::to_pretty (src/libdg_mcts/tree.rs:1620-1746, non-verbose): what Played::from_mcts stores as a move's explanation
// (self_play.rs:136, written as C[...] with newlines as carriage returns, :191-193) --
//   "Nodes: <visits>, Win: <visit-weighted value>%, PV: <greedy path>" and, for the ten best children in compare_children order,
//   "<vertex> -> <child visits> (W: <value>% / <lower confidence bound>%) (N: <prior>%) PV: <vertex> <greedy path below it>".
// Vertices are GTP style (no i column, row 1 at packed y = 0), "pass" for 361.
extern "C++" {
namespace {
std::string pretty_vertex(int i)                                             // PrettyVertex, tree.rs:1620-1640
{
    if (i == kMoves - 1) return "pass";
    static const char letters[] = "abcdefghjklmnopqrst";
    return std::string(1, letters[i % 19]) + std::to_string(i / 19 + 1);
}
int greedy_child(const ChildView& view)                                      // GreedyPath::next, tree.rs:1659-1662
{
    int best = -1;
    for (int i = 0; i < kMoves; ++i) {
        if (view.count(i) == 0) continue;
        if (best < 0 || compare_children(view, best, i) <= 0) best = i;      // max_by keeps the last maximum
    }
    return best < 0 ? kMoves - 1 : best;
}
std::string greedy_path(const TreeNode* node)                                // GreedyPath::new(node, 1) joined by blanks
{
    std::string out;
    while (node) {
        const ChildView view(node);
        const int i = greedy_child(view);
        if (view.count(i) < 1) break;
        if (!out.empty()) out += ' ';
        out += pretty_vertex(i);
        node = view.ptr(i);
    }
    return out;
}
}  // namespace
}  // extern "C++"

size_t dgnn_tree_node_to_pretty(const dgnn_tree_node* handle, char* out, size_t capacity)
{
    if (!handle) return 0;
    const TreeNode* root = reinterpret_cast<const TreeNode*>(handle);
    const ChildView view(root);
    std::vector<int> children;
    for (int i = 0; i < kMoves; ++i) if (view.count(i) != 0) children.push_back(i);
    std::stable_sort(children.begin(), children.end(), [&](int a, int b) { return compare_children(view, b, a) < 0; });   // best first
    if (children.size() > 10) children.resize(10);
    float total_value = 0.0f;
    for (int i = 0; i < kMoves; ++i) {
        const float v = (float)view.count(i) * view.value(i);
        if (std::isfinite(v)) total_value += v;
    }
    const int total = root->total_count.load(std::memory_order_relaxed);
    char line[256];
    std::snprintf(line, sizeof(line), "Nodes: %d, Win: %.1f%%, PV: ", total, (double)(100.0f * (total_value / ((float)total + 1e-5f))));
    std::string text = line + greedy_path(root) + "\n";
    for (int i : children) {
        const TreeNode* child = view.ptr(i);
        const std::string vertex = pretty_vertex(i);
        std::snprintf(line, sizeof(line), "%5s -> %7d (W: %5.2f%% / %5.2f%%) (N: %5.2f%%) PV: %s ", vertex.c_str(),
                      child ? child->total_count.load(std::memory_order_relaxed) : 0, (double)(100.0f * view.value(i)),
                      (double)(100.0f * normal_lcb_m(view.value(i), value_std(view.value_s(i), view.count(i)), view.count(i), total)),
                      (double)(100.0f * root->prior[i]), vertex.c_str());
        text += line + (child ? greedy_path(child) : std::string()) + "\n";
    }
    if (out && capacity > text.size()) std::memcpy(out, text.c_str(), text.size() + 1);
    return text.size() + 1;
}

int dgnn_tree_node_best(const dgnn_tree_node* handle, float temperature, double at, float* value_out, int* index_out)
{
    if (!handle || !value_out || !index_out) return DGNN_BAD_ARGUMENT;
    const TreeNode* node = reinterpret_cast<const TreeNode*>(handle);
    const ChildView view(node);
    if (temperature <= 9e-2f) {                                              // greedy, tree.rs:1263-1268
        int best = -1;
        for (int i = 0; i < kMoves; ++i) {
            if (view.count(i) == 0) continue;                                // children.nonzero()
            if (best < 0 || compare_children(view, best, i) <= 0) best = i;  // Iterator::max_by keeps the LAST maximum
        }
        if (best < 0) best = kMoves - 1;
        *index_out = best;
        *value_out = view.value(best);
        return DGNN_OK;
    }
    int visits[kMoves];
    for (int i = 0; i < kMoves; ++i) visits[i] = view.count(i);
    int chosen = 0;
    if (dgnn_choose(visits, kMoves, 0.5, 1.0 / (double)temperature, at, &chosen, nullptr) == 1) {
        *index_out = chosen;
        *value_out = view.value(chosen);
    } else {
        *index_out = kMoves - 1;                                             // no valid moves, tree.rs:1279
        *value_out = node->initial_value;
    }
    return DGNN_OK;
}

}  // extern "C"
