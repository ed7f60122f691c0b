// Host side of the boundary, row f-1 of the scope table: the many-game leaf batcher, the transposition cache and the
// policy symmetry remap of dream-go's libdg_mcts, as a C++ library behind the C ABI (include/dgnn_host.h; single steps: dgnn_test.h).
//
//   symmetry   src/libdg_go/utils/symmetry.rs:19-128 (the 8 board symmetries and their inverses),
//              Prediction::with_transform, src/libdg_mcts/predictor.rs:30-44
//   cache      src/libdg_mcts/lru_cache.rs:56-150 (doubly linked hash map, insert of a present key is a no-op,
//              the tail is dropped when len > capacity) keyed like NnPredictor's BoardTuple
//              (src/libdg_mcts/predictors/nn.rs:29-45,69-82): entries are stored in the canonical orientation
//              (cache() applies symmetry.inverse(), fetch() applies symmetry)
//   batcher    src/libdg_mcts/pool/batch.rs:24-123: push appends, get_batch(min) hands out the NEWEST max_batch_size
//              entries once at least `min` are queued and fewer than max_batches batches are alive
//
// Differences by design: the reference has ONE global mutex around the LRU (nn.rs:48) and one around the batcher list
// (batch.rs:60-73); here the cache is sharded by key (shards = 1 reproduces the reference's eviction order exactly) and
// a batcher is per (network, device) so that games sharded over the GPUs of a node never share a lock; every batch size
// 1..max is served from workspaces of capacity max (dgnn_forward_n) instead of one workspace per size.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>
#include <list>
#include <memory>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "dgnn_internal.h"
#include "fiber.h"

namespace dgnn {

std::atomic<int> g_fail_push_countdown{-1};
namespace {

// ------------------------------------------------------------------------------------------------ symmetry

struct SymmetryTables {
    uint16_t map[8][DGNN_NUM_MOVES];   // map[t][i] = packed index of transform t applied to point i; pass maps to pass

    SymmetryTables()
    {
        for (int t = 0; t < 8; ++t) {
            for (int i = 0; i < DGNN_NUM_POINTS; ++i) {
                const int x = i % 19 - 9, y = i / 19 - 9;
                int tx = x, ty = y;
                switch (t) {                       // symmetry.rs:39-62, same order as symmetry::ALL (:119-128)
                    case 0: tx = x; ty = y; break;         // Identity
                    case 1: tx = -x; ty = y; break;        // FlipLR
                    case 2: tx = x; ty = -y; break;        // FlipUD
                    case 3: tx = y; ty = x; break;         // Transpose (main diagonal)
                    case 4: tx = -y; ty = -x; break;       // TransposeAnti
                    case 5: tx = y; ty = -x; break;        // Rot90
                    case 6: tx = -x; ty = -y; break;       // Rot180
                    case 7: tx = -y; ty = x; break;        // Rot270
                }
                map[t][i] = (uint16_t)((ty + 9) * 19 + (tx + 9));
            }
            map[t][DGNN_NUM_POINTS] = DGNN_NUM_POINTS;
        }
    }
};

const SymmetryTables& tables()
{
    static const SymmetryTables t;
    return t;
}

const int kInverse[8] = {0, 1, 2, 3, 4, 7, 6, 5};   // symmetry.rs:79-90

void with_transform(const uint16_t* policy, int transform, uint16_t* out)
{
    const uint16_t* map = tables().map[transform];
    for (int i = 0; i < DGNN_NUM_MOVES; ++i) out[map[i]] = policy[i];   // predictor.rs:34-38
}

// ------------------------------------------------------------------------------------------------ LRU cache

struct CacheKey {
    uint64_t hash;
    int to_move;
    bool operator==(const CacheKey& o) const { return hash == o.hash && to_move == o.to_move; }
};

struct CacheKeyHash {
    size_t operator()(const CacheKey& k) const
    {
        uint64_t h = k.hash ^ (0x9e3779b97f4a7c15ull * (uint64_t)(k.to_move + 1));
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
        return (size_t)h;
    }
};

struct CacheEntry {
    CacheKey key;
    uint32_t newer, older;          // the recency list (lru_cache.rs:45-67), as entry numbers
    uint16_t value;
    uint16_t policy[DGNN_NUM_MOVES];
};

// One shard of the table: the entries in an arena that grows block by block up to `capacity` (an entry keeps its number for life;
// the evicted tail's number is taken over by the newcomer), the recency list threaded through them, and an open-addressing index
// (linear probing, backward-shift deletion) of 8-byte buckets that carry 32 bits of the hash -- a probe touches an entry only when
// those match.  (Round 4: the std::list + std::unordered_map form, one malloc per insert and a 64-bit division per lookup, was a
// sixth of the host's time per rollout with 256 games in flight.)
struct CacheShard {
    static constexpr uint32_t kNone = 0xffffffffu;
    static constexpr int kBlockBits = 9;                                       // 512 entries = 380 KB per block
    struct Bucket { uint32_t id, hash; };

    std::mutex mutex;
    MutexLife mutex_life{&mutex};
    size_t capacity = 0, size = 0;
    std::vector<Bucket> buckets;
    uint32_t mask = 0;
    std::vector<std::unique_ptr<CacheEntry[]>> blocks;
    uint32_t newest = kNone, oldest = kNone;

    // (the shard is chosen by hash % shards: the index takes other bits of the same hash)
    static uint32_t index_hash(const CacheKey& k) { return (uint32_t)((CacheKeyHash()(k) * 0x9e3779b97f4a7c15ull) >> 32); }
    CacheEntry& entry(uint32_t id) { return blocks[id >> kBlockBits][id & ((1u << kBlockBits) - 1)]; }

    bool init(size_t entries)
    {
        capacity = entries;
        size_t n = 16;
        while (n < 2 * entries) n <<= 1;
        if (n > (1ull << 31)) return false;
        buckets.assign(n, Bucket{kNone, 0});
        mask = (uint32_t)(n - 1);
        return true;
    }
    void clear()
    {
        std::fill(buckets.begin(), buckets.end(), Bucket{kNone, 0});
        size = 0;
        newest = oldest = kNone;
    }
    // the bucket that holds `key`, or the empty bucket its probe sequence ends in
    uint32_t probe(const CacheKey& key, uint32_t h)
    {
        uint32_t pos = h & mask;
        while (buckets[pos].id != kNone && !(buckets[pos].hash == h && entry(buckets[pos].id).key == key)) pos = (pos + 1) & mask;
        return pos;
    }
    void erase_bucket(uint32_t hole)
    {
        for (uint32_t next = (hole + 1) & mask; buckets[next].id != kNone; next = (next + 1) & mask) {
            const uint32_t ideal = buckets[next].hash & mask;
            if (((next - ideal) & mask) >= ((next - hole) & mask)) { buckets[hole] = buckets[next]; hole = next; }   // it may move up
        }
        buckets[hole].id = kNone;
    }
    void detach(uint32_t id)
    {
        CacheEntry& e = entry(id);
        if (e.newer != kNone) entry(e.newer).older = e.older; else newest = e.older;
        if (e.older != kNone) entry(e.older).newer = e.newer; else oldest = e.newer;
    }
    void attach_newest(uint32_t id)
    {
        CacheEntry& e = entry(id);
        e.newer = kNone;
        e.older = newest;
        if (newest != kNone) entry(newest).newer = id; else oldest = id;
        newest = id;
    }
};

}  // namespace

struct Cache {
    std::vector<std::unique_ptr<CacheShard>> shards;
    std::atomic<uint64_t> hits{0}, misses{0};
    // bumped by dgnn_cache_clear BEFORE the shards are emptied: an answer that was requested under an older generation (a miss seen
    // before a weight refresh) is not stored afterwards (dgnn_cache_insert_at)
    std::atomic<uint64_t> generation{0};

    CacheShard& shard(const CacheKey& k) { return *shards[CacheKeyHash()(k) % shards.size()]; }
};

// ------------------------------------------------------------------------------------------------ batcher

struct Batch {
    struct Batcher* owner = nullptr;
    std::vector<uint16_t> features;
    std::vector<uint64_t> tags;
    bool alive = true;              // still counted in owner->num_batches
};

struct Batcher {
    dgnn_network* net = nullptr;
    int device = 0;
    int max_batch_size = 0;
    int max_batches = 0;
    // What one queued position is: 11552 fp16 feature values (the reference's form) or a compact board record of
    // DGNN_BOARD_RECORD_BYTES (dgnn_forward_boards: the planes are built on the device).  Stored in units of uint16_t either way.
    bool records = false;
    size_t item = DGNN_FEATURE_SIZE;

    std::mutex mutex;               // batch.rs:60-73: the list lock
    MutexLife mutex_life{&mutex};
    std::vector<uint16_t> features;
    std::vector<uint64_t> tags;
    std::vector<std::vector<uint16_t>> spare_features;   // buffers of retired batches, so that the queue does not grow from zero
    std::vector<std::vector<uint64_t>> spare_tags;       // (and fault its pages in) under the lock after every hand-over
    std::atomic<int> num_batches{0};
    std::atomic<int> queued{0};     // == tags.size(), readable without the lock (the pool workers' wake-up predicate)
    std::atomic<long long> first_queued_us{0};   // when the oldest board now waiting was queued (steady clock; LeafPool::min_fill)
    std::atomic<uint64_t> boards_evaluated{0}, batches_evaluated{0};
    std::atomic<int> refs{1};       // the handle + every batch split off it: a batch may outlive dgnn_batcher_destroy

    // ---- how many batches may be alive: the reference's rule is a constant, 2 per device (predictors/nn.rs:64-67).  A third keeps a
    // device supplied when batches are partial (129 .. 200 boards: one workgroup per board leaves CUs idle) -- and
    // it means more leaves in flight, i.e. more positions evaluated twice before the first answer reaches the transposition table.
    // Which effect wins depends on the phase of the games (openings share positions, middle games do not), so the rule can be made
    // ADAPTIVE (dgnn_pool_set_adaptive_batches): the batcher counts the positions it is handed for the first time (a hash of the
    // stones and the colour to move of every board record against a table of recent ones), measures UNIQUE positions per second under
    // `lo` and under `hi` batches alive per lane in alternating windows, stays with the better one for a while, and probes again.
    int lanes = 1;
    std::atomic<int> adaptive_lo{0}, adaptive_hi{0};          // per lane; 0 = the constant rule (read by the workers without the lock)
    std::atomic<int> limit{0};                               // batches alive allowed right now (== max_batches without the adaptive rule)
    static constexpr int kRecent = 1 << 15;
    std::vector<uint64_t> recent;                            // direct-mapped: the last position hash that fell on each slot
    struct Window { uint64_t pushed = 0, repeated = 0; std::chrono::steady_clock::time_point start; } window;
    static constexpr uint64_t kWindowBoards = 1 << 15;       // a window: about 0.4 s at 80 k boards per second
    double rate_at[2] = {0.0, 0.0};                          // unique positions per second last measured under lo / hi
    int probing = 0;                                         // 0, 1: measuring lo / hi; 2: holding `choice`
    int choice = 1, hold_windows = 0;
    std::atomic<uint64_t> switches{0}, boards_repeated{0}, boards_hashed{0};

    // The simpler rule (dgnn_pool_set_extra_batch_queue): `lo` batches alive per lane, and up to `hi` only while at least
    // `extra_min_queued` boards are WAITING -- an extra batch that comes out full keeps a device supplied (256 games in flight: +5 %
    // games per hour), one that splits the same few leaves over more batches only makes every batch smaller (96 games: -9 %).
    std::atomic<int> extra_min_queued{0}, extra_lo{0}, extra_hi{0};
    int alive_limit() const
    {
        const int threshold = extra_min_queued.load(std::memory_order_acquire);
        if (threshold > 0) return (queued.load(std::memory_order_acquire) >= threshold ? extra_hi.load() : extra_lo.load()) * lanes;
        return adaptive_hi.load(std::memory_order_acquire) > 0 ? limit.load(std::memory_order_acquire) : max_batches;
    }
    void note_record(const uint16_t* record);                // under `mutex`
    void release() { if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this; }
};

// position identity of a board record: the colours of the 361 cells (bits 0-1; the block ids next to them depend on the order
// the stones were played in) and the colour to move -- not the symmetry, not the move history: what the transposition table keys on
static uint64_t record_position_hash(const uint16_t* record)
{
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 361; i += 3) {
        const uint64_t three = (uint64_t)(record[i] & 3u) | (uint64_t)(record[i + 1 < 361 ? i + 1 : 360] & 3u) << 2 | (uint64_t)(record[i + 2 < 361 ? i + 2 : 360] & 3u) << 4;
        h = (h ^ three) * 1099511628211ull;
    }
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(record);
    h = (h ^ bytes[368 * 2 + 3 * 48]) * 1099511628211ull;     // to_move: the first byte behind cell[368] and the three 48-byte masks
    h ^= h >> 29;
    return h;
}
static int record_symmetry(const uint16_t* record) { return reinterpret_cast<const uint8_t*>(record)[368 * 2 + 3 * 48 + 1] & 7; }

void Batcher::note_record(const uint16_t* record)
{
    const uint64_t h = record_position_hash(record);
    uint64_t& slot = recent[h & (kRecent - 1)];
    const bool repeat = slot == h;
    slot = h;
    if (window.pushed == 0) window.start = std::chrono::steady_clock::now();
    ++window.pushed;
    window.repeated += repeat ? 1 : 0;
    if (window.pushed < kWindowBoards) return;
    // ---- a window is full: unique positions per second under the limit it ran with
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - window.start).count();
    const double rate = (double)(window.pushed - window.repeated) / (seconds > 1e-6 ? seconds : 1e-6);
    boards_hashed.fetch_add(window.pushed, std::memory_order_relaxed);
    boards_repeated.fetch_add(window.repeated, std::memory_order_relaxed);
    window = Window();
    int next = choice;
    if (probing < 2) {
        rate_at[probing] = rate;
        if (probing == 0) { probing = 1; next = 1; }                     // lo measured: now hi
        else {                                                           // both measured: keep the better one for a while (3 % hysteresis)
            const int better = rate_at[1] > 1.03 * rate_at[0] ? 1 : rate_at[0] > 1.03 * rate_at[1] ? 0 : choice;
            probing = 2; hold_windows = 12; next = better;
        }
    } else {
        rate_at[choice] = rate;
        if (--hold_windows <= 0) { probing = 0; next = 0; }             // probe again, lo first
    }
    if (next != choice) switches.fetch_add(1, std::memory_order_relaxed);
    choice = next;
    limit.store((choice ? adaptive_hi.load() : adaptive_lo.load()) * lanes, std::memory_order_release);
}

// ------------------------------------------------------------------------------------------------ leaf pool
//
// The reference coalesces leaf evaluations only inside ONE search (pool/worker_thread.rs:88-99); the <= 8-board
// full_forward calls of its game threads go to the GPU one by one (lib.rs:114, self_play.rs:370-405), which makes
// policy-play (BASELINE.json configs[4]) a small-batch latency path.  LeafPool gives every caller the blocking
// Predictor::predict signature (predictor.rs:92) and coalesces ACROSS callers: requests are pushed into the batcher,
// worker threads turn whatever has accumulated into device batches whenever fewer than max_batches are in flight.
struct LeafSlot {
    struct LeafRequest* request;
    int index;
    // coalescing (dgnn_pool_set_coalescing): the position this slot asks for and the symmetry it wants the answer in; a slot that was
    // pushed into the batcher LEADS every later slot for the same position until its answer is delivered
    uint64_t position = 0;
    uint8_t symmetry = 0;
    bool leads = false;
};

struct LeafRequest {
    uint16_t* value_out;
    uint16_t* policy_out;
    int remaining;
    int status;
    Fiber* fiber = nullptr;     // the caller is a fiber (fiber.h): made ready by whoever completes the request, instead of `done`
    std::mutex mutex;
    MutexLife mutex_life{&mutex};
    std::condition_variable done;
};

static int batcher_get_batch(struct Batcher* b, int min_batch_size, int cap, dgnn_batch** out);

// One slot's answer (or the status that stands in for it) reaches its request; the last slot of a request wakes the caller.
static void complete_slot(LeafSlot* slot, int status, uint16_t v, const uint16_t* p)
{
    LeafRequest* r = slot->request;
    if (status == DGNN_OK) {
        r->value_out[slot->index] = v;
        std::memcpy(r->policy_out + (size_t)slot->index * DGNN_NUM_MOVES, p, DGNN_NUM_MOVES * sizeof(uint16_t));
    }
    Fiber* resume = nullptr;
    {
        std::lock_guard<std::mutex> lock(r->mutex);
        if (status != DGNN_OK) r->status = status;
        if (--r->remaining == 0) { if (r->fiber) resume = r->fiber; else r->done.notify_one(); }
    }
    if (resume) fiber_make_ready(resume);      // (the request lives on the fiber's stack: not touched after this)
}

struct LeafPool {
    Batcher* batcher = nullptr;     // holds a reference (Batcher::refs) for the pool's lifetime
    bool owns_batcher = false;      // created by dgnn_pool_create_devices / _callbacks: nobody else has a handle
    // Lanes: where a worker evaluates its batches.  Either devices of the network (NnPredictor::predict round-robins
    // Device::all(), src/libdg_mcts/predictors/nn.rs:84-108: here worker w is bound to lane w % lanes, so a batch goes to
    // whichever device has a worker free) or, for tests and dry runs without a GPU, a predictor callback per lane.
    std::vector<int> devices;
    dgnn_predict_fn callback = nullptr;
    dgnn_predict_boards_fn board_callback = nullptr;   // a callback that takes the records as they are
    void* callback_ctx = nullptr;
    int num_lanes = 1;
    std::unique_ptr<std::atomic<uint64_t>[]> lane_boards, lane_batches;

    std::vector<std::thread> workers;
    std::mutex mutex;
    MutexLife mutex_life{&mutex};
    std::condition_variable work;
    bool stopping = false;
    bool paused = false;            // dgnn_pool_pause: workers take no new batch (callers keep queueing and wait)
    int evaluating = 0;             // workers between get_batch and the end of evaluate (under `mutex`)
    std::condition_variable idle;   // signalled when `evaluating` drops to zero
    std::atomic<uint64_t> generation{0};   // completed pauses: what a game compares per move to drop its PRIVATE table after a refresh
    // Coalescing of positions IN FLIGHT (records pools; off unless dgnn_pool_set_coalescing): the transposition table only knows
    // positions whose evaluation has come back; with hundreds of games searching the same opening most "misses" are positions
    // somebody else has on the device right now (measured: 60 % of the boards a pool is handed in the first six moves of 256 games
    // repeat a position it was handed within the last half second).  A request for a position that is in flight FOLLOWS the slot
    // that leads it and receives that answer, re-oriented from the leader's symmetry to its own -- exactly what a table hit returns
    // (predictors/nn.rs:72-81) -- instead of being evaluated again.
    std::atomic<bool> coalesce{false};
    std::mutex inflight_mutex;
    MutexLife inflight_mutex_life{&inflight_mutex};
    struct InFlight { const LeafRequest* owner; std::vector<LeafSlot*> followers; };
    std::unordered_map<uint64_t, InFlight> inflight;                    // position -> who leads it, and the slots that follow
    std::atomic<uint64_t> coalesced{0};
    std::atomic<uint64_t> calls{0};
    // Minimum fill (dgnn_pool_set_min_fill): a workgroup per board makes a batch of 130 boards take as long as one of 256, so while
    // ANOTHER batch keeps the device busy a free slot is not filled with whatever happens to be queued the moment it frees (a handful
    // of boards: the games the finished batch woke have not produced their next leaves yet) but waits until `min_fill` boards are
    // there -- or `max_wait_us` have passed since the oldest of them was queued, or nothing else is alive.
    std::atomic<int> min_fill{0};
    std::atomic<int> max_wait_us{0};
    // Back to back (dgnn_pool_set_back_to_back; pools of one lane): a partial batch is not sent to run BESIDE the batch that occupies
    // the device -- a workgroup per board makes 120 boards take what 128 take and 190 what 256 take -- but `lead_us` before that batch
    // is expected to finish (how long a batch of its size took lately), so that its upload and its first kernels fall into the
    // other's tail and the device is never idle, with everything that has queued up until then.  A FULL batch still goes at once.
    // With snapping a batch of snap_lo < n < snap_hi boards is cut to its snap_lo OLDEST (the rest follows a moment later): alone on
    // the device 129 .. 200 boards on whole-board workgroups are the worst sizes there are.
    static constexpr long long kMaxHoldUs = 5000;
    std::atomic<int> lead_us{0}, snap_lo{0}, snap_hi{0};
    std::atomic<long long> expected_done_us{0};      // of the batch most recently sent to the device
    std::atomic<int> took_us[17] = {};                  // by size / 16: moving average of evaluate()'s wall time, 0 = not seen yet

    bool ready() const
    {
        const int queued = batcher->queued.load(std::memory_order_acquire);
        const int alive = batcher->num_batches.load(std::memory_order_acquire);
        if (queued <= 0 || alive >= batcher->alive_limit()) return false;
        const int lead = lead_us.load(std::memory_order_relaxed);
        if (lead > 0 && alive > 0 && queued < batcher->max_batch_size) {
            // (never longer than kMaxHoldUs behind the oldest board waiting: an estimate spoilt by a first batch that created its
            // workspace must not stall the games)
            const long long now = now_us();
            return now >= expected_done_us.load(std::memory_order_acquire) - lead ||
                   now - batcher->first_queued_us.load(std::memory_order_acquire) >= kMaxHoldUs;
        }
        const int fill = min_fill.load(std::memory_order_relaxed);
        if (fill <= 0 || alive == 0 || queued >= fill) return true;
        const long long waited = now_us() - batcher->first_queued_us.load(std::memory_order_acquire);
        return waited >= max_wait_us.load(std::memory_order_relaxed);
    }
    static long long now_us()
    {
        return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    // state the predicate reads changes outside `mutex` (under the batcher's): passing through the mutex before the notify
    // closes the window between a worker's predicate check and its going to sleep
    void wake()
    {
        { std::lock_guard<std::mutex> lock(mutex); }
        work.notify_one();
    }

    int evaluate(int lane, dgnn_batch* batch, int n, uint16_t* value, uint16_t* policy);

    void run(int lane)
    {
        std::vector<uint16_t> value((size_t)batcher->max_batch_size), policy((size_t)batcher->max_batch_size * DGNN_NUM_MOVES);
        std::vector<uint64_t> tags((size_t)batcher->max_batch_size);
        for (;;) {
            {
                // sleep until there is something queued AND a batch slot free (signalled by dgnn_pool_predict and by every
                // batch that retires); the timeout only covers boards pushed into the batcher behind the pool's back
                std::unique_lock<std::mutex> lock(mutex);
                // (with a minimum fill the predicate also turns true by the clock: short naps while boards are waiting for company)
                const bool holding = (min_fill.load(std::memory_order_relaxed) > 0 || lead_us.load(std::memory_order_relaxed) > 0) &&
                                     batcher->queued.load(std::memory_order_acquire) > 0;
                wait_for_us(work, lock, holding ? 100 : 20000, [&] { return stopping || (!paused && ready()); });
                if (stopping) return;
                if (paused || !ready()) continue;
                ++evaluating;       // from here to done(): a pause waits for us
            }
            auto done = [&] {
                std::lock_guard<std::mutex> lock(mutex);
                if (--evaluating == 0) idle.notify_all();
            };
            dgnn_batch* batch = nullptr;
            // take whatever is queued as soon as a batch slot is free: with max_batches batches in flight the queue
            // fills up behind them, which is what makes the batches large under load and keeps latency low when idle
            int cap = 0;
            if (const int lo = snap_lo.load(std::memory_order_relaxed)) {
                const int waiting = batcher->queued.load(std::memory_order_acquire);
                if (waiting > lo && waiting < snap_hi.load(std::memory_order_relaxed)) cap = lo;
            }
            (void)batcher_get_batch(batcher, 1, cap, &batch);
            if (!batch) { done(); continue; }
            const int n = dgnn_batch_size(batch);
            const long long sent = now_us();
            if (lead_us.load(std::memory_order_relaxed) > 0)
                expected_done_us.store(sent + took_us[std::min(n >> 4, 16)].load(std::memory_order_relaxed), std::memory_order_release);
            if (ready()) work.notify_one();         // more than one batch was queued: the next idle worker takes the rest
            (void)dgnn_batch_tags(batch, tags.data());
            const int status = evaluate(lane, batch, n, value.data(), policy.data());
            {
                const int took = (int)std::min<long long>(now_us() - sent, 20000), before = took_us[std::min(n >> 4, 16)].load(std::memory_order_relaxed);
                took_us[std::min(n >> 4, 16)].store(before ? (3 * before + took) / 4 : took, std::memory_order_relaxed);
            }
            dgnn_batch_destroy(batch);
            wake();                                 // a batch slot is free again
            auto deliver = [&](LeafSlot* slot, uint16_t v, const uint16_t* p) { complete_slot(slot, status, v, p); };
            // followers of the batch's leaders: ONE pass under the table's lock for the whole batch (a lock per slot was 256 lock / unlock
            // pairs per batch against the game threads that register their own positions), before the first answer is handed out --
            // after it a leader's request may return and take its slots with it
            std::vector<std::pair<int, std::vector<LeafSlot*>>> followed;      // (index in the batch, the slots that followed it)
            std::vector<uint8_t> leader_symmetry;
            // (keyed on the slots, not on the switch: dgnn_pool_set_coalescing(0) may fall between a request's look at the switch and this
            // pass -- a leader registered in that window must still be taken out of the table, or every later request for its position
            // would follow an entry nobody answers.  ADVICE r05)
            bool any_leads = false;
            for (int i = 0; i < n && !any_leads; ++i) any_leads = reinterpret_cast<LeafSlot*>((uintptr_t)tags[i])->leads;
            if (any_leads) {
                leader_symmetry.assign((size_t)n, 0);
                std::lock_guard<std::mutex> lock(inflight_mutex);
                for (int i = 0; i < n; ++i) {
                    LeafSlot* slot = reinterpret_cast<LeafSlot*>((uintptr_t)tags[i]);
                    if (!slot->leads) continue;
                    leader_symmetry[(size_t)i] = slot->symmetry;
                    auto it = inflight.find(slot->position);
                    if (it == inflight.end()) continue;
                    if (!it->second.followers.empty()) { followed.emplace_back(i, std::vector<LeafSlot*>()); followed.back().second.swap(it->second.followers); }
                    inflight.erase(it);
                }
            }
            size_t next_followed = 0;
            for (int i = 0; i < n; ++i) {
                LeafSlot* slot = reinterpret_cast<LeafSlot*>((uintptr_t)tags[i]);
                const uint16_t* p = policy.data() + (size_t)i * DGNN_NUM_MOVES;
                deliver(slot, value[i], p);
                if (next_followed < followed.size() && followed[next_followed].first == i) {
                    uint16_t canonical[DGNN_NUM_MOVES], oriented[DGNN_NUM_MOVES];
                    if (status == DGNN_OK) with_transform(p, kInverse[leader_symmetry[(size_t)i]], canonical);        // nn.rs:78-81
                    for (LeafSlot* f : followed[next_followed].second) {
                        if (status == DGNN_OK) with_transform(canonical, f->symmetry, oriented);                       // nn.rs:72-74
                        deliver(f, value[i], oriented);
                    }
                    ++next_followed;
                }
            }
            done();     // after the answers are with their requests: a pause returns with no old-weights answer still on its way
        }
    }
};

// get_batch with a cap below the batcher's own: the `cap` OLDEST entries go (cap <= 0: the reference's rule)
static int batcher_get_batch(Batcher* b, int min_batch_size, int cap, dgnn_batch** out)
{
    *out = nullptr;

    int current = b->num_batches.load(std::memory_order_acquire);               // batch.rs:93-99
    if (current >= b->alive_limit()) return DGNN_OK;

    std::lock_guard<std::mutex> lock(b->mutex);
    const size_t size = b->tags.size();
    if (size == 0 || size < (size_t)(min_batch_size > 0 ? min_batch_size : 0)) return DGNN_OK;   // an empty batch is never useful
    if (!b->num_batches.compare_exchange_strong(current, current + 1, std::memory_order_acquire, std::memory_order_relaxed))
        return DGNN_OK;                                                         // batch.rs:105: lost the race, try again later

    std::unique_ptr<Batch> batch(new (std::nothrow) Batch());
    if (!batch) { b->num_batches.fetch_sub(1); return DGNN_OUT_OF_MEMORY; }
    const size_t split = size >= (size_t)b->max_batch_size ? size - b->max_batch_size : 0;     // batch.rs:106
    batch->owner = b;
    b->refs.fetch_add(1, std::memory_order_relaxed);
    if (cap > 0 && (size_t)cap < size && cap < b->max_batch_size) {             // a cut batch: the entries that have waited longest
        batch->features.assign(b->features.begin(), b->features.begin() + (size_t)cap * b->item);
        batch->tags.assign(b->tags.begin(), b->tags.begin() + cap);
        b->features.erase(b->features.begin(), b->features.begin() + (size_t)cap * b->item);
        b->tags.erase(b->tags.begin(), b->tags.begin() + cap);
        b->first_queued_us.store(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(),
                                 std::memory_order_release);
    } else if (split == 0) {                                                           // everything goes: hand the buffers over, no copy under the lock
        batch->features.swap(b->features);
        batch->tags.swap(b->tags);
        if (!b->spare_features.empty()) { b->features.swap(b->spare_features.back()); b->spare_features.pop_back(); }
        if (!b->spare_tags.empty()) { b->tags.swap(b->spare_tags.back()); b->spare_tags.pop_back(); }
    } else {
        batch->features.assign(b->features.begin() + split * b->item, b->features.end());   // split_off: the NEWEST entries
        batch->tags.assign(b->tags.begin() + split, b->tags.end());
        b->features.resize(split * b->item);
        b->tags.resize(split);
    }
    b->queued.store((int)b->tags.size(), std::memory_order_release);
    *out = reinterpret_cast<dgnn_batch*>(batch.release());
    return DGNN_OK;
}

}  // namespace dgnn

const uint16_t* dgnn::symmetry_map(int transform) { return tables().map[transform & 7]; }
bool dgnn::pool_coalesces(const dgnn_pool* handle) { return handle && reinterpret_cast<const LeafPool*>(handle)->coalesce.load(std::memory_order_relaxed); }

using namespace dgnn;

extern "C" {

// ---- symmetry

int dgnn_symmetry_inverse(int transform) { return (transform >= 0 && transform < 8) ? kInverse[transform] : -1; }

int dgnn_symmetry_apply(int transform, int packed_index)
{
    if (transform < 0 || transform >= 8 || packed_index < 0 || packed_index >= DGNN_NUM_MOVES) return -1;
    return tables().map[transform][packed_index];
}

int dgnn_policy_with_transform(const uint16_t* policy, int transform, uint16_t* out)
{
    if (!policy || !out || policy == out || transform < 0 || transform >= 8) return DGNN_BAD_ARGUMENT;
    with_transform(policy, transform, out);
    return DGNN_OK;
}

// ---- cache

int dgnn_cache_create(size_t capacity, int num_shards, dgnn_cache** out)
{
    if (!out || capacity == 0 || num_shards <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<Cache> c(new (std::nothrow) Cache());
    if (!c) return DGNN_OUT_OF_MEMORY;
    for (int i = 0; i < num_shards; ++i) {
        std::unique_ptr<CacheShard> s(new (std::nothrow) CacheShard());
        if (!s) return DGNN_OUT_OF_MEMORY;
        try {
            if (!s->init((capacity + num_shards - 1) / num_shards)) return DGNN_BAD_ARGUMENT;
        } catch (const std::bad_alloc&) { return DGNN_OUT_OF_MEMORY; }
        c->shards.push_back(std::move(s));
    }
    *out = reinterpret_cast<dgnn_cache*>(c.release());
    return DGNN_OK;
}

void dgnn_cache_destroy(dgnn_cache* cache) { delete reinterpret_cast<Cache*>(cache); }

int dgnn_cache_fetch(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t* value_out, uint16_t* policy_out)
{
    if (!cache || !value_out || !policy_out || symmetry < 0 || symmetry >= 8) return DGNN_BAD_ARGUMENT;
    Cache* c = reinterpret_cast<Cache*>(cache);
    const CacheKey key{board_hash, to_move};
    CacheShard& s = c->shard(key);
    std::lock_guard<std::mutex> lock(s.mutex);
    const uint32_t id = s.buckets[s.probe(key, CacheShard::index_hash(key))].id;
    if (id == CacheShard::kNone) { c->misses.fetch_add(1, std::memory_order_relaxed); return 0; }
    if (s.newest != id) { s.detach(id); s.attach_newest(id); }                  // lru_cache.rs:106-109: detach + attach
    const CacheEntry& e = s.entry(id);
    *value_out = e.value;
    with_transform(e.policy, symmetry, policy_out);                             // nn.rs:72-74
    c->hits.fetch_add(1, std::memory_order_relaxed);
    return 1;
}

static int cache_insert(Cache* c, bool checked, uint64_t generation, uint64_t board_hash, int to_move, int symmetry, uint16_t value,
                        const uint16_t* policy)
{
    const CacheKey key{board_hash, to_move};
    CacheShard& s = c->shard(key);
    std::lock_guard<std::mutex> lock(s.mutex);
    // under the shard's lock: a clear that bumped the generation before this load drops the answer here; one that bumps it later
    // takes this lock after us and drops the entry with the rest of the shard
    if (checked && c->generation.load(std::memory_order_seq_cst) != generation) return DGNN_OK;
    const uint32_t h = CacheShard::index_hash(key);
    uint32_t pos = s.probe(key, h);
    if (s.buckets[pos].id != CacheShard::kNone) return DGNN_OK;                 // lru_cache.rs:121: present keys are left alone
    uint32_t id;
    if (s.size >= s.capacity) {                                                 // lru_cache.rs:136-147: the tail goes, the newcomer takes its place
        id = s.oldest;
        const CacheKey gone = s.entry(id).key;
        s.detach(id);
        s.erase_bucket(s.probe(gone, CacheShard::index_hash(gone)));
        pos = s.probe(key, h);                                                  // (the deletion may have moved buckets)
    } else {
        id = (uint32_t)s.size++;
        if ((id >> CacheShard::kBlockBits) >= s.blocks.size()) {
            std::unique_ptr<CacheEntry[]> block(new (std::nothrow) CacheEntry[1u << CacheShard::kBlockBits]);
            if (block) { try { s.blocks.push_back(std::move(block)); } catch (const std::bad_alloc&) { block.reset(); } }
            if ((id >> CacheShard::kBlockBits) >= s.blocks.size()) { --s.size; return DGNN_OUT_OF_MEMORY; }
        }
    }
    CacheEntry& e = s.entry(id);
    e.key = key;
    e.value = value;
    with_transform(policy, kInverse[symmetry], e.policy);                       // nn.rs:78-81: symmetry.inverse()
    s.buckets[pos] = CacheShard::Bucket{id, h};
    s.attach_newest(id);
    return DGNN_OK;
}

int dgnn_cache_insert(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t value, const uint16_t* policy)
{
    if (!cache || !policy || symmetry < 0 || symmetry >= 8) return DGNN_BAD_ARGUMENT;
    return cache_insert(reinterpret_cast<Cache*>(cache), false, 0, board_hash, to_move, symmetry, value, policy);
}

uint64_t dgnn_cache_generation(const dgnn_cache* cache)
{
    return cache ? reinterpret_cast<const Cache*>(cache)->generation.load(std::memory_order_seq_cst) : 0;
}

// The insert of an answer whose MISS was seen at `generation` (dgnn_cache_generation read before the fetch): dropped when the table
// has been cleared since -- the answer may come from the weights the clear was for (a request in flight across
// pause -> set weights -> clear -> resume used to put the old network's output back into the emptied table).
int dgnn_cache_insert_at(dgnn_cache* cache, uint64_t generation, uint64_t board_hash, int to_move, int symmetry, uint16_t value,
                         const uint16_t* policy)
{
    if (!cache || !policy || symmetry < 0 || symmetry >= 8) return DGNN_BAD_ARGUMENT;
    return cache_insert(reinterpret_cast<Cache*>(cache), true, generation, board_hash, to_move, symmetry, value, policy);
}

// Drops every entry (the counters stay): what a weight refresh needs -- the table holds the OLD network's outputs.
int dgnn_cache_clear(dgnn_cache* cache)
{
    if (!cache) return DGNN_BAD_ARGUMENT;
    Cache* c = reinterpret_cast<Cache*>(cache);
    c->generation.fetch_add(1, std::memory_order_seq_cst);          // first: answers requested before this point are no longer stored
    for (auto& s : c->shards) {
        std::lock_guard<std::mutex> lock(s->mutex);
        s->clear();
    }
    return DGNN_OK;
}

int dgnn_cache_stats(dgnn_cache* cache, uint64_t* entries, uint64_t* hits, uint64_t* misses)
{
    if (!cache) return DGNN_BAD_ARGUMENT;
    Cache* c = reinterpret_cast<Cache*>(cache);
    if (entries) {
        uint64_t n = 0;
        for (auto& s : c->shards) { std::lock_guard<std::mutex> lock(s->mutex); n += s->size; }
        *entries = n;
    }
    if (hits) *hits = c->hits.load();
    if (misses) *misses = c->misses.load();
    return DGNN_OK;
}

// ---- batcher

static int batcher_create(dgnn_network* net, int device, int max_batch_size, int max_batches, bool records, dgnn_batcher** out)
{
    if (!out || max_batch_size <= 0 || max_batches <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<Batcher> b(new (std::nothrow) Batcher());
    if (!b) return DGNN_OUT_OF_MEMORY;
    b->net = net;            // may be NULL: a batcher without a predictor (queueing logic only; forward is then refused)
    b->device = device;
    b->max_batch_size = max_batch_size;
    b->max_batches = max_batches;
    b->records = records;
    b->item = records ? DGNN_BOARD_RECORD_BYTES / sizeof(uint16_t) : DGNN_FEATURE_SIZE;
    b->features.reserve((size_t)2 * max_batch_size * b->item);                  // batch.rs:51-56
    b->tags.reserve((size_t)2 * max_batch_size);
    *out = reinterpret_cast<dgnn_batcher*>(b.release());
    return DGNN_OK;
}

int dgnn_batcher_create(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out)
{
    return batcher_create(net, device, max_batch_size, max_batches, false, out);
}

int dgnn_batcher_create_boards(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out)
{
    return batcher_create(net, device, max_batch_size, max_batches, true, out);
}

void dgnn_batcher_destroy(dgnn_batcher* batcher) { if (batcher) reinterpret_cast<Batcher*>(batcher)->release(); }

static int batcher_push(Batcher* b, const uint16_t* item, uint64_t tag)
{
    // test hook (dgnn_test_fail_push_after, include/dgnn_test.h): the n-th push from now fails like an allocation that failed
    if (g_fail_push_countdown.load(std::memory_order_relaxed) >= 0 && g_fail_push_countdown.fetch_sub(1, std::memory_order_relaxed) == 0)
        return DGNN_OUT_OF_MEMORY;
    std::lock_guard<std::mutex> lock(b->mutex);                                 // batch.rs:80-84
    if (b->adaptive_hi.load(std::memory_order_relaxed) > 0 && b->records) b->note_record(item);
    try {
        b->features.insert(b->features.end(), item, item + b->item);
        b->tags.push_back(tag);
    } catch (const std::bad_alloc&) {
        b->features.resize(b->tags.size() * b->item);                           // (the board without its tag goes again)
        return DGNN_OUT_OF_MEMORY;
    }
    if (b->tags.size() == 1)
        b->first_queued_us.store(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(),
                                 std::memory_order_release);
    b->queued.store((int)b->tags.size(), std::memory_order_release);
    return DGNN_OK;
}

int dgnn_batcher_push(dgnn_batcher* batcher, const uint16_t* features, uint64_t tag)
{
    if (!batcher || !features || reinterpret_cast<Batcher*>(batcher)->records) return DGNN_BAD_ARGUMENT;
    return batcher_push(reinterpret_cast<Batcher*>(batcher), features, tag);
}

int dgnn_batcher_push_board(dgnn_batcher* batcher, const uint8_t* record, uint64_t tag)
{
    if (!batcher || !record || !reinterpret_cast<Batcher*>(batcher)->records) return DGNN_BAD_ARGUMENT;
    return batcher_push(reinterpret_cast<Batcher*>(batcher), reinterpret_cast<const uint16_t*>(record), tag);
}

int dgnn_batcher_pending(dgnn_batcher* batcher, int* queued, int* batches_alive)
{
    if (!batcher) return DGNN_BAD_ARGUMENT;
    Batcher* b = reinterpret_cast<Batcher*>(batcher);
    if (queued) { std::lock_guard<std::mutex> lock(b->mutex); *queued = (int)b->tags.size(); }
    if (batches_alive) *batches_alive = b->num_batches.load(std::memory_order_acquire);
    return DGNN_OK;
}

int dgnn_batcher_get_batch(dgnn_batcher* batcher, int min_batch_size, dgnn_batch** out)
{
    if (!batcher || !out) return DGNN_BAD_ARGUMENT;
    return batcher_get_batch(reinterpret_cast<Batcher*>(batcher), min_batch_size, 0, out);
}

int dgnn_batcher_push_and_get_batch(dgnn_batcher* batcher, const uint16_t* features, uint64_t tag, dgnn_batch** out)
{
    const int status = dgnn_batcher_push(batcher, features, tag);               // batch.rs:86-89
    if (status != DGNN_OK) return status;
    return dgnn_batcher_get_batch(batcher, reinterpret_cast<Batcher*>(batcher)->max_batch_size, out);
}

int dgnn_batch_size(const dgnn_batch* batch) { return batch ? (int)reinterpret_cast<const Batch*>(batch)->tags.size() : 0; }

int dgnn_batch_tags(const dgnn_batch* batch, uint64_t* tags_out)
{
    if (!batch || !tags_out) return DGNN_BAD_ARGUMENT;
    const Batch* bt = reinterpret_cast<const Batch*>(batch);
    std::memcpy(tags_out, bt->tags.data(), bt->tags.size() * sizeof(uint64_t));
    return DGNN_OK;
}

const uint16_t* dgnn_batch_features(const dgnn_batch* batch)
{
    return (batch && !reinterpret_cast<const Batch*>(batch)->owner->records) ? reinterpret_cast<const Batch*>(batch)->features.data() : nullptr;
}
const uint8_t* dgnn_batch_records(const dgnn_batch* batch)
{
    return (batch && reinterpret_cast<const Batch*>(batch)->owner->records) ? reinterpret_cast<const uint8_t*>(reinterpret_cast<const Batch*>(batch)->features.data()) : nullptr;
}

static void retire(Batch* bt)
{
    if (bt->alive) {
        bt->alive = false;
        bt->owner->num_batches.fetch_sub(1, std::memory_order_acq_rel);         // batch.rs:37
    }
}

int dgnn_batch_forward(dgnn_batch* batch, uint16_t* value_out, uint16_t* policy_out)
{
    if (!batch) return DGNN_BAD_ARGUMENT;
    return dgnn_batch_forward_on(batch, reinterpret_cast<Batch*>(batch)->owner->device, value_out, policy_out);
}

int dgnn_batch_forward_on(dgnn_batch* batch, int device, uint16_t* value_out, uint16_t* policy_out)
{
    if (!batch || !value_out || !policy_out) return DGNN_BAD_ARGUMENT;
    Batch* bt = reinterpret_cast<Batch*>(batch);
    Batcher* b = bt->owner;
    if (!b->net) return DGNN_BAD_ARGUMENT;
    const int n = (int)bt->tags.size();

    dgnn_workspace* ws = nullptr;
    int status = dgnn_network_get_workspace(b->net, device, b->max_batch_size, &ws);
    if (status == DGNN_OK) {
        status = b->records ? dgnn_forward_boards(ws, reinterpret_cast<const uint8_t*>(bt->features.data()), n, value_out, policy_out)
                            : dgnn_forward_n(ws, bt->features.data(), n, value_out, policy_out);   // batch.rs:36 -> nn.rs:84-108
        dgnn_workspace_release(ws);
    }
    retire(bt);
    if (status == DGNN_OK) {
        b->boards_evaluated.fetch_add((uint64_t)n, std::memory_order_relaxed);
        b->batches_evaluated.fetch_add(1, std::memory_order_relaxed);
    }
    return status;
}

void dgnn_batch_destroy(dgnn_batch* batch)
{
    if (!batch) return;
    Batch* bt = reinterpret_cast<Batch*>(batch);
    retire(bt);
    {
        Batcher* b = bt->owner;
        std::lock_guard<std::mutex> lock(b->mutex);
        if (b->spare_features.size() < (size_t)b->max_batches + 1 && bt->features.capacity() > 0) {
            bt->features.clear();
            bt->tags.clear();
            b->spare_features.emplace_back(std::move(bt->features));
            b->spare_tags.emplace_back(std::move(bt->tags));
        }
    }
    bt->owner->release();
    delete bt;
}

// ---- leaf pool

int LeafPool::evaluate(int lane, dgnn_batch* batch, int n, uint16_t* value, uint16_t* policy)
{
    int status;
    if (board_callback) {
        status = board_callback(callback_ctx, dgnn_batch_records(batch), n, value, policy);
        if (status == DGNN_OK) {
            batcher->boards_evaluated.fetch_add((uint64_t)n, std::memory_order_relaxed);
            batcher->batches_evaluated.fetch_add(1, std::memory_order_relaxed);
        }
    } else if (callback) {
        if (batcher->records) {
            // lanes that are predictor CALLBACKS (dry runs, the oracle as the predictor in the tests) take fp16 planes: the host
            // statement of the device kernel expands the records (go.cpp: record_to_packed).  A pool over devices never gets here.
            std::vector<uint32_t> packed((size_t)n * DGNN_PACKED_WORDS);
            std::vector<uint16_t> planes((size_t)n * DGNN_FEATURE_SIZE);
            status = dgnn_board_records_to_packed(dgnn_batch_records(batch), n, packed.data());
            if (status == DGNN_OK) status = dgnn_features_unpack(packed.data(), n, planes.data());
            if (status == DGNN_OK) status = callback(callback_ctx, planes.data(), n, value, policy);
        } else
        status = callback(callback_ctx, dgnn_batch_features(batch), n, value, policy);
        if (status == DGNN_OK) {
            batcher->boards_evaluated.fetch_add((uint64_t)n, std::memory_order_relaxed);
            batcher->batches_evaluated.fetch_add(1, std::memory_order_relaxed);
        }
    } else {
        status = dgnn_batch_forward_on(batch, devices[lane], value, policy);
    }
    if (status == DGNN_OK) {
        lane_boards[lane].fetch_add((uint64_t)n, std::memory_order_relaxed);
        lane_batches[lane].fetch_add(1, std::memory_order_relaxed);
    }
    return status;
}

static int start_pool(std::unique_ptr<LeafPool>& pool, int workers_per_lane, dgnn_pool** out)
{
    pool->lane_boards.reset(new (std::nothrow) std::atomic<uint64_t>[pool->num_lanes]);
    pool->lane_batches.reset(new (std::nothrow) std::atomic<uint64_t>[pool->num_lanes]);
    if (!pool->lane_boards || !pool->lane_batches) return DGNN_OUT_OF_MEMORY;
    for (int i = 0; i < pool->num_lanes; ++i) { pool->lane_boards[i] = 0; pool->lane_batches[i] = 0; }
    pool->batcher->refs.fetch_add(1, std::memory_order_relaxed);   // the workers poll the batcher until they have joined
    LeafPool* raw = pool.get();
    for (int w = 0; w < workers_per_lane * pool->num_lanes; ++w) {
        const int lane = w % pool->num_lanes;
        pool->workers.emplace_back([raw, lane] { raw->run(lane); });
    }
    *out = reinterpret_cast<dgnn_pool*>(pool.release());
    return DGNN_OK;
}

int dgnn_pool_create(dgnn_batcher* batcher, int num_workers, dgnn_pool** out)
{
    if (!batcher || !out || num_workers <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    Batcher* b = reinterpret_cast<Batcher*>(batcher);
    if (!b->net) return DGNN_BAD_ARGUMENT;
    std::unique_ptr<LeafPool> pool(new (std::nothrow) LeafPool());
    if (!pool) return DGNN_OUT_OF_MEMORY;
    pool->batcher = b;
    pool->devices.assign(1, b->device);
    pool->num_lanes = 1;
    return start_pool(pool, num_workers, out);
}

static int pool_create_devices(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                               int workers_per_device, bool records, dgnn_pool** out);

int dgnn_pool_create_devices(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                             int workers_per_device, dgnn_pool** out)
{
    return pool_create_devices(net, devices, num_devices, max_batch_size, batches_per_device, workers_per_device, false, out);
}

int dgnn_pool_create_devices_boards(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                                    int workers_per_device, dgnn_pool** out)
{
    return pool_create_devices(net, devices, num_devices, max_batch_size, batches_per_device, workers_per_device, true, out);
}

static int pool_create_devices(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                               int workers_per_device, bool records, dgnn_pool** out)
{
    if (!net || !out || max_batch_size <= 0 || batches_per_device <= 0 || workers_per_device <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    std::vector<int> lanes;
    if (devices) {
        if (num_devices <= 0) return DGNN_BAD_ARGUMENT;
        lanes.assign(devices, devices + num_devices);
    } else {                                                   // Device::all(), src/libdg_cuda/devices.rs:55-63
        int count = 0;
        const int status = dgnn_device_count(&count);
        if (status != DGNN_OK) return status;
        if (count <= 0) return DGNN_HIP_ERROR;
        for (int d = 0; d < count; ++d) lanes.push_back(d);
    }
    int count = 0;
    if (dgnn_device_count(&count) == DGNN_OK)
        for (int d : lanes) if (d < 0 || d >= count) return DGNN_BAD_ARGUMENT;

    dgnn_batcher* batcher = nullptr;                           // ONE queue for all devices like the reference's (batch.rs:60-73),
    int status = batcher_create(net, lanes[0], max_batch_size, batches_per_device * (int)lanes.size(), records, &batcher);   // 2 x devices alive (nn.rs:64-67)
    if (status != DGNN_OK) return status;
    std::unique_ptr<LeafPool> pool(new (std::nothrow) LeafPool());
    if (!pool) { dgnn_batcher_destroy(batcher); return DGNN_OUT_OF_MEMORY; }
    pool->batcher = reinterpret_cast<Batcher*>(batcher);
    pool->owns_batcher = true;
    pool->devices = lanes;
    pool->num_lanes = (int)lanes.size();
    status = start_pool(pool, workers_per_device, out);
    if (status != DGNN_OK) dgnn_batcher_destroy(batcher);
    return status;
}

static int pool_create_callbacks(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                 int workers_per_lane, bool records, dgnn_pool** out);

int dgnn_pool_create_callbacks(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                               int workers_per_lane, dgnn_pool** out)
{
    return pool_create_callbacks(predict, predict_ctx, num_lanes, max_batch_size, batches_per_lane, workers_per_lane, false, out);
}

int dgnn_pool_create_callbacks_boards(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                      int workers_per_lane, dgnn_pool** out)
{
    return pool_create_callbacks(predict, predict_ctx, num_lanes, max_batch_size, batches_per_lane, workers_per_lane, true, out);
}

int dgnn_pool_create_board_callbacks(dgnn_predict_boards_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                     int workers_per_lane, dgnn_pool** out)
{
    if (!predict || !out || num_lanes <= 0 || max_batch_size <= 0 || batches_per_lane <= 0 || workers_per_lane <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    dgnn_batcher* batcher = nullptr;
    int status = batcher_create(nullptr, 0, max_batch_size, batches_per_lane * num_lanes, true, &batcher);
    if (status != DGNN_OK) return status;
    std::unique_ptr<LeafPool> pool(new (std::nothrow) LeafPool());
    if (!pool) { dgnn_batcher_destroy(batcher); return DGNN_OUT_OF_MEMORY; }
    pool->batcher = reinterpret_cast<Batcher*>(batcher);
    pool->owns_batcher = true;
    pool->board_callback = predict;
    pool->callback_ctx = predict_ctx;
    pool->num_lanes = num_lanes;
    status = start_pool(pool, workers_per_lane, out);
    if (status != DGNN_OK) dgnn_batcher_destroy(batcher);
    return status;
}

static int pool_create_callbacks(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                 int workers_per_lane, bool records, dgnn_pool** out)
{
    if (!predict || !out || num_lanes <= 0 || max_batch_size <= 0 || batches_per_lane <= 0 || workers_per_lane <= 0) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    dgnn_batcher* batcher = nullptr;
    int status = batcher_create(nullptr, 0, max_batch_size, batches_per_lane * num_lanes, records, &batcher);
    if (status != DGNN_OK) return status;
    std::unique_ptr<LeafPool> pool(new (std::nothrow) LeafPool());
    if (!pool) { dgnn_batcher_destroy(batcher); return DGNN_OUT_OF_MEMORY; }
    pool->batcher = reinterpret_cast<Batcher*>(batcher);
    pool->owns_batcher = true;
    pool->callback = predict;
    pool->callback_ctx = predict_ctx;
    pool->num_lanes = num_lanes;
    status = start_pool(pool, workers_per_lane, out);
    if (status != DGNN_OK) dgnn_batcher_destroy(batcher);
    return status;
}

void dgnn_pool_destroy(dgnn_pool* handle)
{
    if (!handle) return;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    {
        std::lock_guard<std::mutex> lock(pool->mutex);
        pool->stopping = true;
    }
    pool->work.notify_all();
    for (std::thread& t : pool->workers) t.join();
    Batcher* b = pool->batcher;
    if (pool->owns_batcher) b->release();      // the handle dgnn_batcher_create returned
    b->release();                              // the pool's own reference (start_pool)
    delete pool;
}

// Quiesce for a weight refresh: returns when no worker holds a batch (or a workspace) any more; predict calls made meanwhile
// queue up and are served after dgnn_pool_resume -- by the new weights.  The sequence of a refresh inside a running job:
//   dgnn_pool_pause -> dgnn_broadcast_weights / dgnn_network_set_weight_image -> dgnn_cache_clear -> dgnn_pool_resume
// (the reference restarts the worker process instead, contrib/distr-env/run_worker.py:66-77).
// The adaptive rule above for a pool of board records: between `lo` and `hi` batches alive per lane (hi <= what the pool was created
// with: its workers and the batcher's cap).  lo == hi == 0 switches back to the constant the pool was created with.
int dgnn_pool_set_adaptive_batches(dgnn_pool* handle, int lo, int hi)
{
    if (!handle || lo < 0 || hi < lo) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    Batcher* b = pool->batcher;
    if (hi > 0 && (!b->records || lo == 0 || hi * pool->num_lanes > b->max_batches || hi * pool->num_lanes > (int)pool->workers.size()))
        return DGNN_BAD_ARGUMENT;
    {
        std::lock_guard<std::mutex> lock(b->mutex);
        b->lanes = pool->num_lanes;
        if (hi > 0 && b->recent.empty()) b->recent.assign(Batcher::kRecent, 0);
        b->window = Batcher::Window();
        b->probing = 0;
        b->choice = 0;
        b->limit.store(hi > 0 ? lo * pool->num_lanes : b->max_batches, std::memory_order_release);
        b->adaptive_lo.store(lo, std::memory_order_release);
        b->adaptive_hi.store(hi, std::memory_order_release);
    }
    pool->work.notify_all();
    return DGNN_OK;
}

// lo batches alive per lane; up to hi while at least min_queued boards are waiting in the queue (0: off, back to the constant)
int dgnn_pool_set_extra_batch_queue(dgnn_pool* handle, int lo, int hi, int min_queued)
{
    if (!handle || lo < 0 || hi < lo || min_queued < 0) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    Batcher* b = pool->batcher;
    if (min_queued > 0 && (lo == 0 || hi * pool->num_lanes > b->max_batches || hi * pool->num_lanes > (int)pool->workers.size())) return DGNN_BAD_ARGUMENT;
    {
        std::lock_guard<std::mutex> lock(b->mutex);
        b->lanes = pool->num_lanes;
        b->extra_lo.store(lo);
        b->extra_hi.store(hi);
        b->extra_min_queued.store(min_queued, std::memory_order_release);
    }
    pool->work.notify_all();
    return DGNN_OK;
}

int dgnn_pool_set_min_fill(dgnn_pool* handle, int min_boards, int max_wait_us)
{
    if (!handle || min_boards < 0 || max_wait_us < 0) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if (min_boards > pool->batcher->max_batch_size) min_boards = pool->batcher->max_batch_size;
    pool->max_wait_us.store(max_wait_us, std::memory_order_relaxed);
    pool->min_fill.store(min_boards, std::memory_order_release);
    pool->work.notify_all();
    return DGNN_OK;
}

int dgnn_pool_set_back_to_back(dgnn_pool* handle, int lead_us, int snap_lo, int snap_hi)
{
    if (!handle || lead_us < 0 || snap_lo < 0 || (snap_lo > 0 && snap_hi <= snap_lo)) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if ((lead_us > 0 || snap_lo > 0) && pool->num_lanes != 1) return DGNN_BAD_ARGUMENT;     // one device: the clock is that device's
    if (snap_lo >= pool->batcher->max_batch_size) snap_lo = 0;
    pool->snap_hi.store(snap_hi, std::memory_order_relaxed);
    pool->snap_lo.store(snap_lo, std::memory_order_relaxed);
    pool->lead_us.store(lead_us, std::memory_order_release);
    pool->work.notify_all();
    return DGNN_OK;
}

// What evaluate() took lately, by batch size: took_us[k] is the moving average over batches of 16 k .. 16 k + 15 boards (k = 16: 256 and
// more), 0 = none seen.  Wall time of the worker from handing the batch to the device to having the answers, i.e. including what it
// shared the device with.
int dgnn_pool_batch_times(const dgnn_pool* handle, int* took_us)
{
    if (!handle || !took_us) return DGNN_BAD_ARGUMENT;
    const LeafPool* pool = reinterpret_cast<const LeafPool*>(handle);
    for (int k = 0; k < 17; ++k) took_us[k] = pool->took_us[k].load(std::memory_order_relaxed);
    return DGNN_OK;
}

int dgnn_pool_set_coalescing(dgnn_pool* handle, int enabled)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if (enabled && !pool->batcher->records) return DGNN_BAD_ARGUMENT;       // positions are recognised in board records only
    std::lock_guard<std::mutex> lock(pool->inflight_mutex);
    if (!pool->inflight.empty()) return DGNN_BUSY;                          // not while leaders are out
    pool->coalesce = enabled != 0;
    return DGNN_OK;
}

uint64_t dgnn_pool_coalesced(const dgnn_pool* handle) { return handle ? reinterpret_cast<const LeafPool*>(handle)->coalesced.load() : 0; }

int dgnn_pool_adaptive_stats(dgnn_pool* handle, int* alive_per_lane, uint64_t* switches, uint64_t* boards_hashed, uint64_t* boards_repeated,
                             double* unique_rate_lo, double* unique_rate_hi)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    Batcher* b = pool->batcher;
    std::lock_guard<std::mutex> lock(b->mutex);
    if (alive_per_lane) *alive_per_lane = b->alive_limit() / (pool->num_lanes > 0 ? pool->num_lanes : 1);
    if (switches) *switches = b->switches.load();
    if (boards_hashed) *boards_hashed = b->boards_hashed.load() + b->window.pushed;
    if (boards_repeated) *boards_repeated = b->boards_repeated.load() + b->window.repeated;
    if (unique_rate_lo) *unique_rate_lo = b->rate_at[0];
    if (unique_rate_hi) *unique_rate_hi = b->rate_at[1];
    return DGNN_OK;
}

int dgnn_pool_pause(dgnn_pool* handle)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    std::unique_lock<std::mutex> lock(pool->mutex);
    pool->paused = true;
    pool->idle.wait(lock, [&] { return pool->evaluating == 0; });
    pool->generation.fetch_add(1, std::memory_order_seq_cst);
    return DGNN_OK;
}

uint64_t dgnn_pool_generation(const dgnn_pool* handle)
{
    return handle ? reinterpret_cast<const LeafPool*>(handle)->generation.load(std::memory_order_seq_cst) : 0;
}

int dgnn_pool_resume(dgnn_pool* handle)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    {
        std::lock_guard<std::mutex> lock(pool->mutex);
        pool->paused = false;
    }
    pool->work.notify_all();
    return DGNN_OK;
}

int dgnn_pool_stats(dgnn_pool* handle, int* num_lanes, uint64_t* calls, uint64_t* boards, uint64_t* batches,
                    uint64_t* lane_boards, uint64_t* lane_batches, int lane_capacity)
{
    if (!handle) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if (num_lanes) *num_lanes = pool->num_lanes;
    if (calls) *calls = pool->calls.load();
    if (boards) *boards = pool->batcher->boards_evaluated.load();
    if (batches) *batches = pool->batcher->batches_evaluated.load();
    for (int i = 0; i < pool->num_lanes && i < lane_capacity; ++i) {
        if (lane_boards) lane_boards[i] = pool->lane_boards[i].load();
        if (lane_batches) lane_batches[i] = pool->lane_batches[i].load();
    }
    return DGNN_OK;
}

static int pool_predict(LeafPool* pool, const uint16_t* features, int batch_size, uint16_t* value_out, uint16_t* policy_out);

int dgnn_pool_predict(dgnn_pool* handle, const uint16_t* features, int batch_size, uint16_t* value_out, uint16_t* policy_out)
{
    if (!handle || !features || !value_out || !policy_out || batch_size <= 0) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if (pool->batcher->records) return DGNN_BAD_ARGUMENT;      // this pool queues board records: dgnn_pool_predict_boards
    return pool_predict(pool, features, batch_size, value_out, policy_out);
}

int dgnn_pool_predict_boards(dgnn_pool* handle, const uint8_t* records, int batch_size, uint16_t* value_out, uint16_t* policy_out)
{
    if (!handle || !records || !value_out || !policy_out || batch_size <= 0) return DGNN_BAD_ARGUMENT;
    LeafPool* pool = reinterpret_cast<LeafPool*>(handle);
    if (!pool->batcher->records) return DGNN_BAD_ARGUMENT;
    return pool_predict(pool, reinterpret_cast<const uint16_t*>(records), batch_size, value_out, policy_out);
}

int dgnn_pool_takes_boards(const dgnn_pool* handle) { return handle && reinterpret_cast<const LeafPool*>(handle)->batcher->records ? 1 : 0; }

static int pool_predict(LeafPool* pool, const uint16_t* features, int batch_size, uint16_t* value_out, uint16_t* policy_out)
{
    LeafRequest request;
    request.fiber = fiber_current();
    request.value_out = value_out;
    request.policy_out = policy_out;
    request.remaining = batch_size;
    request.status = DGNN_OK;
    std::vector<LeafSlot> slots((size_t)batch_size);
    for (int i = 0; i < batch_size; ++i) {
        slots[i].request = &request;
        slots[i].index = i;
        const uint16_t* item = features + (size_t)i * pool->batcher->item;
        int status = DGNN_OK;
        bool follows = false;
        try {
            if (pool->coalesce) {
                slots[i].position = record_position_hash(item);
                slots[i].symmetry = (uint8_t)record_symmetry(item);
                std::lock_guard<std::mutex> lock(pool->inflight_mutex);
                auto found = pool->inflight.find(slots[i].position);
                if (found == pool->inflight.end()) {
                    pool->inflight.emplace(slots[i].position, LeafPool::InFlight{&request, {}});
                    slots[i].leads = true;
                } else if (found->second.owner != &request) {   // somebody ELSE has this position on its way: follow
                    found->second.followers.push_back(&slots[i]);
                    pool->coalesced.fetch_add(1, std::memory_order_relaxed);
                    follows = true;
                }
                // (the same position twice in ONE request is the root ensemble of lib.rs:83-130 -- eight symmetries that are meant to be
                // eight evaluations: the reference's table would not have served them either, it is filled after the predict call)
            }
        } catch (const std::bad_alloc&) {
            status = DGNN_OUT_OF_MEMORY;
        }
        if (follows) continue;
        if (status == DGNN_OK) status = batcher_push(pool->batcher, item, (uint64_t)(uintptr_t)&slots[i]);
        if (status != DGNN_OK) {       // an allocation failed: slots i.. never reach a device; the part that was pushed or follows still completes
            // A slot that registered as the LEADER of its position and then was not pushed would leave an entry nobody ever answers:
            // every later request for that position would follow it and wait for ever.  Take the entry back and hand the
            // status to whoever followed it meanwhile.
            std::vector<LeafSlot*> orphans;
            if (slots[i].leads) {
                std::lock_guard<std::mutex> lock(pool->inflight_mutex);
                auto it = pool->inflight.find(slots[i].position);
                if (it != pool->inflight.end() && it->second.owner == &request) { orphans.swap(it->second.followers); pool->inflight.erase(it); }
                slots[i].leads = false;
            }
            for (LeafSlot* f : orphans) complete_slot(f, status, 0, nullptr);
            std::unique_lock<std::mutex> lock(request.mutex);
            request.remaining -= batch_size - i;
            request.status = status;
            if (request.remaining == 0) return status;     // nothing of this request is on its way
            break;
        }
    }
    pool->calls.fetch_add(1, std::memory_order_relaxed);
    pool->wake();
    if (request.fiber) {
        fiber_yield();                                     // resumed by the pool worker that completes the request
        std::lock_guard<std::mutex> lock(request.mutex);   // (that worker may still be inside its critical section)
        return request.status;
    }
    std::unique_lock<std::mutex> lock(request.mutex);
    request.done.wait(lock, [&] { return request.remaining == 0; });
    return request.status;
}

void dgnn_test_fail_push_after(int n) { g_fail_push_countdown.store(n, std::memory_order_relaxed); }

int dgnn_batcher_stats(dgnn_batcher* batcher, uint64_t* boards, uint64_t* batches)
{
    if (!batcher) return DGNN_BAD_ARGUMENT;
    Batcher* b = reinterpret_cast<Batcher*>(batcher);
    if (boards) *boards = b->boards_evaluated.load();
    if (batches) *batches = b->batches_evaluated.load();
    return DGNN_OK;
}

}  // extern "C"
