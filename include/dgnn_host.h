/*
 * dgnn_host.h -- the host side of libdgnn_hip.so: what sits either side of the leaf-evaluation path in the reference, rebuilt in C++
 * behind the same C ABI conventions as dgnn.h (SURVEY.md section 8f, rows f-1 .. f-4):
 *
 *   f-1  the many-game leaf pool and the transposition table   src/libdg_mcts/pool/batch.rs, predictors/nn.rs, lru_cache.rs
 *   f-4  the formats either side of the path                    src/libdg_utils/b85.rs, src/libdg_mcts/self_play.rs:187-213
 *   f-2  the search tree and the search driver                  src/libdg_mcts/tree.rs, pool/worker_thread.rs, lib.rs
 *   f-3  the Go board and the V1 feature planes                 src/libdg_go
 *        whole self-play games                                   src/libdg_mcts/self_play.rs:423-499
 *
 * A host that keeps dream-go's own Rust crates for these rows needs dgnn.h only.  Entry points that exist to pin
 * the rows to the reference's unit tests -- accessors, single steps, fault injection -- are in dgnn_test.h and are exported by the
 * test build of the library only.
 */
#ifndef DGNN_HOST_H
#define DGNN_HOST_H

#include "dgnn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ============================================================================================================
 * Host side of the boundary (scope row f-1): policy symmetries, transposition cache, many-game leaf batcher.
 * Pure host code (batcher.cpp); only a pool over devices touches a GPU.
 * ============================================================================================================ */

/* The eight board symmetries in the order of symmetry::ALL (src/libdg_go/utils/symmetry.rs:119-128). */
#define DGNN_SYM_IDENTITY        0
#define DGNN_SYM_FLIP_LR         1
#define DGNN_SYM_FLIP_UD         2
#define DGNN_SYM_TRANSPOSE       3
#define DGNN_SYM_TRANSPOSE_ANTI  4
#define DGNN_SYM_ROT90           5
#define DGNN_SYM_ROT180          6
#define DGNN_SYM_ROT270          7

/* (Transform::inverse / apply and Prediction::with_transform themselves: dgnn_test.h.) */

/* LruCache + NnPredictor::{fetch, cache} (src/libdg_mcts/lru_cache.rs:56-150, predictors/nn.rs:29-82).  Keyed by
 * (zobrist hash, colour to move); entries are stored in the canonical orientation.  `num_shards` = 1 reproduces the
 * reference's single list (and its eviction order) exactly; more shards remove the global lock.  Thread safe. */
typedef struct dgnn_cache dgnn_cache;
int dgnn_cache_create(size_t capacity, int num_shards, dgnn_cache** out);        /* reference: 200 000 + 1, nn.rs:29,57 */
void dgnn_cache_destroy(dgnn_cache* cache);
/* NnPredictor::{fetch, cache} (nn.rs:60-82).  fetch: 1 on a hit (value_out / policy_out[362] filled, remapped by `symmetry`), 0 on a
 * miss, < 0 on error.  insert: stores the prediction remapped by symmetry.inverse(); a key that is already present is left untouched. */
int dgnn_cache_fetch(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t* value_out, uint16_t* policy_out);
int dgnn_cache_insert(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t value, const uint16_t* policy);
int dgnn_cache_stats(dgnn_cache* cache, uint64_t* entries, uint64_t* hits, uint64_t* misses);
int dgnn_cache_clear(dgnn_cache* cache);     /* drops every entry: the table holds the outputs of the weights that are being replaced */
/* A clear also starts a new GENERATION of the table: the search driver reads the generation before a fetch and stores the answer of a
 * miss only if it still holds, so an answer whose miss predates a clear is dropped instead of putting the replaced network's output
 * back into the emptied table. */

/* Batcher / Batch (src/libdg_mcts/pool/batch.rs:24-123): the queue inside a leaf pool -- leaf positions from any number of games
 * and threads, handed out as the NEWEST max_batch_size of them while fewer than `max_batches` batches are alive (the reference:
 * 2 x devices, nn.rs:64-67).  A pool owns its batcher; the queue's own operations are in dgnn_test.h. */
typedef struct dgnn_batcher dgnn_batcher;
typedef struct dgnn_batch dgnn_batch;

/* Cross-caller coalescing (no reference counterpart: the reference sends the <= 8-board full_forward calls of its
 * game threads to the GPU one by one, src/libdg_mcts/lib.rs:114).  dgnn_pool_predict has the blocking
 * Predictor::predict signature (src/libdg_mcts/predictor.rs:92) and may be called from any number of threads; worker
 * threads of the pool turn everything that has been pushed meanwhile into device batches (<= max_batch_size boards,
 * <= max_batches in flight). */
typedef struct dgnn_pool dgnn_pool;
void dgnn_pool_destroy(dgnn_pool* pool);   /* every dgnn_pool_predict call must have returned */
int dgnn_pool_predict(dgnn_pool* pool, const uint16_t* features, int batch_size, uint16_t* value_out, uint16_t* policy_out);

/* Predictor::predict as a C callback (src/libdg_mcts/predictor.rs:92): n boards of 11552 fp16 features -> value[n],
 * policy[n * 362] fp16; returns a DGNN_* status. */
typedef int (*dgnn_predict_fn)(void* ctx, const uint16_t* features, int n, uint16_t* value_out, uint16_t* policy_out);

/* The same signature over compact board records (n * DGNN_BOARD_RECORD_BYTES, 2-byte aligned): what a pool of board records and
 * dgnn_forward_boards take. */
typedef int (*dgnn_predict_boards_fn)(void* ctx, const uint8_t* records, int n, uint16_t* value_out, uint16_t* policy_out);

/* NnPredictor over SEVERAL devices in one process (src/libdg_mcts/predictors/nn.rs:46-108: one queue, `2 x devices` batches
 * alive, every batch sent to the next device of Device::all()).  One queue shared by all devices; `workers_per_device`
 * threads are bound to each device and a batch goes to whichever device has a worker free (the reference's round robin
 * without its head-of-line blocking); `batches_per_device` x devices batches alive (the reference: 2.  A third keeps an MI355X
 * supplied while the batches are partial -- 76-81 k instead of 57-75 k evals/s at 129 .. 200 boards of 256 -- but a search behind a
 * shared transposition table then evaluates more positions twice: measure moves per second, not evals: DESIGN.md 11.7).  devices == NULL: every
 * visible device.  The pool owns its batcher.  dgnn_pool_predict / dgnn_pool_predictor() work on it unchanged, so one host
 * process (dgnn_selfplay_go, a GTP engine) drives all GPUs of a node. */
int dgnn_pool_create_devices(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                             int workers_per_device, dgnn_pool** out);
/* The same pool over `num_lanes` copies of a predictor callback instead of devices: the queueing, coalescing and lane logic
 * without a GPU (CPU tests, dry runs of the harness). */
int dgnn_pool_create_callbacks(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                               int workers_per_lane, dgnn_pool** out);
/* The same batcher and pools queueing COMPACT BOARD RECORDS instead of fp16 planes (dgnn_forward_boards): 896 instead of 23 104
 * bytes per position through the queue and over PCIe, and no feature extraction on the host.  A record pool is fed with
 * dgnn_pool_predict_boards / dgnn_pool_boards_predictor(); dgnn_pool_predict on it (and vice versa) is DGNN_BAD_ARGUMENT. */
int dgnn_pool_create_devices_boards(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                                    int workers_per_device, dgnn_pool** out);
/* ... and over copies of a predictor that takes the records themselves (no expansion on the host). */
int dgnn_pool_create_board_callbacks(dgnn_predict_boards_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                     int workers_per_lane, dgnn_pool** out);
int dgnn_pool_predict_boards(dgnn_pool* pool, const uint8_t* records, int batch_size, uint16_t* value_out, uint16_t* policy_out);
int dgnn_pool_takes_boards(const dgnn_pool* pool);               /* 1 for a pool of board records */

/* Coalescing of positions that are IN FLIGHT (a pool of board records; off by default).  The transposition table
 * (NnPredictor::{fetch, cache}, predictors/nn.rs:29-82) knows a position once its evaluation has come back; until then every search
 * that reaches it evaluates it again -- with hundreds of games in the same opening that is most of the device's work.  With
 * coalescing a request for a position (same stones, same colour to move) that the pool already has queued or on a device FOLLOWS it
 * and receives that answer re-oriented from the leader's symmetry to its own: what a table hit would have returned
 * (nn.rs:72-81), one evaluation earlier.  Like the shared table it makes WHICH evaluation a search sees a matter of timing: games that
 * must be a function of their seed (cache_entries_per_game > 0) keep it off -- dgnn_selfplay_go refuses that combination with
 * DGNN_BAD_ARGUMENT.  DGNN_BUSY while requests are in flight. */
int dgnn_pool_set_coalescing(dgnn_pool* pool, int enabled);
/* Back to back (pools of ONE lane; off by default): while a batch occupies the device, a PARTIAL batch is held back until
 * `lead_us` before that batch is expected to finish (the pool keeps a moving average of how long a batch of each size took) and
 * then goes with everything that has queued up meanwhile -- its upload and first kernels fall into the other's tail, the device is
 * never idle, and the batches come out as full as the leaves in flight allow (a workgroup per board makes 120 boards take what 128
 * take, and 190 what 256 take).  A full batch is sent at once, as before.  snap_lo > 0: a batch of snap_lo < n < snap_hi boards is
 * cut to its snap_lo OLDEST boards, the rest goes with the next batch (alone on the device, 129 .. 200 boards are the least
 * efficient sizes).  lead_us 0 and snap_lo 0 switch both off.  The reference sends what is queued the moment a slot is free
 * (pool/worker_thread.rs:131-140, predictors/nn.rs:64-67). */
int dgnn_pool_set_back_to_back(dgnn_pool* pool, int lead_us, int snap_lo, int snap_hi);
/* What a batch took lately, by size: took_us[k] = moving average (microseconds, wall time of the worker from handing the batch over to
 * having its answers) over batches of 16 k .. 16 k + 15 boards, k = 0 .. 16 (16: 256 and more); 0 = no such batch yet.  The estimate
 * back-to-back batching runs on, and a measured handle on the fixed cost per batch. */
int dgnn_pool_batch_times(const dgnn_pool* pool, int* took_us /* 17 */);
uint64_t dgnn_pool_coalesced(const dgnn_pool* pool);        /* boards that followed instead of being evaluated */

/* (How many batches a pool keeps alive per lane is the `batches_per_device` it was created with -- the reference's constant 2,
 * predictors/nn.rs:64-67.  The rules that were measured against the constant and lost -- a minimum fill, 2 or 3 by the unique positions
 * per second, a third batch only behind a full queue -- are kept for the record in dgnn_test.h.) */

/* Weight refresh inside a running job (f-4: "closes the RL loop"; the reference restarts the worker process,
 * contrib/distr-env/run_worker.py:66-77):  dgnn_pool_pause -> dgnn_broadcast_weights or dgnn_network_set_weight_image ->
 * dgnn_cache_clear -> dgnn_pool_resume.  pause returns when no worker holds a batch or a workspace any more (so the weights may be
 * replaced: no DGNN_BUSY); predict calls made meanwhile queue up and are answered after resume -- by the new weights. */
int dgnn_pool_pause(dgnn_pool* pool);
int dgnn_pool_resume(dgnn_pool* pool);
/* Completed pauses so far.  Games with a transposition table of their own (dgnn_selfplay_params.cache_entries_per_game > 0), which
 * dgnn_cache_clear cannot reach, compare it before every move and empty their table when it has changed. */
uint64_t dgnn_pool_generation(const dgnn_pool* pool);

/* Counters: dgnn_pool_predict calls, boards / device batches evaluated, and the same per lane (device) for the first
 * lane_capacity lanes.  Any out pointer may be NULL. */
int dgnn_pool_stats(dgnn_pool* pool, int* num_lanes, uint64_t* calls, uint64_t* boards, uint64_t* batches,
                    uint64_t* lane_boards, uint64_t* lane_batches, int lane_capacity);

/* ============================================================================================================
 * Formats either side of the path (scope row f-4), formats.cpp.
 * ============================================================================================================ */

/* b85::encode (src/libdg_utils/b85.rs:141-165): `count` (even) fp32 values -> fp16 -> 5 characters per pair, NUL
 * terminated.  Returns the number of bytes needed including the NUL (also when out is NULL or too small: size query);
 * 0 for invalid arguments. */
size_t dgnn_b85_encode_f16(const float* values, size_t count, char* out, size_t capacity);
/* b85::decode (b85.rs:101-132): raw little-endian bytes, 4 per 5 characters (a trailing partial group is dropped).
 * out may be NULL to query *num_bytes.  DGNN_MALFORMED_WEIGHTS for a character outside the alphabet. */
int dgnn_b85_decode(const char* text, size_t len, uint8_t* out, size_t capacity, size_t* num_bytes);
/* Played::fmt (src/libdg_mcts/self_play.rs:187-213): one move of a self-play record,
 * ";B[dd]" [C[explain]] [TR[prior]] [TV[rollouts]P[b85 of the 362 search probabilities]] [V[win rate, black's side]].
 * color 0 = black, 1 = white; packed_index 361 (or any value outside 0..360) = pass; prior_index < 0 = none;
 * value = win rate of the side to move in [0, 1].  Returns the bytes needed including the NUL; writes when it fits. */
size_t dgnn_sgf_played(char* out, size_t capacity, int color, int packed_index, int num_rollout, const float* softmax,
                       int has_value, float value, int prior_index, const char* explain);

/* ============================================================================================================
 * Search tree (scope row f-2, first part), tree.cpp: dream-go's PUCT tree with first-play urgency, virtual loss and
 * the "expanding" flag (src/libdg_mcts/tree.rs), without any Go knowledge: legality is in the prior (-inf), the board
 * is a callback that is told which stones the descent places.
 * ============================================================================================================ */
#define DGNN_COLOR_BLACK 0
#define DGNN_COLOR_WHITE 1
#define DGNN_PROBE_FOUND      1     /* ProbeResult::Found    (tree.rs:996-1004) */
#define DGNN_PROBE_NO_RESULT  0     /* ProbeResult::NoResult: no child with a finite value */
#define DGNN_PROBE_CONFLICT   2     /* ProbeResult::Conflict: the chosen child is being expanded by somebody else (a result, not an error) */

typedef struct dgnn_tree_node dgnn_tree_node;   /* Node, tree.rs:1027-1051 */
typedef struct dgnn_trace dgnn_trace;           /* NodeTrace = Vec<(node, colour, child index)>, tree.rs:1387 */
typedef struct dgnn_board_ops {
    void (*place)(void* board, int color, int packed_index);   /* Board::place during the descent, tree.rs:1452 */
} dgnn_board_ops;

/* (The constants and single steps the reference's tree tests pin -- config::get_uct_exp ..., asm::argmax_f32, Child accessors, undo,
 * choose, the stopping rule -- are in dgnn_test.h.) */

/* Node::new(to_move, value, prior[>= 362]) (tree.rs:1070-1090); a node owns its sub-tree. */
dgnn_tree_node* dgnn_tree_node_new(int to_move, float value, const float* prior);
void dgnn_tree_node_free(dgnn_tree_node* node);
/* probe (tree.rs:1421-1470): returns DGNN_PROBE_FOUND with *out = the trace, DGNN_PROBE_NO_RESULT or DGNN_PROBE_CONFLICT
 * (virtual losses of the partial descent undone).  ops / board may be NULL. */
int dgnn_tree_probe(dgnn_tree_node* root, const dgnn_board_ops* ops, void* board, dgnn_trace** out);
void dgnn_trace_free(dgnn_trace* trace);
/* insert + UCT::update (tree.rs:1482-1510, 125-160); does not free the trace. */
int dgnn_tree_insert(const dgnn_trace* trace, int color, float value /* win rate of `color`, 0..1 */, const float* prior);
/* Node accessors (tree.rs:1027-1051).  Any out pointer may be NULL. */
int dgnn_tree_node_info(const dgnn_tree_node* node, int* to_move, int* total_count, int* vtotal_count, int* pass_count, float* initial_value);
int dgnn_tree_node_softmax(const dgnn_tree_node* node, float* out /* 362 */);                 /* tree.rs:1293-1307 */
/* Node::best (tree.rs:1262-1283): greedy (LCB / visits / prior / value order) when temperature <= 0.09, else a visit-
 * proportional draw with `at` in [0, 1) standing in for thread_rng(). */
int dgnn_tree_node_best(const dgnn_tree_node* node, float temperature, double at, float* value_out, int* index_out);
/* tree::to_pretty (tree.rs:1620-1746, non-verbose): the human-readable summary of a search ("Nodes: .., Win: ..%, PV: .." and one
 * line per child for the ten best) that Played::from_mcts stores with a move and Played::fmt writes as C[...] (self_play.rs:136,
 * 191-193).  Returns the bytes needed including the NUL; writes when it fits. */
size_t dgnn_tree_node_to_pretty(const dgnn_tree_node* node, char* out, size_t capacity);

/* ---- search driver (scope row f-2, second part), search.cpp ------------------------------------------------- */

/* dgnn_predict_fn (declared with the leaf pool above) is the predictor behind a search. */
dgnn_predict_fn dgnn_pool_predictor(void);    /* ctx = a dgnn_pool*: the HIP engine behind the cross-caller batcher */
dgnn_predict_fn dgnn_hash_predictor(void);    /* ctx unused: deterministic CPU stand-in (role of predictors/{random,fake}.rs) */
dgnn_predict_boards_fn dgnn_pool_boards_predictor(void);   /* ctx = a dgnn_pool* of board records */
/* The CPU stand-in on board records.  ctx NULL: returns at once; ctx = a dgnn_dry_device*: a call also takes
 * base_us + per_board_us * n of wall time (the calling thread sleeps; calls are served one at a time, like batches on one GPU) --
 * the host side of self-play measured without a GPU (bench.py --dry). */
typedef struct dgnn_dry_device { double base_us, per_board_us; } dgnn_dry_device;
dgnn_predict_boards_fn dgnn_hash_boards_predictor(void);

/* The game behind the search: what libdg_go provides in the reference (Board::clone / place,
 * features::V1::get_features, the policy checker of pool/policy_helper.rs:36-48, symmetry::is_symmetric). */
typedef struct dgnn_game_ops {
    void* (*clone)(void* board);
    void (*destroy)(void* board);
    void (*place)(void* board, int color, int packed_index);                       /* 361 = pass is never placed */
    void (*features)(void* board, int to_move, int symmetry, uint16_t* out /* 11552 */);
    void (*candidates)(void* board, int to_move, unsigned char* out /* 362: 1 = may be searched */);
    int (*is_symmetric)(void* board, int symmetry);                                /* may be NULL: no elimination */
    uint64_t (*hash)(void* board);                                                 /* may be NULL: no transposition cache */
    void (*record)(void* board, int to_move, int symmetry, uint8_t* out /* DGNN_BOARD_RECORD_BYTES */);   /* may be NULL: only fp16 features */
    void (*candidate_bits)(void* board, int to_move, uint64_t* out /* 6 words: bit i = move i may be searched */);   /* may be NULL: `candidates` is used */
} dgnn_game_ops;

/* One search on `root` from `board` (not modified): rollouts until root.total_count >= num_rollouts or the reference's early stop --
 * the runner-up cannot overtake the most visited child in the rollouts that remain (pool/worker_thread.rs:50-176,
 * time_control/mod.rs:86-101, rollout_limit.rs:35-44) -- with the transposition cache of NnPredictor in front of the predictor
 * (pool/event.rs:50-51, pool/worker_thread.rs:111: needs ops->hash; cache may be NULL; *evals_out counts predictor calls only).
 * All randomness derives from `seed`; with num_threads == 1 the resulting visit counts are a pure function of the seed and the
 * predictor's outputs.  Up to `leaves_per_thread` (1..64) leaves per worker are in flight: a worker probes that many leaves one
 * after the other (each probe's virtual loss steers the next one elsewhere), evaluates them in ONE predictor call and inserts them.
 * 1 is the reference's worker.  One thread with K leaves keeps the search a pure function of (seed, network outputs) while K
 * boards are on the device, and on a host with few cores it replaces K threads and their wake-ups by one. */
int dgnn_search_run_batched(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                            dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                            uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out);
/* dgnn_search_run_batched / dgnn_selfplay_go behind a predictor of board records: the game's `record` callback
 * is used instead of `features` (pool/event.rs:54 without the feature extraction), everything else is the same code. */
int dgnn_search_run_boards(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_boards_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                           uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out);
/* Node::forward (tree.rs:1225-1250): the sub-tree below `index` becomes the root and is returned (NULL when it was never
 * expanded, except for a pass); `node` and the rest of its tree are freed.  set_prior: lib.rs:176-183. */
dgnn_tree_node* dgnn_tree_node_forward(dgnn_tree_node* node, int index);
int dgnn_tree_node_set_prior(dgnn_tree_node* node, const float* prior /* 362 */);

/* What a self-play job counted (dgnn_selfplay_go below). */
typedef struct dgnn_selfplay_stats { double seconds; uint64_t games, moves, evals, conflicts, cache_hits, games_ended, ex_it_searches; } dgnn_selfplay_stats;

/* ============================================================================================================
 * Go board, ladders and feature planes (scope row f-3), go.cpp: what produces the input of the hot path.
 * Colours: DGNN_COLOR_BLACK / DGNN_COLOR_WHITE; coordinates x, y in 0..18 like Point::new (src/libdg_go/point.rs:26-34).
 * ============================================================================================================ */
typedef struct dgnn_go_board dgnn_go_board;                                  /* Board, src/libdg_go/board.rs:28-50 */
dgnn_go_board* dgnn_go_board_new(float komi);
dgnn_go_board* dgnn_go_board_clone(const dgnn_go_board* board);
void dgnn_go_board_free(dgnn_go_board* board);
int dgnn_go_place(dgnn_go_board* board, int color, int x, int y);            /* Board::place: no legality check (board.rs:152-176) */
int dgnn_go_is_valid(const dgnn_go_board* board, int color, int x, int y);   /* Board::is_valid: no suicide, no super-ko */
/* features::V1::get_features::<HWC, f16> (src/libdg_go/utils/features.rs:99-251): 361 * 32 fp16, the `features` argument of
 * dgnn_forward for one board. */
int dgnn_go_features(const dgnn_go_board* board, int to_move, int symmetry, uint16_t* out);
/* The compact record of the position for dgnn_forward_boards (layout there). */
int dgnn_go_board_record(const dgnn_go_board* board, int to_move, int symmetry, uint8_t* out /* DGNN_BOARD_RECORD_BYTES */);
/* The board as the game behind dgnn_search_run_batched (a dgnn_go_board* is the `board` argument). */
const dgnn_game_ops* dgnn_go_game_ops(void);        /* StandardSearch: every legal move and pass (options.rs:44-82) */
const dgnn_game_ops* dgnn_go_scoring_ops(void);     /* ScoringSearch: no pass, no point inside a Benson eye of either colour, no filling of
                                                       one's own simple eye (options.rs:110-214) */

/* (Liberties, ladders, Benson's unconditional life, candidate moves, scoring: what the reference's libdg_go tests look at is in
 * dgnn_test.h.)  Score::get_stone_status (utils/score.rs:146-186) reports a bit mask per point: */
#define DGNN_STONE_ALIVE 1
#define DGNN_STONE_DEAD 2
#define DGNN_STONE_SEKI 4
#define DGNN_TERRITORY_BLACK 8
#define DGNN_TERRITORY_WHITE 16
/* Self-play on the real board following self_play_one / Player::predict / predict (src/libdg_mcts/self_play.rs:423-459,
 * 331-420, lib.rs:144-200): random komi, per-player trees re-used across moves, win-rate driven rollout budget, 8-symmetry
 * root evaluation through `cache` (may be NULL); a deterministic ScoringSearch without pass until the board is scorable, a
 * StandardSearch with dirichlet noise after; two passes on a scorable board end the game (stats->games_ended counts those).
 * max_moves caps the stones per game (0 = the reference's 722).  moves_out (num_games * max_moves packed indices, -1 = not
 * played) and sgf_out (one GameResult::Ended record per game, game_result.rs:35-41, newline separated) may be NULL.
 * The call plays the global game ids first_game, first_game + game_stride, ... (num_games of them); a game's random
 * stream depends on (seed, global id) only, so ranks sharding games `id mod world` play what one rank would.
 * min(num_parallel, num_games) games run at a time, each game thread taking the next unplayed game (self_play.rs:478-495;
 * num_parallel <= 0: all at once).  num_rollouts <= 10 gives the policy-play mode (the budget falls to one rollout,
 * self_play.rs:369-405); num_ex_it_rollouts > 0 is --ex-it: one move in twenty is re-searched with that budget and the
 * search's distribution replaces the record's (self_play.rs:277-319).  With threads_per_search = 1 a game is a function of
 * (seed, game id, network outputs) for any leaves_per_thread: deterministic self-play with many leaves on the device --
 * provided the games do not share a transposition table (cache == NULL or cache_entries_per_game > 0): the table is keyed
 * like the reference's, by stones and colour only, while the feature planes also hold the last moves and the ko state. */
struct dgnn_selfplay_window;
typedef struct dgnn_selfplay_params {     /* zero the whole struct first: fields are appended between versions and 0 means "off" */
    int num_games;            /* games this call plays */
    int num_parallel;         /* games in flight (--num-games of the reference); <= 0: all at once */
    int first_game;           /* global id of the first game, */
    int game_stride;          /*   and the distance to the next one (sharding `id mod world`); <= 0: 1 */
    int max_moves;            /* cap on the stones per game; 0 = the reference's 722 */
    int num_rollouts;         /* --num-rollout; <= 10 gives policy play */
    int num_ex_it_rollouts;   /* --ex-it with this --num-ex-it-rollout; 0 = off */
    int threads_per_search;   /* search threads per game (--num-threads); <= 0: 1 */
    int leaves_per_thread;    /* leaves a search thread sends to the predictor per call (dgnn_search_run_batched); <= 0: 1 */
    int cache_entries_per_game; /* > 0: every game gets a transposition table of its own with this many entries instead of the
                                   shared `cache` -- needed for reproducible games, see below */
    uint64_t seed;
    int host_threads;         /* 0: one OS thread per game in flight (the reference's model, self_play.rs:478-495).  > 0: the games in flight
                                 are FIBERS on this many host threads (csrc/fiber.h): a game that waits for the leaf pool switches, in
                                 user space, to the next game of its thread instead of sleeping in the kernel -- hundreds of games per
                                 core.  Same games, move for move (a game is a function of its seed and the network's outputs);
                                 ignored (= 0) where fibers are not available. */
    int opening_len;          /* > 0: every game starts from a given position: the first opening_len entries of its row of `opening` are
                                 placed, black first, before the first move is searched (a -1 ends a row early) */
    const int* opening;       /* [num_games][opening_len] packed indices, e.g. the moves_out of an earlier call; NULL = empty boards.
                                 (A row holds stones only and is placed with strictly alternating colours, ending at the first entry
                                 that is not a legal move then: the moves_out of a game with a pass in it gives a position to start
                                 from, not that game's position.) */
    int opening_games;        /* > 0: `opening` holds rows for the first opening_games games only, later games start from the empty board */
    struct dgnn_selfplay_window* window;   /* non-NULL with seconds > 0: a timed window over a steady state of num_parallel games, see below */
    void (*on_game)(void* ctx, int game_id, const char* record, size_t bytes, int whole);
                              /* non-NULL: the reference's channel (self_play.rs:471-497: every game thread sends its GameResult, the caller
                                 prints it and lets go of it).  Called when a game leaves, with its global id, its record (GameResult::Ended,
                                 game_result.rs:35-41; not NUL-terminated, valid during the call) and whether it was a WHOLE game (two passes on
                                 a scorable board, or the stone cap) rather than one a window's end or an error cut short.  Calls are
                                 serialised by the library (one at a time, from the games' threads).  A job that runs for hours takes its
                                 records this way: sgf_out holds every record until the call returns (~1.5 MB per searched game). */
    void* on_game_ctx;
} dgnn_selfplay_params;
/* A measurement window over a running self-play job: games per hour on somebody's clock needs whole games, and a whole game is
 * minutes.  The job starts num_parallel games (each from its row of `opening`, so that the games in flight can be at EVERY stage of
 * a game from the first second on), replaces every game that ends (self_play.rs:478-495) -- num_games is then only the upper bound
 * on the games that may be started -- and a watcher thread notes the counters `ramp_seconds` after the start and `seconds` later;
 * then the job stops: every game finishes the move it is searching and leaves, unfinished.  on_edge (may be NULL) is called from
 * the watcher thread at the window's start (edge 0) and end (edge 1): the place to read what the library does not count (the leaf
 * pool's device boards and batches, the process's CPU time).  A game COMPLETED when two passes ended it on a scorable board or
 * when it reached the stone cap (self_play.rs:437-450); stones_completed sums the stones on the boards of those games, the stones
 * of their openings included. */
typedef struct dgnn_selfplay_window {
    double ramp_seconds, seconds;                               /* in */
    void (*on_edge)(void* ctx, int edge);                       /* in */
    void* ctx;                                                  /* in */
    double measured_seconds;                                    /* out: the window as the watcher's clock saw it */
    uint64_t moves, evals, cache_hits, conflicts, games_started, games_completed, games_ended, stones_completed;   /* out: inside the window */
    uint64_t ex_it_searches;                                    /* out: expert-iteration searches that ENDED inside the window (appended in round 6) */
} dgnn_selfplay_window;
int dgnn_selfplay_go(dgnn_predict_fn predict, void* predict_ctx, dgnn_cache* cache, const dgnn_selfplay_params* params,
                     dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity);

int dgnn_selfplay_go_boards(dgnn_predict_boards_fn predict, void* predict_ctx, dgnn_cache* cache, const dgnn_selfplay_params* params,
                            dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity);

#ifdef __cplusplus
}
#endif

#endif /* DGNN_HOST_H */
