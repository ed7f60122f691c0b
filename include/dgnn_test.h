/*
 * dgnn_test.h -- entry points that exist for the TESTS and for measurements: single layers on host buffers, accessors into the search
 * tree and the Go board that the reference's own unit tests look at, the raw batcher behind the leaf pool, fault injection.  They are
 * compiled into every build of the library (several are used inside it), but EXPORTED only by the test build
 * (dream-go_amd/lib/libdgnn_hip_test.so: the same kernel objects, the host sources compiled with -DDGNN_TEST_ENTRIES); the shipped
 * libdgnn_hip.so exports what dgnn.h and dgnn_host.h declare and nothing else (tests/test_cabi_host.py holds it to that).
 * The reference item each one mirrors is named with it, as in the other two headers.
 */
#ifndef DGNN_TEST_H
#define DGNN_TEST_H

#include "dgnn_host.h"

#if defined(DGNN_TEST_ENTRIES) || defined(DGNN_ABLATIONS)
#define DGNN_TEST_API __attribute__((visibility("default")))
#else
#define DGNN_TEST_API __attribute__((visibility("hidden")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- single layers on host buffers: the reference's layer KATs go through the HIP kernels ------------------------------------ */

/* One Conv2d::forward / forward_skip (src/libdg_nn/layers/conv2d.rs:171-220) on host buffers, for the layer-level parity tests that
 * mirror the reference's own (conv2d.rs:252-291):  y = act(alpha1 * conv3x3(x, w) + alpha2 * z + bias)
 *   x [batch][board][board][cin] fp16 NHWC, w [cout][3][3][cin] fp16 KRSC, bias [cout], z/y [..][cout];
 *   cin must be a multiple of 32 (pad with zero channels); z is read only when alpha2 != 0;
 *   relu != 0 selects ReLU, 0 the identity.  Allocates, runs and frees on `device`; blocking. */
DGNN_TEST_API int dgnn_layer_conv3x3(int device, const uint16_t* x, int batch, int board_size, int cin,
                       const uint16_t* w_krsc, const uint16_t* bias, int cout,
                       float alpha1, float alpha2, const uint16_t* z, int relu, uint16_t* y);
/* One Dense::forward (src/libdg_nn/layers/dense.rs:173-220) on host buffers through the MFMA GEMM of the policy head (heads.hip:
 * policy_gemm_kernel), for the layer-level tests that mirror the reference's own (dense.rs:243-299):
 *   y[b][m] = act(fp16(alpha * sum_k x[b][k] * w[k][m] + bias[m]))      fp32 accumulate
 *   x [batch][k_in] fp16, w [k_in][n_out] fp16 AS STORED in the weights file ([in][out]; the transpose of dense.rs:182-189 is done
 *   here), bias [n_out] fp16 (already scaled by the caller where the reference scales it), y [batch][n_out]; k_in a multiple of 8;
 *   relu != 0 selects ReLU (dense.rs:108-116), 0 the identity. */
DGNN_TEST_API int dgnn_layer_dense(int device, const uint16_t* x, int batch, int k_in, const uint16_t* w_in_out, const uint16_t* bias,
                     int n_out, float alpha, int relu, uint16_t* y);
/* The tail of both heads (heads.hip: softmax_value_kernel) on host buffers: cudnnSoftmaxForward ACCURATE over the 362 logits of a
 * board (src/libdg_cuda/cudnn/softmax.rs:58-79; policy_head.rs:98-100) and the value head's Dense 722 -> 1 followed by tanh
 * (value_head.rs:80-81).  logits [batch][362], value_features [batch][722], value_w [722], value_b [1]. */
DGNN_TEST_API int dgnn_layer_heads(int device, const uint16_t* logits, const uint16_t* value_features, int batch, const uint16_t* value_w,
                     const uint16_t* value_b, uint16_t* policy_out, uint16_t* value_out);
/* The two fp16 tensors the reference materialises IN FRONT of the output activations, as the last forward on `ws` left them -- the
 * temperature-scaled policy logits, input of cudnnSoftmaxForward (policy_head.rs:95-100; [batch][362]) and the value head's dense
 * output, input of tanh (value_head.rs:80-81; [batch]).  The parity tests bound the error there, in fp16 ulps. */
DGNN_TEST_API int dgnn_workspace_download_pre(dgnn_workspace* ws, uint16_t* logits_out, uint16_t* value_pre_out);
/* The planes the DEVICE builds from `count` board records (features.hip), downloaded in packed form (allocates, runs, frees). */
DGNN_TEST_API int dgnn_board_records_to_packed_device(int device, const uint8_t* records, int count, uint32_t* packed);
/* Where the wall time of one device batch goes: `steps` submissions of `count` board records with HIP events between the stages of a
 * submission on the workspace's stream, `in_flight` (1..8) distinct workspaces kept busy the way a leaf pool keeps batches alive.
 * stage_us[7]: mean microseconds per stage -- H2D of the records, planes kernel, tower, head convolution, policy GEMM, soft-max +
 * value, D2H of the outputs (a stage that waits for compute units held by another submission's tower shows as long);  host_us[4]:
 * the submit call, submit -> answers with the caller, the device's span first -> last stamp, and the difference of the last two. */
DGNN_TEST_API int dgnn_forward_stage_times(dgnn_workspace* const* workspaces, int in_flight, const uint8_t* records, int count, int steps,
                             float* stage_us /* 7 */, double* host_us /* 4 */);
/* The workspace's own HBM buffers: features [batch*11552] fp16, value [batch] fp16,
 * policy [batch*362] fp16.  Valid until the workspace is destroyed. */
DGNN_TEST_API int dgnn_workspace_device_buffers(dgnn_workspace* ws, void** d_features, void** d_value, void** d_policy);
/* dgnn_forward_pipelined_timed_n with two workspaces (rounds 2-4 called this form). */
DGNN_TEST_API int dgnn_forward_pipelined_timed(dgnn_workspace* a, dgnn_workspace* b, int form, const void* input, int count, int steps, double* seconds);

/* ---- f-1: symmetries, the transposition table's single operations, the raw batcher ------------------------------------------ */

/* Transform::inverse / apply (src/libdg_go/utils/symmetry.rs:78-90, 39-62; 361 = pass maps to itself; -1 for invalid arguments)
 * and Prediction::with_transform (src/libdg_mcts/predictor.rs:30-44): out[apply(t, i)] = policy[i]; out != policy. */
DGNN_TEST_API int dgnn_symmetry_inverse(int transform);
DGNN_TEST_API int dgnn_symmetry_apply(int transform, int packed_index);
DGNN_TEST_API int dgnn_policy_with_transform(const uint16_t* policy, int transform, uint16_t* out);
/* The table's generation (dgnn_cache_clear starts a new one) and an insert that drops an answer whose miss predates a clear
 * (generation = dgnn_cache_generation read before the fetch): what the library's search driver does inside. */
DGNN_TEST_API uint64_t dgnn_cache_generation(const dgnn_cache* cache);
DGNN_TEST_API int dgnn_cache_insert_at(dgnn_cache* cache, uint64_t generation, uint64_t board_hash, int to_move, int symmetry, uint16_t value,
                         const uint16_t* policy);
/* Batcher / Batch (src/libdg_mcts/pool/batch.rs:24-123), the queue a leaf pool owns.  Leaf positions are pushed with a caller-chosen
 * 64-bit tag; get_batch hands out the NEWEST max_batch_size of them once at least `min_batch_size` are queued and fewer than
 * `max_batches` batches are alive (the reference: 2 x devices, nn.rs:64-67).  `net` may be NULL for a batcher that only queues.
 * Batch::forward (batch.rs:35-40) evaluates through a pooled workspace of capacity max_batch_size.  dgnn_pool_create puts worker
 * threads over a batcher the caller made (the pool holds a reference: the two may be destroyed in either order). */
DGNN_TEST_API int dgnn_batcher_create(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out);
DGNN_TEST_API int dgnn_batcher_create_boards(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out);
DGNN_TEST_API void dgnn_batcher_destroy(dgnn_batcher* batcher);
DGNN_TEST_API int dgnn_batcher_push(dgnn_batcher* batcher, const uint16_t* features /* 11552 fp16 */, uint64_t tag);       /* batch.rs:80-84 */
DGNN_TEST_API int dgnn_batcher_push_board(dgnn_batcher* batcher, const uint8_t* record /* DGNN_BOARD_RECORD_BYTES */, uint64_t tag);
DGNN_TEST_API int dgnn_batcher_get_batch(dgnn_batcher* batcher, int min_batch_size, dgnn_batch** out /* NULL: none */);    /* batch.rs:91-121 */
DGNN_TEST_API int dgnn_batcher_push_and_get_batch(dgnn_batcher* batcher, const uint16_t* features, uint64_t tag, dgnn_batch** out);  /* :86-89 */
DGNN_TEST_API int dgnn_batcher_pending(dgnn_batcher* batcher, int* queued, int* batches_alive);
DGNN_TEST_API int dgnn_batcher_stats(dgnn_batcher* batcher, uint64_t* boards_evaluated, uint64_t* batches_evaluated);
DGNN_TEST_API int dgnn_batch_size(const dgnn_batch* batch);
DGNN_TEST_API int dgnn_batch_tags(const dgnn_batch* batch, uint64_t* tags_out /* dgnn_batch_size entries, in push order */);
DGNN_TEST_API const uint16_t* dgnn_batch_features(const dgnn_batch* batch);
DGNN_TEST_API const uint8_t* dgnn_batch_records(const dgnn_batch* batch);       /* NULL for a batch of fp16 planes (and dgnn_batch_features vice versa) */
DGNN_TEST_API int dgnn_batch_forward(dgnn_batch* batch, uint16_t* value_out, uint16_t* policy_out);
DGNN_TEST_API int dgnn_batch_forward_on(dgnn_batch* batch, int device, uint16_t* value_out, uint16_t* policy_out);
DGNN_TEST_API void dgnn_batch_destroy(dgnn_batch* batch);
DGNN_TEST_API int dgnn_pool_create(dgnn_batcher* batcher, int num_workers, dgnn_pool** out);
/* A records pool over copies of an fp16-PLANES predictor: its lanes expand the records with the host statement of the device kernel
 * (dgnn_board_records_to_packed) -- tests with the CPU oracle as the predictor. */
DGNN_TEST_API int dgnn_pool_create_callbacks_boards(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                      int workers_per_lane, dgnn_pool** out);
/* Fault injection: the n-th board pushed into any batcher's queue from now on (n = 0: the next one) is refused with
 * DGNN_OUT_OF_MEMORY, as if the queue's allocation had failed; n < 0 switches the hook off (its state at load). */
DGNN_TEST_API void dgnn_test_fail_push_after(int n);
/* Batching rules that were measured against the reference's constant and LOST (kept for the record and their tests):
 * a minimum fill per batch, 2 or 3 batches alive by the unique positions per second the pool measures, a third batch only while a
 * full batch of boards is waiting. */
DGNN_TEST_API int dgnn_pool_set_min_fill(dgnn_pool* pool, int min_boards, int max_wait_us);
DGNN_TEST_API int dgnn_pool_set_adaptive_batches(dgnn_pool* pool, int lo, int hi);
DGNN_TEST_API int dgnn_pool_set_extra_batch_queue(dgnn_pool* pool, int lo, int hi, int min_queued);
DGNN_TEST_API int dgnn_pool_adaptive_stats(dgnn_pool* pool, int* alive_per_lane, uint64_t* switches, uint64_t* boards_hashed, uint64_t* boards_repeated,
                             double* unique_rate_lo, double* unique_rate_hi);

/* ---- f-2: what the reference's tree tests look at (src/libdg_mcts/tree.rs:1773-1946, config.rs:352-375, choose.rs:108-140) ------ */

/* config::get_uct_exp / get_fpu_reduce / get_lcb_critical_value / VLOSS_CNT and the interpolation they use
 * (src/libdg_utils/config.rs:181-195, 289-330); asm::argmax_f32 (src/libdg_mcts/asm/argmax.rs:24-78; n a multiple of 8; -1 when
 * the maximum is not finite). */
DGNN_TEST_API float dgnn_config_uct_exp(int visits);
DGNN_TEST_API float dgnn_config_fpu_reduce(int visits);
DGNN_TEST_API float dgnn_config_lcb_critical_value(int visits);
DGNN_TEST_API int dgnn_config_vloss_cnt(void);
DGNN_TEST_API float dgnn_intp_value(const int* xs, const float* ys, int n, int x);
DGNN_TEST_API int dgnn_argmax_f32(const float* values, int n);
/* NodeTrace entries (tree.rs:1387), undo (tree.rs:1397-1409), Child accessors (tree.rs:326-358), the prior, min_promote_rollouts
 * (time_control/mod.rs:48-73), time_control::is_done for a RolloutLimit (mod.rs:86-101, rollout_limit.rs:35-44), disqualify. */
DGNN_TEST_API int dgnn_trace_len(const dgnn_trace* trace);
DGNN_TEST_API int dgnn_trace_get(const dgnn_trace* trace, int i, dgnn_tree_node** node, int* color, int* index);
DGNN_TEST_API int dgnn_tree_undo(dgnn_trace* trace, int undo_expanding);
DGNN_TEST_API int dgnn_tree_node_child(const dgnn_tree_node* node, int index, int* count, int* vcount, float* value, float* value_std, int* has_child);
DGNN_TEST_API int dgnn_tree_node_prior(const dgnn_tree_node* node, float* prior_out /* 362 */);
DGNN_TEST_API int dgnn_tree_node_min_promote(const dgnn_tree_node* node);
DGNN_TEST_API int dgnn_rollout_limit_is_done(const dgnn_tree_node* node, int limit);
DGNN_TEST_API int dgnn_tree_node_disqualify(dgnn_tree_node* node, int index);                              /* tree.rs:1315-1320 */
/* choose (src/libdg_mcts/choose.rs:54-88) on integer and on real weights; dirichlet::add_ex (dirichlet.rs:40-70) with a seeded generator. */
DGNN_TEST_API int dgnn_choose(const int* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out);
DGNN_TEST_API int dgnn_choose_f64(const double* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out);
DGNN_TEST_API void dgnn_dirichlet_add(float* x, int n, float shape, float beta, uint64_t seed);
/* The search driver's older forms (one leaf per worker; with the table in front of the predictor), full_forward (lib.rs:83-130: the
 * eight-symmetry root evaluation) on its own, and a rule-free synthetic game (NOT Go) with a self-play loop over it. */
DGNN_TEST_API int dgnn_search_run(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                    int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out, uint64_t* conflicts_out);
DGNN_TEST_API int dgnn_search_run_cached(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out,
                           uint64_t* conflicts_out, uint64_t* cache_hits_out);
DGNN_TEST_API int dgnn_full_forward(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_fn predict, void* predict_ctx,
                      dgnn_cache* cache, float* value_out, float* policy_out);
DGNN_TEST_API int dgnn_full_forward_boards(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_boards_fn predict, void* predict_ctx,
                             dgnn_cache* cache, float* value_out, float* policy_out);
DGNN_TEST_API const dgnn_game_ops* dgnn_synthetic_game_ops(void);
DGNN_TEST_API void* dgnn_synthetic_board_new(uint64_t seed);
DGNN_TEST_API void dgnn_synthetic_board_free(void* board);
DGNN_TEST_API int dgnn_selfplay_synthetic(dgnn_predict_fn predict, void* predict_ctx, int num_games, int num_moves, int num_rollouts,
                            int threads_per_search, uint64_t seed, dgnn_selfplay_stats* stats);

/* ---- f-3: what the reference's board, ladder, Benson and score tests look at (src/libdg_go) ----------------------------------- */

DGNN_TEST_API int dgnn_go_count(const dgnn_go_board* board);
DGNN_TEST_API int dgnn_go_at(const dgnn_go_board* board, int x, int y);                    /* Board::at: -1 empty, else the colour */
DGNN_TEST_API int dgnn_go_to_move(const dgnn_go_board* board);                             /* board.rs:100-105 */
DGNN_TEST_API uint64_t dgnn_go_hash(const dgnn_go_board* board);                           /* position hash (own zobrist numbers) */
DGNN_TEST_API int dgnn_go_n_liberty(const dgnn_go_board* board, int x, int y);             /* BoardFast::get_n_liberty; -1 on an empty point */
DGNN_TEST_API int dgnn_go_n_liberty_if(const dgnn_go_board* board, int color, int x, int y);   /* BoardFast::get_n_liberty_if; -1 if not valid */
DGNN_TEST_API int dgnn_go_is_ladder_capture(const dgnn_go_board* board, int color, int x, int y);   /* utils/ladder.rs:124-136; -1 if not valid */
DGNN_TEST_API int dgnn_go_is_ladder_escape(const dgnn_go_board* board, int color, int x, int y);    /* utils/ladder.rs:138-170 */
DGNN_TEST_API int dgnn_go_is_symmetric(const dgnn_go_board* board, int symmetry);          /* symmetry::is_symmetric */
/* Benson's unconditional life as the reference computes it (utils/benson.rs:62-190): per point 0 = neither, 1 = stone of an
 * unconditionally alive block of `color`, 2 = point of a region vital to one; _flood: the same by the reference's own steps, the
 * checker of the bit-board form.  candidates: the moves a search may try (pool/policy_helper.rs:36-48; scoring != 0: the
 * ScoringSearch rule, options.rs:110-214).  check: 1 when the incremental bookkeeping equals a from-scratch flood fill. */
DGNN_TEST_API int dgnn_go_benson(const dgnn_go_board* board, int color, unsigned char* status_out /* 361 */);
DGNN_TEST_API int dgnn_go_benson_flood(const dgnn_go_board* board, int color, unsigned char* status_out /* 361 */);
DGNN_TEST_API int dgnn_go_candidates(const dgnn_go_board* board, int to_move, int scoring, unsigned char* out);
DGNN_TEST_API int dgnn_go_check(const dgnn_go_board* board);
/* Score::is_scorable / get_score / get_stone_status (utils/score.rs:105-239; the DGNN_STONE_* / DGNN_TERRITORY_* bits of dgnn_host.h)
 * and the RE[] / TB[] TW[] text of GameResult::Ended (src/libdg_mcts/game_result.rs:35-93). */
DGNN_TEST_API int dgnn_go_is_scorable(const dgnn_go_board* board);                             /* Score::is_scorable, utils/score.rs:105-117 */
DGNN_TEST_API int dgnn_go_score(const dgnn_go_board* board, int* black_out, int* white_out);   /* Score::get_score (Tromp-Taylor, no komi), :130-136, 220-239 */
DGNN_TEST_API int dgnn_go_stone_status(const dgnn_go_board* board, unsigned char* status_out /* 361 */);
DGNN_TEST_API int dgnn_go_result_sgf(const dgnn_go_board* board, char* winner_out, size_t winner_capacity, char* territory_out, size_t territory_capacity);

#ifdef __cplusplus
}
#endif

#endif /* DGNN_TEST_H */
