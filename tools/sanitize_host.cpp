// Driver for tools/sanitize_host.sh: whole self-play jobs on the CPU stand-in predictor (whole games with per-game tables and
// batched leaves; expert iteration with four racing search threads and a shared table; games started as others end; the
// synthetic game with eight threads per search), so that AddressSanitizer / UBSan / ThreadSanitizer see the host rows at work.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "dgnn_test.h"
int main() {
    for (int mode = 0; mode < 3; ++mode) {
        dgnn_selfplay_params p{};
        p.num_games = 6; p.num_parallel = mode == 2 ? 2 : 0; p.first_game = 0; p.game_stride = 1; p.max_moves = mode == 0 ? 0 : 60;
        p.num_rollouts = mode == 0 ? 24 : 40; p.num_ex_it_rollouts = mode == 1 ? 30 : 0; p.threads_per_search = mode == 0 ? 1 : 4;
        p.leaves_per_thread = mode == 0 ? 6 : 2; p.cache_entries_per_game = mode == 0 ? 3000 : 0; p.seed = 5 + mode;
        dgnn_cache* cache = nullptr;
        if (mode != 0) dgnn_cache_create(20000, 4, &cache);
        dgnn_selfplay_stats st;
        std::vector<int> moves((size_t)6 * 722, -1);
        std::vector<char> sgf((size_t)6 * (722 * 2600 + 40000));
        int rc = dgnn_selfplay_go(dgnn_hash_predictor(), nullptr, cache, &p, &st, moves.data(), sgf.data(), sgf.size());
        std::printf("mode %d rc %d games %llu moves %llu evals %llu hits %llu conflicts %llu ended %llu exit %llu\n", mode, rc,
                    (unsigned long long)st.games, (unsigned long long)st.moves, (unsigned long long)st.evals, (unsigned long long)st.cache_hits,
                    (unsigned long long)st.conflicts, (unsigned long long)st.games_ended, (unsigned long long)st.ex_it_searches);
        if (cache) dgnn_cache_destroy(cache);
    }
    {   // through the leaf pool: batcher, worker threads, wake-ups (the engine behind it is the script's CPU stand-in)
        dgnn_batcher* batcher = nullptr;
        dgnn_pool* pool = nullptr;
        dgnn_cache* cache = nullptr;
        dgnn_batcher_create(reinterpret_cast<dgnn_network*>(0x10), 0, 32, 3, &batcher);
        dgnn_pool_create(batcher, 3, &pool);
        dgnn_cache_create(20000, 4, &cache);
        dgnn_selfplay_params p{};
        p.num_games = 12; p.num_parallel = 8; p.game_stride = 1; p.max_moves = 40; p.num_rollouts = 48; p.threads_per_search = 3;
        p.leaves_per_thread = 3; p.seed = 77;
        dgnn_selfplay_stats st;
        int rc = dgnn_selfplay_go(dgnn_pool_predictor(), pool, cache, &p, &st, nullptr, nullptr, 0);
        uint64_t boards = 0, batches = 0;
        dgnn_batcher_stats(batcher, &boards, &batches);
        std::printf("pool rc %d moves %llu evals %llu boards %llu batches %llu\n", rc, (unsigned long long)st.moves, (unsigned long long)st.evals,
                    (unsigned long long)boards, (unsigned long long)batches);
        dgnn_pool_destroy(pool);
        dgnn_batcher_destroy(batcher);
        dgnn_cache_destroy(cache);
    }
    {   // round 3: board records through a records pool (the dry device), pause / cache clear / resume while games run
        dgnn_pool* pool = nullptr;
        dgnn_cache* cache = nullptr;
        dgnn_dry_device device{100.0, 1.0};
        dgnn_pool_create_board_callbacks(dgnn_hash_boards_predictor(), &device, 1, 32, 2, 2, &pool);
        dgnn_cache_create(20000, 4, &cache);
        dgnn_selfplay_params p{};
        p.num_games = 16; p.num_parallel = 16; p.game_stride = 1; p.max_moves = 30; p.num_rollouts = 48; p.threads_per_search = 2;
        p.leaves_per_thread = 4; p.seed = 78; p.host_threads = 0;
        // the records leave over the channel (params.on_game): sixteen game threads, one call at a time, into plain unguarded state
        struct Taken { int records = 0; size_t bytes = 0; } taken;
        p.on_game = [](void* ctx, int, const char* record, size_t bytes, int) { Taken* t = static_cast<Taken*>(ctx); ++t->records; t->bytes += bytes + (size_t)(record[0] == '('); };
        p.on_game_ctx = &taken;
        dgnn_selfplay_stats st;
        std::thread refresher([&] {
            for (int i = 0; i < 5; ++i) {
                std::this_thread::sleep_for(std::chrono::milliseconds(30));
                dgnn_pool_pause(pool);
                dgnn_cache_clear(cache);
                dgnn_pool_resume(pool);
            }
        });
        int rc = dgnn_selfplay_go_boards(dgnn_pool_boards_predictor(), pool, cache, &p, &st, nullptr, nullptr, 0);
        refresher.join();
        std::printf("records pool rc %d moves %llu evals %llu hits %llu; %d records over the channel, %zu bytes\n", rc, (unsigned long long)st.moves,
                    (unsigned long long)st.evals, (unsigned long long)st.cache_hits, taken.records, taken.bytes);
        // ... and every record of a game against the host statement of the device kernel
        dgnn_go_board* board = dgnn_go_board_new(6.5f);
        std::vector<uint8_t> record(DGNN_BOARD_RECORD_BYTES);
        std::vector<uint32_t> packed(DGNN_PACKED_WORDS), expect(DGNN_PACKED_WORDS);
        std::vector<uint16_t> planes(DGNN_FEATURE_SIZE);
        unsigned rng = 12345, mismatches = 0;
        for (int m = 0, color = 0; m < 200; ++m) {
            for (int tries = 0; tries < 50; ++tries) {
                rng = rng * 1664525u + 1013904223u;
                const int pnt = (int)((rng >> 8) % 361);
                if (dgnn_go_is_valid(board, color, pnt % 19, pnt / 19)) { dgnn_go_place(board, color, pnt % 19, pnt / 19); break; }
            }
            color ^= 1;
            dgnn_go_board_record(board, color, m % 8, record.data());
            dgnn_board_records_to_packed(record.data(), 1, packed.data());
            dgnn_go_features(board, color, m % 8, planes.data());
            dgnn_features_pack(planes.data(), 1, expect.data());
            mismatches += packed != expect;
        }
        std::printf("records vs planes: %u mismatches\n", mismatches);
        dgnn_go_board_free(board);
        dgnn_pool_destroy(pool);
        dgnn_cache_destroy(cache);
    }
    {   // round 4: what SHIPS -- 256 games as fibers on two host threads (csrc/fiber.cpp: a hand-written stack switch, announced to
        // ASan with __sanitizer_start/finish_switch_fiber and to TSan with __tsan_create_fiber / __tsan_switch_to_fiber), board records
        // through a records pool whose workers make the fibers ready again, a refresh (pause / clear / resume) while they run
        dgnn_pool* pool = nullptr;
        dgnn_cache* cache = nullptr;
        dgnn_dry_device device{200.0, 2.0};
        dgnn_pool_create_board_callbacks(dgnn_hash_boards_predictor(), &device, 1, 256, 3, 3, &pool);
        dgnn_pool_set_coalescing(pool, 1);              // positions in flight are followed, not evaluated again (shared table below)
        dgnn_pool_set_adaptive_batches(pool, 2, 3);     // ... and 2 or 3 batches alive by what the pool measures
        dgnn_cache_create(200001, 8, &cache);
        dgnn_selfplay_params p{};
        p.num_games = 256; p.num_parallel = 256; p.game_stride = 1; p.max_moves = 4; p.num_rollouts = 64; p.threads_per_search = 1;
        p.leaves_per_thread = 8; p.seed = 79; p.host_threads = 2;
        dgnn_selfplay_stats st;
        std::thread refresher([&] {
            for (int i = 0; i < 3; ++i) {
                std::this_thread::sleep_for(std::chrono::milliseconds(200));
                dgnn_pool_pause(pool);
                dgnn_cache_clear(cache);
                dgnn_pool_resume(pool);
            }
        });
        int rc = dgnn_selfplay_go_boards(dgnn_pool_boards_predictor(), pool, cache, &p, &st, nullptr, nullptr, 0);
        refresher.join();
        std::printf("fibers: 256 games on 2 host threads rc %d moves %llu evals %llu hits %llu followed %llu\n", rc, (unsigned long long)st.moves,
                    (unsigned long long)st.evals, (unsigned long long)st.cache_hits, (unsigned long long)dgnn_pool_coalesced(pool));
        dgnn_pool_set_coalescing(pool, 0);              // (the private-table job below must stay a function of its seeds)
        dgnn_pool_set_adaptive_batches(pool, 0, 0);
        dgnn_pool_set_back_to_back(pool, 300, 16, 48);  // partial batches held until the running one is due, 17 .. 47 boards cut to 16
        // the same with a table per game (the reproducible mode) and games started as others end
        p.num_games = 48; p.num_parallel = 32; p.max_moves = 12; p.cache_entries_per_game = 2000; p.seed = 80; p.host_threads = 3;
        rc = dgnn_selfplay_go_boards(dgnn_pool_boards_predictor(), pool, nullptr, &p, &st, nullptr, nullptr, 0);
        std::printf("fibers: 48 games (32 in flight) on 3 host threads, private tables rc %d moves %llu evals %llu\n", rc,
                    (unsigned long long)st.moves, (unsigned long long)st.evals);
        int took[17];
        dgnn_pool_batch_times(pool, took);
        dgnn_pool_destroy(pool);
        dgnn_cache_destroy(cache);
    }
    {   // twelve threads probing ONE root while its child slots grow past the inline eight, the first chunk and the point
        // where the index -> slot map appears (tree.cpp: TreeNode::find / find_or_create) -- every probe's virtual loss sends the
        // next one to a new child of a flat prior
        std::vector<float> prior(362, 1.0f / 362.0f);
        for (int attempt = 0; attempt < 40; ++attempt) {
            dgnn_tree_node* root = dgnn_tree_node_new(0, 0.5f, prior.data());
            std::vector<std::thread> threads;
            for (int t = 0; t < 12; ++t)
                threads.emplace_back([&] {
                    for (int i = 0; i < 6; ++i) {
                        dgnn_trace* trace = nullptr;
                        if (dgnn_tree_probe(root, nullptr, nullptr, &trace) != DGNN_PROBE_FOUND) continue;
                        dgnn_tree_insert(trace, 1, 0.5f, prior.data());
                        dgnn_trace_free(trace);
                    }
                });
            for (std::thread& t : threads) t.join();
            int total = 0, vtotal = 0;
            dgnn_tree_node_info(root, nullptr, &total, &vtotal, nullptr, nullptr);
            if (vtotal != 0 || total <= 0) std::printf("one root, twelve threads: total %d vtotal %d\n", total, vtotal);
            dgnn_tree_node_free(root);
        }
        std::printf("one root probed by twelve threads across the slot-map boundary: 40 roots\n");
    }
    // synthetic self-play with many threads
    dgnn_selfplay_stats st;
    int rc = dgnn_selfplay_synthetic(dgnn_hash_predictor(), nullptr, 4, 4, 60, 8, 3, &st);
    std::printf("synthetic rc %d evals %llu\n", rc, (unsigned long long)st.evals);
    return 0;
}
