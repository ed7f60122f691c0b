/* Self-play from plain C: board, search, pool and engine behind one call.
 *
 *   gcc -std=c99 -Iinclude examples/selfplay.c -Ldream-go_amd/lib -ldgnn_hip -Wl,-rpath,$PWD/dream-go_amd/lib -o selfplay
 *   ./selfplay                          (no weights: the deterministic CPU stand-in predictor, no GPU needed)
 *   ./selfplay dream_go.json 64 800     (weights file, games, rollouts: HIP engine behind the leaf pool)
 *
 * Prints one GameResult::Ended record per game (src/libdg_mcts/game_result.rs:35-41), the format dream-go's trainer reads, as each
 * game ends -- the reference's reader of self_play()'s channel (src/libdg_mcts/self_play.rs:471-497) -- and keeps none of them. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dgnn_host.h"

static void print_record(void* ctx, int game_id, const char* record, size_t bytes, int whole)
{
    (void)game_id;
    (void)whole;
    fwrite(record, 1, bytes, (FILE*)ctx);       /* (one call at a time: the library serialises them) */
    fputc('\n', (FILE*)ctx);
}

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : NULL;
    dgnn_selfplay_params params;
    dgnn_selfplay_stats stats;
    dgnn_network* net = NULL;
    dgnn_pool* pool = NULL;
    dgnn_predict_fn predict = dgnn_hash_predictor();
    void* predict_ctx = NULL;
    int status;

    memset(&params, 0, sizeof params);          /* fields added later (host_threads, opening) default to "off" */
    params.num_games = argc > 2 ? atoi(argv[2]) : 2;
    params.num_parallel = 0;
    params.first_game = 0;
    params.game_stride = 1;
    params.max_moves = path ? 0 : 30;          /* whole games on the engine, a short demonstration without it */
    params.num_rollouts = argc > 3 ? atoi(argv[3]) : 40;
    params.num_ex_it_rollouts = 0;
    params.threads_per_search = 1;             /* one thread with several leaves in flight: reproducible for a fixed seed */
    params.leaves_per_thread = 8;
    params.cache_entries_per_game = 20001;     /* a table per game, not a shared one: see dgnn_selfplay_go in dgnn_host.h */
    params.seed = 2024;
    params.on_game = print_record;             /* a job of any length holds no record longer than it takes to print it */
    params.on_game_ctx = stdout;

    if (path) {
        if ((status = dgnn_network_load(path, &net)) != DGNN_OK) { fprintf(stderr, "%s: %s\n", path, dgnn_status_string(status)); return 1; }
        if ((status = dgnn_pool_create_devices(net, NULL, 0, 256, 2, 2, &pool)) != DGNN_OK) {      /* every visible GPU, two batches alive on each */
            fprintf(stderr, "pool: %s (hip %d)\n", dgnn_status_string(status), dgnn_last_hip_error());
            return 1;
        }
        predict = dgnn_pool_predictor();
        predict_ctx = pool;
    }
    status = dgnn_selfplay_go(predict, predict_ctx, NULL, &params, &stats, NULL, NULL, 0);
    if (status != DGNN_OK) { fprintf(stderr, "self-play: %s\n", dgnn_status_string(status)); return 1; }
    fprintf(stderr, "%llu games, %llu stones, %llu evaluations (%llu from the cache) in %.2f s\n", (unsigned long long)stats.games,
            (unsigned long long)stats.moves, (unsigned long long)stats.evals, (unsigned long long)stats.cache_hits, stats.seconds);

    if (pool) dgnn_pool_destroy(pool);
    if (net) dgnn_network_destroy(net);
    return 0;
}
