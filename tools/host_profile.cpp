// A sampling profile of the HOST rows (board, search, tree, table, pool) without a GPU and without perf, which the image lacks:
// the host sources linked with the sanitizer script's CPU stand-in for the engine (tools/sanitize_host.sh), whole self-play
// jobs under a SIGPROF timer whose handler records the interrupted program counter; tools/host_profile.sh builds, runs and turns
// pcs.txt / maps.txt into a table per function (addr2line for the binary, the nearest exported symbol for libc).
//   host_profile [games] [stones]                   one game after the other, the stand-in called directly (warm caches)
//   NOCACHE=1 host_profile ...                      without the transposition table: the miss path alone
//   POOL=256 host_profile 256 12                    256 games in flight as fibers on ONE host thread, through the leaf pool over a
//                                                   stand-in device (500 + 10.8 n us per batch): the memory behaviour of production
#include <chrono>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <sys/resource.h>
#include <sys/time.h>
#include <ucontext.h>
#include <vector>
#include "dgnn_test.h"
static uint64_t* g_pcs; static volatile size_t g_n; static const size_t kMax = 1 << 20;
static void on_prof(int, siginfo_t*, void* ctx) { if (g_n < kMax) g_pcs[g_n++] = ((ucontext_t*)ctx)->uc_mcontext.gregs[REG_RIP]; }
int main(int argc, char** argv) {
    g_pcs = (uint64_t*)malloc(kMax * 8);
    struct sigaction sa{}; sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction(SIGPROF, &sa, nullptr);
    itimerval it{{0, 500}, {0, 500}}; setitimer(ITIMER_PROF, &it, nullptr);
    dgnn_selfplay_params p{};
    p.num_games = argc > 1 ? atoi(argv[1]) : 8; p.num_parallel = 1; p.first_game = 0; p.game_stride = 1; p.max_moves = argc > 2 ? atoi(argv[2]) : 80;
    p.num_rollouts = 800; p.threads_per_search = 1; p.leaves_per_thread = 16; p.cache_entries_per_game = 0; p.seed = 11;
    dgnn_cache* cache = nullptr; if (!getenv("NOCACHE")) dgnn_cache_create(200001, 8, &cache);
    dgnn_selfplay_stats st;
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (getenv("POOL")) {
        static dgnn_dry_device dev{500.0, 10.8};
        dgnn_pool* pool = nullptr;
        dgnn_pool_create_board_callbacks(dgnn_hash_boards_predictor(), &dev, 1, 256, 2, 2, &pool);
        dgnn_pool_set_coalescing(pool, 1);
        p.num_parallel = atoi(getenv("POOL")); p.host_threads = 1; p.opening_len = 0;
        rc = dgnn_selfplay_go_boards(dgnn_pool_boards_predictor(), pool, cache, &p, &st, nullptr, nullptr, 0);
    } else
    rc = dgnn_selfplay_go_boards(dgnn_hash_boards_predictor(), nullptr, cache, &p, &st, nullptr, nullptr, 0);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    it = itimerval{}; setitimer(ITIMER_PROF, &it, nullptr);
    rusage ru{}; getrusage(RUSAGE_SELF, &ru);
    const double cpu = ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec + ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec;
    std::printf("CPU %.2f us per rollout (user + system, every thread); ", 1e6 * cpu / (double)(st.evals + st.cache_hits));
    std::printf("rc %d games %llu moves %llu evals %llu hits %llu: %.2f s, %.2f us per rollout\n", rc, (unsigned long long)st.games, (unsigned long long)st.moves,
                (unsigned long long)st.evals, (unsigned long long)st.cache_hits, sec, 1e6 * sec / (double)(st.evals + st.cache_hits));
    { FILE* m = fopen("/proc/self/maps", "r"); FILE* o = fopen("maps.txt", "w"); char line[512]; while (fgets(line, sizeof line, m)) fputs(line, o); fclose(m); fclose(o); }
    FILE* f = fopen("pcs.txt", "w"); for (size_t i = 0; i < g_n; ++i) fprintf(f, "%llx\n", (unsigned long long)g_pcs[i]); fclose(f);
    return 0;
}
