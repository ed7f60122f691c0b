// Host-side cost per leaf of the rows above the engine (board, feature planes, candidates, Benson), timed through the C ABI
// on random mid-game positions.  Build + run:  tools/host_bench.sh
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dgnn_test.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const dgnn_game_ops* ops = dgnn_go_game_ops();
    const dgnn_game_ops* scoring = dgnn_go_scoring_ops();
    std::vector<uint16_t> features(DGNN_FEATURE_SIZE);
    unsigned char candidates[362], status[361];
    uint64_t rng = 12345;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng >> 33); };
    std::printf("%8s %10s %10s %10s %10s %10s %10s %10s\n", "stones", "features", "clone+free", "place", "candidates", "cand.score", "benson", "scorable");
    for (int stones : {20, 80, 150, 220, 300}) {
        const int kBoards = 32;
        std::vector<dgnn_go_board*> boards;
        std::vector<int> to_move;
        for (int i = 0; i < kBoards; ++i) {
            dgnn_go_board* b = dgnn_go_board_new(7.5f);
            int color = 0, placed = 0, tries = 0;
            while (placed < stones && tries < 100000) {
                ++tries;
                const int p = (int)(next() % 361);
                if (dgnn_go_is_valid(b, color, p % 19, p / 19)) { dgnn_go_place(b, color, p % 19, p / 19); color ^= 1; ++placed; }
            }
            boards.push_back(b);
            to_move.push_back(color);
        }
        const int reps = 20;
        double t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) ops->features(boards[i], to_move[i], r % 8, features.data());
        const double t_features = (now() - t0) / (reps * kBoards);
        t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) ops->destroy(ops->clone(boards[i]));
        const double t_clone = (now() - t0) / (reps * kBoards);
        t0 = now();
        int n_place = 0;
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) {
            void* c = ops->clone(boards[i]);
            for (int k = 0; k < 8; ++k) { const int p = (int)(next() % 361); if (dgnn_go_is_valid((dgnn_go_board*)c, to_move[i], p % 19, p / 19)) { ops->place(c, to_move[i], p); ++n_place; break; } }
            ops->destroy(c);
        }
        const double t_place = (now() - t0) / (reps * kBoards) - t_clone;
        t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) ops->candidates(boards[i], to_move[i], candidates);
        const double t_cand = (now() - t0) / (reps * kBoards);
        t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) scoring->candidates(boards[i], to_move[i], candidates);
        const double t_cand_s = (now() - t0) / (reps * kBoards);
        t0 = now();
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) dgnn_go_benson(boards[i], r & 1, status);
        const double t_benson = (now() - t0) / (reps * kBoards);
        t0 = now();
        int n_scorable = 0;
        for (int r = 0; r < reps; ++r) for (int i = 0; i < kBoards; ++i) n_scorable += dgnn_go_is_scorable(boards[i]);
        const double t_scorable = (now() - t0) / (reps * kBoards);
        std::printf("%8d %9.1fus %9.1fus %9.1fus %9.1fus %9.1fus %9.1fus %9.1fus\n", stones, t_features * 1e6, t_clone * 1e6, t_place * 1e6,
                    t_cand * 1e6, t_cand_s * 1e6, t_benson * 1e6, t_scorable * 1e6);
        (void)n_place; (void)n_scorable;
        for (dgnn_go_board* b : boards) dgnn_go_board_free(b);
    }
    return 0;
}
