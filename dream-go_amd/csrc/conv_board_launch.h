// conv3x3_board_kernel (one launch per layer of the board kernel, conv_board_body.h) and the launcher of one work split.  Included by
// conv_board.hip (instantiates the splits <16> and <1>) and conv_board_b.hip (<4> and <2>): the 24 instantiations of the body are
// the longest compile of the library, and two translation units halve it.
#pragma once
#include "conv_board_body.h"

namespace dgnn {

template <int NS, int SKIP, bool RELU, int ABL>
__global__ __launch_bounds__(Split<NS>::Threads)
void conv3x3_board_kernel(ConvBoardLaunch a)
{
    __shared__ __attribute__((aligned(16))) char lds[Split<NS>::LdsBytes];
    half8 af[2][Split<NS>::CT][2];
    conv_board_body<NS, SKIP, RELU, ABL>(a, lds, blockIdx.x, blockIdx.y, af);
}


template <int NS>
void launch_board_variant(const ConvBoardLaunch& args, int skip, bool relu, int ablate, hipStream_t stream)
{
    const dim3 grid(args.batch, args.cout / (32 * Split<NS>::SL));
#define DGNN_BOARD(S, R, A) hipLaunchKernelGGL((conv3x3_board_kernel<NS, S, R, A>), grid, dim3(Split<NS>::Threads), 0, stream, args)
    // The ablated kernels (tools/ablate.sh, tools/ablate2.sh: timing only, garbage results) are 34 more instantiations of the
    // body -- half the compile time of this file.  They are built only with -DDGNN_ABLATIONS (DGNN_ABLATIONS=1 python
    // dream-go_amd/build.py); a library without them ignores DGNN_CONV_ABLATE.
#ifdef DGNN_ABLATIONS
    if constexpr (NS == 1) {
        if (ablate == 1 && relu && skip != 1) {      // timing only: the chunked kernel without its MFMAs
            if (skip == 2) DGNN_BOARD(2, true, 1); else DGNN_BOARD(0, true, 1);
            return;
        }
    }
    if constexpr (NS == 4) {
        // measurement only (tools/ablate.sh, tools/ablate2.sh; garbage results), residual shapes of the four-slice split:
        // 1 = no MFMAs, 3 = neither MFMAs nor in-loop memory instructions (filter loads, activation pieces), 16 = no barrier at the
        // chunk boundaries.  (Round 2's 2 / 8 / 24 / 32 / 64 priced the wave-private filter ring, which this split no longer has.)
        if ((ablate == 1 || ablate == 3 || ablate == 16) && relu && skip != 1) {
            if (skip == 2) {
                switch (ablate) { case 1: DGNN_BOARD(2, true, 1); break; case 3: DGNN_BOARD(2, true, 3); break; default: DGNN_BOARD(2, true, 16); break; }
            } else {
                switch (ablate) { case 1: DGNN_BOARD(0, true, 1); break; case 3: DGNN_BOARD(0, true, 3); break; default: DGNN_BOARD(0, true, 16); break; }
            }
            return;
        }
    }
#else
    (void)ablate;
#endif
    if (skip == 2 && relu) DGNN_BOARD(2, true, 0);
    else if (skip == 2) DGNN_BOARD(2, false, 0);
    else if (skip == 1 && relu) DGNN_BOARD(1, true, 0);
    else if (skip == 1) DGNN_BOARD(1, false, 0);
    else if (relu) DGNN_BOARD(0, true, 0);
    else DGNN_BOARD(0, false, 0);
#undef DGNN_BOARD
}


}  // namespace dgnn
