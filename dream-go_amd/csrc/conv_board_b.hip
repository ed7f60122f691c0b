// The four-slice and two-slice work splits of conv3x3_board_kernel (conv_board_launch.h): a translation unit of their own, so that the
// longest compile of the library runs as two halves side by side.
#include "conv_board_launch.h"

namespace dgnn {

template void launch_board_variant<4>(const ConvBoardLaunch&, int, bool, int, hipStream_t);
template void launch_board_variant<2>(const ConvBoardLaunch&, int, bool, int, hipStream_t);

}  // namespace dgnn
