/* Board records instead of feature planes (include/dgnn.h: dgnn_forward_boards), from strict C99: what a host that keeps its own
 * board (the reference's libdg_go) hands to the engine -- 896 bytes per position instead of 23 104, the 32 planes
 * are built on the device.
 *
 *   gcc -std=c99 -Iinclude examples/forward_boards.c -Ldream-go_amd/lib -ldgnn_hip -Wl,-rpath,$PWD/dream-go_amd/lib -o forward_boards
 *   ./forward_boards [weights.json]
 *
 * Without a GPU (or a weights file) it still shows the host half: the record of a small position, expanded by the HOST statement of
 * the device kernel (dgnn_board_records_to_packed), equals the planes features::V1 gives for it (dgnn_go_features). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dgnn_host.h"

int main(int argc, char** argv)
{
    static uint8_t record[DGNN_BOARD_RECORD_BYTES];
    static uint32_t from_record[DGNN_PACKED_WORDS], from_planes[DGNN_PACKED_WORDS];
    static uint16_t planes[DGNN_FEATURE_SIZE], value[1], policy[DGNN_NUM_MOVES];
    const int stones[][3] = {{0, 3, 3}, {1, 15, 15}, {0, 2, 15}, {1, 16, 3}, {0, 15, 16}, {1, 3, 2}, {0, 2, 2}};
    dgnn_go_board* board = dgnn_go_board_new(7.5f);
    dgnn_network* net = NULL;
    dgnn_workspace* ws = NULL;
    size_t i;
    int status, set = 0;

    if (!board) return 2;
    for (i = 0; i < sizeof stones / sizeof stones[0]; ++i) dgnn_go_place(board, stones[i][0], stones[i][1], stones[i][2]);

    /* white to move, seen through symmetry 5 (rot 90) */
    if (dgnn_go_board_record(board, DGNN_COLOR_WHITE, DGNN_SYM_ROT90, record) != DGNN_OK) return 2;
    if (dgnn_board_records_to_packed(record, 1, from_record) != DGNN_OK) return 2;
    if (dgnn_go_features(board, DGNN_COLOR_WHITE, DGNN_SYM_ROT90, planes) != DGNN_OK) return 2;
    if (dgnn_features_pack(planes, 1, from_planes) != DGNN_OK) return 2;
    for (i = 0; i < DGNN_NUM_POINTS; ++i) set += from_record[i] != 0;
    printf("record: %d bytes; planes from the record %s the planes of features::V1 (%d points with a plane set)\n",
           DGNN_BOARD_RECORD_BYTES, memcmp(from_record, from_planes, sizeof from_record) == 0 ? "equal" : "DIFFER FROM", set);

    status = dgnn_network_load(argc > 1 ? argv[1] : "dream_go.json", &net);
    if (status != DGNN_OK) {
        fprintf(stderr, "no network (%s): the device half is skipped\n", dgnn_status_string(status));
        dgnn_go_board_free(board);
        return memcmp(from_record, from_planes, sizeof from_record) == 0 ? 0 : 1;
    }
    if ((status = dgnn_network_get_workspace(net, 0, 1, &ws)) != DGNN_OK ||
        (status = dgnn_forward_boards(ws, record, 1, value, policy)) != DGNN_OK) {
        fprintf(stderr, "forward_boards: %s (hipError_t %d)\n", dgnn_status_string(status), dgnn_last_hip_error());
        return 1;
    }
    printf("value bits 0x%04x, pass probability bits 0x%04x\n", value[0], policy[DGNN_NUM_POINTS]);
    dgnn_workspace_release(ws);
    dgnn_network_destroy(net);
    dgnn_go_board_free(board);
    return 0;
}
