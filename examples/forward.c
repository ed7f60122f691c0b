/* Minimal C host of libdgnn_hip.so: what a non-Python embedder (or the Rust binding of INTEGRATION.md) does.
 *
 *   gcc -std=c99 -Iinclude examples/forward.c -Ldream-go_amd/lib -ldgnn_hip -Wl,-rpath,$PWD/dream-go_amd/lib -o forward
 *   ./forward tests/golden/net_2x32_s2.json 4
 *
 * Loads a weights file in the reference's JSON / base85 format, evaluates `batch` all-zero boards through
 * dg_nn::forward's replacement and prints the value and the three largest policy entries of board 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dgnn.h"

static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    float f;
    if (exp == 0) {
        if (man == 0) { bits = sign; }
        else { exp = 113; while (!(man & 0x400u)) { man <<= 1; --exp; } bits = sign | (exp << 23) | ((man & 0x3ffu) << 13); }
    } else if (exp == 31) { bits = sign | 0x7f800000u | (man << 13); }
    else { bits = sign | ((exp + 112) << 23) | (man << 13); }
    memcpy(&f, &bits, sizeof f);
    return f;
}

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "dream_go.json";
    const int batch = argc > 2 ? atoi(argv[2]) : 1;
    dgnn_network* net = NULL;
    dgnn_workspace* ws = NULL;
    int status, blocks, channels, samples, i, k;

    if ((status = dgnn_network_load(path, &net)) != DGNN_OK) {
        fprintf(stderr, "%s: %s\n", path, dgnn_status_string(status));
        return 1;
    }
    dgnn_network_info(net, &blocks, &channels, &samples);
    printf("%s: %d residual blocks, %d channels, %d policy planes (%s)\n", path, blocks, channels, samples, dgnn_version());

    if ((status = dgnn_network_get_workspace(net, 0, batch, &ws)) != DGNN_OK) {
        fprintf(stderr, "workspace: %s (hip %d)\n", dgnn_status_string(status), dgnn_last_hip_error());
        return 1;
    }
    {
        uint16_t* features = (uint16_t*)calloc((size_t)batch * DGNN_FEATURE_SIZE, sizeof(uint16_t));
        uint16_t* value = (uint16_t*)malloc((size_t)batch * sizeof(uint16_t));
        uint16_t* policy = (uint16_t*)malloc((size_t)batch * DGNN_NUM_MOVES * sizeof(uint16_t));
        status = dgnn_forward(ws, features, (size_t)batch * DGNN_FEATURE_SIZE, value, policy);
        if (status != DGNN_OK) {
            fprintf(stderr, "forward: %s (hip %d)\n", dgnn_status_string(status), dgnn_last_hip_error());
            return 1;
        }
        printf("kernel %s; value[0] = %.4f; top policy entries of board 0:", dgnn_workspace_conv_kernel(ws), half_to_float(value[0]));
        for (k = 0; k < 3; ++k) {
            int best = 0;
            for (i = 1; i < DGNN_NUM_MOVES; ++i) if (policy[i] > policy[best]) best = i;   /* non-negative fp16: bit order = value order */
            printf(" %d:%.4f", best, half_to_float(policy[best]));
            policy[best] = 0;
        }
        printf("\n");
        free(features); free(value); free(policy);
    }
    dgnn_workspace_release(ws);
    dgnn_network_destroy(net);
    return 0;
}
