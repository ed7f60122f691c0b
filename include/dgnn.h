/*
 * dgnn.h -- C ABI of libdgnn_hip.so, the MI355X-native replacement for dream-go's
 * neural-network leaf evaluation (crate `dg_nn` + the parts of `dg_cuda` it uses).
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference checkout, kblomdahl/dream-go).  The reference-side binding a maintainer would
 * add (Rust `extern "C"` block + safe wrappers keeping `Network` / `WorkspaceGuard` /
 * `forward` / `OutputMap` source-compatible) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; fp16 values travel as their IEEE binary16 bit patterns (uint16_t),
 *     which is the in-memory representation of the reference's `f16` (repr(C) u16,
 *     src/libdg_utils/types/fp16.rs:16-20);
 *   - every function returns a status: 0 on success, negative on failure; nothing aborts or
 *     panics (the reference's `Result<_, Error>`, src/libdg_nn/error.rs:19-24);
 *   - no process-global "current device": the device is an argument, and a call may be made
 *     from any thread (the reference relies on thread-local cudaSetDevice,
 *     src/libdg_mcts/predictors/nn.rs:87-89);
 *   - a workspace is used by one thread at a time (`&mut Workspace`, graph.rs:123); distinct
 *     workspaces may run concurrently from distinct threads; the network is immutable and shared.
 */
#ifndef DGNN_H
#define DGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: src/libdg_nn/error.rs:19-24 flattened to integers ---------------------- */
#define DGNN_OK                  0
#define DGNN_MISSING_WEIGHTS    -1  /* Error::MissingWeights: no file / empty document (non-fatal while searching paths, network.rs:110-118) */
#define DGNN_MALFORMED_WEIGHTS  -2  /* Error::MalformedWeights: bad JSON / base85 / dtype tag, or a required tensor absent or too short */
#define DGNN_HIP_ERROR          -3  /* Error::Cuda(..) / Error::CuDNN(..): a HIP runtime call failed; see dgnn_last_hip_error() */
#define DGNN_BAD_ARGUMENT       -4  /* null pointer, batch <= 0, feature count != batch * 11552 (the debug_assert!s of graph.rs:124-125) */
#define DGNN_OUT_OF_MEMORY      -5  /* host allocation failed */
#define DGNN_UNSUPPORTED        -6  /* architecture outside what the kernels cover (num_channels not a multiple of 32, > 1024, ...) */
#define DGNN_BUSY               -7  /* the object is in use: weights cannot be replaced while workspaces are checked out; a timed pass gave up and must be repeated */
#define DGNN_COMM_ERROR         -8  /* librccl missing or an RCCL call failed; see dgnn_last_rccl_error() */

#define DGNN_NUM_FEATURES   32      /* features::V1::num_features(), src/libdg_go/utils/features.rs:88-90 */
#define DGNN_NUM_POINTS     361     /* 19 x 19, src/libdg_nn/layers/conv2d.rs:50 */
#define DGNN_FEATURE_SIZE   11552   /* features::V1::size() = 32 * 361, features.rs:94-96 */
#define DGNN_NUM_MOVES      362     /* 361 points + pass, src/libdg_nn/layers/policy_head.rs:54 */
#define DGNN_PACKED_WORDS   368     /* packed features of one board, 32-bit words: see dgnn_forward_packed */
#define DGNN_BOARD_RECORD_BYTES 896 /* compact board record of one position: see dgnn_forward_boards */

typedef struct dgnn_network dgnn_network;     /* replaces dg_nn::Network  (src/libdg_nn/network.rs:83-86)  */
typedef struct dgnn_workspace dgnn_workspace; /* replaces dg_nn::Workspace (src/libdg_nn/graph.rs:99-113) */

/* Human readable text for a status code.  Never NULL. */
const char* dgnn_status_string(int status);

/* The hipError_t of the most recent DGNN_HIP_ERROR on the calling thread (0 if none).
 * Replaces the payload of Error::Cuda(cuda::Error), src/libdg_cuda/error.rs:27-41. */
int dgnn_last_hip_error(void);

/* ---- devices: dg_cuda::Device, src/libdg_cuda/devices.rs:55-73 ----------------------------- */

/* Device::len() -- number of visible GPUs. */
int dgnn_device_count(int* count);
/* Which physical GPU device `device` of THIS process is ("0000:c1:00.0"): HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES re-number the
 * devices a process sees, and a launcher that binds rank -> GPU wants to see that the library opens the GPU it means
 * (dg_cuda::Device::set_current takes the same runtime index, src/libdg_cuda/devices.rs:62-73).  capacity >= 16. */
int dgnn_device_pci_bus_id(int device, char* out, size_t capacity);

/* ---- network: dg_nn::Network + loader ------------------------------------------------------ */

/* loader::load(path) + graph::Builder::new (src/libdg_nn/loader.rs:102-116, graph.rs:35-41).
 * Parses the weights JSON ({name: {"s": b85 f32, "t": "i1|i4|f2|f4", "v": b85 bytes}}),
 * discovers the topology by name probing (graph.rs:76-96) and prepares the packed weight image
 * (alpha * b2, tau * b_l, kernel-native filter layout) on the host.  Touches no GPU.
 * Returns DGNN_MISSING_WEIGHTS if the file does not exist or holds no tensor. */
int dgnn_network_load(const char* json_path, dgnn_network** out);

/* Same, from a document already in memory (the reference's loader::load_aux, loader.rs:36-100). */
int dgnn_network_from_json(const char* json_text, size_t json_len, dgnn_network** out);

/* Network::new() (network.rs:92-127): tries, in order, `<exe_path with extension .json>`
 * (skipped when exe_path is NULL), "dream_go.json", "models/dream_go.json",
 * "/usr/share/dreamgo/dream_go.json", "/usr/share/dream_go/dream_go.json"; the first file that
 * loads wins; MISSING_WEIGHTS moves on to the next path, any other error is returned. */
int dgnn_network_load_default(const char* exe_path, dgnn_network** out);

/* Drop of the last Network clone.  Drains and frees every workspace and device copy. */
void dgnn_network_destroy(dgnn_network* net);

/* Topology as the reference derives it: residual block count (graph.rs:76-96),
 * get_num_channels / get_num_samples (layers/common.rs:26-50).  Any out pointer may be NULL. */
int dgnn_network_info(const dgnn_network* net, int* num_blocks, int* num_channels, int* num_samples);

/* config::SOFTMAX_TEMPERATURE (src/libdg_utils/config.rs:176-177; read by policy_head.rs:46 when a
 * workspace is built).  The loader initialises it from the SOFTMAX_TEMPERATURE environment variable
 * exactly like the reference (default 0.709888).  Must be called before the first workspace is created. */
int dgnn_network_set_softmax_temperature(dgnn_network* net, float temperature);

/* Network::get_workspace(batch_size) (network.rs:132-142) with the device made explicit instead of
 * thread-local: pops a pooled workspace for (device, batch_size) or builds one.  The first workspace
 * on a device uploads the weight image to it (the lazy Tensor::copy_to_device of tensor.rs:123-141). */
int dgnn_network_get_workspace(dgnn_network* net, int device, int batch_size, dgnn_workspace** out);

/* Drop of the WorkspaceGuard (network.rs:71-79): returns the workspace to its pool. */
void dgnn_workspace_release(dgnn_workspace* ws);

/* Network::synchronize() (network.rs:145-159): waits for every device, then drops all pooled workspaces. */
int dgnn_network_synchronize(dgnn_network* net);

/* ---- the hot path: dg_nn::forward, src/libdg_nn/graph.rs:123-158 --------------------------- */

/* forward(&mut Workspace, &[f16]) -> OutputMap<f16>.
 *   features     batch * 11552 fp16, HWC per board (index 32 * (19 * y + x) + c), borrowed for the call
 *   num_features must equal batch_size(ws) * 11552 (graph.rs:124-125), else DGNN_BAD_ARGUMENT
 *   value_out    batch fp16          tanh output in [-1, 1]           (OutputMap::value)
 *   policy_out   batch * 362 fp16    soft-maxed with temperature, index 361 = pass; NOT masked, NOT
 *                renormalised (callers do that, src/libdg_mcts/pool/policy_helper.rs:87-134)
 * Synchronous like the reference (to_vec + cudaStreamSynchronize, src/libdg_cuda/memory.rs:138-153):
 * returns after both outputs are in the caller's host buffers. */
int dgnn_forward(dgnn_workspace* ws, const uint16_t* features, size_t num_features,
                 uint16_t* value_out, uint16_t* policy_out);

/* Split form of the same call for callers that overlap tree work with the GPU (no reference
 * counterpart; the reference blocks).  submit copies the features out before returning. */
int dgnn_forward_submit(dgnn_workspace* ws, const uint16_t* features, size_t num_features);
int dgnn_forward_wait(dgnn_workspace* ws, uint16_t* value_out, uint16_t* policy_out);

/* The same calls for a batch SMALLER than the workspace's batch size: `count` boards (1 <= count <= batch size)
 * are evaluated, count * 11552 features are read and count values / count * 362 policy entries are written.  The
 * reference keeps one workspace per distinct batch size (network.rs:132-142) because cuDNN descriptors are per
 * shape; the HIP kernels take the board count at launch, so a batcher can serve every batch 1..max from ONE
 * workspace (and one arena) of capacity max. */
int dgnn_forward_n(dgnn_workspace* ws, const uint16_t* features, int count, uint16_t* value_out, uint16_t* policy_out);
int dgnn_forward_submit_n(dgnn_workspace* ws, const uint16_t* features, int count);

/* The same evaluation from PACKED features.  features::V1 (src/libdg_go/utils/features.rs:99-251) only ever produces, per
 * board, 29 binary planes (channels 2..31, value 0 or 1) and two planes that are 0 or ONE board-wide constant (channels 0 / 1:
 * the komi-scaled colour planes, features.rs:236-239).  Packed form of one board, DGNN_PACKED_WORDS 32-bit words:
 *   words 0..360   bit c of word 19 * y + x is set  <=>  channel c of that point is non-zero
 *   word  361      fp16 bits of the channel-0 constant | fp16 bits of the channel-1 constant << 16
 *   words 362..367 zero (16-byte multiple)
 * 1 472 bytes instead of 23 104 over PCIe and through the batcher; the device expands it directly into the up-layer's
 * input layout.  Outputs are bit-identical to dgnn_forward on the fp16 tensor the words describe (dgnn_features_unpack). */
int dgnn_forward_packed(dgnn_workspace* ws, const uint32_t* packed, int count, uint16_t* value_out, uint16_t* policy_out);
int dgnn_forward_submit_packed(dgnn_workspace* ws, const uint32_t* packed, int count);   /* then dgnn_forward_wait */
/* fp16 HWC features -> packed words (count boards).  DGNN_BAD_ARGUMENT when a board is not representable: a channel >= 2
 * holding something other than 0 / 1, or channel 0 / 1 holding two different non-zero values. */
int dgnn_features_pack(const uint16_t* features, int count, uint32_t* packed);
int dgnn_features_unpack(const uint32_t* packed, int count, uint16_t* features);
/* The workspace's pinned staging buffer for packed features (batch * DGNN_PACKED_WORDS words): a caller that assembles
 * its batch there and passes the pointer skips the staging memcpy. */
int dgnn_workspace_packed_buffer(dgnn_workspace* ws, uint32_t** h_packed);

/* The same evaluation from COMPACT BOARD RECORDS: the feature planes themselves are built on the device.
 * features::V1::get_features (src/libdg_go/utils/features.rs:99-251) runs on the CPU of the reference, once per leaf, on the
 * search worker's thread (src/libdg_mcts/pool/event.rs:46-59): 10-45 us and 23 KB per leaf.  Here the host hands over what it
 * keeps incrementally anyway and the device computes liberties (features.rs:130-140), liberties-after-move and validity for
 * both colours (:141-158), the move-history, colour / komi and any-ko planes (:103-128, :236-250) -- bit-identical planes, so
 * bit-identical outputs to dgnn_forward on dgnn_go_features of the same position.  One record, DGNN_BOARD_RECORD_BYTES, little endian:
 *   u16 cell[368]            per point 19 * y + x (identity orientation): bits 0-1 = 0 empty / 1 black / 2 white, bits 2-10 = block id
 *                            (the index of any one point of the block; every stone of a block carries the same id); 361..367 zero
 *   u32 ko[12]               bit p set: p is empty, not suicide for the side to move, and playing there repeats one of the last positions
 *                            (Board::_is_ko, src/libdg_go/board.rs:120-130 -- needs the hash ring, so it stays on the host)
 *   u32 ladder_capture[12]   bit p set: ... and the stone captures in a ladder (src/libdg_go/utils/ladder.rs:124-136)
 *   u32 ladder_escape[12]    bit p set: ... and the stone escapes a ladder    (ladder.rs:138-170)
 *   u8 to_move, u8 symmetry  colour to move (0 black / 1 white), symmetry 0..7 applied to the OUTPUT planes (symmetry::ALL order)
 *   u16 komi                 fp16 bits of clamp(0.5 + 0.5 * komi / 7.5, 0, 1) (features.rs:236-239)
 *   i16 last[2]              the two most recent moves as points, -1 = none / pass (features.rs:103-110)
 *   u8 reserved[8]           zero
 * dgnn_go_board_record fills one from a dgnn_go_board; dgnn_board_records_to_packed is the HOST statement of the device algorithm
 * (records -> the packed form of dgnn_forward_packed), what the CPU tests compare with dgnn_go_features. */
int dgnn_forward_boards(dgnn_workspace* ws, const uint8_t* records, int count, uint16_t* value_out, uint16_t* policy_out);
int dgnn_forward_submit_boards(dgnn_workspace* ws, const uint8_t* records, int count);   /* then dgnn_forward_wait */
/* The workspace's pinned staging buffer for records (batch * DGNN_BOARD_RECORD_BYTES): assembling the batch there skips a memcpy. */
int dgnn_workspace_records_buffer(dgnn_workspace* ws, uint8_t** h_records);
int dgnn_board_records_to_packed(const uint8_t* records, int count, uint32_t* packed /* count * DGNN_PACKED_WORDS */);
/* Test hook: the planes the DEVICE builds from `count` records, downloaded in packed form (allocates, runs, frees; blocking). */
int dgnn_board_records_to_packed_device(int device, const uint8_t* records, int count, uint32_t* packed);

/* ---- device-resident variants (measurement harness; inputs already in HBM) ----------------- */

int dgnn_workspace_batch_size(const dgnn_workspace* ws, int* batch_size, int* device);

/* The workspace's own HBM buffers: features [batch*11552] fp16, value [batch] fp16,
 * policy [batch*362] fp16.  Valid until the workspace is destroyed. */
int dgnn_workspace_device_buffers(dgnn_workspace* ws, void** d_features, void** d_value, void** d_policy);

/* The workspace's pinned host staging buffers (same shapes).  A batcher may assemble features directly in
 * h_features and pass that pointer to dgnn_forward / dgnn_forward_submit: the staging memcpy is then skipped
 * (likewise for value_out == h_value, policy_out == h_policy). */
int dgnn_workspace_host_buffers(dgnn_workspace* ws, uint16_t** h_features, uint16_t** h_value, uint16_t** h_policy);

/* Name of the residual-convolution kernel this workspace launches ("conv3x3_board_kernel", "conv3x3_tile2_kernel fm=6",
 * "conv3x3_kernel"): the harness quotes it next to the roofline numbers.  Static string, never NULL. */
const char* dgnn_workspace_conv_kernel(const dgnn_workspace* ws);

/* The HIP stream (hipStream_t) every kernel of this workspace is launched on. */
int dgnn_workspace_stream(dgnn_workspace* ws, void** hip_stream);

/* Host -> HBM copy of the features without running anything (blocking). */
int dgnn_workspace_upload(dgnn_workspace* ws, const uint16_t* features, size_t num_features);
/* HBM -> host copy of the outputs (blocking). */
int dgnn_workspace_download(dgnn_workspace* ws, uint16_t* value_out, uint16_t* policy_out);

/* Enqueue one pass of the hot path over the features resident in HBM; asynchronous. */
int dgnn_forward_resident(dgnn_workspace* ws);
/* Block until everything enqueued on the workspace's stream has finished. */
int dgnn_workspace_sync(dgnn_workspace* ws);

/* `steps` back-to-back resident passes bracketed by HIP events on the workspace's stream.
 *   total_ms    elapsed time of all passes
 *   tower_ms    summed elapsed time of the residual-tower launches only (event pairs recorded on the
 *               same stream around the residual convolutions of every pass)
 *   tower_launches  number of kernel launches those events bracket: 2 * blocks * steps with one launch per convolution,
 *               steps where the whole tower is one launch (dgnn_workspace_conv_kernel() == "tower_board_kernel")
 * Any out pointer may be NULL. */
int dgnn_forward_timed(dgnn_workspace* ws, int steps, float* total_ms, float* tower_ms, int* tower_launches);

/* `steps` PCIe-inclusive evaluations of `count` boards from the same host buffer, alternating between two workspaces so that one
 * submission is always queued behind the running one -- how a leaf pool drives a device (two batches alive, predictors/nn.rs:64-67).
 * Host buffer in, host buffers out, no interpreter in the loop.  form: 0 = fp16 planes (dgnn_forward_submit_n), 1 = packed words
 * (dgnn_forward_submit_packed), 2 = board records (dgnn_forward_submit_boards).  *seconds = wall time of the `steps` evaluations. */
int dgnn_forward_pipelined_timed(dgnn_workspace* a, dgnn_workspace* b, int form, const void* input, int count, int steps, double* seconds);
/* The same with `in_flight` (1..8) distinct workspaces: as many submissions are kept queued on the device. */
int dgnn_forward_pipelined_timed_n(dgnn_workspace* const* workspaces, int in_flight, int form, const void* input, int count, int steps,
                                   double* seconds);

/* Where the wall time of one device batch goes: `steps` submissions of `count` board records with HIP events between the stages of a
 * submission on the workspace's stream, `in_flight` (1..8) distinct workspaces kept busy the way a leaf pool keeps batches alive.
 * stage_us[7]: mean microseconds per stage -- H2D of the records, planes kernel, tower, head convolution, policy GEMM, soft-max +
 * value, D2H of the outputs (a stage that waits for compute units held by another submission's tower shows as long);  host_us[4]:
 * the submit call, submit -> answers with the caller, the device's span first -> last stamp, and the difference of the last two. */
int dgnn_forward_stage_times(dgnn_workspace* const* workspaces, int in_flight, const uint8_t* records, int count, int steps,
                             float* stage_us /* 7 */, double* host_us /* 4 */);

/* ---- layer-level entry point (parity tests) --------------------------------------------------- */

/* One Conv2d::forward / forward_skip (src/libdg_nn/layers/conv2d.rs:171-220) on host buffers, for the
 * layer-level parity tests that mirror the reference's own (conv2d.rs:252-291):
 *   y = act(alpha1 * conv3x3(x, w) + alpha2 * z + bias)
 *   x [batch][board][board][cin] fp16 NHWC, w [cout][3][3][cin] fp16 KRSC, bias [cout], z/y [..][cout];
 *   cin must be a multiple of 32 (pad with zero channels); z is read only when alpha2 != 0;
 *   relu != 0 selects ReLU, 0 the identity.  Allocates, runs and frees on `device`; blocking. */
int dgnn_layer_conv3x3(int device, const uint16_t* x, int batch, int board_size, int cin,
                       const uint16_t* w_krsc, const uint16_t* bias, int cout,
                       float alpha1, float alpha2, const uint16_t* z, int relu, uint16_t* y);

/* One Dense::forward (src/libdg_nn/layers/dense.rs:173-220) on host buffers through the MFMA GEMM of the policy head
 * (heads.hip: policy_gemm_kernel), for the layer-level tests that mirror the reference's own (dense.rs:243-299):
 *   y[b][m] = act(fp16(alpha * sum_k x[b][k] * w[k][m] + bias[m]))      fp32 accumulate
 *   x [batch][k_in] fp16, w [k_in][n_out] fp16 AS STORED in the weights file ([in][out]; the transpose of dense.rs:182-189
 *   is done here), bias [n_out] fp16 (already scaled by the caller where the reference scales it), y [batch][n_out];
 *   k_in a multiple of 8; relu != 0 selects ReLU (dense.rs:108-116), 0 the identity. */
int dgnn_layer_dense(int device, const uint16_t* x, int batch, int k_in, const uint16_t* w_in_out, const uint16_t* bias,
                     int n_out, float alpha, int relu, uint16_t* y);

/* The tail of both heads (heads.hip: softmax_value_kernel) on host buffers: cudnnSoftmaxForward ACCURATE over the 362 logits
 * of a board (src/libdg_cuda/cudnn/softmax.rs:58-79; policy_head.rs:98-100) and the value head's Dense 722 -> 1 followed by
 * tanh (value_head.rs:80-81).  logits [batch][362], value_features [batch][722], value_w [722], value_b [1]. */
int dgnn_layer_heads(int device, const uint16_t* logits, const uint16_t* value_features, int batch, const uint16_t* value_w,
                     const uint16_t* value_b, uint16_t* policy_out, uint16_t* value_out);

/* Test-only: the two fp16 tensors the reference materialises IN FRONT of the output activations, as the last forward on `ws` left
 * them -- the temperature-scaled policy logits, input of cudnnSoftmaxForward (policy_head.rs:95-100; [batch][362]) and the value
 * head's dense output, input of tanh (value_head.rs:80-81; [batch]).  The parity tests bound the error there, in fp16 ulps: a
 * probability moves by p (1 - p) dlogit, so a tolerance in probability space says little about a peaked soft-max. */
int dgnn_workspace_download_pre(dgnn_workspace* ws, uint16_t* logits_out, uint16_t* value_pre_out);

/* ---- multi-GPU: weight image for an RCCL broadcast ------------------------------------------ */

/* The packed, kernel-native weight image of `net` on `device` (uploaded on first use) and its size.
 * One ncclBroadcast of this range from the rank that loaded the file is the only collective the
 * path needs (replaces the per-device host upload of tensor.rs:123-141 on weight refresh). */
int dgnn_network_weight_image(dgnn_network* net, int device, void** d_image, size_t* num_bytes);

/* A 64-bit FNV-1a checksum of the host-side weight image, to verify a broadcast. */
int dgnn_network_weight_checksum(const dgnn_network* net, uint64_t* checksum);

/* The same range without the upload when `device` is touched for the first time: what a RECEIVER of a broadcast needs
 * (contents undefined until the collective has landed and dgnn_network_set_weight_image adopted it). */
int dgnn_network_weight_image_reserve(dgnn_network* net, int device, void** d_image, size_t* num_bytes);
/* ... the caller has filled that range with this network's image (e.g. a broadcast that is known to have succeeded).  Until then the
 * range counts as empty: the first workspace on the device uploads the host image instead of running on whatever the range holds. */
int dgnn_network_weight_image_commit(dgnn_network* net, int device);

/* Weight refresh.  `src` holds a whole image (`num_bytes` == the image size) produced by a network of the SAME architecture
 * and packing -- the image starts with a header (architecture, packing, residual gates `NN_residual/alpha:0`, soft-max
 * temperature) that is checked first: a foreign image is refused with DGNN_BAD_ARGUMENT / DGNN_MALFORMED_WEIGHTS and nothing is
 * changed.  `src` is a device pointer on `device` when src_on_device != 0 (e.g. the buffer an RCCL broadcast landed in; the
 * pointer dgnn_network_weight_image returned makes the device copy a no-op) or a host pointer otherwise.  Afterwards the
 * network launches with the SENDER's gates and temperature, its host image (dgnn_network_weight_image_host / _checksum), every
 * device copy and every cached launch descriptor are the new ones.  Returns DGNN_BUSY, changing nothing, while any workspace of
 * the network is checked out (a forward may be in flight): release them -- pause a leaf pool with dgnn_pool_pause -- first. */
int dgnn_network_set_weight_image(dgnn_network* net, int device, const void* src, size_t num_bytes, int src_on_device);

/* What the ranks of a broadcast must agree on before a byte moves: descriptor[0] = image bytes, descriptor[1] = a hash of the
 * architecture and packing (not of the weights, gates or temperature).  dgnn_broadcast_weights exchanges it itself. */
int dgnn_network_image_descriptor(const dgnn_network* net, uint64_t* descriptor /* 2 */);

/* FNV-1a checksum of the image as it currently sits in `device` memory (downloads it). */
int dgnn_network_weight_checksum_device(dgnn_network* net, int device, uint64_t* checksum);

/* The host-side weight image itself (read-only, owned by the network). */
int dgnn_network_weight_image_host(const dgnn_network* net, const void** image, size_t* num_bytes);

/* ---- the collective itself: RCCL inside the library ------------------------------------------------------------
 * librccl is opened with dlopen on first use (a copy already in the process is shared); a host that never calls these
 * functions needs only the HIP runtime.  Replaces, at load / weight refresh, the host upload every device performs on its
 * own in the reference (Tensor::copy_to_device, src/libdg_nn/tensor.rs:123-141, once per device and tensor). */
#define DGNN_RCCL_UNIQUE_ID_BYTES 128
typedef struct dgnn_comm dgnn_comm;
int dgnn_rccl_available(void);                         /* 1 when librccl could be opened */
int dgnn_last_rccl_error(void);                        /* the ncclResult_t of the most recent DGNN_COMM_ERROR on this thread */
const char* dgnn_rccl_error_string(int nccl_result);
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks by whatever channel the host has (a file, a
 * socket, MPI, a torch.distributed store). */
int dgnn_rccl_unique_id(uint8_t* id_out /* DGNN_RCCL_UNIQUE_ID_BYTES */);
/* ncclCommInitRank on `device`: collective over the num_ranks callers (one process per GPU, or one thread per GPU). */
int dgnn_rccl_comm_create(int device, int num_ranks, int rank, const uint8_t* id, dgnn_comm** out);
/* Use a communicator the host application already has (an ncclComm_t); not destroyed by dgnn_rccl_comm_destroy. */
int dgnn_rccl_comm_wrap(void* nccl_comm, int device, dgnn_comm** out);
int dgnn_rccl_comm_info(const dgnn_comm* comm, int* num_ranks, int* rank, int* device);
void dgnn_rccl_comm_destroy(dgnn_comm* comm);
/* ONE ncclBroadcast of the packed weight image from rank `root` into every rank's device copy, then
 * dgnn_network_set_weight_image on what arrived (header check, gates / temperature adopted).  Collective: every rank of the
 * communicator calls it with a network of the same architecture.  Blocking. */
int dgnn_broadcast_weights(dgnn_network* net, dgnn_comm* comm, int root);
/* The single-process form (the reference's own process model: one host process, all devices, predictors/nn.rs:84-108): the
 * image is uploaded to devices[0] over PCIe and broadcast to the others over xGMI (ncclCommInitAll + one grouped
 * ncclBroadcast).  devices == NULL: devices 0..num_devices-1. */
int dgnn_broadcast_weights_devices(dgnn_network* net, const int* devices, int num_devices);

/* ============================================================================================================
 * Host side of the boundary (scope row f-1): policy symmetries, transposition cache, many-game leaf batcher.
 * Pure host code (batcher.cpp); only dgnn_batch_forward touches a GPU.
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

/* Transform::inverse (symmetry.rs:78-90); -1 for an invalid transform. */
int dgnn_symmetry_inverse(int transform);
/* Transform::apply on a packed index 0..361 (361 = pass maps to itself); -1 for invalid arguments. */
int dgnn_symmetry_apply(int transform, int packed_index);
/* Prediction::with_transform (src/libdg_mcts/predictor.rs:30-44): out[apply(t, i)] = policy[i]; out != policy. */
int dgnn_policy_with_transform(const uint16_t* policy, int transform, uint16_t* out);

/* LruCache + NnPredictor::{fetch, cache} (src/libdg_mcts/lru_cache.rs:56-150, predictors/nn.rs:29-82).  Keyed by
 * (zobrist hash, colour to move); entries are stored in the canonical orientation.  `num_shards` = 1 reproduces the
 * reference's single list (and its eviction order) exactly; more shards remove the global lock.  Thread safe. */
typedef struct dgnn_cache dgnn_cache;
int dgnn_cache_create(size_t capacity, int num_shards, dgnn_cache** out);        /* reference: 200 000 + 1, nn.rs:29,57 */
void dgnn_cache_destroy(dgnn_cache* cache);
/* fetch: returns 1 on a hit (value_out / policy_out[362] filled, remapped by `symmetry`), 0 on a miss, < 0 on error. */
int dgnn_cache_fetch(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t* value_out, uint16_t* policy_out);
/* cache: stores the prediction remapped by symmetry.inverse(); a key that is already present is left untouched. */
int dgnn_cache_insert(dgnn_cache* cache, uint64_t board_hash, int to_move, int symmetry, uint16_t value, const uint16_t* policy);
int dgnn_cache_stats(dgnn_cache* cache, uint64_t* entries, uint64_t* hits, uint64_t* misses);
int dgnn_cache_clear(dgnn_cache* cache);     /* drops every entry: the table holds the outputs of the weights that are being replaced */
/* A clear also starts a new GENERATION of the table.  A caller that evaluates its misses asynchronously reads the generation before
 * the fetch and stores the answer with dgnn_cache_insert_at: an answer whose miss predates a clear is dropped instead of putting
 * the replaced network's output back into the emptied table (the search driver and dgnn_full_forward do this). */
uint64_t dgnn_cache_generation(const dgnn_cache* cache);
int dgnn_cache_insert_at(dgnn_cache* cache, uint64_t generation, uint64_t board_hash, int to_move, int symmetry, uint16_t value,
                         const uint16_t* policy);

/* Batcher / Batch (src/libdg_mcts/pool/batch.rs:24-123).  Leaf positions from any number of games and threads are
 * pushed with a caller-chosen 64-bit tag; get_batch hands out the NEWEST max_batch_size of them once at least
 * `min_batch_size` are queued and fewer than `max_batches` batches are alive (the reference: 2 x devices,
 * nn.rs:64-67).  `net` may be NULL for a batcher that only queues (dgnn_batch_forward is then refused). */
typedef struct dgnn_batcher dgnn_batcher;
typedef struct dgnn_batch dgnn_batch;
int dgnn_batcher_create(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out);
void dgnn_batcher_destroy(dgnn_batcher* batcher);
int dgnn_batcher_push(dgnn_batcher* batcher, const uint16_t* features /* 11552 fp16 */, uint64_t tag);       /* batch.rs:80-84 */
int dgnn_batcher_get_batch(dgnn_batcher* batcher, int min_batch_size, dgnn_batch** out /* NULL: none */);    /* batch.rs:91-121 */
int dgnn_batcher_push_and_get_batch(dgnn_batcher* batcher, const uint16_t* features, uint64_t tag, dgnn_batch** out);  /* :86-89 */
int dgnn_batcher_pending(dgnn_batcher* batcher, int* queued, int* batches_alive);
int dgnn_batcher_stats(dgnn_batcher* batcher, uint64_t* boards_evaluated, uint64_t* batches_evaluated);
int dgnn_batch_size(const dgnn_batch* batch);
int dgnn_batch_tags(const dgnn_batch* batch, uint64_t* tags_out /* dgnn_batch_size entries, in push order */);
const uint16_t* dgnn_batch_features(const dgnn_batch* batch);
/* Batch::forward (batch.rs:35-40): evaluates the batch on the batcher's device through a pooled workspace of
 * capacity max_batch_size; value_out[n], policy_out[n * 362] in tag order; the batch stops counting as alive. */
int dgnn_batch_forward(dgnn_batch* batch, uint16_t* value_out, uint16_t* policy_out);
/* The same on an explicit device (NnPredictor::predict picks one per call, predictors/nn.rs:87-89). */
int dgnn_batch_forward_on(dgnn_batch* batch, int device, uint16_t* value_out, uint16_t* policy_out);
void dgnn_batch_destroy(dgnn_batch* batch);

/* Cross-caller coalescing (no reference counterpart: the reference sends the <= 8-board full_forward calls of its
 * game threads to the GPU one by one, src/libdg_mcts/lib.rs:114).  dgnn_pool_predict has the blocking
 * Predictor::predict signature (src/libdg_mcts/predictor.rs:92) and may be called from any number of threads; worker
 * threads of the pool turn everything that has been pushed meanwhile into device batches (<= max_batch_size boards,
 * <= max_batches in flight). */
typedef struct dgnn_pool dgnn_pool;
/* Ownership: the pool holds a reference on the batcher until dgnn_pool_destroy has joined its workers, so the two may be
 * destroyed in either order. */
int dgnn_pool_create(dgnn_batcher* batcher, int num_workers, dgnn_pool** out);
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
 * dgnn_pool_predict_boards / dgnn_pool_boards_predictor(); dgnn_pool_predict on it (and vice versa) is DGNN_BAD_ARGUMENT.  The
 * callback form takes an fp16-plane predictor (dgnn_predict_fn): its lanes expand the records with the host statement of the device
 * kernel (dgnn_board_records_to_packed) -- for dry runs and for tests with the CPU oracle as the predictor. */
int dgnn_batcher_create_boards(dgnn_network* net, int device, int max_batch_size, int max_batches, dgnn_batcher** out);
int dgnn_batcher_push_board(dgnn_batcher* batcher, const uint8_t* record /* DGNN_BOARD_RECORD_BYTES */, uint64_t tag);
const uint8_t* dgnn_batch_records(const dgnn_batch* batch);       /* NULL for a batch of fp16 planes (and dgnn_batch_features vice versa) */
int dgnn_pool_create_devices_boards(dgnn_network* net, const int* devices, int num_devices, int max_batch_size, int batches_per_device,
                                    int workers_per_device, dgnn_pool** out);
int dgnn_pool_create_callbacks_boards(dgnn_predict_fn predict, void* predict_ctx, int num_lanes, int max_batch_size, int batches_per_lane,
                                      int workers_per_lane, dgnn_pool** out);
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
 * must be a function of their seed (cache_entries_per_game > 0) keep it off.  DGNN_BUSY while requests are in flight. */
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
/* Minimum fill: while ANOTHER batch keeps a device busy, a free batch slot waits until `min_boards` are queued -- or `max_wait_us` have
 * passed since the oldest waiting board was queued -- instead of taking the handful that is there the moment the slot frees.  A
 * workgroup per board makes 130 boards take as long as 256, so fuller batches are more evaluations per second.  0: off (the reference
 * takes whatever is queued, batch.rs:93-121). */
int dgnn_pool_set_min_fill(dgnn_pool* pool, int min_boards, int max_wait_us);
uint64_t dgnn_pool_coalesced(const dgnn_pool* pool);        /* boards that followed instead of being evaluated */

/* How many batches a pool keeps alive per lane.  The reference's rule is the constant 2 per device (predictors/nn.rs:64-67); a third
 * keeps a device supplied when batches are partial (one workgroup per board: 129 .. 200 boards leave CUs idle) but puts more leaves in
 * flight, i.e. more positions evaluated twice before the first answer reaches the transposition table.  With lo < hi (a pool of board
 * records created with at least `hi` batches and workers per lane) the pool measures UNIQUE positions per second -- it hashes the
 * stones and the colour to move of every record against a table of recent ones -- under both settings in alternating windows of 32 768
 * boards and stays with the better one, probing again every dozen windows.  lo == hi == 0: back to the constant. */
int dgnn_pool_set_adaptive_batches(dgnn_pool* pool, int lo, int hi);
/* The rule that measured better than both constants: `lo` batches alive per lane, up to `hi` only while at least `min_queued` boards
 * are waiting -- an extra batch that comes out full keeps the device supplied, one that splits the same few leaves over more batches
 * only makes every batch smaller.  min_queued == 0: off. */
int dgnn_pool_set_extra_batch_queue(dgnn_pool* pool, int lo, int hi, int min_queued);
int dgnn_pool_adaptive_stats(dgnn_pool* pool, int* alive_per_lane, uint64_t* switches, uint64_t* boards_hashed, uint64_t* boards_repeated,
                             double* unique_rate_lo, double* unique_rate_hi);

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

/* config::get_uct_exp / get_fpu_reduce / get_lcb_critical_value / VLOSS_CNT and the interpolation they use
 * (src/libdg_utils/config.rs:181-195, 289-330). */
float dgnn_config_uct_exp(int visits);
float dgnn_config_fpu_reduce(int visits);
float dgnn_config_lcb_critical_value(int visits);
int dgnn_config_vloss_cnt(void);
float dgnn_intp_value(const int* xs, const float* ys, int n, int x);
/* asm::argmax_f32 (src/libdg_mcts/asm/argmax.rs:24-78), n a multiple of 8; -1 when the maximum is not finite. */
int dgnn_argmax_f32(const float* values, int n);

/* Node::new(to_move, value, prior[>= 362]) (tree.rs:1070-1090); a node owns its sub-tree. */
dgnn_tree_node* dgnn_tree_node_new(int to_move, float value, const float* prior);
void dgnn_tree_node_free(dgnn_tree_node* node);
/* probe (tree.rs:1421-1470): returns DGNN_PROBE_FOUND with *out = the trace, DGNN_PROBE_NO_RESULT or DGNN_PROBE_CONFLICT
 * (virtual losses of the partial descent undone).  ops / board may be NULL. */
int dgnn_tree_probe(dgnn_tree_node* root, const dgnn_board_ops* ops, void* board, dgnn_trace** out);
int dgnn_trace_len(const dgnn_trace* trace);
int dgnn_trace_get(const dgnn_trace* trace, int i, dgnn_tree_node** node, int* color, int* index);
void dgnn_trace_free(dgnn_trace* trace);
/* undo (tree.rs:1397-1409) and insert + UCT::update (tree.rs:1482-1510, 125-160); neither frees the trace. */
int dgnn_tree_undo(dgnn_trace* trace, int undo_expanding);
int dgnn_tree_insert(const dgnn_trace* trace, int color, float value /* win rate of `color`, 0..1 */, const float* prior);
/* Child / Node accessors (tree.rs:326-358, 1027-1051). */
int dgnn_tree_node_child(const dgnn_tree_node* node, int index, int* count, int* vcount, float* value, float* value_std, int* has_child);
int dgnn_tree_node_info(const dgnn_tree_node* node, int* to_move, int* total_count, int* vtotal_count, int* pass_count, float* initial_value);
int dgnn_tree_node_prior(const dgnn_tree_node* node, float* prior_out /* 362 */);
/* min_promote_rollouts (src/libdg_mcts/time_control/mod.rs:48-73): visits of the most visited child minus the runner-up's. */
int dgnn_tree_node_min_promote(const dgnn_tree_node* node);
/* time_control::is_done for a RolloutLimit (time_control/mod.rs:86-101, rollout_limit.rs:35-44): 1 when root.total_count has
 * reached `limit`, or (EARLY-C) when the runner-up cannot overtake the most visited child in the rollouts that remain;
 * 0 otherwise and for an unvisited root.  The search workers stop on it (pool/worker_thread.rs:149). */
int dgnn_rollout_limit_is_done(const dgnn_tree_node* node, int limit);
int dgnn_tree_node_disqualify(dgnn_tree_node* node, int index);                              /* tree.rs:1315-1320 */
int dgnn_tree_node_softmax(const dgnn_tree_node* node, float* out /* 362 */);                 /* tree.rs:1293-1307 */
/* Node::best (tree.rs:1262-1283): greedy (LCB / visits / prior / value order) when temperature <= 0.09, else a visit-
 * proportional draw with `at` in [0, 1) standing in for thread_rng(). */
int dgnn_tree_node_best(const dgnn_tree_node* node, float temperature, double at, float* value_out, int* index_out);
/* tree::to_pretty (tree.rs:1620-1746, non-verbose): the human-readable summary of a search ("Nodes: .., Win: ..%, PV: .." and one
 * line per child for the ten best) that Played::from_mcts stores with a move and Played::fmt writes as C[...] (self_play.rs:136,
 * 191-193).  Returns the bytes needed including the NUL; writes when it fits. */
size_t dgnn_tree_node_to_pretty(const dgnn_tree_node* node, char* out, size_t capacity);
/* choose (src/libdg_mcts/choose.rs:54-88) on integer weights: 1 and *index_out / *cum_out when an item was chosen. */
int dgnn_choose(const int* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out);
/* The same on real weights (a policy with -inf for illegal moves, as the policy play of self_play.rs:383-388 calls it). */
int dgnn_choose_f64(const double* items, int n, double cutoff_percentile, double temperature, double at, int* index_out, double* cum_out);

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

/* One search on `root` from `board` (not modified): rollouts until root.total_count >= num_rollouts
 * (pool/worker_thread.rs:50-176 + time_control/rollout_limit.rs:35-44).  All randomness derives from `seed`; with
 * num_threads == 1 the resulting visit counts are a pure function of the seed and the predictor's outputs. */
int dgnn_search_run(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                    int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out, uint64_t* conflicts_out);

/* The same with the transposition cache of NnPredictor in front of the predictor (pool/event.rs:50-51,
 * pool/worker_thread.rs:111): needs ops->hash; cache may be NULL.  *evals_out counts predictor calls only. */
int dgnn_search_run_cached(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, uint64_t seed, uint64_t* evals_out,
                           uint64_t* conflicts_out, uint64_t* cache_hits_out);
/* The same with up to `leaves_per_thread` (1..64) leaves per worker in flight: a worker probes that many leaves one after the
 * other (each probe's virtual loss steers the next one elsewhere), evaluates them in ONE predictor call and inserts them.
 * 1 is the reference's worker.  One thread with K leaves keeps the search a pure function of (seed, network outputs) while K
 * boards are on the device, and on a host with few cores it replaces K threads and their wake-ups by one. */
int dgnn_search_run_batched(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_fn predict, void* predict_ctx,
                            dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                            uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out);
/* dgnn_search_run_batched / dgnn_full_forward / dgnn_selfplay_go behind a predictor of board records: the game's `record` callback
 * is used instead of `features` (pool/event.rs:54 without the feature extraction), everything else is the same code. */
int dgnn_search_run_boards(dgnn_tree_node* root, const dgnn_game_ops* ops, void* board, dgnn_predict_boards_fn predict, void* predict_ctx,
                           dgnn_cache* cache, int num_rollouts, int num_threads, int leaves_per_thread, uint64_t seed,
                           uint64_t* evals_out, uint64_t* conflicts_out, uint64_t* cache_hits_out);
int dgnn_full_forward_boards(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_boards_fn predict, void* predict_ctx,
                             dgnn_cache* cache, float* value_out, float* policy_out);
/* full_forward (src/libdg_mcts/lib.rs:83-130): win rate and prior (368 floats, -inf = not a candidate) of a position as the
 * average over the eight symmetries, the uncached ones evaluated as ONE batch.  Returns the number of symmetries evaluated
 * (>= 0) or a negative status. */
int dgnn_full_forward(const dgnn_game_ops* ops, void* board, int to_move, dgnn_predict_fn predict, void* predict_ctx,
                      dgnn_cache* cache, float* value_out, float* policy_out);
/* dirichlet::add_ex (src/libdg_mcts/dirichlet.rs:40-70) with a seeded generator; the reference calls it with shape 0.03 and
 * beta = DIRICHLET_NOISE (0.25 in self-play, config.rs:165-166). */
void dgnn_dirichlet_add(float* x, int n, float shape, float beta, uint64_t seed);
/* Node::forward (tree.rs:1225-1250): the sub-tree below `index` becomes the root and is returned (NULL when it was never
 * expanded, except for a pass); `node` and the rest of its tree are freed.  set_prior: lib.rs:176-183. */
dgnn_tree_node* dgnn_tree_node_forward(dgnn_tree_node* node, int index);
int dgnn_tree_node_set_prior(dgnn_tree_node* node, const float* prior /* 362 */);

/* A rule-free synthetic game for throughput measurements (NOT Go: every empty point is legal, nothing is captured,
 * features are pseudo-random bits of the position hash) and a self-play loop over it. */
const dgnn_game_ops* dgnn_synthetic_game_ops(void);
void* dgnn_synthetic_board_new(uint64_t seed);
void dgnn_synthetic_board_free(void* board);
typedef struct dgnn_selfplay_stats { double seconds; uint64_t games, moves, evals, conflicts, cache_hits, games_ended, ex_it_searches; } dgnn_selfplay_stats;
int dgnn_selfplay_synthetic(dgnn_predict_fn predict, void* predict_ctx, int num_games, int num_moves, int num_rollouts,
                            int threads_per_search, uint64_t seed, dgnn_selfplay_stats* stats);

/* ============================================================================================================
 * Go board, ladders and feature planes (scope row f-3), go.cpp: what produces the input of the hot path.
 * Colours: DGNN_COLOR_BLACK / DGNN_COLOR_WHITE; coordinates x, y in 0..18 like Point::new (src/libdg_go/point.rs:26-34).
 * ============================================================================================================ */
typedef struct dgnn_go_board dgnn_go_board;                                  /* Board, src/libdg_go/board.rs:28-50 */
dgnn_go_board* dgnn_go_board_new(float komi);
dgnn_go_board* dgnn_go_board_clone(const dgnn_go_board* board);
void dgnn_go_board_free(dgnn_go_board* board);
int dgnn_go_place(dgnn_go_board* board, int color, int x, int y);            /* Board::place: no legality check (board.rs:152-176) */
int dgnn_go_at(const dgnn_go_board* board, int x, int y);                    /* Board::at: -1 empty, else the colour */
int dgnn_go_is_valid(const dgnn_go_board* board, int color, int x, int y);   /* Board::is_valid: no suicide, no super-ko */
int dgnn_go_to_move(const dgnn_go_board* board);                             /* board.rs:100-105 */
int dgnn_go_count(const dgnn_go_board* board);
uint64_t dgnn_go_hash(const dgnn_go_board* board);                           /* position hash (own zobrist numbers) */
int dgnn_go_n_liberty(const dgnn_go_board* board, int x, int y);             /* BoardFast::get_n_liberty; -1 on an empty point */
int dgnn_go_n_liberty_if(const dgnn_go_board* board, int color, int x, int y);   /* BoardFast::get_n_liberty_if; -1 if not valid */
int dgnn_go_is_ladder_capture(const dgnn_go_board* board, int color, int x, int y);   /* utils/ladder.rs:124-136; -1 if not valid */
int dgnn_go_is_ladder_escape(const dgnn_go_board* board, int color, int x, int y);    /* utils/ladder.rs:138-170 */
int dgnn_go_is_symmetric(const dgnn_go_board* board, int symmetry);          /* symmetry::is_symmetric */
/* features::V1::get_features::<HWC, f16> (src/libdg_go/utils/features.rs:99-251): 361 * 32 fp16, the `features` argument of
 * dgnn_forward for one board. */
int dgnn_go_features(const dgnn_go_board* board, int to_move, int symmetry, uint16_t* out);
/* The compact record of the position for dgnn_forward_boards (layout there). */
int dgnn_go_board_record(const dgnn_go_board* board, int to_move, int symmetry, uint8_t* out /* DGNN_BOARD_RECORD_BYTES */);
/* The board as the game behind dgnn_search_run (a dgnn_go_board* is the `board` argument). */
const dgnn_game_ops* dgnn_go_game_ops(void);        /* StandardSearch: every legal move and pass (options.rs:44-82) */
const dgnn_game_ops* dgnn_go_scoring_ops(void);     /* ScoringSearch: no pass, no point inside a Benson eye of either colour, no filling of
                                                       one's own simple eye (options.rs:110-214) */

/* Benson's unconditional life as the reference computes it (src/libdg_go/utils/benson.rs:62-190): per point 0 = neither,
 * 1 = stone of an unconditionally alive block of `color` (is_alive), 2 = point of a region vital to one (is_eye). */
int dgnn_go_benson(const dgnn_go_board* board, int color, unsigned char* status_out /* 361 */);
/* The same by the reference's own steps (flood fills over points): the checker of the bit-board form dgnn_go_benson runs. */
int dgnn_go_benson_flood(const dgnn_go_board* board, int color, unsigned char* status_out /* 361 */);
/* The candidate moves of a search as the game callbacks give them (pool/policy_helper.rs:36-48): scoring == 0 the
 * StandardSearch rule (any valid move, and pass), else the ScoringSearch rule (options.rs:110-214).  out: 362 flags. */
int dgnn_go_candidates(const dgnn_go_board* board, int to_move, int scoring, unsigned char* out);
/* 1 when the incremental block / liberty bookkeeping equals a from-scratch flood fill of the position (a self-test hook). */
int dgnn_go_check(const dgnn_go_board* board);
int dgnn_go_is_scorable(const dgnn_go_board* board);                             /* Score::is_scorable, utils/score.rs:105-117 */
int dgnn_go_score(const dgnn_go_board* board, int* black_out, int* white_out);   /* Score::get_score (Tromp-Taylor, no komi), :130-136, 220-239 */
/* Score::get_stone_status(board, board) (utils/score.rs:146-186) as a bit mask per point. */
#define DGNN_STONE_ALIVE 1
#define DGNN_STONE_DEAD 2
#define DGNN_STONE_SEKI 4
#define DGNN_TERRITORY_BLACK 8
#define DGNN_TERRITORY_WHITE 16
int dgnn_go_stone_status(const dgnn_go_board* board, unsigned char* status_out /* 361 */);
/* The RE[] value ("B+3.5", "W+0.5", "0") and the TB[..]TW[..] properties of GameResult::Ended
 * (src/libdg_mcts/game_result.rs:35-93).  Either output may be NULL. */
int dgnn_go_result_sgf(const dgnn_go_board* board, char* winner_out, size_t winner_capacity, char* territory_out, size_t territory_capacity);
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
    const int* opening;       /* [num_games][opening_len] packed indices, e.g. the moves_out of an earlier call; NULL = empty boards */
    int opening_games;        /* > 0: `opening` holds rows for the first opening_games games only, later games start from the empty board */
    struct dgnn_selfplay_window* window;   /* non-NULL with seconds > 0: a timed window over a steady state of num_parallel games, see below */
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
} dgnn_selfplay_window;
int dgnn_selfplay_go(dgnn_predict_fn predict, void* predict_ctx, dgnn_cache* cache, const dgnn_selfplay_params* params,
                     dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity);

int dgnn_selfplay_go_boards(dgnn_predict_boards_fn predict, void* predict_ctx, dgnn_cache* cache, const dgnn_selfplay_params* params,
                            dgnn_selfplay_stats* stats, int* moves_out, char* sgf_out, size_t sgf_capacity);

/* Test hook: the n-th board pushed into any batcher's queue from now on (n = 0: the next one) is refused with DGNN_OUT_OF_MEMORY, as if
 * the queue's allocation had failed; n < 0 switches the hook off (its state at load). */
void dgnn_test_fail_push_after(int n);

/* Library version / build info string (static). */
const char* dgnn_version(void);

#ifdef __cplusplus
}
#endif

#endif /* DGNN_H */
