/*
 * dgnn.h -- C ABI of libdgnn_hip.so, the MI355X-native replacement for dream-go's
 * neural-network leaf evaluation (crate `dg_nn` + the parts of `dg_cuda` it uses).
 *
 * This header is the drop-in boundary: what the reference's `dg_nn` crate offers its callers, plus the packed /
 * board-record input forms, the weight image for an RCCL broadcast and the resident-input variants a measurement harness needs.
 * The C++ host rows around the path (leaf pool, search, board, self-play) are in dgnn_host.h, test-only entry points in dgnn_test.h.
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

/* ---- device-resident variants (measurement harness; inputs already in HBM) ----------------- */

int dgnn_workspace_batch_size(const dgnn_workspace* ws, int* batch_size, int* device);

/* The workspace's pinned host staging buffers (same shapes).  A batcher may assemble features directly in
 * h_features and pass that pointer to dgnn_forward / dgnn_forward_submit: the staging memcpy is then skipped
 * (likewise for value_out == h_value, policy_out == h_policy). */
int dgnn_workspace_host_buffers(dgnn_workspace* ws, uint16_t** h_features, uint16_t** h_value, uint16_t** h_policy);

/* Name of the residual-convolution kernel this workspace launches ("conv3x3_board_kernel", "conv3x3_tile2_kernel fm=6",
 * "conv3x3_kernel"): the harness quotes it next to the roofline numbers.  Static string, never NULL. */
const char* dgnn_workspace_conv_kernel(const dgnn_workspace* ws);

/* Which form the residual tower of the passes enqueued so far took, over all workspaces of `net` on `device` (-1: every device):
 * counts[0] one launch per layer, [1] one launch with a workgroup per board (tower_board_kernel), [2] one launch with a board split
 * over several workgroups (tower_split_kernel: batches too small to fill the chip), [3] split launches that gave up and were repeated
 * per layer (expected: 0 -- see dgnn_status_string / stderr).  What a harness quotes next to a leaf pool's throughput. */
int dgnn_network_tower_stats(dgnn_network* net, int device, uint64_t* counts /* 4 */);

/* The HIP stream (hipStream_t) every kernel of this workspace is launched on. */
int dgnn_workspace_stream(dgnn_workspace* ws, void** hip_stream);

/* Host -> HBM copy of the features without running anything (blocking). */
int dgnn_workspace_upload(dgnn_workspace* ws, const uint16_t* features, size_t num_features);
/* The outputs of the most recent pass on this workspace, whichever entry point ran it (blocking): an HBM -> host copy after
 * dgnn_forward_resident / dgnn_forward_timed, the pinned host buffers' contents after a dgnn_forward* call (whose outputs never sit in HBM). */
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

/* `steps` PCIe-inclusive evaluations of `count` boards from the same host buffer with `in_flight` (1..8) distinct workspaces, so that
 * in_flight - 1 submissions are always queued behind the running one -- how a leaf pool drives a device (two batches alive,
 * predictors/nn.rs:64-67).  Host buffer in, host buffers out, no interpreter in the loop.  form: 0 = fp16 planes
 * (dgnn_forward_submit_n), 1 = packed words (dgnn_forward_submit_packed), 2 = board records (dgnn_forward_submit_boards).
 * *seconds = wall time of the `steps` evaluations. */
int dgnn_forward_pipelined_timed_n(dgnn_workspace* const* workspaces, int in_flight, int form, const void* input, int count, int steps,
                                   double* seconds);

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

/* Library version / build info string (static). */
const char* dgnn_version(void);

#ifdef __cplusplus
}
#endif

#endif /* DGNN_H */
