"""Host-side mirror of the reference's `dg_nn` public surface over the C ABI (include/dgnn.h).

Names, argument meaning and error behaviour follow the reference so that the parity tests read like
its own tests:

    reference (Rust)                                         here
    ------------------------------------------------------   -----------------------------------------
    dg_nn::Network::new() -> Option<Network>                  Network.new(exe_path=None) -> Network | None
      (src/libdg_nn/network.rs:92-127)
    loader::load(path)   (src/libdg_nn/loader.rs:102-116)     Network.load(path) / Network.from_json(text)
    Network::get_workspace(batch) -> Result<WorkspaceGuard>   Network.get_workspace(batch, device=0) -> WorkspaceGuard
      (network.rs:132-142)                                      (context manager; leaving it = Drop, network.rs:71-79)
    dg_nn::forward(&mut ws, &[f16]) -> Result<OutputMap<f16>> forward(ws, features) -> OutputMap
      (src/libdg_nn/graph.rs:123-158)
    OutputMap::unwrap() -> (Vec<f16>, Vec<f16>)               OutputMap.unwrap() -> (value, policy) np.float16 arrays
      (src/libdg_nn/output_map.rs:29-33)
    Network::synchronize()   (network.rs:145-159)             Network.synchronize()
    dg_nn::Error  (src/libdg_nn/error.rs:19-24)               Error(status) with .status in {MISSING_WEIGHTS, ...}
    dg_cuda::Device::{len, all} (src/libdg_cuda/devices.rs)   Device.len() / Device.all()

There is NO fallback: if libdgnn_hip.so is missing or does not load, importing this module raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Which build of the library this module binds.  Default: the SHIPPED one, lib/libdgnn_hip.so (exports include/dgnn.h + dgnn_host.h).
# DGNN_LIBRARY=<path>: another build (A/B measurements of two builds on one box; the ablations build).  load_full() returns a second
# instance of this module bound to lib/libdgnn_hip_test.so, the same kernel objects with the test-only entry points of
# include/dgnn_test.h exported as well (tests/conftest.py: api_full).
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libdgnn_hip_test.so")
LIB_PATH = globals().get("_FORCED_LIB_PATH") or os.environ.get("DGNN_LIBRARY") or os.path.join(_HERE, "lib", "libdgnn_hip.so")

OK = 0
MISSING_WEIGHTS = -1
MALFORMED_WEIGHTS = -2
HIP_ERROR = -3
BAD_ARGUMENT = -4
OUT_OF_MEMORY = -5
UNSUPPORTED = -6
BUSY = -7
COMM_ERROR = -8

NUM_FEATURES = 32
NUM_POINTS = 361
FEATURE_SIZE = 11552
NUM_MOVES = 362
PACKED_WORDS = 368
BOARD_RECORD_BYTES = 896
RCCL_UNIQUE_ID_BYTES = 128

# every symbol the three headers declare (include/dgnn.h, dgnn_host.h, dgnn_test.h): (name, restype, argtypes)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)
_ip = ctypes.POINTER(ctypes.c_int)
_fp = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
SYMBOLS = [
    ("dgnn_status_string", ctypes.c_char_p, [ctypes.c_int]),
    ("dgnn_last_hip_error", ctypes.c_int, []),
    ("dgnn_version", ctypes.c_char_p, []),
    ("dgnn_device_count", ctypes.c_int, [_ip]),
    ("dgnn_network_load", ctypes.c_int, [ctypes.c_char_p, _vpp]),
    ("dgnn_network_from_json", ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, _vpp]),
    ("dgnn_network_load_default", ctypes.c_int, [ctypes.c_char_p, _vpp]),
    ("dgnn_network_destroy", None, [_vp]),
    ("dgnn_network_info", ctypes.c_int, [_vp, _ip, _ip, _ip]),
    ("dgnn_network_set_softmax_temperature", ctypes.c_int, [_vp, ctypes.c_float]),
    ("dgnn_network_get_workspace", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_workspace_release", None, [_vp]),
    ("dgnn_network_synchronize", ctypes.c_int, [_vp]),
    ("dgnn_forward", ctypes.c_int, [_vp, _u16p, ctypes.c_size_t, _u16p, _u16p]),
    ("dgnn_forward_submit", ctypes.c_int, [_vp, _u16p, ctypes.c_size_t]),
    ("dgnn_forward_wait", ctypes.c_int, [_vp, _u16p, _u16p]),
    ("dgnn_workspace_batch_size", ctypes.c_int, [_vp, _ip, _ip]),
    ("dgnn_workspace_device_buffers", ctypes.c_int, [_vp, _vpp, _vpp, _vpp]),
    ("dgnn_workspace_host_buffers", ctypes.c_int, [_vp, ctypes.POINTER(_u16p), ctypes.POINTER(_u16p), ctypes.POINTER(_u16p)]),
    ("dgnn_workspace_stream", ctypes.c_int, [_vp, _vpp]),
    ("dgnn_network_tower_stats", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_workspace_conv_kernel", ctypes.c_char_p, [_vp]),
    ("dgnn_workspace_upload", ctypes.c_int, [_vp, _u16p, ctypes.c_size_t]),
    ("dgnn_workspace_download", ctypes.c_int, [_vp, _u16p, _u16p]),
    ("dgnn_workspace_download_pre", ctypes.c_int, [_vp, _u16p, _u16p]),
    ("dgnn_forward_resident", ctypes.c_int, [_vp]),
    ("dgnn_workspace_sync", ctypes.c_int, [_vp]),
    ("dgnn_forward_timed", ctypes.c_int, [_vp, ctypes.c_int, _fp, _fp, _ip]),
    ("dgnn_layer_conv3x3", ctypes.c_int, [ctypes.c_int, _u16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u16p, _u16p,
                                          ctypes.c_int, ctypes.c_float, ctypes.c_float, _u16p, ctypes.c_int, _u16p]),
    ("dgnn_network_weight_image", ctypes.c_int, [_vp, ctypes.c_int, _vpp, ctypes.POINTER(ctypes.c_size_t)]),
    ("dgnn_network_weight_checksum", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_network_set_weight_image", ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_size_t, ctypes.c_int]),
    ("dgnn_network_weight_checksum_device", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_network_weight_image_host", ctypes.c_int, [_vp, _vpp, ctypes.POINTER(ctypes.c_size_t)]),
    ("dgnn_forward_n", ctypes.c_int, [_vp, _u16p, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_forward_submit_n", ctypes.c_int, [_vp, _u16p, ctypes.c_int]),
    # host side of the boundary (scope row f-1)
    ("dgnn_symmetry_inverse", ctypes.c_int, [ctypes.c_int]),
    ("dgnn_symmetry_apply", ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    ("dgnn_policy_with_transform", ctypes.c_int, [_u16p, ctypes.c_int, _u16p]),
    ("dgnn_cache_create", ctypes.c_int, [ctypes.c_size_t, ctypes.c_int, _vpp]),
    ("dgnn_cache_destroy", None, [_vp]),
    ("dgnn_cache_fetch", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_cache_insert", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, _u16p]),
    ("dgnn_cache_stats", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_batcher_create", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_batcher_destroy", None, [_vp]),
    ("dgnn_batcher_push", ctypes.c_int, [_vp, _u16p, ctypes.c_uint64]),
    ("dgnn_batcher_get_batch", ctypes.c_int, [_vp, ctypes.c_int, _vpp]),
    ("dgnn_batcher_push_and_get_batch", ctypes.c_int, [_vp, _u16p, ctypes.c_uint64, _vpp]),
    ("dgnn_batcher_pending", ctypes.c_int, [_vp, _ip, _ip]),
    ("dgnn_batcher_stats", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_batch_size", ctypes.c_int, [_vp]),
    ("dgnn_batch_tags", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_batch_features", _u16p, [_vp]),
    ("dgnn_batch_forward", ctypes.c_int, [_vp, _u16p, _u16p]),
    ("dgnn_batch_destroy", None, [_vp]),
    ("dgnn_b85_encode_f16", ctypes.c_size_t, [_fp, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    ("dgnn_b85_decode", ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_size_t)]),
    ("dgnn_sgf_played", ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                          ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_char_p]),
    ("dgnn_config_uct_exp", ctypes.c_float, [ctypes.c_int]),
    ("dgnn_config_fpu_reduce", ctypes.c_float, [ctypes.c_int]),
    ("dgnn_config_lcb_critical_value", ctypes.c_float, [ctypes.c_int]),
    ("dgnn_config_vloss_cnt", ctypes.c_int, []),
    ("dgnn_intp_value", ctypes.c_float, [_ip, _fp, ctypes.c_int, ctypes.c_int]),
    ("dgnn_argmax_f32", ctypes.c_int, [_fp, ctypes.c_int]),
    ("dgnn_tree_node_new", _vp, [ctypes.c_int, ctypes.c_float, _fp]),
    ("dgnn_tree_node_free", None, [_vp]),
    ("dgnn_tree_probe", ctypes.c_int, [_vp, _vp, _vp, _vpp]),
    ("dgnn_trace_len", ctypes.c_int, [_vp]),
    ("dgnn_trace_get", ctypes.c_int, [_vp, ctypes.c_int, _vpp, _ip, _ip]),
    ("dgnn_trace_free", None, [_vp]),
    ("dgnn_tree_undo", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_tree_insert", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_float, _fp]),
    ("dgnn_tree_node_child", ctypes.c_int, [_vp, ctypes.c_int, _ip, _ip, _fp, _fp, _ip]),
    ("dgnn_tree_node_info", ctypes.c_int, [_vp, _ip, _ip, _ip, _ip, _fp]),
    ("dgnn_tree_node_prior", ctypes.c_int, [_vp, _fp]),
    ("dgnn_tree_node_disqualify", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_tree_node_softmax", ctypes.c_int, [_vp, _fp]),
    ("dgnn_tree_node_best", ctypes.c_int, [_vp, ctypes.c_float, ctypes.c_double, _fp, _ip]),
    ("dgnn_choose", ctypes.c_int, [_ip, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, _ip,
                                   ctypes.POINTER(ctypes.c_double)]),
    ("dgnn_pool_predictor", _vp, []),
    ("dgnn_hash_predictor", _vp, []),
    ("dgnn_search_run", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_synthetic_game_ops", _vp, []),
    ("dgnn_synthetic_board_new", _vp, [ctypes.c_uint64]),
    ("dgnn_synthetic_board_free", None, [_vp]),
    ("dgnn_selfplay_synthetic", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, _vp]),
    ("dgnn_go_board_new", _vp, [ctypes.c_float]),
    ("dgnn_go_board_clone", _vp, [_vp]),
    ("dgnn_go_board_free", None, [_vp]),
    ("dgnn_go_place", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_at", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_is_valid", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_to_move", ctypes.c_int, [_vp]),
    ("dgnn_go_count", ctypes.c_int, [_vp]),
    ("dgnn_go_hash", ctypes.c_uint64, [_vp]),
    ("dgnn_go_n_liberty", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_n_liberty_if", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_is_ladder_capture", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_is_ladder_escape", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_go_is_symmetric", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_go_features", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _u16p]),
    ("dgnn_go_game_ops", _vp, []),
    ("dgnn_selfplay_go", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _ip, ctypes.c_char_p, ctypes.c_size_t]),
    ("dgnn_search_run_batched", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_search_run_cached", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_go_scoring_ops", _vp, []),
    ("dgnn_go_benson", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte)]),
    ("dgnn_go_is_scorable", ctypes.c_int, [_vp]),
    ("dgnn_go_check", ctypes.c_int, [_vp]),
    ("dgnn_go_score", ctypes.c_int, [_vp, _ip, _ip]),
    ("dgnn_go_stone_status", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_ubyte)]),
    ("dgnn_go_result_sgf", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    ("dgnn_choose_f64", ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       _ip, ctypes.POINTER(ctypes.c_double)]),
    ("dgnn_full_forward", ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _fp, _fp]),
    ("dgnn_dirichlet_add", None, [_fp, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint64]),
    ("dgnn_tree_node_forward", _vp, [_vp, ctypes.c_int]),
    ("dgnn_tree_node_min_promote", ctypes.c_int, [_vp]),
    ("dgnn_rollout_limit_is_done", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_tree_node_set_prior", ctypes.c_int, [_vp, _fp]),
    ("dgnn_pool_create", ctypes.c_int, [_vp, ctypes.c_int, _vpp]),
    ("dgnn_pool_destroy", None, [_vp]),
    ("dgnn_pool_predict", ctypes.c_int, [_vp, _u16p, ctypes.c_int, _u16p, _u16p]),
    # round 2: packed features, layer-level dense / heads, RCCL inside the library, the multi-device pool
    ("dgnn_forward_packed", ctypes.c_int, [_vp, _u32p, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_forward_submit_packed", ctypes.c_int, [_vp, _u32p, ctypes.c_int]),
    ("dgnn_features_pack", ctypes.c_int, [_u16p, ctypes.c_int, _u32p]),
    ("dgnn_features_unpack", ctypes.c_int, [_u32p, ctypes.c_int, _u16p]),
    ("dgnn_workspace_packed_buffer", ctypes.c_int, [_vp, ctypes.POINTER(_u32p)]),
    ("dgnn_layer_dense", ctypes.c_int, [ctypes.c_int, _u16p, ctypes.c_int, ctypes.c_int, _u16p, _u16p, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_int, _u16p]),
    ("dgnn_layer_heads", ctypes.c_int, [ctypes.c_int, _u16p, _u16p, ctypes.c_int, _u16p, _u16p, _u16p, _u16p]),
    ("dgnn_network_weight_image_reserve", ctypes.c_int, [_vp, ctypes.c_int, _vpp, ctypes.POINTER(ctypes.c_size_t)]),
    ("dgnn_rccl_available", ctypes.c_int, []),
    ("dgnn_last_rccl_error", ctypes.c_int, []),
    ("dgnn_rccl_error_string", ctypes.c_char_p, [ctypes.c_int]),
    ("dgnn_rccl_unique_id", ctypes.c_int, [_u8p]),
    ("dgnn_rccl_comm_create", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _u8p, _vpp]),
    ("dgnn_rccl_comm_wrap", ctypes.c_int, [_vp, ctypes.c_int, _vpp]),
    ("dgnn_rccl_comm_info", ctypes.c_int, [_vp, _ip, _ip, _ip]),
    ("dgnn_rccl_comm_destroy", None, [_vp]),
    ("dgnn_broadcast_weights", ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    ("dgnn_broadcast_weights_devices", ctypes.c_int, [_vp, _ip, ctypes.c_int]),
    ("dgnn_batch_forward_on", ctypes.c_int, [_vp, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_pool_create_devices", ctypes.c_int, [_vp, _ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_pool_create_callbacks", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_pool_stats", ctypes.c_int, [_vp, _ip, _u64p, _u64p, _u64p, _u64p, _u64p, ctypes.c_int]),
    # round 3: compact board records (the feature planes are built on the device)
    ("dgnn_forward_boards", ctypes.c_int, [_vp, _u8p, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_forward_submit_boards", ctypes.c_int, [_vp, _u8p, ctypes.c_int]),
    ("dgnn_workspace_records_buffer", ctypes.c_int, [_vp, ctypes.POINTER(_u8p)]),
    ("dgnn_board_records_to_packed", ctypes.c_int, [_u8p, ctypes.c_int, _u32p]),
    ("dgnn_board_records_to_packed_device", ctypes.c_int, [ctypes.c_int, _u8p, ctypes.c_int, _u32p]),
    ("dgnn_go_board_record", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _u8p]),
    ("dgnn_forward_pipelined_timed", ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    ("dgnn_tree_node_to_pretty", ctypes.c_size_t, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    ("dgnn_cache_clear", ctypes.c_int, [_vp]),
    ("dgnn_cache_generation", ctypes.c_uint64, [_vp]),
    ("dgnn_cache_insert_at", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, _u16p]),
    ("dgnn_pool_generation", ctypes.c_uint64, [_vp]),
    ("dgnn_pool_set_adaptive_batches", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("dgnn_pool_set_coalescing", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_pool_set_min_fill", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    ("dgnn_pool_set_back_to_back", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_pool_batch_times", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    ("dgnn_pool_set_extra_batch_queue", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("dgnn_pool_coalesced", ctypes.c_uint64, [_vp]),
    ("dgnn_pool_adaptive_stats", ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    ("dgnn_pool_pause", ctypes.c_int, [_vp]),
    ("dgnn_pool_resume", ctypes.c_int, [_vp]),
    ("dgnn_network_weight_image_commit", ctypes.c_int, [_vp, ctypes.c_int]),
    ("dgnn_network_image_descriptor", ctypes.c_int, [_vp, _u64p]),
    ("dgnn_forward_pipelined_timed_n", ctypes.c_int, [_vpp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    ("dgnn_go_benson_flood", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte)]),
    ("dgnn_go_candidates", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte)]),
    ("dgnn_batcher_create_boards", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_batcher_push_board", ctypes.c_int, [_vp, _u8p, ctypes.c_uint64]),
    ("dgnn_batch_records", _u8p, [_vp]),
    ("dgnn_pool_create_devices_boards", ctypes.c_int, [_vp, _ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_pool_create_callbacks_boards", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_pool_create_board_callbacks", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    ("dgnn_hash_boards_predictor", _vp, []),
    ("dgnn_pool_predict_boards", ctypes.c_int, [_vp, _u8p, ctypes.c_int, _u16p, _u16p]),
    ("dgnn_pool_takes_boards", ctypes.c_int, [_vp]),
    ("dgnn_pool_boards_predictor", _vp, []),
    ("dgnn_search_run_boards", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dgnn_full_forward_boards", ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _fp, _fp]),
    ("dgnn_selfplay_go_boards", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _ip, ctypes.c_char_p, ctypes.c_size_t]),
    # round 5
    ("dgnn_device_pci_bus_id", ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    ("dgnn_test_fail_push_after", None, [ctypes.c_int]),
    ("dgnn_forward_stage_times", ctypes.c_int, [_vpp, ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_int, _fp, ctypes.POINTER(ctypes.c_double)]),
]


# include/dgnn_test.h: exported by the test build (and the ablations build) only; calling one of them through a binding of the shipped
# library raises AttributeError -- loudly, at the call
TEST_ONLY = frozenset('dgnn_forward_stage_times dgnn_workspace_device_buffers dgnn_pool_create_callbacks_boards dgnn_go_count dgnn_argmax_f32 dgnn_batch_destroy dgnn_batch_features dgnn_batch_forward dgnn_batch_forward_on dgnn_batch_records dgnn_batch_size dgnn_batch_tags dgnn_batcher_create dgnn_batcher_create_boards dgnn_batcher_destroy dgnn_batcher_get_batch dgnn_batcher_pending dgnn_batcher_push dgnn_batcher_push_and_get_batch dgnn_batcher_push_board dgnn_batcher_stats dgnn_board_records_to_packed_device dgnn_cache_generation dgnn_cache_insert_at dgnn_choose dgnn_choose_f64 dgnn_config_fpu_reduce dgnn_config_lcb_critical_value dgnn_config_uct_exp dgnn_config_vloss_cnt dgnn_dirichlet_add dgnn_forward_pipelined_timed dgnn_full_forward dgnn_full_forward_boards dgnn_go_at dgnn_go_benson dgnn_go_benson_flood dgnn_go_candidates dgnn_go_check dgnn_go_hash dgnn_go_is_ladder_capture dgnn_go_is_ladder_escape dgnn_go_is_scorable dgnn_go_is_symmetric dgnn_go_n_liberty dgnn_go_n_liberty_if dgnn_go_result_sgf dgnn_go_score dgnn_go_stone_status dgnn_go_to_move dgnn_intp_value dgnn_layer_conv3x3 dgnn_layer_dense dgnn_layer_heads dgnn_policy_with_transform dgnn_pool_adaptive_stats dgnn_pool_create dgnn_pool_set_adaptive_batches dgnn_pool_set_extra_batch_queue dgnn_pool_set_min_fill dgnn_rollout_limit_is_done dgnn_search_run dgnn_search_run_cached dgnn_selfplay_synthetic dgnn_symmetry_apply dgnn_symmetry_inverse dgnn_synthetic_board_free dgnn_synthetic_board_new dgnn_synthetic_game_ops dgnn_test_fail_push_after dgnn_trace_get dgnn_trace_len dgnn_tree_node_child dgnn_tree_node_disqualify dgnn_tree_node_min_promote dgnn_tree_node_prior dgnn_tree_undo dgnn_workspace_download_pre'.split())


def load_full():
    """A second instance of this module bound to the test build of the library (every symbol of the three headers)."""
    import importlib.util
    import sys
    name = __name__ + "_full"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    module = importlib.util.module_from_spec(spec)
    module.__dict__["_FORCED_LIB_PATH"] = os.environ.get("DGNN_LIBRARY") or TEST_LIB_PATH
    sys.modules[name] = module
    spec.loader.exec_module(module)
    return module


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python dream-go_amd/build.py` (hipcc, gfx950). "
                          "There is no CPU fallback for the product path.")
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's).  Two HIP runtimes in
    # one process do not work, and the dynamic linker only shares one copy if torch's is loaded FIRST (ours is
    # then resolved by SONAME against it).  So when torch is installed, import it before loading the library;
    # set DGNN_NO_TORCH=1 for a torch-free process (the library itself only needs the HIP runtime).
    if os.environ.get("DGNN_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)     # (local: the shipped and the test build may live in one process)
    full = hasattr(lib, "dgnn_layer_conv3x3")               # the test or the ablations build
    for name, restype, argtypes in SYMBOLS:
        if name in TEST_ONLY and not full:
            if hasattr(lib, name):
                raise ImportError(f"{LIB_PATH} exports {name}, which include/dgnn_test.h reserves for the test build")
            continue
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


class Error(Exception):
    """dg_nn::Error (src/libdg_nn/error.rs:19-24) as a status code."""

    def __init__(self, status, where=""):
        self.status = status
        self.hip_error = lib.dgnn_last_hip_error() if status == HIP_ERROR else 0
        self.rccl_error = lib.dgnn_last_rccl_error() if status == COMM_ERROR else 0
        text = lib.dgnn_status_string(status).decode("ascii")
        detail = f" (hipError_t {self.hip_error})" if self.hip_error else ""
        if status == COMM_ERROR:
            detail = f" (ncclResult_t {self.rccl_error}: {lib.dgnn_rccl_error_string(self.rccl_error).decode('ascii', 'replace')})"
        super().__init__(f"{where}: {text}{detail}" if where else text + detail)


def _check(status, where):
    if status != OK:
        raise Error(status, where)


def _as_u16(array, name):
    a = np.ascontiguousarray(array)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    if a.dtype != np.uint16:
        raise TypeError(f"{name} must be float16 (or its uint16 bit patterns), got {a.dtype}")
    return a.reshape(-1)


class Device:
    """dg_cuda::Device (src/libdg_cuda/devices.rs:38-73)."""

    def __init__(self, device_id):
        self.device_id = int(device_id)

    def id(self):
        return self.device_id

    @staticmethod
    def len():
        count = ctypes.c_int(0)
        _check(lib.dgnn_device_count(ctypes.byref(count)), "Device::len")
        return count.value

    @staticmethod
    def all():
        return [Device(i) for i in range(Device.len())]

    def pci_bus_id(self):
        """The physical GPU behind this runtime index ("0000:c1:00.0")."""
        out = ctypes.create_string_buffer(32)
        _check(lib.dgnn_device_pci_bus_id(int(self.device_id), out, len(out)), "Device::pci_bus_id")
        return out.value.decode("ascii")


class OutputMap:
    """dg_nn::OutputMap<f16> (src/libdg_nn/output_map.rs:15-34)."""

    def __init__(self, value, policy):
        self.value = value
        self.policy = policy

    def unwrap(self):
        return self.value, self.policy


class WorkspaceGuard:
    """dg_nn::WorkspaceGuard (src/libdg_nn/network.rs:54-79): returns the workspace to its pool when dropped."""

    def __init__(self, network, handle, batch_size, device):
        self._network = network      # keeps the network alive
        self._handle = handle
        self.batch_size = batch_size
        self.device = device

    @property
    def handle(self):
        if self._handle is None:
            raise ValueError("workspace already released")
        return self._handle

    def release(self):
        if self._handle is not None:
            lib.dgnn_workspace_release(self._handle)
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ---- measurement harness (device-resident) ----
    def upload(self, features):
        f = _as_u16(features, "features")
        _check(lib.dgnn_workspace_upload(self.handle, f.ctypes.data_as(_u16p), f.size), "upload")

    def download(self):
        value = np.empty(self.batch_size, dtype=np.uint16)
        policy = np.empty(self.batch_size * NUM_MOVES, dtype=np.uint16)
        _check(lib.dgnn_workspace_download(self.handle, value.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p)), "download")
        return OutputMap(value.view(np.float16), policy.view(np.float16))

    def download_pre(self):
        """(logits fp16 [batch, 362], pre-tanh value fp16 [batch]) of the last forward on this workspace (test-only entry)."""
        logits = np.empty(self.batch_size * NUM_MOVES, dtype=np.uint16)
        value = np.empty(self.batch_size, dtype=np.uint16)
        _check(lib.dgnn_workspace_download_pre(self.handle, logits.ctypes.data_as(_u16p), value.ctypes.data_as(_u16p)), "download_pre")
        return logits.view(np.float16).reshape(self.batch_size, NUM_MOVES), value.view(np.float16)

    def forward_resident(self):
        _check(lib.dgnn_forward_resident(self.handle), "forward_resident")

    def sync(self):
        _check(lib.dgnn_workspace_sync(self.handle), "sync")

    def forward_timed(self, steps):
        """-> (total_ms, tower_ms, tower_launches) measured with HIP events on the workspace's stream."""
        total, tower, launches = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int(0)
        _check(lib.dgnn_forward_timed(self.handle, steps, ctypes.byref(total), ctypes.byref(tower), ctypes.byref(launches)),
               "forward_timed")
        return total.value, tower.value, launches.value

    def device_buffers(self):
        f, v, p = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _check(lib.dgnn_workspace_device_buffers(self.handle, ctypes.byref(f), ctypes.byref(v), ctypes.byref(p)), "device_buffers")
        return f.value, v.value, p.value

    def conv_kernel(self):
        """Name of the residual-convolution kernel this workspace launches."""
        return lib.dgnn_workspace_conv_kernel(self.handle).decode("ascii")

    def stream(self):
        s = ctypes.c_void_p()
        _check(lib.dgnn_workspace_stream(self.handle, ctypes.byref(s)), "stream")
        return s.value


class Network:
    """dg_nn::Network (src/libdg_nn/network.rs:83-160)."""

    def __init__(self, handle):
        self._handle = handle

    @property
    def handle(self):
        if self._handle is None:
            raise ValueError("network already destroyed")
        return self._handle

    # -- constructors
    @classmethod
    def new(cls, exe_path=None):
        """Network::new(): search the default paths; None when no weights file is found (network.rs:92-127)."""
        out = ctypes.c_void_p()
        status = lib.dgnn_network_load_default(exe_path.encode() if exe_path else None, ctypes.byref(out))
        if status == MISSING_WEIGHTS:
            return None
        _check(status, "Network::new")      # the reference panics here
        return cls(out)

    @classmethod
    def load(cls, path):
        out = ctypes.c_void_p()
        _check(lib.dgnn_network_load(os.fsencode(path), ctypes.byref(out)), f"load({path})")
        return cls(out)

    @classmethod
    def from_json(cls, text):
        raw = text.encode("ascii") if isinstance(text, str) else bytes(text)
        out = ctypes.c_void_p()
        _check(lib.dgnn_network_from_json(raw, len(raw), ctypes.byref(out)), "from_json")
        return cls(out)

    # -- queries
    def info(self):
        b, c, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.dgnn_network_info(self.handle, ctypes.byref(b), ctypes.byref(c), ctypes.byref(s)), "info")
        return {"num_blocks": b.value, "num_channels": c.value, "num_samples": s.value}

    def set_softmax_temperature(self, temperature):
        _check(lib.dgnn_network_set_softmax_temperature(self.handle, float(temperature)), "set_softmax_temperature")

    def weight_checksum(self):
        c = ctypes.c_uint64()
        _check(lib.dgnn_network_weight_checksum(self.handle, ctypes.byref(c)), "weight_checksum")
        return c.value

    def weight_image_host(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib.dgnn_network_weight_image_host(self.handle, ctypes.byref(p), ctypes.byref(n)), "weight_image_host")
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,))

    def weight_image(self, device=0):
        """-> (device pointer, bytes) of the packed weight image on `device` (uploads it on first use)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib.dgnn_network_weight_image(self.handle, device, ctypes.byref(p), ctypes.byref(n)), "weight_image")
        return p.value, n.value

    def set_weight_image(self, device, src_ptr, num_bytes, src_on_device=True):
        _check(lib.dgnn_network_set_weight_image(self.handle, device, ctypes.c_void_p(src_ptr), num_bytes, int(src_on_device)),
               "set_weight_image")

    def weight_image_reserve(self, device=0):
        """-> (device pointer, bytes) without the upload when the device is touched for the first time (a broadcast receiver)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(lib.dgnn_network_weight_image_reserve(self.handle, device, ctypes.byref(p), ctypes.byref(n)), "weight_image_reserve")
        return p.value, n.value

    def broadcast_weights(self, comm, root=0):
        """One RCCL broadcast of the packed weight image inside the library; the receivers adopt the sender's gates / temperature."""
        _check(lib.dgnn_broadcast_weights(self.handle, comm.handle, int(root)), "broadcast_weights")

    def broadcast_weights_devices(self, devices=None, num_devices=None):
        """Single-process form: upload to devices[0], RCCL broadcast over xGMI to the others."""
        if devices is None:
            n = Device.len() if num_devices is None else int(num_devices)
            _check(lib.dgnn_broadcast_weights_devices(self.handle, None, n), "broadcast_weights_devices")
        else:
            arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib.dgnn_broadcast_weights_devices(self.handle, arr, len(devices)), "broadcast_weights_devices")

    def image_descriptor(self):
        """(image bytes, hash of architecture and packing): what the ranks of a broadcast must agree on before anything moves."""
        d = (ctypes.c_uint64 * 2)()
        _check(lib.dgnn_network_image_descriptor(self.handle, d), "image_descriptor")
        return int(d[0]), int(d[1])

    def weight_checksum_device(self, device=0):
        c = ctypes.c_uint64()
        _check(lib.dgnn_network_weight_checksum_device(self.handle, device, ctypes.byref(c)), "weight_checksum_device")
        return c.value

    # -- workspaces
    def tower_stats(self, device=-1):
        """{per_layer, whole_board, split, split_gave_up}: passes by the form their residual tower took (dgnn_network_tower_stats)."""
        counts = (ctypes.c_uint64 * 4)()
        _check(lib.dgnn_network_tower_stats(self.handle, int(device), counts), "tower_stats")
        return dict(zip(("per_layer", "whole_board", "split", "split_gave_up"), (int(c) for c in counts)))

    def get_workspace(self, batch_size, device=0):
        out = ctypes.c_void_p()
        _check(lib.dgnn_network_get_workspace(self.handle, int(device), int(batch_size), ctypes.byref(out)), "get_workspace")
        return WorkspaceGuard(self, out, int(batch_size), int(device))

    def synchronize(self):
        _check(lib.dgnn_network_synchronize(self.handle), "synchronize")

    def destroy(self):
        if self._handle is not None:
            lib.dgnn_network_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def forward(workspace, features):
    """dg_nn::forward (src/libdg_nn/graph.rs:123-158): features [batch * 11552] fp16 HWC -> OutputMap."""
    f = _as_u16(features, "features")
    value = np.empty(workspace.batch_size, dtype=np.uint16)
    policy = np.empty(workspace.batch_size * NUM_MOVES, dtype=np.uint16)
    _check(lib.dgnn_forward(workspace.handle, f.ctypes.data_as(_u16p), f.size,
                            value.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p)), "forward")
    return OutputMap(value.view(np.float16), policy.view(np.float16))


def layer_conv3x3(x, batch, board_size, cin, w_krsc, bias, cout, alpha1=1.0, alpha2=0.0, z=None, relu=True, device=0):
    """One Conv2d::forward / forward_skip (src/libdg_nn/layers/conv2d.rs:171-220) through the HIP kernel."""
    x = _as_u16(x, "x"); w = _as_u16(w_krsc, "w"); b = _as_u16(bias, "bias")
    zp = None
    if z is not None:
        z = _as_u16(z, "z")
        zp = z.ctypes.data_as(_u16p)
    y = np.empty(batch * board_size * board_size * cout, dtype=np.uint16)
    _check(lib.dgnn_layer_conv3x3(device, x.ctypes.data_as(_u16p), batch, board_size, cin, w.ctypes.data_as(_u16p),
                                  b.ctypes.data_as(_u16p), cout, alpha1, alpha2, zp, int(relu), y.ctypes.data_as(_u16p)),
           "layer_conv3x3")
    return y


def forward_n(workspace, features, count):
    """forward on the first `count` boards' worth of a workspace of larger capacity (include/dgnn.h: dgnn_forward_n)."""
    f = _as_u16(features, "features")
    if f.size != count * FEATURE_SIZE:
        raise Error(BAD_ARGUMENT, "forward_n")
    value = np.empty(count, dtype=np.uint16)
    policy = np.empty(count * NUM_MOVES, dtype=np.uint16)
    _check(lib.dgnn_forward_n(workspace.handle, f.ctypes.data_as(_u16p), int(count),
                              value.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p)), "forward_n")
    return OutputMap(value.view(np.float16), policy.view(np.float16))


def features_pack(features, count):
    """fp16 HWC features -> packed words [count][368] (include/dgnn.h: dgnn_forward_packed); Error(BAD_ARGUMENT) if not representable."""
    f = _as_u16(features, "features")
    if f.size != count * FEATURE_SIZE:
        raise Error(BAD_ARGUMENT, "features_pack")
    out = np.empty(count * PACKED_WORDS, dtype=np.uint32)
    _check(lib.dgnn_features_pack(f.ctypes.data_as(_u16p), int(count), out.ctypes.data_as(_u32p)), "features_pack")
    return out


def features_unpack(packed, count):
    p = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1)
    out = np.empty(count * FEATURE_SIZE, dtype=np.uint16)
    _check(lib.dgnn_features_unpack(p.ctypes.data_as(_u32p), int(count), out.ctypes.data_as(_u16p)), "features_unpack")
    return out.view(np.float16)


def forward_packed(workspace, packed, count):
    """dg_nn::forward from packed features (1 472 bytes per board instead of 23 104)."""
    p = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1)
    if p.size != count * PACKED_WORDS:
        raise Error(BAD_ARGUMENT, "forward_packed")
    value = np.empty(count, dtype=np.uint16)
    policy = np.empty(count * NUM_MOVES, dtype=np.uint16)
    _check(lib.dgnn_forward_packed(workspace.handle, p.ctypes.data_as(_u32p), int(count),
                                   value.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p)), "forward_packed")
    return OutputMap(value.view(np.float16), policy.view(np.float16))


def _as_records(records, count):
    r = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1)
    if r.size != count * BOARD_RECORD_BYTES:
        raise Error(BAD_ARGUMENT, "board records")
    return r


def forward_boards(workspace, records, count):
    """dg_nn::forward from compact board records (include/dgnn.h: dgnn_forward_boards): the planes are built on the device."""
    r = _as_records(records, count)
    value = np.empty(count, dtype=np.uint16)
    policy = np.empty(count * NUM_MOVES, dtype=np.uint16)
    _check(lib.dgnn_forward_boards(workspace.handle, r.ctypes.data_as(_u8p), int(count), value.ctypes.data_as(_u16p),
                                   policy.ctypes.data_as(_u16p)), "forward_boards")
    return OutputMap(value.view(np.float16), policy.view(np.float16))


def board_records_to_packed(records, count, device=None):
    """Records -> packed plane words [count * 368]: the host statement (device=None) or what the DEVICE kernel builds."""
    r = _as_records(records, count)
    out = np.empty(count * PACKED_WORDS, dtype=np.uint32)
    if device is None:
        _check(lib.dgnn_board_records_to_packed(r.ctypes.data_as(_u8p), int(count), out.ctypes.data_as(_u32p)), "board_records_to_packed")
    else:
        _check(lib.dgnn_board_records_to_packed_device(int(device), r.ctypes.data_as(_u8p), int(count), out.ctypes.data_as(_u32p)),
               "board_records_to_packed_device")
    return out


def forward_pipelined_timed(ws_a, ws_b, form, data, count, steps):
    """PCIe-inclusive evaluations per second with two workspaces in flight, the loop inside the library (include/dgnn.h).
    form: "planes" (fp16 [count * 11552]), "packed" (uint32 [count * 368]) or "records" (uint8 [count * 896])."""
    code = {"planes": 0, "packed": 1, "records": 2}[form]
    a = np.ascontiguousarray(data)
    seconds = ctypes.c_double()
    _check(lib.dgnn_forward_pipelined_timed(ws_a.handle, ws_b.handle, code, a.ctypes.data_as(ctypes.c_void_p), int(count), int(steps),
                                            ctypes.byref(seconds)), "forward_pipelined_timed")
    return count * steps / seconds.value


def forward_pipelined_timed_n(workspaces, form, data, count, steps):
    """forward_pipelined_timed with len(workspaces) submissions kept in flight."""
    code = {"planes": 0, "packed": 1, "records": 2}[form]
    a = np.ascontiguousarray(data)
    handles = (ctypes.c_void_p * len(workspaces))(*[w.handle for w in workspaces])
    seconds = ctypes.c_double()
    _check(lib.dgnn_forward_pipelined_timed_n(handles, len(workspaces), code, a.ctypes.data_as(ctypes.c_void_p), int(count), int(steps),
                                              ctypes.byref(seconds)), "forward_pipelined_timed_n")
    return count * steps / seconds.value


STAGES = ("h2d_records", "planes_kernel", "tower", "head_conv", "policy_gemm", "softmax_value", "d2h_outputs")


def forward_stage_times(workspaces, records, count, steps):
    """dgnn_forward_stage_times -> {stage: microseconds, ..., "host": {...}}: where one device batch's wall time goes."""
    handles = (ctypes.c_void_p * len(workspaces))(*[w.handle for w in workspaces])
    r = _as_records(records, count)
    stage = (ctypes.c_float * 7)()
    host = (ctypes.c_double * 4)()
    _check(lib.dgnn_forward_stage_times(handles, len(workspaces), r.ctypes.data_as(_u8p), int(count), int(steps), stage, host), "forward_stage_times")
    out = {name: float(stage[i]) for i, name in enumerate(STAGES)}
    out["host"] = {"submit_call_us": host[0], "submit_to_answers_us": host[1], "device_span_us": host[2], "host_on_top_us": host[3]}
    return out


def layer_dense(x, batch, k_in, w_in_out, bias, n_out, alpha=1.0, relu=False, device=0):
    """One Dense::forward (src/libdg_nn/layers/dense.rs:173-220) through the MFMA GEMM of the policy head."""
    x = _as_u16(x, "x"); w = _as_u16(w_in_out, "w"); b = _as_u16(bias, "bias")
    y = np.empty(batch * n_out, dtype=np.uint16)
    _check(lib.dgnn_layer_dense(device, x.ctypes.data_as(_u16p), int(batch), int(k_in), w.ctypes.data_as(_u16p), b.ctypes.data_as(_u16p),
                                int(n_out), float(alpha), int(relu), y.ctypes.data_as(_u16p)), "layer_dense")
    return y


def layer_heads(logits, value_features, batch, value_w, value_b, device=0):
    """Soft-max over 362 logits + value Dense 722 -> 1 + tanh through heads.hip's softmax_value_kernel -> (policy, value) uint16."""
    lg = _as_u16(logits, "logits"); vf = _as_u16(value_features, "value_features"); vw = _as_u16(value_w, "value_w"); vb = _as_u16(value_b, "value_b")
    policy = np.empty(batch * NUM_MOVES, dtype=np.uint16)
    value = np.empty(batch, dtype=np.uint16)
    _check(lib.dgnn_layer_heads(device, lg.ctypes.data_as(_u16p), vf.ctypes.data_as(_u16p), int(batch), vw.ctypes.data_as(_u16p),
                                vb.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p), value.ctypes.data_as(_u16p)), "layer_heads")
    return policy, value


class Comm:
    """An RCCL communicator owned by libdgnn_hip.so (include/dgnn.h: dgnn_rccl_comm_create)."""

    def __init__(self, handle):
        self.handle = handle

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES)()
        _check(lib.dgnn_rccl_unique_id(buf), "rccl_unique_id")
        return bytes(buf)

    @classmethod
    def create(cls, device, num_ranks, rank, unique_id):
        buf = (ctypes.c_uint8 * RCCL_UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        out = ctypes.c_void_p()
        _check(lib.dgnn_rccl_comm_create(int(device), int(num_ranks), int(rank), buf, ctypes.byref(out)), "rccl_comm_create")
        return cls(out)

    def info(self):
        n, r, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib.dgnn_rccl_comm_info(self.handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "rccl_comm_info")
        return {"num_ranks": n.value, "rank": r.value, "device": d.value}

    def destroy(self):
        if self.handle is not None:
            lib.dgnn_rccl_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------------
# Host side of the boundary (scope row f-1): the reference's libdg_mcts names over batcher.cpp
# ------------------------------------------------------------------------------------------------------------------

class Transform:
    """dg_go::utils::symmetry::Transform (src/libdg_go/utils/symmetry.rs:65-128), in the order of symmetry::ALL."""
    Identity, FlipLR, FlipUD, Transpose, TransposeAnti, Rot90, Rot180, Rot270 = range(8)
    ALL = tuple(range(8))

    @staticmethod
    def inverse(t):
        return lib.dgnn_symmetry_inverse(int(t))

    @staticmethod
    def apply(t, packed_index):
        return lib.dgnn_symmetry_apply(int(t), int(packed_index))


class Prediction:
    """dg_mcts::Prediction (src/libdg_mcts/predictor.rs:19-57): value fp16, policy 362 x fp16."""

    def __init__(self, value, policy):
        self._value = np.float16(value)
        self._policy = np.ascontiguousarray(policy, dtype=np.float16).reshape(NUM_MOVES)

    @staticmethod
    def with_transform(other, transform):
        out = np.empty(NUM_MOVES, dtype=np.uint16)
        src = other._policy.view(np.uint16)
        _check(lib.dgnn_policy_with_transform(src.ctypes.data_as(_u16p), int(transform), out.ctypes.data_as(_u16p)), "with_transform")
        return Prediction(other._value, out.view(np.float16))

    def value(self):
        return float(self._value)

    def winrate(self):
        return 0.5 * self.value() + 0.5

    def policy(self):
        return self._policy.astype(np.float32)


class LruCache:
    """The transposition table of NnPredictor (src/libdg_mcts/predictors/nn.rs:29-82 over lru_cache.rs:56-150)."""

    def __init__(self, capacity=200_001, num_shards=1):
        out = ctypes.c_void_p()
        _check(lib.dgnn_cache_create(int(capacity), int(num_shards), ctypes.byref(out)), "LruCache")
        self._handle = out

    def fetch(self, board_hash, to_move, symmetry):
        value = ctypes.c_uint16()
        policy = np.empty(NUM_MOVES, dtype=np.uint16)
        hit = lib.dgnn_cache_fetch(self._handle, int(board_hash), int(to_move), int(symmetry), ctypes.byref(value),
                                   policy.ctypes.data_as(_u16p))
        if hit < 0:
            raise Error(hit, "LruCache.fetch")
        if hit == 0:
            return None
        return Prediction(np.uint16(value.value).view(np.float16), policy.view(np.float16))

    def cache(self, board_hash, to_move, symmetry, response):
        bits = np.asarray(response._value).view(np.uint16)
        _check(lib.dgnn_cache_insert(self._handle, int(board_hash), int(to_move), int(symmetry), int(bits),
                                     response._policy.view(np.uint16).ctypes.data_as(_u16p)), "LruCache.cache")

    def clear(self):
        """Drops every entry (weight refresh: the table holds the old network's outputs) and starts a new generation."""
        _check(lib.dgnn_cache_clear(self._handle), "LruCache.clear")

    def generation(self):
        return int(lib.dgnn_cache_generation(self._handle))

    def cache_at(self, generation, board_hash, to_move, symmetry, response):
        """cache() of an answer whose miss was seen at `generation`: dropped when the table has been cleared since."""
        bits = np.asarray(response._value).view(np.uint16)
        _check(lib.dgnn_cache_insert_at(self._handle, int(generation), int(board_hash), int(to_move), int(symmetry), int(bits),
                                        response._policy.view(np.uint16).ctypes.data_as(_u16p)), "LruCache.cache_at")

    def stats(self):
        e, h, m = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib.dgnn_cache_stats(self._handle, ctypes.byref(e), ctypes.byref(h), ctypes.byref(m)), "LruCache.stats")
        return {"entries": e.value, "hits": h.value, "misses": m.value}

    def __len__(self):
        return self.stats()["entries"]

    def __del__(self):
        try:
            if self._handle is not None:
                lib.dgnn_cache_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class Batch:
    """dg_mcts::pool::Batch (src/libdg_mcts/pool/batch.rs:24-41)."""

    def __init__(self, handle):
        self._handle = handle

    def __len__(self):
        return lib.dgnn_batch_size(self._handle)

    def tags(self):
        out = np.empty(len(self), dtype=np.uint64)
        _check(lib.dgnn_batch_tags(self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))), "Batch.tags")
        return out

    def features(self):
        p = lib.dgnn_batch_features(self._handle)
        return np.ctypeslib.as_array(p, shape=(len(self) * FEATURE_SIZE,)).copy().view(np.float16)

    def forward(self):
        """-> (tags, OutputMap); the batch stops counting as alive (batch.rs:35-40)."""
        n = len(self)
        value = np.empty(n, dtype=np.uint16)
        policy = np.empty(n * NUM_MOVES, dtype=np.uint16)
        _check(lib.dgnn_batch_forward(self._handle, value.ctypes.data_as(_u16p), policy.ctypes.data_as(_u16p)), "Batch.forward")
        return self.tags(), OutputMap(value.view(np.float16), policy.view(np.float16))

    def destroy(self):
        if self._handle is not None:
            lib.dgnn_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Batcher:
    """dg_mcts::pool::Batcher (src/libdg_mcts/pool/batch.rs:58-123), one per (network, device)."""

    def __init__(self, network, max_batch_size, max_batches, device=0):
        self._network = network
        out = ctypes.c_void_p()
        _check(lib.dgnn_batcher_create(network.handle if network is not None else None, int(device), int(max_batch_size),
                                       int(max_batches), ctypes.byref(out)), "Batcher")
        self._handle = out
        self.max_batch_size = int(max_batch_size)

    def push(self, tag, features):
        f = _as_u16(features, "features")
        if f.size != FEATURE_SIZE:
            raise Error(BAD_ARGUMENT, "Batcher.push")
        _check(lib.dgnn_batcher_push(self._handle, f.ctypes.data_as(_u16p), int(tag)), "Batcher.push")

    def get_batch(self, min_batch_size):
        out = ctypes.c_void_p()
        _check(lib.dgnn_batcher_get_batch(self._handle, int(min_batch_size), ctypes.byref(out)), "Batcher.get_batch")
        return Batch(out) if out.value else None

    def push_and_get_batch(self, tag, features):
        f = _as_u16(features, "features")
        out = ctypes.c_void_p()
        _check(lib.dgnn_batcher_push_and_get_batch(self._handle, f.ctypes.data_as(_u16p), int(tag), ctypes.byref(out)),
               "Batcher.push_and_get_batch")
        return Batch(out) if out.value else None

    def pending(self):
        q, a = ctypes.c_int(), ctypes.c_int()
        _check(lib.dgnn_batcher_pending(self._handle, ctypes.byref(q), ctypes.byref(a)), "Batcher.pending")
        return q.value, a.value

    def stats(self):
        b, n = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib.dgnn_batcher_stats(self._handle, ctypes.byref(b), ctypes.byref(n)), "Batcher.stats")
        return {"boards": b.value, "batches": n.value}

    def __del__(self):
        try:
            if self._handle is not None:
                lib.dgnn_batcher_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class LeafPool:
    """A Predictor (src/libdg_mcts/predictor.rs:59-92) whose predict() coalesces across every calling thread."""

    def __init__(self, batcher, num_workers=2, _handle=None, _keep=None):
        self._batcher = batcher
        self._keep = _keep
        if _handle is not None:
            self._handle = _handle
            return
        out = ctypes.c_void_p()
        _check(lib.dgnn_pool_create(batcher._handle, int(num_workers), ctypes.byref(out)), "LeafPool")
        self._handle = out

    @classmethod
    def over_devices(cls, network, devices=None, max_batch_size=256, batches_per_device=2, workers_per_device=2, boards=False):
        """NnPredictor over several devices in ONE process (predictors/nn.rs:46-108): one queue, a batch goes to whichever
        device has a worker free.  devices=None: Device.all().  boards=True: the queue holds compact board records and the
        feature planes are built on the device (include/dgnn.h: dgnn_forward_boards)."""
        out = ctypes.c_void_p()
        arr, n = (None, 0) if devices is None else ((ctypes.c_int * len(devices))(*[int(d) for d in devices]), len(devices))
        create = lib.dgnn_pool_create_devices_boards if boards else lib.dgnn_pool_create_devices
        _check(create(network.handle, arr, n, int(max_batch_size), int(batches_per_device), int(workers_per_device), ctypes.byref(out)),
               "LeafPool.over_devices")
        return cls(None, _handle=out, _keep=network)

    @classmethod
    def over_callbacks(cls, predict_fn, predict_ctx=None, num_lanes=1, max_batch_size=256, batches_per_lane=2, workers_per_lane=1, boards=False):
        """The same pool over copies of a dgnn_predict_fn (e.g. lib.dgnn_hash_predictor()): no GPU needed.  boards=True: the queue
        holds board records, expanded for the callback by the host statement of the device kernel."""
        out = ctypes.c_void_p()
        create = lib.dgnn_pool_create_callbacks_boards if boards else lib.dgnn_pool_create_callbacks
        _check(create(predict_fn, predict_ctx, int(num_lanes), int(max_batch_size), int(batches_per_lane), int(workers_per_lane), ctypes.byref(out)),
               "LeafPool.over_callbacks")
        return cls(None, _handle=out, _keep=(predict_fn, predict_ctx))

    @classmethod
    def over_dry_device(cls, base_us=0.0, per_board_us=0.0, max_batch_size=256, batches_per_lane=2, workers_per_lane=2):
        """A pool of board records over the CPU stand-in (dgnn_hash_boards_predictor) behind a stand-in for the device: a batch of n
        boards takes base_us + per_board_us * n of wall time, one batch at a time.  No GPU: the host side of self-play alone."""
        out = ctypes.c_void_p()
        device = DryDevice(float(base_us), float(per_board_us)) if (base_us or per_board_us) else None
        _check(lib.dgnn_pool_create_board_callbacks(lib.dgnn_hash_boards_predictor(), ctypes.byref(device) if device is not None else None, 1,
                                                    int(max_batch_size), int(batches_per_lane), int(workers_per_lane), ctypes.byref(out)),
               "LeafPool.over_dry_device")
        return cls(None, _handle=out, _keep=device)

    def pause(self):
        """Returns when no worker holds a batch or a workspace (weights may be replaced); callers queue up until resume()."""
        _check(lib.dgnn_pool_pause(self._handle), "LeafPool.pause")

    def resume(self):
        _check(lib.dgnn_pool_resume(self._handle), "LeafPool.resume")

    def generation(self):
        """Completed pauses (weight refreshes) so far."""
        return int(lib.dgnn_pool_generation(self._handle))

    def set_extra_batch_queue(self, lo, hi, min_queued):
        """`lo` batches alive per lane, up to `hi` while at least `min_queued` boards are waiting (include/dgnn.h); min_queued 0: off."""
        _check(lib.dgnn_pool_set_extra_batch_queue(self._handle, int(lo), int(hi), int(min_queued)), "LeafPool.set_extra_batch_queue")

    def set_min_fill(self, min_boards, max_wait_us):
        """While another batch keeps the device busy a free slot waits for `min_boards` (or `max_wait_us`); 0: off (include/dgnn.h)."""
        _check(lib.dgnn_pool_set_min_fill(self._handle, int(min_boards), int(max_wait_us)), "LeafPool.set_min_fill")

    def set_back_to_back(self, lead_us, snap_lo=0, snap_hi=0):
        """A partial batch waits until `lead_us` before the batch on the device is expected to finish; with snap_lo a batch of
        snap_lo < n < snap_hi boards is cut to its snap_lo oldest (include/dgnn.h).  (0, 0): off."""
        _check(lib.dgnn_pool_set_back_to_back(self._handle, int(lead_us), int(snap_lo), int(snap_hi)), "LeafPool.set_back_to_back")

    def batch_times(self):
        """{first size of the bucket: microseconds a batch of that size took lately} for the sizes seen (include/dgnn.h)."""
        took = (ctypes.c_int * 17)()
        _check(lib.dgnn_pool_batch_times(self._handle, took), "LeafPool.batch_times")
        return {16 * k: int(took[k]) for k in range(17) if took[k] > 0}

    def set_coalescing(self, enabled=True):
        """Requests for a position that is already in flight follow it instead of being evaluated again (include/dgnn.h)."""
        _check(lib.dgnn_pool_set_coalescing(self._handle, 1 if enabled else 0), "LeafPool.set_coalescing")

    def coalesced(self):
        return int(lib.dgnn_pool_coalesced(self._handle))

    def set_adaptive_batches(self, lo, hi):
        """Between `lo` and `hi` batches alive per lane, chosen by the pool from the unique positions per second it measures
        (include/dgnn.h: dgnn_pool_set_adaptive_batches); (0, 0): the constant the pool was created with."""
        _check(lib.dgnn_pool_set_adaptive_batches(self._handle, int(lo), int(hi)), "LeafPool.set_adaptive_batches")

    def adaptive_stats(self):
        alive, sw, hashed, rep = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        lo, hi = ctypes.c_double(), ctypes.c_double()
        _check(lib.dgnn_pool_adaptive_stats(self._handle, ctypes.byref(alive), ctypes.byref(sw), ctypes.byref(hashed), ctypes.byref(rep),
                                            ctypes.byref(lo), ctypes.byref(hi)), "LeafPool.adaptive_stats")
        return {"alive_per_lane": alive.value, "switches": sw.value, "boards_hashed": hashed.value, "boards_repeated": rep.value,
                "unique_per_sec_lo": lo.value, "unique_per_sec_hi": hi.value}

    @property
    def takes_boards(self):
        return bool(lib.dgnn_pool_takes_boards(self._handle))

    def predict_boards(self, records, batch_size):
        r = _as_records(records, batch_size)
        value = np.empty(batch_size, dtype=np.uint16)
        policy = np.empty(batch_size * NUM_MOVES, dtype=np.uint16)
        _check(lib.dgnn_pool_predict_boards(self._handle, r.ctypes.data_as(_u8p), int(batch_size), value.ctypes.data_as(_u16p),
                                            policy.ctypes.data_as(_u16p)), "LeafPool.predict_boards")
        v, p = value.view(np.float16), policy.view(np.float16).reshape(batch_size, NUM_MOVES)
        return [Prediction(v[i], p[i]) for i in range(batch_size)]

    def stats(self):
        n, calls, boards, batches = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        lane_boards, lane_batches = (ctypes.c_uint64 * 64)(), (ctypes.c_uint64 * 64)()
        _check(lib.dgnn_pool_stats(self._handle, ctypes.byref(n), ctypes.byref(calls), ctypes.byref(boards), ctypes.byref(batches),
                                   lane_boards, lane_batches, 64), "LeafPool.stats")
        k = min(n.value, 64)
        return {"lanes": n.value, "calls": calls.value, "boards": boards.value, "batches": batches.value,
                "lane_boards": list(lane_boards[:k]), "lane_batches": list(lane_batches[:k])}

    def predict(self, features, batch_size):
        f = _as_u16(features, "features")
        if f.size != batch_size * FEATURE_SIZE:
            raise Error(BAD_ARGUMENT, "LeafPool.predict")
        value = np.empty(batch_size, dtype=np.uint16)
        policy = np.empty(batch_size * NUM_MOVES, dtype=np.uint16)
        _check(lib.dgnn_pool_predict(self._handle, f.ctypes.data_as(_u16p), int(batch_size), value.ctypes.data_as(_u16p),
                                     policy.ctypes.data_as(_u16p)), "LeafPool.predict")
        v, p = value.view(np.float16), policy.view(np.float16).reshape(batch_size, NUM_MOVES)
        return [Prediction(v[i], p[i]) for i in range(batch_size)]

    def close(self):
        if self._handle is not None:
            lib.dgnn_pool_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------------
# Formats either side of the path (scope row f-4)
# ------------------------------------------------------------------------------------------------------------------

def b85_encode(values):
    """dg_utils::b85::encode (src/libdg_utils/b85.rs:141-165): fp32 values -> fp16 -> RFC 1924 base85 text."""
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    need = lib.dgnn_b85_encode_f16(v.ctypes.data_as(_fp), v.size, None, 0)
    if need == 0:
        raise Error(BAD_ARGUMENT, "b85_encode")
    buf = ctypes.create_string_buffer(need)
    lib.dgnn_b85_encode_f16(v.ctypes.data_as(_fp), v.size, buf, need)
    return buf.value.decode("ascii")


def b85_decode(text, dtype=np.float16):
    """dg_utils::b85::decode (b85.rs:101-132) -> array of `dtype` (float16 or float32), or None for an invalid character."""
    raw = text.encode("ascii") if isinstance(text, str) else bytes(text)
    n = ctypes.c_size_t()
    if lib.dgnn_b85_decode(raw, len(raw), None, 0, ctypes.byref(n)) != OK:
        return None
    out = np.empty(n.value, dtype=np.uint8)
    _check(lib.dgnn_b85_decode(raw, len(raw), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), out.size, ctypes.byref(n)), "b85_decode")
    return out.view(dtype)


def sgf_played(color, packed_index, value=None, num_rollout=1, softmax=None, prior_index=-1, explain=""):
    """Display of dg_mcts::self_play::Played (src/libdg_mcts/self_play.rs:187-213)."""
    sm = None
    if softmax is not None:
        sm = np.ascontiguousarray(softmax, dtype=np.float32).reshape(NUM_MOVES)
    args = (int(color), int(packed_index), int(num_rollout), sm.ctypes.data_as(_fp) if sm is not None else None,
            int(value is not None), float(value if value is not None else 0.0), int(prior_index), explain.encode("utf-8"))
    need = lib.dgnn_sgf_played(None, 0, *args)
    buf = ctypes.create_string_buffer(need)
    lib.dgnn_sgf_played(buf, need, *args)
    return buf.value.decode("utf-8")


# ------------------------------------------------------------------------------------------------------------------
# Search tree (scope row f-2, first part): dg_mcts::tree over tree.cpp
# ------------------------------------------------------------------------------------------------------------------

BLACK, WHITE = 0, 1
PROBE_FOUND, PROBE_NO_RESULT, PROBE_CONFLICT = 1, 0, 2


class config:
    """dg_utils::config accessors used by the tree (src/libdg_utils/config.rs:181-195, 289-330)."""
    VLOSS_CNT = lib.dgnn_config_vloss_cnt() if hasattr(lib, "dgnn_config_vloss_cnt") else 32      # (config.rs:187; the accessor is a test-build export)
    get_uct_exp = staticmethod(lambda visits: lib.dgnn_config_uct_exp(int(visits)))
    get_fpu_reduce = staticmethod(lambda visits: lib.dgnn_config_fpu_reduce(int(visits)))
    get_lcb_critical_value = staticmethod(lambda visits: lib.dgnn_config_lcb_critical_value(int(visits)))

    @staticmethod
    def get_intp_value(points, x):
        xs = np.asarray([p[0] for p in points], dtype=np.int32)
        ys = np.asarray([p[1] for p in points], dtype=np.float32)
        return lib.dgnn_intp_value(xs.ctypes.data_as(_ip), ys.ctypes.data_as(_fp), len(points), int(x))


def argmax_f32(values):
    """dg_mcts::asm::argmax_f32 (src/libdg_mcts/asm/argmax.rs:24-78); None when the maximum is not finite."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    i = lib.dgnn_argmax_f32(v.ctypes.data_as(_fp), v.size)
    if i == -2:
        raise Error(BAD_ARGUMENT, "argmax_f32")
    return None if i < 0 else i


def choose(items, cutoff_percentile, temperature, at):
    """dg_mcts::choose::choose (src/libdg_mcts/choose.rs:54-88) -> (index, cumulative weight) or None."""
    v = np.ascontiguousarray(items, dtype=np.int32)
    index, cum = ctypes.c_int(), ctypes.c_double()
    r = lib.dgnn_choose(v.ctypes.data_as(_ip), v.size, cutoff_percentile, temperature, at, ctypes.byref(index), ctypes.byref(cum))
    if r < 0:
        raise Error(r, "choose")
    return (index.value, cum.value) if r == 1 else None


class Child:
    def __init__(self, count, vcount, value, value_std, has_child):
        self._c, self._v, self._value, self._std, self._has = count, vcount, value, value_std, has_child

    def count(self): return self._c
    def vcount(self): return self._v
    def value(self): return self._value
    def value_std(self): return self._std
    def has_node(self): return bool(self._has)


class Trace:
    """NodeTrace (src/libdg_mcts/tree.rs:1387): [(node handle, colour, child index)]."""

    def __init__(self, handle):
        self._handle = handle

    def __len__(self):
        return lib.dgnn_trace_len(self._handle)

    def __getitem__(self, i):
        node, color, index = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        _check(lib.dgnn_trace_get(self._handle, int(i), ctypes.byref(node), ctypes.byref(color), ctypes.byref(index)), "Trace[]")
        return node.value, color.value, index.value

    def __del__(self):
        try:
            if self._handle is not None:
                lib.dgnn_trace_free(self._handle)
                self._handle = None
        except Exception:
            pass


class Node:
    """dg_mcts::tree::Node (src/libdg_mcts/tree.rs:1027-1384); owns its sub-tree."""

    def __init__(self, to_move, value, prior):
        p = np.ascontiguousarray(prior, dtype=np.float32).reshape(-1)
        if p.size < NUM_MOVES:
            raise Error(BAD_ARGUMENT, "Node")
        self._handle = lib.dgnn_tree_node_new(int(to_move), float(value), p.ctypes.data_as(_fp))
        if not self._handle:
            raise Error(OUT_OF_MEMORY, "Node")

    def with_(self, index):
        c, v, has = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        val, std = ctypes.c_float(), ctypes.c_float()
        _check(lib.dgnn_tree_node_child(self._handle, int(index), ctypes.byref(c), ctypes.byref(v), ctypes.byref(val),
                                        ctypes.byref(std), ctypes.byref(has)), "Node.with")
        return Child(c.value, v.value, val.value, std.value, has.value)

    def _info(self):
        tm, tc, vc, pc, iv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_float()
        _check(lib.dgnn_tree_node_info(self._handle, ctypes.byref(tm), ctypes.byref(tc), ctypes.byref(vc), ctypes.byref(pc),
                                       ctypes.byref(iv)), "Node.info")
        return tm.value, tc.value, vc.value, pc.value, iv.value

    to_move = property(lambda self: self._info()[0])
    total_count = property(lambda self: self._info()[1])
    vtotal_count = property(lambda self: self._info()[2])
    pass_count = property(lambda self: self._info()[3])
    initial_value = property(lambda self: self._info()[4])

    @property
    def prior(self):
        out = np.empty(NUM_MOVES, dtype=np.float32)
        _check(lib.dgnn_tree_node_prior(self._handle, out.ctypes.data_as(_fp)), "Node.prior")
        return out

    def disqualify(self, index):
        _check(lib.dgnn_tree_node_disqualify(self._handle, int(index)), "Node.disqualify")

    def forward(self, index):
        """Node::forward (tree.rs:1225-1250): consumes this node, returns the sub-tree below `index` as the new root (or None)."""
        handle = lib.dgnn_tree_node_forward(self._handle, int(index))
        self._handle = None
        if not handle:
            return None
        node = Node.__new__(Node)
        node._handle = handle
        return node

    def min_promote_rollouts(self):
        return lib.dgnn_tree_node_min_promote(self._handle)

    def is_done(self, limit):
        """time_control::is_done with RolloutLimit::new(limit) (time_control/mod.rs:86-101)."""
        return lib.dgnn_rollout_limit_is_done(self._handle, int(limit)) == 1

    def set_prior(self, prior):
        p = np.ascontiguousarray(prior, dtype=np.float32).reshape(-1)
        _check(lib.dgnn_tree_node_set_prior(self._handle, p.ctypes.data_as(_fp)), "Node.set_prior")

    def softmax(self):
        out = np.empty(NUM_MOVES, dtype=np.float32)
        _check(lib.dgnn_tree_node_softmax(self._handle, out.ctypes.data_as(_fp)), "Node.softmax")
        return out

    def to_pretty(self):
        """tree::to_pretty (tree.rs:1683-1731): the search summary stored with a move as C[]."""
        buf = ctypes.create_string_buffer(4096)
        lib.dgnn_tree_node_to_pretty(self._handle, buf, len(buf))
        return buf.value.decode("ascii")

    def best(self, temperature, at=0.0):
        value, index = ctypes.c_float(), ctypes.c_int()
        _check(lib.dgnn_tree_node_best(self._handle, float(temperature), float(at), ctypes.byref(value), ctypes.byref(index)), "Node.best")
        return value.value, index.value

    def __del__(self):
        try:
            if self._handle:
                lib.dgnn_tree_node_free(self._handle)
                self._handle = None
        except Exception:
            pass


def probe(root):
    """dg_mcts::tree::probe (src/libdg_mcts/tree.rs:1421-1470) without a board: -> Trace, or PROBE_NO_RESULT / PROBE_CONFLICT."""
    out = ctypes.c_void_p()
    r = lib.dgnn_tree_probe(root._handle, None, None, ctypes.byref(out))
    if r == PROBE_FOUND:
        return Trace(out)
    if r in (PROBE_NO_RESULT, PROBE_CONFLICT):
        return r
    raise Error(r, "probe")


def insert(trace, color, value, prior):
    """dg_mcts::tree::insert (tree.rs:1482-1510)."""
    p = np.ascontiguousarray(prior, dtype=np.float32).reshape(-1)
    _check(lib.dgnn_tree_insert(trace._handle, int(color), float(value), p.ctypes.data_as(_fp)), "insert")


def undo(trace, undo_expanding):
    _check(lib.dgnn_tree_undo(trace._handle, int(bool(undo_expanding))), "undo")


# ---- search driver (scope row f-2, second part): search.cpp ----

class DryDevice(ctypes.Structure):
    _fields_ = [("base_us", ctypes.c_double), ("per_board_us", ctypes.c_double)]


class SelfplayParams(ctypes.Structure):
    _fields_ = [("num_games", ctypes.c_int), ("num_parallel", ctypes.c_int), ("first_game", ctypes.c_int), ("game_stride", ctypes.c_int),
                ("max_moves", ctypes.c_int), ("num_rollouts", ctypes.c_int), ("num_ex_it_rollouts", ctypes.c_int),
                ("threads_per_search", ctypes.c_int), ("leaves_per_thread", ctypes.c_int), ("cache_entries_per_game", ctypes.c_int),
                ("seed", ctypes.c_uint64), ("host_threads", ctypes.c_int), ("opening_len", ctypes.c_int), ("opening", ctypes.c_void_p),
                ("opening_games", ctypes.c_int), ("window", ctypes.c_void_p), ("on_game", ctypes.c_void_p), ("on_game_ctx", ctypes.c_void_p)]


WINDOW_EDGE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)
GAME_RECORD_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)


class SelfplayWindow(ctypes.Structure):
    """dgnn_selfplay_window (include/dgnn.h): a timed window over a steady state of games in flight."""
    _fields_ = [("ramp_seconds", ctypes.c_double), ("seconds", ctypes.c_double), ("on_edge", WINDOW_EDGE_FN), ("ctx", ctypes.c_void_p),
                ("measured_seconds", ctypes.c_double), ("moves", ctypes.c_uint64), ("evals", ctypes.c_uint64), ("cache_hits", ctypes.c_uint64),
                ("conflicts", ctypes.c_uint64), ("games_started", ctypes.c_uint64), ("games_completed", ctypes.c_uint64),
                ("games_ended", ctypes.c_uint64), ("stones_completed", ctypes.c_uint64), ("ex_it_searches", ctypes.c_uint64)]


class SelfplayStats(ctypes.Structure):
    _fields_ = [("seconds", ctypes.c_double), ("games", ctypes.c_uint64), ("moves", ctypes.c_uint64),
                ("evals", ctypes.c_uint64), ("conflicts", ctypes.c_uint64), ("cache_hits", ctypes.c_uint64),
                ("games_ended", ctypes.c_uint64), ("ex_it_searches", ctypes.c_uint64)]


class SyntheticBoard:
    """The rule-free synthetic game of search.cpp (NOT Go; for driving the search and the pipeline)."""

    def __init__(self, seed):
        self._handle = lib.dgnn_synthetic_board_new(int(seed))

    def __del__(self):
        try:
            if self._handle:
                lib.dgnn_synthetic_board_free(self._handle)
                self._handle = None
        except Exception:
            pass


PREDICT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, _u16p, ctypes.c_int, _u16p, _u16p)


def python_predictor(fn):
    """Wraps `fn(features uint16 [n * 11552], n) -> (value fp16 [n], policy fp16 [n * 362])` as a dgnn_predict_fn
    (tests: the CPU oracle as the predictor behind the search)."""
    def thunk(_ctx, features, n, value_out, policy_out):
        try:
            f = np.ctypeslib.as_array(features, shape=(n * FEATURE_SIZE,))
            value, policy = fn(f, n)
            np.ctypeslib.as_array(value_out, shape=(n,))[:] = np.asarray(value).view(np.uint16).reshape(n)
            np.ctypeslib.as_array(policy_out, shape=(n * NUM_MOVES,))[:] = np.asarray(policy).view(np.uint16).reshape(n * NUM_MOVES)
            return OK
        except Exception:   # noqa: BLE001
            return BAD_ARGUMENT
    return PREDICT_FN(thunk)


def _predictor(pool):
    """(callback, context) for search_run / selfplay_synthetic: a LeafPool, a python_predictor, or None for the
    deterministic CPU stand-in."""
    if pool is None:
        return lib.dgnn_hash_predictor(), None
    if isinstance(pool, PREDICT_FN):
        return ctypes.cast(pool, ctypes.c_void_p), None
    if pool.takes_boards:
        return lib.dgnn_pool_boards_predictor(), pool._handle
    return lib.dgnn_pool_predictor(), pool._handle


def _takes_boards(pool):
    return isinstance(pool, LeafPool) and pool.takes_boards


def _game_ops(board, scoring=False):
    if isinstance(board, Board):
        return lib.dgnn_go_scoring_ops() if scoring else lib.dgnn_go_game_ops()
    return lib.dgnn_synthetic_game_ops()


def search_run(root, board, num_rollouts, num_threads=1, seed=0, pool=None, cache=None, scoring=False, leaves_per_thread=1):
    """One search on `root` from a Board or a SyntheticBoard (worker_thread.rs:50-176 for a single search) ->
    (predictor calls, probe conflicts), or (predictor calls, probe conflicts, cache hits) when a LruCache is given.
    scoring=True searches a Board as a ScoringSearch (options.rs:141-163) instead of a StandardSearch."""
    fn, ctx = _predictor(pool)
    evals, conflicts, hits = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    run = lib.dgnn_search_run_boards if _takes_boards(pool) else lib.dgnn_search_run_batched
    _check(run(root._handle, _game_ops(board, scoring), board._handle, fn, ctx,
               cache._handle if cache is not None else None, int(num_rollouts), int(num_threads), int(leaves_per_thread),
               int(seed), ctypes.byref(evals), ctypes.byref(conflicts), ctypes.byref(hits)), "search_run")
    return (evals.value, conflicts.value) if cache is None else (evals.value, conflicts.value, hits.value)


def full_forward(board, to_move, pool=None, cache=None, scoring=False):
    """dg_mcts::full_forward (src/libdg_mcts/lib.rs:83-130) -> (win rate, prior [362] with -inf for non-candidates, symmetries evaluated)."""
    fn, ctx = _predictor(pool)
    value = ctypes.c_float()
    policy = np.empty(368, dtype=np.float32)
    call = lib.dgnn_full_forward_boards if _takes_boards(pool) else lib.dgnn_full_forward
    n = call(_game_ops(board, scoring), board._handle, int(to_move), fn, ctx, cache._handle if cache is not None else None,
             ctypes.byref(value), policy.ctypes.data_as(_fp))
    if n < 0:
        raise Error(n, "full_forward")
    return value.value, policy[:NUM_MOVES].copy(), n


def dirichlet_add(x, shape, beta, seed):
    """dg_mcts::dirichlet::add_ex (src/libdg_mcts/dirichlet.rs:40-70) with a seeded generator; returns a new array."""
    v = np.array(x, dtype=np.float32).reshape(-1)
    lib.dgnn_dirichlet_add(v.ctypes.data_as(_fp), v.size, float(shape), float(beta), int(seed))
    return v


def selfplay_synthetic(num_games, num_moves, num_rollouts, threads_per_search=1, seed=0, pool=None):
    fn, ctx = _predictor(pool)
    stats = SelfplayStats()
    _check(lib.dgnn_selfplay_synthetic(fn, ctx, int(num_games), int(num_moves), int(num_rollouts), int(threads_per_search),
                                       int(seed), ctypes.byref(stats)), "selfplay_synthetic")
    return {"seconds": stats.seconds, "games": stats.games, "moves": stats.moves, "evals": stats.evals, "conflicts": stats.conflicts}


# ------------------------------------------------------------------------------------------------------------------
# Go board, ladders, feature planes (scope row f-3): dg_go over go.cpp
# ------------------------------------------------------------------------------------------------------------------

class Board:
    """dg_go::Board (src/libdg_go/board.rs:28-176) + BoardFast liberties + the ladder reader + features::V1."""

    def __init__(self, komi=7.5, _handle=None):
        self._handle = _handle if _handle is not None else lib.dgnn_go_board_new(float(komi))
        if not self._handle:
            raise Error(OUT_OF_MEMORY, "Board")

    def clone(self):
        return Board(_handle=lib.dgnn_go_board_clone(self._handle))

    def place(self, color, x, y):
        _check(lib.dgnn_go_place(self._handle, int(color), int(x), int(y)), "Board.place")

    def at(self, x, y):
        c = lib.dgnn_go_at(self._handle, int(x), int(y))
        return None if c < 0 else c

    def is_valid(self, color, x, y):
        return bool(lib.dgnn_go_is_valid(self._handle, int(color), int(x), int(y)))

    def to_move(self):
        return lib.dgnn_go_to_move(self._handle)

    def count(self):
        return lib.dgnn_go_count(self._handle)

    def zobrist_hash(self):
        return lib.dgnn_go_hash(self._handle)

    def get_n_liberty(self, x, y):
        return lib.dgnn_go_n_liberty(self._handle, int(x), int(y))

    def get_n_liberty_if(self, color, x, y):
        return lib.dgnn_go_n_liberty_if(self._handle, int(color), int(x), int(y))

    def is_ladder_capture(self, color, x, y):
        return bool(lib.dgnn_go_is_ladder_capture(self._handle, int(color), int(x), int(y)) == 1)

    def is_ladder_escape(self, color, x, y):
        return bool(lib.dgnn_go_is_ladder_escape(self._handle, int(color), int(x), int(y)) == 1)

    def is_symmetric(self, transform):
        return bool(lib.dgnn_go_is_symmetric(self._handle, int(transform)))

    def benson(self, color):
        """BensonImpl::new(board, color) (utils/benson.rs:62-190) -> uint8 [19, 19] indexed [y, x]: 1 = is_alive, 2 = is_eye."""
        out = np.empty(NUM_POINTS, dtype=np.uint8)
        _check(lib.dgnn_go_benson(self._handle, int(color), out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))), "Board.benson")
        return out.reshape(19, 19)

    def benson_flood(self, color):
        """The same by the reference's own flood-fill steps (the checker of the bit-board form)."""
        out = np.empty(NUM_POINTS, dtype=np.uint8)
        _check(lib.dgnn_go_benson_flood(self._handle, int(color), out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))), "Board.benson_flood")
        return out.reshape(19, 19)

    def candidates(self, to_move, scoring=False):
        """The candidate moves of a StandardSearch / ScoringSearch (options.rs:44-214) -> bool [362], index 361 = pass."""
        out = np.empty(NUM_MOVES, dtype=np.uint8)
        _check(lib.dgnn_go_candidates(self._handle, int(to_move), int(bool(scoring)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))), "Board.candidates")
        return out.astype(bool)

    def is_scorable(self):
        return bool(lib.dgnn_go_is_scorable(self._handle))

    def check(self):
        """True when the incremental block / liberty bookkeeping equals a from-scratch flood fill."""
        return bool(lib.dgnn_go_check(self._handle))

    def get_score(self):
        black, white = ctypes.c_int(), ctypes.c_int()
        _check(lib.dgnn_go_score(self._handle, ctypes.byref(black), ctypes.byref(white)), "Board.get_score")
        return black.value, white.value

    def get_stone_status(self):
        """Score::get_stone_status(board, board) -> uint8 [19, 19] bit masks (STONE_ALIVE / DEAD / SEKI, TERRITORY_BLACK / WHITE)."""
        out = np.empty(NUM_POINTS, dtype=np.uint8)
        _check(lib.dgnn_go_stone_status(self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))), "Board.get_stone_status")
        return out.reshape(19, 19)

    def result_sgf(self):
        """(RE[] value, "TB[..]TW[..]") of GameResult::Ended (game_result.rs:35-93)."""
        winner, territory = ctypes.create_string_buffer(32), ctypes.create_string_buffer(4 * 361 + 8)
        _check(lib.dgnn_go_result_sgf(self._handle, winner, len(winner), territory, len(territory)), "Board.result_sgf")
        return winner.value.decode("ascii"), territory.value.decode("ascii")

    def record(self, to_move, symmetry=0):
        """The compact record of the position for forward_boards (include/dgnn.h) -> uint8 [896]."""
        out = np.empty(BOARD_RECORD_BYTES, dtype=np.uint8)
        _check(lib.dgnn_go_board_record(self._handle, int(to_move), int(symmetry), out.ctypes.data_as(_u8p)), "Board.record")
        return out

    def get_features(self, to_move, symmetry=0):
        """features::V1::get_features::<HWC, f16> -> float16 [361, 32]."""
        out = np.empty(FEATURE_SIZE, dtype=np.uint16)
        _check(lib.dgnn_go_features(self._handle, int(to_move), int(symmetry), out.ctypes.data_as(_u16p)), "Board.get_features")
        return out.view(np.float16).reshape(NUM_POINTS, NUM_FEATURES)

    def __del__(self):
        try:
            if self._handle:
                lib.dgnn_go_board_free(self._handle)
                self._handle = None
        except Exception:
            pass


STONE_ALIVE, STONE_DEAD, STONE_SEKI, TERRITORY_BLACK, TERRITORY_WHITE = 1, 2, 4, 8, 16


def choose_f64(items, cutoff_percentile, temperature, at):
    """choose (src/libdg_mcts/choose.rs:54-88) on real weights -> (index, cumulative value) or None."""
    v = np.ascontiguousarray(items, dtype=np.float64)
    index, cum = ctypes.c_int(), ctypes.c_double()
    r = lib.dgnn_choose_f64(v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size, float(cutoff_percentile), float(temperature), float(at),
                            ctypes.byref(index), ctypes.byref(cum))
    if r < 0:
        raise Error(r, "choose_f64")
    return (index.value, cum.value) if r == 1 else None


def selfplay_go(num_games, num_moves, num_rollouts, threads_per_search=1, seed=0, pool=None, cache=None, return_moves=False,
                return_sgf=False, first_game=0, game_stride=1, num_parallel=0, ex_it_rollouts=0, leaves_per_thread=1, cache_per_game=0,
                host_threads=0, opening=None, window=None, on_game=None):
    """self_play (src/libdg_mcts/self_play.rs:423-505) for num_games games in parallel; num_moves caps the stones per game
    (0 = the reference's 722).  The games are the global ids first_game, first_game + game_stride, ... (sharding across ranks);
    num_parallel of them run at a time (0 = all); host_threads > 0 runs them as fibers on that many OS threads (csrc/fiber.h).
    opening: [rows][len] packed indices, -1 ends a row; fewer rows than games: the later games start from the empty board.
    window = (ramp_seconds, seconds[, on_edge(edge)]): the job runs num_parallel games WITH REPLACEMENT (num_games bounds how many may
    be started) and is stopped at the end of a timed window; out["window"] holds what happened inside it (dgnn_selfplay_window).
    on_game(game_id, record_bytes, whole): the reference's channel (self_play.rs:471-497) -- called as each game leaves, one call at a
    time; how a long job takes its records (return_sgf holds them all until the call returns)."""
    fn, ctx = _predictor(pool)
    stats = SelfplayStats()
    if num_moves <= 0 or num_moves > 722:
        num_moves = 722
    moves = np.full(num_games * num_moves, -1, dtype=np.int32) if return_moves else None
    sgf = ctypes.create_string_buffer(num_games * (num_moves * 2600 + 4 * 722 * 6 + 2048)) if return_sgf else None
    params = SelfplayParams(int(num_games), int(num_parallel), int(first_game), int(game_stride), int(num_moves), int(num_rollouts),
                            int(ex_it_rollouts), int(threads_per_search), int(leaves_per_thread), int(cache_per_game), int(seed),
                            int(host_threads), 0, None, 0, None, None, None)
    game_thunk = None
    if on_game is not None:
        game_thunk = GAME_RECORD_FN(lambda _ctx, game_id, text, size, whole: on_game(int(game_id), ctypes.string_at(text, size), bool(whole)))
        params.on_game = ctypes.cast(game_thunk, ctypes.c_void_p)
    if opening is not None:                                    # [rows][opening_len] packed indices: game g < rows starts from the position of row g
        opening = np.ascontiguousarray(opening, dtype=np.int32)
        opening = opening.reshape(opening.shape[0] if opening.ndim == 2 else num_games, -1)
        if opening.shape[0] > num_games:
            opening = np.ascontiguousarray(opening[:num_games])
        params.opening_len = opening.shape[1]
        params.opening = opening.ctypes.data_as(ctypes.c_void_p)
        params.opening_games = opening.shape[0]
    win = None
    if window is not None:
        on_edge = window[2] if len(window) > 2 and window[2] is not None else None
        thunk = WINDOW_EDGE_FN((lambda _ctx, edge: on_edge(int(edge))) if on_edge else (lambda _ctx, edge: None))
        win = SelfplayWindow(float(window[0]), float(window[1]), thunk, None)
        params.window = ctypes.addressof(win)
    play = lib.dgnn_selfplay_go_boards if _takes_boards(pool) else lib.dgnn_selfplay_go
    _check(play(fn, ctx, cache._handle if cache is not None else None, ctypes.byref(params), ctypes.byref(stats),
                moves.ctypes.data_as(_ip) if moves is not None else None, sgf, len(sgf) if sgf is not None else 0), "selfplay_go")
    out = {"seconds": stats.seconds, "games": stats.games, "moves": stats.moves, "evals": stats.evals, "conflicts": stats.conflicts,
           "cache_hits": stats.cache_hits, "games_ended": stats.games_ended, "ex_it_searches": stats.ex_it_searches}
    if win is not None:
        out["window"] = {"seconds": win.measured_seconds, "moves": win.moves, "evals": win.evals, "cache_hits": win.cache_hits,
                         "conflicts": win.conflicts, "games_started": win.games_started, "games_completed": win.games_completed,
                         "games_ended": win.games_ended, "stones_completed": win.stones_completed, "ex_it_searches": win.ex_it_searches}
    if return_moves:
        out["record"] = moves.reshape(num_games, num_moves)
    if return_sgf:
        out["sgf"] = [g for g in sgf.value.decode("ascii").split("\n") if g]     # (a record holds carriage returns: C[] text, self_play.rs:192)
    return out
