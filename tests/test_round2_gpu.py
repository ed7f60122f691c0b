"""Round-2 parity and boundary tests on a real MI355X (everything here calls through the C ABI).

What they pin, item by item of VERDICT r01 / ADVICE r01:
  * all 256 boards of BASELINE.json configs[2] against the oracle (was 4 of 256 in a test, all 256 only as a bench side effect)
  * visit counts under a fixed seed, HIP predictor vs oracle predictor: EQUAL, not "<= 6 moved"
  * the reference's two dense KATs (dense.rs:243-299) through the MFMA GEMM of the policy head; soft-max / value tail vs the oracle
  * policy play + --ex-it through the pool on the engine (configs[4])
  * residual gates of 0 and 1 on the one-launch tower; weight refresh from a network with OTHER gates and temperature
  * packed features: bit-identical outputs; RCCL inside the library (one rank); the multi-device pool
"""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, f16_to_f32, policy_kl

pytestmark = pytest.mark.gpu

VALUE_TOL = 1e-3
POLICY_TOL = 1e-3


def _oracle_outputs(oracle, text, features, batch):
    tensors, _ = oracle.load_weights(text)
    rv, rp = oracle.OracleNetwork(tensors).forward(np.ascontiguousarray(features).view(np.uint16).reshape(-1), batch)
    return f16_to_f32(rv), f16_to_f32(rp).reshape(batch, 362)


# --------------------------------------------------------------------------------------------- whole network, every board

def test_full_size_network_every_board(api, oracle, synth):
    """BASELINE.json configs[2] at full size (20 x 256, batch 256): ALL 256 boards against the oracle (OpenMP, a few
    seconds on the box's host cores), tolerance 1e-3 on value and on every policy entry, KL reported."""
    blocks, channels, batch = 20, 256, 256
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=2020))
    features = synth.make_features(batch, seed=77)
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        assert ws.conv_kernel() == "tower_board_kernel"
        value, policy = api.forward(ws, features).unwrap()
    rv, rp = _oracle_outputs(oracle, text, features, batch)
    gv, gp = value.astype(np.float32), policy.astype(np.float32).reshape(batch, 362)
    dv, dp, kl = np.abs(gv - rv), np.abs(gp - rp).max(axis=1), policy_kl(rp, gp)
    print(f"20x256 B=256, all boards: max|dvalue| {dv.max():.3e}  max|dpolicy| {dp.max():.3e}  max KL {kl.max():.3e}  "
          f"boards with a differing fp16 value {int(np.count_nonzero(dv))}")
    assert dv.max() <= VALUE_TOL and dp.max() <= POLICY_TOL
    assert kl.max() <= 1e-4
    net.destroy()


# --------------------------------------------------------------------------------------------- search: visit counts

@pytest.mark.parametrize("seed", [3, 11, 29])
def test_search_visit_counts_are_identical_hip_vs_oracle(api_full, oracle, synth, seed):
    """north_star: "MCTS visit counts must be bit-identical under a fixed RNG seed".  One search thread, the same seed and game;
    the HIP engine behind the pool vs the CPU oracle as the predictor.  Equality, not a tolerance."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_tensors(2, 64, seed=71))
    net = api.Network.from_json(text)
    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    oracle_fn = api.python_predictor(lambda f, n: onet.forward(f, n))

    def run(predictor):
        root = api.Node(api.BLACK, 0.5, np.full(362, 1.0 / 362, dtype=np.float32))
        api.search_run(root, api.SyntheticBoard(99 + seed), num_rollouts=96, num_threads=1, seed=seed, pool=predictor)
        return np.array([root.with_(i).count() for i in range(362)])

    pool = api.LeafPool.over_devices(net, [0], max_batch_size=8, batches_per_device=1, workers_per_device=1)
    hip_a, hip_b, ref = run(pool), run(pool), run(oracle_fn)
    pool.close()
    assert np.array_equal(hip_a, hip_b)
    assert 0 < ref.sum() <= 96
    assert np.array_equal(hip_a, ref), f"{int(np.abs(hip_a - ref).sum()) // 2} of {int(ref.sum())} visits differ"


# --------------------------------------------------------------------------------------------- dense / heads, layer level

def test_dense_kats_through_the_policy_gemm(api_full):
    """The reference's own Dense tests (src/libdg_nn/layers/dense.rs:243-299) through policy_gemm_kernel."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    kats = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
    eye = np.eye(8, dtype=np.float16)
    x = np.arange(1, 9, dtype=np.float16)
    y = api.layer_dense(x, 1, 8, eye, np.zeros(8, np.float16), 8)                      # matmul_identity_8x8
    assert f16_to_f32(y).tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    y = api.layer_dense(x, 1, 8, np.ones((8, 1), np.float16), np.array([-36.0], np.float16), 1)   # matmul_identity_1x8
    assert f16_to_f32(y).tolist() == [0.0]
    if "dense_matmul_identity_8x8" in kats:                                            # the same numbers as the oracle's KAT file holds them
        assert f16_to_f32(api.layer_dense(np.asarray(kats["dense_matmul_identity_8x8"]["input"], np.float16), 1, 8, eye,
                                          np.zeros(8, np.float16), 8)).tolist() == [float(v) for v in kats["dense_matmul_identity_8x8"]["expected"]]


@pytest.mark.parametrize("batch,k_in,n_out,alpha,relu", [
    (5, 2888, 362, 1.40867, False),     # the policy head's shape and temperature scale (policy_head.rs:46,54)
    (33, 1444, 362, 1.0, False),        # S = 4; batch not a multiple of the 32-board tile
    (3, 64, 40, 0.5, True),             # ReLU Dense (dense.rs:108-116), ragged output tile
    (70, 722, 1, 1.0, False),           # the value head's Dense as a GEMM
])
def test_dense_layer_against_oracle_arithmetic(api_full, batch, k_in, n_out, alpha, relu):
    """y = act(fp16(alpha * (x . w) + b)) with fp32 accumulation: against float64 numpy rounded once to fp16 (the summation
    order differs, so a few outputs may flip one fp16 rounding)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng([batch, k_in, n_out])
    x = np.maximum(rng.normal(0, 1, (batch, k_in)), 0).astype(np.float16)
    w = rng.normal(0, np.sqrt(1.0 / k_in), (k_in, n_out)).astype(np.float16)
    b = rng.normal(0, 0.5, (n_out,)).astype(np.float16)
    ref = np.float32(alpha) * (x.astype(np.float64) @ w.astype(np.float64)) + b.astype(np.float64)
    if relu:
        ref = np.maximum(ref, 0)
    ref16 = ref.astype(np.float16)
    got = api.layer_dense(x, batch, k_in, w, b, n_out, alpha=alpha, relu=relu).view(np.float16).reshape(batch, n_out)
    diff = np.abs(got.astype(np.float32) - ref16.astype(np.float32))
    assert diff.max() <= 4e-3 * max(1.0, float(np.abs(ref).max()))
    assert np.count_nonzero(got.view(np.uint16) != ref16.view(np.uint16)) <= 0.02 * got.size


def test_softmax_and_value_tail_against_oracle_arithmetic(api_full):
    """cudnnSoftmaxForward ACCURATE over 362 logits and tanh(fp16(w . v + b)) (softmax.rs:58-79, value_head.rs:80-81)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(41)
    batch = 9
    logits = rng.normal(-6.5, 2.0, (batch, 362)).astype(np.float16)
    logits[0] = -60000.0; logits[0, 361] = 3.0                 # one dominant move: the max-subtraction must hold
    logits[1] = 11.0                                            # uniform
    vf = np.maximum(rng.normal(0, 1, (batch, 722)), 0).astype(np.float16)
    vw = rng.normal(0, 0.05, 722).astype(np.float16)
    vb = np.array([-0.00502319782], np.float16)
    policy, value = api.layer_heads(logits, vf, batch, vw, vb)
    policy = policy.view(np.float16).astype(np.float32).reshape(batch, 362)
    l64 = logits.astype(np.float64)
    e = np.exp(l64 - l64.max(axis=1, keepdims=True))
    ref_p = (e / e.sum(axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    assert np.abs(policy - ref_p).max() <= 1e-3 and abs(policy[0, 361] - 1.0) < 1e-3 and np.allclose(policy[1], 1 / 362, atol=2e-5)
    pre = (vf.astype(np.float64) @ vw.astype(np.float64) + float(vb[0])).astype(np.float16).astype(np.float64)
    assert np.abs(value.view(np.float16).astype(np.float64) - np.tanh(pre)).max() <= 2e-3


# --------------------------------------------------------------------------------------------- configs[4]: policy play, --ex-it

def test_policy_play_and_expert_iteration_on_the_engine(api, oracle, synth):
    """BASELINE.json configs[4] (--ex-it --num-rollout 1) through the leaf pool on the HIP engine: hundreds of game threads each
    asking for <= 8 boards (full_forward, lib.rs:83-130), coalesced across callers into device batches.  Same invariants as the
    CPU test of the driver (tests/test_go_host.py) -- plus: the games ARE the ones the CPU oracle predictor plays."""
    text = synth.to_json(synth.make_tensors(2, 64, seed=61))
    net = api.Network.from_json(text)
    pool = api.LeafPool.over_devices(net, [0], max_batch_size=64, batches_per_device=2, workers_per_device=2)
    kw = dict(num_games=24, num_moves=30, num_rollouts=1, threads_per_search=1, seed=17, return_moves=True, return_sgf=True,
              cache_per_game=4000)
    plain = api.selfplay_go(pool=pool, **kw)
    st = pool.stats()
    assert plain["ex_it_searches"] == 0 and plain["conflicts"] == 0 and plain["moves"] == 24 * 30
    for game in plain["sgf"]:
        assert "TV[" not in game and "TR[" not in game and game.count("V[") == 30
    assert plain["evals"] <= 24 * 30 * 8 + 24 * 16                                   # no search ran: at most 8 evaluations per move
    assert st["boards"] == plain["evals"] and st["batches"] < st["boards"] / 4       # the 8-board calls of 24 games shared device batches
    print(f"policy play: {plain['evals']} evals in {st['batches']} device batches (mean {st['boards'] / st['batches']:.1f})")

    again = api.selfplay_go(pool=pool, **kw)                                           # batch composition is timing; the games are not
    assert np.array_equal(plain["record"], again["record"])

    ex_it = api.selfplay_go(pool=pool, ex_it_rollouts=40, **kw)
    assert ex_it["ex_it_searches"] >= 1
    sizes = [int(v) for game in ex_it["sgf"] for v in re.findall(r"TV\[(\d+)\]", game)]
    assert len(sizes) == ex_it["ex_it_searches"] and all(1 < n <= 40 for n in sizes)
    assert sum(game.count("TR[") for game in ex_it["sgf"]) == ex_it["ex_it_searches"]
    pool.close()

    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    few = dict(kw, num_games=3, num_moves=12)
    on_oracle = api.selfplay_go(pool=api.python_predictor(lambda f, n: onet.forward(f, n)), **few)
    pool = api.LeafPool.over_devices(net, [0], max_batch_size=64)
    on_engine = api.selfplay_go(pool=pool, **few)
    pool.close()
    same = int(np.count_nonzero(on_oracle["record"] == on_engine["record"]))
    print(f"policy play, engine vs oracle predictor: {same} of {on_oracle['record'].size} moves equal")
    assert same >= 0.9 * on_oracle["record"].size      # moves are DRAWN from the policy: an fp16 ulp can move a draw across a boundary


# --------------------------------------------------------------------------------------------- gates, refresh

@pytest.mark.parametrize("alpha", [0.0, 1.0, 0.25])
def test_residual_gate_extremes_on_the_tower_shapes(api, oracle, synth, alpha):
    """a gate of 0 made the one-launch tower compute alpha2 / alpha1 = inf.  Such networks now take the per-layer
    launches (epilogue skip); gate 1 (no skip contribution) and an ordinary gate stay on the tower.  All against the oracle."""
    blocks, channels, batch = 2, 64, 6
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=131, alpha=alpha))
    features = synth.make_features(batch, seed=132)
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        assert ws.conv_kernel() == ("conv3x3_board_kernel" if alpha == 0.0 else "tower_board_kernel")
        value, policy = api.forward(ws, features).unwrap()
    assert np.all(np.isfinite(value.astype(np.float32))) and np.all(np.isfinite(policy.astype(np.float32)))
    rv, rp = _oracle_outputs(oracle, text, features, batch)
    assert np.abs(value.astype(np.float32) - rv).max() <= VALUE_TOL
    assert np.abs(policy.astype(np.float32).reshape(batch, 362) - rp).max() <= POLICY_TOL
    net.destroy()


def test_weight_refresh_adopts_the_senders_gates_and_temperature(api, oracle, synth):
    """the image carried conv_2's bias pre-scaled by the SENDER's gate while the receiver launched with its
    own.  The image now carries gates and temperature in a header; a refresh adopts them, refreshes the host copy and every
    cached descriptor, and refuses an image of another architecture."""
    batch = 4
    features = synth.make_features(batch, seed=141)
    text_a = synth.to_json(synth.make_tensors(2, 64, seed=142, alpha=0.5))
    text_b = synth.to_json(synth.make_tensors(2, 64, seed=143, alpha=0.125))
    net_a = api.Network.from_json(text_a)
    net_b = api.Network.from_json(text_b)
    net_b.set_softmax_temperature(1.25)
    with net_a.get_workspace(batch) as ws:
        assert ws.conv_kernel() == "tower_board_kernel"
        before = api.forward(ws, features).unwrap()
    image_b = net_b.weight_image_host().copy()
    net_a.set_weight_image(0, image_b.ctypes.data, image_b.size, src_on_device=False)
    assert net_a.weight_checksum() == net_b.weight_checksum() == net_a.weight_checksum_device(0)
    with net_a.get_workspace(batch) as ws:                      # the pooled workspace with its cached tower descriptors
        after = api.forward(ws, features).unwrap()
    with net_b.get_workspace(batch) as ws:
        want = api.forward(ws, features).unwrap()
    assert np.array_equal(after[0].view(np.uint16), want[0].view(np.uint16))
    assert np.array_equal(after[1].view(np.uint16), want[1].view(np.uint16))
    assert not np.array_equal(after[1].view(np.uint16), before[1].view(np.uint16))

    other = api.Network.from_json(synth.to_json(synth.make_tensors(3, 64, seed=144))).weight_image_host().copy()
    with pytest.raises(api.Error) as e:
        net_a.set_weight_image(0, other.ctypes.data, other.size, src_on_device=False)
    assert e.value.status == api.BAD_ARGUMENT
    garbage = np.zeros_like(image_b)
    with pytest.raises(api.Error) as e:
        net_a.set_weight_image(0, garbage.ctypes.data, garbage.size, src_on_device=False)
    assert e.value.status == api.MALFORMED_WEIGHTS
    with net_a.get_workspace(batch) as ws:                      # a refused image changed nothing
        still = api.forward(ws, features).unwrap()
    assert np.array_equal(still[1].view(np.uint16), want[1].view(np.uint16))


# --------------------------------------------------------------------------------------------- packed features

@pytest.mark.parametrize("blocks,channels,batch,env", [(2, 64, 7, None), (1, 256, 33, None), (1, 256, 5, "tile2"), (1, 32, 3, None)])
def test_packed_features_give_bit_identical_outputs(api, synth, monkeypatch, blocks, channels, batch, env):
    """dgnn_forward_packed (1 472 bytes per board) vs dgnn_forward (23 104 bytes) on the same positions: the same bits, on the
    board path and on the NHWC paths.  Includes real komi constants in channels 0 / 1."""
    if env:
        monkeypatch.setenv("DGNN_CONV", env)
    f = synth.make_features(batch, seed=151).view(np.float16).reshape(batch, 361, 32).copy()
    komi = np.float16(0.5 + 0.5 * 6.5 / 7.5)
    for b in range(batch):
        f[b, :, 0] = komi if b % 2 == 0 else 0
        f[b, :, 1] = 0 if b % 2 == 0 else komi
        f[b, :, 2] = 1 if b % 3 == 0 else 0
    packed = api.features_pack(f, batch)
    net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, seed=152)))
    with net.get_workspace(batch + 3) as ws:                    # capacity above the count, like the pool's workspaces
        a = api.forward_n(ws, f.reshape(-1), batch).unwrap()
        b = api.forward_packed(ws, packed, batch).unwrap()
    assert np.array_equal(a[0].view(np.uint16), b[0].view(np.uint16))
    assert np.array_equal(a[1].view(np.uint16), b[1].view(np.uint16))
    net.destroy()


# --------------------------------------------------------------------------------------------- RCCL inside the library, pools

def test_rccl_broadcast_inside_the_library_one_rank(api, synth):
    """dgnn_broadcast_weights end to end with the ranks this box has: unique id -> communicator -> ncclBroadcast of
    the image -> adoption.  (N > 1: bench.py --gpus N, which starts every rank > 0 from other weights.)"""
    if not api.lib.dgnn_rccl_available():
        pytest.fail("librccl could not be opened on a GPU box")
    net = api.Network.from_json(synth.to_json(synth.make_tensors(1, 64, seed=161)))
    comm = api.Comm.create(0, 1, 0, api.Comm.unique_id())
    assert comm.info() == {"num_ranks": 1, "rank": 0, "device": 0}
    before = net.weight_checksum()
    net.broadcast_weights(comm, root=0)
    assert net.weight_checksum() == before == net.weight_checksum_device(0)
    with pytest.raises(api.Error) as e:
        net.broadcast_weights(comm, root=1)
    assert e.value.status == api.BAD_ARGUMENT
    comm.destroy()
    net.broadcast_weights_devices([0])                          # the single-process form with the devices this box has
    features = synth.make_features(2, seed=162)
    with net.get_workspace(2) as ws:
        value, policy = api.forward(ws, features).unwrap()
    assert np.all(np.isfinite(policy.astype(np.float32)))


def test_pool_over_device_lanes(api, synth):
    """NnPredictor's process model (predictors/nn.rs:46-108): ONE process, one queue, batches spread over the device list.  The
    box has one GPU, so the list names it twice (two lanes, two workspaces, two streams); results equal the direct forward."""
    import threading
    text = synth.to_json(synth.make_tensors(2, 64, seed=171))
    net = api.Network.from_json(text)
    pool = api.LeafPool.over_devices(net, [0, 0], max_batch_size=32, batches_per_device=2, workers_per_device=1)
    callers, per_call, rounds = 16, 6, 8
    feats = [synth.make_features(per_call, seed=172 + c) for c in range(callers)]
    want = []
    with net.get_workspace(per_call) as ws:
        for c in range(callers):
            v, p = api.forward(ws, feats[c]).unwrap()
            want.append((v.view(np.uint16).copy(), p.view(np.uint16).copy()))
    errors = []

    def caller(c):
        try:
            for _ in range(rounds):
                got = pool.predict(feats[c], per_call)
                v = np.array([np.asarray(g._value).view(np.uint16) for g in got]).reshape(-1)
                p = np.concatenate([g._policy.view(np.uint16) for g in got])
                assert np.array_equal(v, want[c][0]) and np.array_equal(p, want[c][1])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=caller, args=(c,)) for c in range(callers)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    st = pool.stats()
    pool.close()
    assert not errors, errors[0]
    assert st["lanes"] == 2 and st["boards"] == callers * per_call * rounds == sum(st["lane_boards"])
    assert all(n > 0 for n in st["lane_batches"])
    with pytest.raises(api.Error) as e:
        api.LeafPool.over_devices(net, [0, 97])
    assert e.value.status == api.BAD_ARGUMENT
