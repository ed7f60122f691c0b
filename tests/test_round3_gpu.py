"""Round-3 parity tests on a real MI355X: workloads that CAN fail ("Parity tests that can fail").

  * sharpened networks (synth.make_sharp_tensors: policy maxima 0.05 ... 0.99, values at both ends of tanh) on STRUCTURED inputs
    (positions of the three committed reference games, all eight symmetries) at BASELINE.json configs[2] (20 x 256, batch 256)
    and configs[1] (9 x 128, batch 32): the 1e-3 bounds, board identity (every board's HIP output is nearer to its OWN oracle
    output than to any other board's) and the relative error on entries > 0.01;
  * NaN / +inf / -inf features: where they end up in the outputs, HIP vs oracle, and that the other boards of the batch are untouched;
  * scope row f-4 under -m gpu: the `P[b85]` / `TV[]` / `V[]` properties of engine-played moves (self_play.rs:187-213) decoded and
    compared with what the oracle gives for the recorded position.
"""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, f16_to_f32, policy_kl
from parity_bounds import allowances

pytestmark = pytest.mark.gpu

VALUE_TOL = 1e-3     # north_star: |dvalue| <= 1e-3
POLICY_TOL = 1e-3    # north_star: max |dpolicy| <= 1e-3
# Peaked outputs.  Both sides store every tensor as fp16 (layers/common.rs:61-98) and differ only in the order of their fp32 sums,
# i.e. by single-ulp flips of fp16 values.  In front of the soft-max sit fp16 LOGITS: at the magnitudes a peaked policy needs (8 ... 32)
# one ulp is 2^-7 ... 2^-6, and one flipped logit moves a probability by p (1 - p) ulp -- more than the 1e-3 that holds on flat
# outputs, whatever the implementation.  The contract on peaked outputs is therefore "ONE fp16 ulp in front of the soft-max / tanh",
# and what it allows per probability / value is derived per entry (tests/parity_bounds.py, the helper tests/test_round4_gpu.py
# measures the bound with): round 3's blanket 4e-3 / 2.5e-3 widening is gone.  The same networks with their dense gains divided by
# four stay within north_star's 1e-3 outright (second parametrisation); board identity holds with a margin of more than 3x in both.


def real_game_positions(api):
    """Every position of the three committed reference games (tests/golden/real_games.json; src/dg_tests/tests/real_games.rs)
    as (board, colour to move)."""
    out = []
    for game in json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]:
        board = api.Board(7.5)
        for color, vertex in re.findall(r";([BW])\[([a-s]{2})\]", game["moves"]):
            c = 0 if color == "B" else 1
            board.place(c, ord(vertex[0]) - 97, ord(vertex[1]) - 97)
            out.append((board.clone(), 1 - c))
    return out


def real_game_features(api, batch, first=3):
    """`batch` DISTINCT feature tensors: positions spread over the three games, symmetry = index mod 8."""
    positions = real_game_positions(api)
    picks = np.linspace(first, len(positions) - 1, batch).astype(int)
    assert len(set(picks.tolist())) == batch
    planes = [positions[i][0].get_features(positions[i][1], symmetry=k % 8).reshape(-1) for k, i in enumerate(picks)]
    return np.concatenate(planes).view(np.uint16)


def oracle_outputs(oracle, text, features, batch, pre=False):
    """-> value [B], policy [B, 362] (and, with pre=True, the logits [B, 362] and the pre-tanh value [B]) of the fp32 oracle."""
    tensors, _ = oracle.load_weights(text)
    f = np.ascontiguousarray(features).view(np.uint16).reshape(-1)
    if pre:
        rv, rp, rl, rpre = oracle.OracleNetwork(tensors).forward_ex(f, batch)
        return f16_to_f32(rv), f16_to_f32(rp).reshape(batch, 362), f16_to_f32(rl).reshape(batch, 362), f16_to_f32(rpre)
    rv, rp = oracle.OracleNetwork(tensors).forward(f, batch)
    return f16_to_f32(rv), f16_to_f32(rp).reshape(batch, 362)


# --------------------------------------------------------------------------------------------- sharpened nets, board identity

@pytest.mark.parametrize("blocks,channels,batch,gains,kernel,peaked", [
    (20, 256, 256, (24.0, 16.0), "tower_board_kernel", True),      # BASELINE.json configs[2]
    (9, 128, 32, (32.0, 6.0), "tower_split_kernel", True),         # BASELINE.json configs[1]
    (20, 256, 256, (6.0, 4.0), "tower_board_kernel", False),       # the same networks, dense gains / 4: the 1e-3 bounds
    (9, 128, 32, (8.0, 1.5), "tower_split_kernel", False),
])
def test_sharpened_network_on_real_positions_board_identity(api, oracle, synth, blocks, channels, batch, gains, kernel, peaked):
    text = synth.to_json(synth.make_sharp_tensors(blocks, channels, seed=2020, policy_gain=gains[0], value_gain=gains[1]))
    features = real_game_features(api, batch)
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        assert ws.conv_kernel() == kernel
        value, policy = api.forward(ws, features).unwrap()
    net.destroy()
    rv, rp, rl, rpre = oracle_outputs(oracle, text, features, batch, pre=True)
    gv, gp = value.astype(np.float32), policy.astype(np.float32).reshape(batch, 362)

    # the workload discriminates: peaked AND flat soft-maxes, many different best moves, saturated AND small values
    pmax = rp.max(axis=1)
    print(f"{blocks}x{channels} B={batch}: oracle policy max {pmax.min():.3f} ... {pmax.max():.3f}, {len(set(rp.argmax(axis=1).tolist()))} distinct "
          f"arg-maxima, value {rv.min():.3f} ... {rv.max():.3f}")
    if peaked:
        assert pmax.min() <= 0.25 and pmax.max() >= 0.9
        assert np.abs(rv).max() >= 0.9 and np.abs(rv).min() <= 0.5
    else:
        assert pmax.max() >= 3 * pmax.min() and np.abs(rv).max() >= 0.3
    assert len(set(rp.argmax(axis=1).tolist())) >= 5

    dv, dp, kl = np.abs(gv - rv), np.abs(gp - rp).max(axis=1), policy_kl(rp, gp)
    big = rp > 0.01
    rel = (np.abs(gp - rp)[big] / rp[big]).max()
    print(f"   max|dvalue| {dv.max():.3e}  max|dpolicy| {dp.max():.3e}  max KL {kl.max():.3e}  max relative error on entries > 0.01: {rel:.3e}")
    if peaked:
        # one fp16 ulp in front of the soft-max / tanh, and what that allows per entry (parity_bounds.allowances)
        allowed_p, allowed_v, held = allowances(rp, rv, rl, rpre)
        print(f"   {100 * held.mean():.2f} % of the probabilities held to 1e-3, the rest to 2 p (1 - p) x one logit ulp (largest allowance {allowed_p.max():.2e})")
        assert np.all(np.abs(gp - rp) <= allowed_p) and np.all(dv <= allowed_v) and kl.max() <= 1e-3
    else:
        assert dv.max() <= VALUE_TOL and dp.max() <= POLICY_TOL

    # board identity: the HIP output of board i is nearer to the oracle's output of board i than to that of any other board
    hip = np.concatenate([gp, gv[:, None]], axis=1).astype(np.float64)
    ref = np.concatenate([rp, rv[:, None]], axis=1).astype(np.float64)
    distance = np.abs(hip[:, None, :] - ref[None, :, :]).max(axis=2)             # [hip board, oracle board], L-infinity
    own = np.diag(distance).copy()
    np.fill_diagonal(distance, np.inf)
    nearest_other = distance.min(axis=1)
    print(f"   board identity: own distance <= {own.max():.3e}, nearest OTHER board >= {nearest_other.min():.3e}")
    assert np.all(own < nearest_other), np.nonzero(own >= nearest_other)[0]
    assert nearest_other.min() > 3 * max(own.max(), 1e-4)       # neighbouring positions of one game are 2-3 stones apart: still 3x the error


def test_every_symmetry_and_colour_of_a_position_is_told_apart(api, oracle, synth):
    """One mid-game position of every committed game x 8 symmetries x both colours to move = 48 boards that differ ONLY by the
    symmetry / colour planes: a kernel that dropped a symmetry's layout or the komi / colour planes could not keep them apart."""
    positions = real_game_positions(api)
    per_game = len(positions) // 3
    planes = []
    for g in range(3):
        board, _ = positions[g * per_game + 80]
        for color in (0, 1):
            for s in range(8):
                planes.append(board.get_features(color, symmetry=s).reshape(-1))
    batch = len(planes)
    features = np.concatenate(planes).view(np.uint16)
    text = synth.to_json(synth.make_sharp_tensors(6, 128, seed=31, policy_gain=24.0, value_gain=8.0))
    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        value, policy = api.forward(ws, features).unwrap()
    net.destroy()
    rv, rp, rl, rpre = oracle_outputs(oracle, text, features, batch, pre=True)
    gv, gp = value.astype(np.float32), policy.astype(np.float32).reshape(batch, 362)
    print(f"max|dvalue| {np.abs(gv - rv).max():.3e} max|dpolicy| {np.abs(gp - rp).max():.3e}; oracle policy max {rp.max(axis=1).min():.3f} ... {rp.max():.3f}")
    allowed_p, allowed_v, _ = allowances(rp, rv, rl, rpre)
    assert np.all(np.abs(gv - rv) <= allowed_v) and np.all(np.abs(gp - rp) <= allowed_p)
    hip = np.concatenate([gp, gv[:, None]], axis=1)
    ref = np.concatenate([rp, rv[:, None]], axis=1)
    distance = np.abs(hip[:, None, :] - ref[None, :, :]).max(axis=2)
    own = np.diag(distance).copy()
    np.fill_diagonal(distance, np.inf)
    print(f"48 symmetry / colour variants: own distance <= {own.max():.3e}, nearest other >= {distance.min():.3e}")
    assert np.all(own < distance.min(axis=1))


# --------------------------------------------------------------------------------------------- NaN / inf propagation

@pytest.mark.parametrize("blocks", [2, 0])
def test_nan_and_inf_features_propagate_like_the_oracle(api, oracle, synth, blocks):
    """cuDNN's ReLU is NOT_PROPAGATE_NAN (src/libdg_cuda/cudnn/activation_descriptor.rs:52,118-134): a NaN that reaches an
    activation becomes 0, an infinity survives it.  Boards 1 / 3 / 5 of a batch of 8 carry a NaN / +inf / -inf feature: the HIP
    path and the oracle must agree on which outputs are NaN / infinite and on the finite ones, and the other boards must come
    out bit-identical to a clean run (a board is one workgroup: nothing leaks across boards)."""
    channels, batch = 64, 8           # 0 blocks: the up-layer's infinities reach the heads (inf logits -> NaN soft-max, tanh(inf) = 1)
    text = synth.to_json(synth.make_tensors(blocks, channels, seed=404))
    clean = synth.make_features(batch, seed=405).copy()
    dirty = clean.copy().reshape(batch, 361, 32)
    dirty[1, 180, 7] = 0x7e00        # NaN in the middle of board 1
    dirty[3, 0, 3] = 0x7c00          # +inf in a corner of board 3
    dirty[5, 200, 31] = 0xfc00       # -inf on board 5
    dirty = dirty.reshape(-1)

    net = api.Network.from_json(text)
    with net.get_workspace(batch) as ws:
        cv, cp = api.forward(ws, clean).unwrap()
        dv, dp = api.forward(ws, dirty).unwrap()
    net.destroy()
    tensors, _ = oracle.load_weights(text)
    rv, rp = oracle.OracleNetwork(tensors).forward(dirty, batch)
    rv, rp = rv.view(np.float16), rp.view(np.float16).reshape(batch, 362)
    dp, cp = dp.reshape(batch, 362), cp.reshape(batch, 362)

    for b in (0, 2, 4, 6, 7):
        assert dv.view(np.uint16)[b] == cv.view(np.uint16)[b] and np.array_equal(dp[b].view(np.uint16), cp[b].view(np.uint16)), b
    for b in (1, 3, 5):
        g = np.concatenate([dp[b].astype(np.float32), dv[b:b + 1].astype(np.float32)])
        r = np.concatenate([rp[b].astype(np.float32), rv[b:b + 1].astype(np.float32)])
        print(f"board {b}: oracle NaN {int(np.isnan(r).sum())} inf {int(np.isinf(r).sum())} | HIP NaN {int(np.isnan(g).sum())} inf {int(np.isinf(g).sum())}")
        assert np.array_equal(np.isnan(g), np.isnan(r)), b
        assert np.array_equal(np.isinf(g), np.isinf(r)) and np.array_equal(np.sign(g[np.isinf(g)]), np.sign(r[np.isinf(r)])), b
        finite = np.isfinite(r)
        assert np.abs(g[finite] - r[finite]).max(initial=0.0) <= 1e-3, b


# --------------------------------------------------------------------------------------------- f-4: records of engine-played moves

MOVE = re.compile(r";([BW])\[([a-s]{0,2})\](?:C\[[^\]]*\])?(?:TR\[([a-s]{2})\])?(?:TV\[(\d+)\]P\[([^\]]*)\])?(?:V\[(-?[0-9.]+)\])?")


def parse_record(api, sgf):
    """One GameResult::Ended record -> [(colour, packed index, rollouts, P as float32 [362] or None, V or None)]."""
    body = sgf[sgf.index("RE["):]
    moves = []
    for color, vertex, _prior, rollouts, p, v in MOVE.findall(body):
        index = 361 if vertex == "" else (ord(vertex[1]) - 97) * 19 + (ord(vertex[0]) - 97)
        dist = api.b85_decode(p, np.float16).astype(np.float32) if p else None
        moves.append((0 if color == "B" else 1, index, int(rollouts) if rollouts else None, dist, float(v) if v else None))
    return moves


def test_search_records_of_engine_played_games_decode_to_the_oracles(api, oracle, synth):
    """Played::fmt (src/libdg_mcts/self_play.rs:187-213): `TV[rollouts]P[b85 fp16 x 362]V[win rate, black's side]`.  Six games are
    played twice with the same seed -- on the HIP engine behind the leaf pool and on the CPU oracle as the predictor (one search
    thread, a transposition table per game: a game is a function of the seed and the network outputs).  The records are decoded:
    same stones, same TV[], P[] (the visit distribution) within 1e-3 entry by entry, V[] within 1e-3."""
    text = synth.to_json(synth.make_sharp_tensors(2, 64, seed=77, policy_gain=8.0, value_gain=4.0))
    net = api.Network.from_json(text)
    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    oracle_fn = api.python_predictor(lambda f, n: onet.forward(f, n))
    pool = api.LeafPool.over_devices(net, [0], max_batch_size=32, batches_per_device=2, workers_per_device=2)
    kw = dict(num_games=6, num_moves=8, num_rollouts=40, threads_per_search=1, leaves_per_thread=4, seed=1234, cache_per_game=4096,
              return_moves=True, return_sgf=True)
    hip = api.selfplay_go(pool=pool, **kw)
    ref = api.selfplay_go(pool=oracle_fn, **kw)
    pool.close()
    net.destroy()
    assert np.array_equal(hip["record"], ref["record"])
    checked, worst_p, worst_v = 0, 0.0, 0.0
    for game_hip, game_ref in zip(hip["sgf"], ref["sgf"]):
        a, b = parse_record(api, game_hip), parse_record(api, game_ref)
        assert len(a) == len(b) == 8
        for (ca, ia, ra, pa, va), (cb, ib, rb, pb, vb) in zip(a, b):
            assert (ca, ia, ra) == (cb, ib, rb)
            assert pa is not None and pb is not None and va is not None and vb is not None
            assert pa.shape == (362,) and abs(float(pa.sum()) - 1.0) < 5e-3
            worst_p, worst_v = max(worst_p, float(np.abs(pa - pb).max())), max(worst_v, abs(va - vb))
            checked += 1
    print(f"{checked} recorded moves: max |dP| {worst_p:.3e}, max |dV| {worst_v:.3e}")
    assert worst_p <= 1e-3 and worst_v <= 1e-3


def test_policy_play_records_hold_the_oracles_value_of_the_position(api_full, oracle, synth):
    """Policy play (--num-rollout 1, self_play.rs:369-405): a move's record is `V[..]` of the eight-symmetry ensemble
    (lib.rs:83-130).  The stones are replayed on a fresh board and the ensemble is evaluated with the ORACLE as the predictor:
    every V[] the engine wrote must be the oracle's value of that position within 1e-3 (%.4f text included)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    text = synth.to_json(synth.make_sharp_tensors(2, 64, seed=78, policy_gain=8.0, value_gain=4.0))
    net = api.Network.from_json(text)
    tensors, _ = oracle.load_weights(text)
    onet = oracle.OracleNetwork(tensors)
    oracle_fn = api.python_predictor(lambda f, n: onet.forward(f, n))
    pool = api.LeafPool.over_devices(net, [0], max_batch_size=64, batches_per_device=2, workers_per_device=2)
    out = api.selfplay_go(num_games=4, num_moves=10, num_rollouts=1, threads_per_search=1, seed=99, pool=pool, return_moves=True, return_sgf=True)
    pool.close()
    net.destroy()
    checked, worst = 0, 0.0
    for sgf in out["sgf"]:
        komi = float(re.search(r"KM\[([-0-9.]+)\]", sgf).group(1))
        board = api.Board(komi)
        for color, index, rollouts, dist, v in parse_record(api, sgf):
            assert rollouts is None and dist is None and v is not None           # policy play writes no TV[] / P[] (self_play.rs:199)
            scorable = board.is_scorable()
            winrate, prior, _ = api.full_forward(board, color, pool=oracle_fn, scoring=not scorable)
            black_side = 2.0 * winrate - 1.0 if color == 0 else -2.0 * winrate + 1.0
            worst = max(worst, abs(black_side - v))
            assert index == 361 or np.isfinite(prior[index])                     # the stone was a candidate of the oracle's prior
            if index != 361:
                board.place(color, index % 19, index // 19)
            checked += 1
    print(f"{checked} policy-play moves: max |V[] - oracle value| {worst:.3e}")
    assert checked >= 40 and worst <= 1e-3 + 5e-5                                 # + half a unit of the %.4f text


def test_a_slow_pass_does_not_slow_the_synchronous_calls_that_follow(api, synth):
    """dgnn_forward sleeps for most of what the last passes of that size took instead of spinning (engine.cpp: wait_submission).  A
    pass that finished before the sleep ended must not be averaged in as if it had taken that long: the estimate would never come
    down again (seen in a batch sweep as 14 ms per call for a 3.4 ms pass, after one slow first call).  Here the first call of a
    workspace carries the weight upload and the kernel load; the calls that follow must settle near the pass itself."""
    import time
    net = api.Network.from_json(synth.to_json(synth.make_tensors(6, 128, seed=5)))
    batch = 64
    features = synth.make_features(batch, seed=6)
    with net.get_workspace(batch) as ws:
        t0 = time.perf_counter()
        api.forward(ws, features)                       # the slow one
        first = time.perf_counter() - t0
        ws.upload(features)
        total_ms, _, _ = ws.forward_timed(10)
        per_pass = total_ms / 10 * 1e-3
        for _ in range(12):
            api.forward(ws, features)
        laps = []
        for _ in range(8):
            t0 = time.perf_counter()
            api.forward(ws, features)
            laps.append(time.perf_counter() - t0)
    net.destroy()
    laps.sort()
    print(f"first call {first * 1e3:.2f} ms, pass {per_pass * 1e3:.3f} ms, synchronous calls after it: median {laps[4] * 1e3:.3f} ms")
    assert laps[4] <= 2.0 * per_pass + 0.4e-3          # (the old estimator: ~0.6 x the first call's 2.3 ms here)
