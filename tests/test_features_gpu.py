"""The feature planes built ON THE DEVICE (csrc/features.hip) from compact board records against go.cpp's features_v1
(features::V1::get_features, src/libdg_go/utils/features.rs:99-251): bit-identical on every position of the three committed
reference games x 8 symmetries x both colours to move, on dense random positions, and end to end:
dgnn_forward_boards gives the same bits as dgnn_forward on the fp16 planes of the same positions."""
import numpy as np
import pytest

from test_features_host import describe_mismatch, expected_and_records, game_positions, random_positions

pytestmark = pytest.mark.gpu


def test_device_planes_equal_features_v1_on_every_position_of_the_reference_games(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    expected, records, n = expected_and_records(api, list(b.clone() for b in game_positions(api)))
    assert n == 648 * 16
    got = api.board_records_to_packed(records, n, device=0).reshape(n, api.PACKED_WORDS)
    assert np.array_equal(expected, got), describe_mismatch(expected, got)
    host = api.board_records_to_packed(records, n).reshape(n, api.PACKED_WORDS)
    assert np.array_equal(host, got)
    planes = [int(((expected[:, :361] >> c) & 1).sum()) for c in range(32)]
    print(f"{n} boards bit-identical; set bits per plane: {planes}")
    assert all(count > 0 for count in planes)          # every plane occurs (ko, ladder capture and ladder escape included)


@pytest.mark.parametrize("komi", [0.5, -20.0])
def test_device_planes_equal_features_v1_on_dense_random_positions(api_full, komi):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    boards = [b.clone() for b in random_positions(api, seed=11 + int(abs(komi)), games=3, stones=340, komi=komi) if b.count() % 2 == 0]
    expected, records, n = expected_and_records(api, boards)
    got = api.board_records_to_packed(records, n, device=0).reshape(n, api.PACKED_WORDS)
    assert np.array_equal(expected, got), describe_mismatch(expected, got)
    print(f"komi {komi}: {n} boards bit-identical")


@pytest.mark.parametrize("blocks,channels,batch", [(2, 64, 48), (2, 256, 256), (1, 256, 100), (3, 128, 7), (1, 96, 33)])
def test_forward_boards_gives_the_bits_of_forward_on_the_same_planes(api, synth, blocks, channels, batch):
    """Whole path: records -> device planes -> tower -> heads vs fp16 planes from the host -> the same tower.  The shapes cover the
    one-launch tower, the split tower and a generic-kernel channel count (96: the NHWC path expands the words a second time)."""
    positions = [b.clone() for b in game_positions(api, stride=5)]
    picks = [positions[i % len(positions)] for i in range(batch)]
    features = np.concatenate([b.get_features(i % 2, i % 8).reshape(-1).view(np.uint16) for i, b in enumerate(picks)])
    records = np.concatenate([b.record(i % 2, i % 8) for i, b in enumerate(picks)])
    net = api.Network.from_json(synth.to_json(synth.make_tensors(blocks, channels, seed=515)))
    with net.get_workspace(batch) as ws:
        a = api.forward(ws, features).unwrap()
        b = api.forward_boards(ws, records, batch).unwrap()
        half = batch // 2 + 1
        c = api.forward_boards(ws, records[: half * api.BOARD_RECORD_BYTES], half).unwrap()      # fewer boards than the capacity
    net.destroy()
    assert np.array_equal(a[0].view(np.uint16), b[0].view(np.uint16)) and np.array_equal(a[1].view(np.uint16), b[1].view(np.uint16))
    assert np.array_equal(a[0].view(np.uint16)[:half], c[0].view(np.uint16))
    assert np.array_equal(a[1].view(np.uint16)[: half * 362], c[1].view(np.uint16))


def test_device_planes_equal_the_committed_oracle_fixture(api_full):
    """The device kernel against an oracle that is NOT product code: tests/golden/planes_real_games.npz holds the
    planes oracle/features.py computes -- flood fills, simulated moves, whole-position super-ko; made by tests/golden/make_planes.py
    -- for every 5th position of the three reference games x both colours x eight symmetries.  board_records_kernel builds the same
    words from 896-byte records."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import json
    import os
    from conftest import GOLDEN
    from oracle import features as F
    fx = np.load(os.path.join(GOLDEN, "planes_real_games.npz"))
    wanted = {}
    for i in range(len(fx["game"])):
        wanted.setdefault((int(fx["game"][i]), int(fx["stones"][i])), []).append(i)
    records, order = [], []
    for g, game in enumerate(json.load(open(os.path.join(GOLDEN, "real_games.json")))["games"]):
        board = api.Board(7.5)
        for n, (_, c, p) in enumerate(F.replay(game["moves"])):          # (the oracle's replay only supplies the move list here)
            board.place(c - 1, p % 19, p // 19)
            for i in wanted.get((g, n + 1), ()):
                records.append(board.record(int(fx["color"][i]) - 1, int(fx["symmetry"][i])))
                order.append(i)
    n = len(order)
    assert n == len(fx["game"]) == 2064
    got = api.board_records_to_packed(np.concatenate(records), n, device=0).reshape(n, api.PACKED_WORDS)
    masks, komi, color = fx["masks"][order], fx["komi"][order].astype(np.uint32), fx["color"][order]
    bad = np.argwhere(got[:, :361] != masks)
    assert bad.size == 0, (len(bad), bad[0], hex(int(got[bad[0][0], bad[0][1]])), hex(int(masks[bad[0][0], bad[0][1]])))
    assert np.array_equal(got[:, 361], np.where(color == 1, komi, komi << 16))          # the constant of the colour plane, as fp16 bits
    planes = [int(((masks >> c) & 1).sum()) for c in range(32)]
    print(f"{n} boards: device planes equal the oracle fixture; set bits per plane {planes}")
    assert all(count > 0 for count in planes)
