"""The search tree (scope row f-2, first part) through the C ABI, pinned to the reference's own tests:

    src/libdg_mcts/tree.rs:1773-1946      visit_order, virtual_loss, value_update, undo_trace
    src/libdg_utils/config.rs:352-375     intp_out_of_bounds_1/2, intp_lower_edge, intp_upper_edge, intp_mid
    src/libdg_mcts/choose.rs:108-140      choose_1, choose_2

The reference's tests run on an empty board, where every point is a legal move: the prior carries no -inf and the board
callback has nothing to do, so the tree can be exercised without any Go rules.
"""
import re

import numpy as np


def _prior_distribution(rng):
    """get_prior_distribution (tree.rs:1758-1771) on an empty board: random, normalised over the finite entries."""
    prior = rng.random(368).astype(np.float32)
    return prior / prior.sum()


def test_visit_order(api_full):
    """tree.rs:1773-1823."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(1)
    root = api.Node(api.BLACK, 0.5, _prior_distribution(rng))
    vloss = api.config.VLOSS_CNT
    choices = []
    prior = root.prior
    while True:
        trace = api.probe(root)
        if isinstance(trace, api.Trace):
            assert len(trace) == 1
            i = trace[0][2]
            assert i not in choices                     # never re-visit a child that has not finished expanding
            choices.append(i)
            assert root.with_(i).vcount() == vloss      # the virtual loss has been applied
            assert root.vtotal_count == len(choices) * vloss
            assert all(prior[o] >= prior[i] for o in choices)   # visited in the order of the prior
        else:
            assert all(root.with_(o).vcount() == vloss for o in choices)   # no virtual loss was added twice
            assert root.vtotal_count == len(choices) * vloss
            break
        assert len(choices) <= 362
    # the walk ends when a visited high-prior child (bonus / 33) outranks every unvisited one and is still "expanding"
    assert 300 < len(choices) <= 362 and trace == api.PROBE_CONFLICT


def test_virtual_loss(api_full):
    """tree.rs:1825-1860."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(2)
    root = api.Node(api.BLACK, 0.5, _prior_distribution(rng))
    vloss = api.config.VLOSS_CNT
    trace = api.probe(root)
    i = trace[0][2]
    assert root.with_(i).vcount() == vloss and root.vtotal_count == vloss
    api.insert(trace, api.BLACK, 0.9, _prior_distribution(rng))
    assert root.with_(i).vcount() == 0 and root.vtotal_count == 0
    assert root.with_(i).count() == 1 and root.total_count == 1


def test_value_update(api_full):
    """tree.rs:1862-1925: two parallel probes below a pre-expanded child; incremental mean 0.9 -> 0.85 -> 0.8."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    vloss = api.config.VLOSS_CNT
    root = api.Node(api.BLACK, 0.5, [1.0 if i == 60 else 0.0 for i in range(362)])
    other_prior = [0.5 if i in (61, 62) else 0.0 for i in range(362)]
    trace = api.probe(root)
    api.insert(trace, api.BLACK, 0.9, other_prior)
    assert 0.8999 <= root.with_(60).value() <= 0.9001
    assert root.with_(60).count() == 1 and root.total_count == 1
    assert root.with_(60).vcount() == 0 and root.vtotal_count == 0

    trace_1, trace_2 = api.probe(root), api.probe(root)
    assert trace_1[0][2] == 60 and trace_2[0][2] == 60
    assert 61 in (trace_1[1][2], trace_2[1][2]) and 62 in (trace_1[1][2], trace_2[1][2])
    # the value of the root's child is unchanged, only its virtual loss grew
    assert root.with_(60).value() == np.float32(0.9)
    assert root.with_(60).count() == 1 and root.total_count == 1
    assert root.with_(60).vcount() == 2 * vloss and root.vtotal_count == 2 * vloss

    api.insert(trace_1, api.WHITE, 0.2, other_prior)       # white at 0.2 is black at 0.8: (0.9 + 0.8) / 2
    assert root.with_(60).value() == np.float32(0.85)
    assert root.with_(60).count() == 2 and root.total_count == 2
    assert root.with_(60).vcount() == vloss and root.vtotal_count == vloss

    api.insert(trace_2, api.WHITE, 0.3, other_prior)
    assert root.with_(60).value() == np.float32(0.8)
    assert root.with_(60).count() == 3 and root.total_count == 3
    assert root.with_(60).vcount() == 0 and root.vtotal_count == 0


def test_undo_trace(api_full):
    """tree.rs:1927-1945: the second probe hits the child the first one is still expanding and is undone."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = api.Node(api.BLACK, 0.5, [1.0 if i == 60 else 0.0 for i in range(362)])
    assert isinstance(api.probe(root), api.Trace)
    assert api.probe(root) == api.PROBE_CONFLICT
    assert root.vtotal_count == api.config.VLOSS_CNT


def test_undo_restores_the_expanding_flag(api_full):
    """undo(trace, true) (tree.rs:1397-1409, used when a worker gives a probe up): the child can be chosen again."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = api.Node(api.BLACK, 0.5, [1.0 if i == 60 else 0.0 for i in range(362)])
    trace = api.probe(root)
    api.undo(trace, True)
    assert root.vtotal_count == 0 and root.with_(60).vcount() == 0
    again = api.probe(root)
    assert isinstance(again, api.Trace) and again[0][2] == 60


def test_no_result_when_every_prior_is_minus_infinity(api):
    root = api.Node(api.BLACK, 0.5, [-np.inf] * 362)
    assert api.probe(root) == api.PROBE_NO_RESULT


def test_intp(api_full):
    """config.rs:352-375."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    pts = [(0, 0.0), (100, 1.0)]
    g = api.config.get_intp_value
    assert g(pts, -100) == 0.0 and g(pts, 200) == 1.0 and g(pts, 0) == 0.0 and g(pts, 100) == 1.0
    assert g(pts, 40) == np.float32(0.4)
    # the shipped tables (config.rs:181-195) at their knots
    assert api.config.get_uct_exp(0) == np.float32(0.77392) and api.config.get_uct_exp(800) == np.float32(1.05439)
    assert api.config.get_fpu_reduce(6400) == np.float32(0.514494) and api.config.get_fpu_reduce(10 ** 6) == np.float32(0.514494)
    assert api.config.get_lcb_critical_value(1600) == np.float32(1.86943)
    assert api.config.VLOSS_CNT == 32


def test_choose(api_full):
    """choose.rs:108-140."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    assert api.choose([2, 1, 4, 3], 0.5, 1.0, 0.0) == (2, 4.0 / 7.0)
    assert api.choose([4, 3, 2, 1], 0.5, 1.0, 1.0) == (1, 1.0)


def test_argmax(api_full):
    """asm/argmax.rs: first maximal element of the last 8-group that holds one; None when nothing is finite."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    v = np.full(368, -np.inf, dtype=np.float32)
    assert api.argmax_f32(v) is None
    v[5] = 1.0; v[100] = 3.0; v[200] = 2.0
    assert api.argmax_f32(v) == 100
    v[102] = 3.0                                       # same group: the first one wins
    assert api.argmax_f32(v) == 100
    v[300] = 3.0                                       # a later group with the same maximum wins
    assert api.argmax_f32(v) == 300
    rng = np.random.default_rng(1)                     # the vector implementation against the rule, with many ties
    for _ in range(300):
        v = rng.integers(0, 4, 368).astype(np.float32)
        v[rng.random(368) < 0.3] = -np.inf
        so_far, index = -np.inf, 0
        for g in range(46):
            so_far = max(so_far, v[8 * g:8 * g + 8].max())
            hit = np.nonzero(v[8 * g:8 * g + 8] == so_far)[0]
            if hit.size:
                index = 8 * g + int(hit[0])
        assert api.argmax_f32(v) == (index if np.isfinite(v[index]) else None)


def test_best_and_softmax_follow_the_visit_counts(api_full):
    """Node::softmax (tree.rs:1293-1307) and the greedy Node::best (tree.rs:1263-1268, compare_children :1524-1560)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    prior = np.zeros(362, dtype=np.float32); prior[10] = 0.6; prior[20] = 0.4
    root = api.Node(api.BLACK, 0.5, prior)
    child_prior = np.full(362, 1.0 / 362, dtype=np.float32)
    for value in (0.7, 0.6, 0.8, 0.4, 0.5, 0.9):
        trace = api.probe(root)
        api.insert(trace, api.WHITE if len(trace) % 2 else api.BLACK, value, child_prior)
    counts = np.array([root.with_(i).count() for i in range(362)])
    assert counts.sum() == 6 and counts[10] + counts[20] == 6
    assert np.allclose(root.softmax(), counts / 6.0)
    value, index = root.best(0.0)
    assert index == int(np.argmax(counts)) and value == root.with_(index).value()
    root.disqualify(index)                             # tree.rs:1315-1320
    assert root.with_(index).count() == 0 and root.best(0.0)[1] != index
    # temperature: a visit-proportional draw, `at` stands in for thread_rng() (tree.rs:1269-1281)
    assert root.best(1.0, at=0.0)[1] in (10, 20)


# ---- search driver (search.cpp) with the deterministic CPU stand-in for the network -----------------------------------

def _visit_counts(root):
    return np.array([root.with_(i).count() for i in range(362)])


def _fresh_root(api):
    prior = np.full(362, 1.0 / 362, dtype=np.float32)
    return api.Node(api.BLACK, 0.5, prior)


def test_seeded_single_thread_search_is_reproducible(api_full):
    """The determinism the reference does not have (it draws from thread_rng everywhere): with one
    search thread the visit counts are a pure function of the seed and the predictor's outputs."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    runs = []
    for seed in (7, 7, 8):
        root = _fresh_root(api)
        evals, conflicts = api.search_run(root, api.SyntheticBoard(123), num_rollouts=200, num_threads=1, seed=seed)
        # the search ends at the limit, or before it once the runner-up cannot catch up (time_control/mod.rs:86-101)
        assert evals == root.total_count <= 200 and root.is_done(200) and conflicts == 0 and root.vtotal_count == 0
        assert root.total_count == 200 or root.min_promote_rollouts() > 200 - root.total_count
        runs.append(_visit_counts(root))
        assert runs[-1].sum() == evals
    assert np.array_equal(runs[0], runs[1])
    assert not np.array_equal(runs[0], runs[2])           # another seed draws other symmetries


def test_multi_threaded_search_keeps_the_tree_consistent(api_full):
    """Several threads in one tree (virtual loss + the expanding flag): every virtual loss is returned, every rollout
    is counted once, the search stops within threads - 1 rollouts of the limit (worker_thread.rs:144-146)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = _fresh_root(api)
    evals, conflicts = api.search_run(root, api.SyntheticBoard(5), num_rollouts=400, num_threads=8, seed=1)
    assert root.total_count <= 400 + 7 and root.total_count == evals and root.is_done(400)
    assert root.vtotal_count == 0
    counts = _visit_counts(root)
    assert counts.sum() == root.total_count and all(root.with_(i).vcount() == 0 for i in np.nonzero(counts)[0])
    assert np.isclose(root.softmax().sum(), 1.0)


def test_synthetic_selfplay_counts_every_evaluation(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    stats = api.selfplay_synthetic(num_games=3, num_moves=4, num_rollouts=20, threads_per_search=2, seed=11)
    assert stats["games"] == 3 and stats["moves"] == 12
    assert 12 * 2 <= stats["evals"] <= 12 * (21 + 1)       # at most 20 rollouts (+ up to threads - 1) + the root evaluation


def test_forward_keeps_the_sub_tree(api_full):
    """Node::forward (src/libdg_mcts/tree.rs:1225-1250): the child's statistics become the new root."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = _fresh_root(api)
    api.search_run(root, api.SyntheticBoard(9), num_rollouts=300, num_threads=1, seed=2)
    counts = _visit_counts(root)
    best = int(np.argmax(counts))
    below = root.with_(best).count()
    child = root.forward(best)
    assert child is not None and child.to_move == api.WHITE
    assert child.total_count == below - 1                         # the visit that expanded the child is not one of its own
    assert _visit_counts(child).sum() == child.total_count
    assert api.Node(api.BLACK, 0.5, np.full(362, 1.0 / 362, dtype=np.float32)).forward(17) is None    # nothing below: no tree


def test_set_prior_replaces_the_prior_only(api_full):
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = _fresh_root(api)
    api.search_run(root, api.SyntheticBoard(9), num_rollouts=50, num_threads=1, seed=2)
    before = _visit_counts(root)
    prior = np.random.default_rng(0).dirichlet(np.ones(362)).astype(np.float32)
    root.set_prior(prior)
    assert np.array_equal(before, _visit_counts(root))
    assert np.allclose(root.prior, prior)


def test_dirichlet_noise(api_full):
    """dirichlet::add_ex (src/libdg_mcts/dirichlet.rs:40-70, tests 80-101): mixes beta of a Dirichlet(shape) sample into the
    finite entries, keeps -inf entries, preserves the sum."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    x = np.full(362, 1.0 / 300, dtype=np.float32); x[300:] = -np.inf
    y = api.dirichlet_add(x, 0.03, 0.25, seed=5)
    assert np.all(np.isneginf(y[300:])) and np.all(np.isfinite(y[:300])) and np.all(y[:300] >= 0.75 / 300 - 1e-7)
    assert np.isclose(y[:300].sum(), 1.0, atol=1e-4)
    assert np.array_equal(y, api.dirichlet_add(x, 0.03, 0.25, seed=5)) and not np.array_equal(y, api.dirichlet_add(x, 0.03, 0.25, seed=6))
    assert np.array_equal(api.dirichlet_add(x, 0.03, 0.0, seed=5), x)


def test_early_termination_rule(api_full):
    """time_control::is_done (time_control/mod.rs:86-101) with min_promote_rollouts (:48-73): not done before the first
    visit; done at the limit; done early exactly when the lead of the most visited child exceeds the rollouts left."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    root = _fresh_root(api)
    assert not root.is_done(10) and root.min_promote_rollouts() == 0
    uniform = np.full(362, 1.0 / 362, dtype=np.float32)
    for visits_a, visits_b in ((3, 0), (5, 2)):
        root = _fresh_root(api)
        for index, n in ((17, visits_a), (40, visits_b)):
            only = np.full(362, -np.inf, dtype=np.float32); only[index] = 1.0
            for _ in range(n):                                  # steer the descent through child `index` of the root
                root.set_prior(only)
                trace = api.probe(root)
                assert isinstance(trace, api.Trace)
                api.insert(trace, api.WHITE, 0.5, uniform)
        root.set_prior(uniform)
        lead, total = root.min_promote_rollouts(), root.total_count
        assert lead == visits_a - visits_b and total == visits_a + visits_b
        assert root.with_(17).count() == visits_a and root.with_(40).count() == visits_b
        assert root.is_done(total)                              # the limit
        assert root.is_done(total + lead - 1)                   # lead > remaining: the runner-up cannot catch up
        assert not root.is_done(total + lead)                   # lead == remaining: it still can


def test_many_leaves_per_thread_stay_deterministic(api_full):
    """dgnn_search_run_batched: one worker with K leaves in flight probes them one after the other (virtual loss steers each
    probe away from the previous ones), evaluates them in ONE predictor call and inserts them -- still a pure function of the
    seed, unlike K threads; every virtual loss is returned; K = 1 is the plain search."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    calls = []

    def predictor(features, n):
        calls.append(n)
        f = np.asarray(features).view(np.float16).reshape(n, -1).astype(np.float32)
        value = np.tanh(f[:, ::97].sum(axis=1) * 0.05 - 0.5).astype(np.float16)
        logits = f[:, :362 * 8].reshape(n, 362, 8).sum(axis=2) * 0.5
        policy = np.exp(logits - logits.max(axis=1, keepdims=True))
        return value, (policy / policy.sum(axis=1, keepdims=True)).astype(np.float16)

    fn = api.python_predictor(predictor)
    runs = {}
    for leaves in (1, 8, 8):
        calls.clear()
        root = _fresh_root(api)
        evals, conflicts = api.search_run(root, api.SyntheticBoard(77), num_rollouts=120, num_threads=1, seed=3, pool=fn,
                                          leaves_per_thread=leaves)
        assert evals == root.total_count <= 120 and root.vtotal_count == 0 and root.is_done(120)
        assert conflicts == 0 or leaves > 1
        assert sum(calls) == evals and max(calls) <= leaves
        if leaves == 8:
            assert max(calls) == 8 and len(calls) < evals / 4          # the leaves really travel together
        runs.setdefault(leaves, []).append(_visit_counts(root))
    assert np.array_equal(runs[8][0], runs[8][1])
    plain = _fresh_root(api)
    api.search_run(plain, api.SyntheticBoard(77), num_rollouts=120, num_threads=1, seed=3, pool=fn)
    assert np.array_equal(runs[1][0], _visit_counts(plain))


def test_selfplay_with_batched_leaves_is_a_function_of_the_seed(api):
    kw = dict(num_games=3, num_moves=8, num_rollouts=30, threads_per_search=1, leaves_per_thread=6, seed=19, return_moves=True,
              cache_per_game=5000)       # a table per game: a shared one makes games depend on each other's timing (dgnn.h)
    a, b = api.selfplay_go(**kw), api.selfplay_go(num_parallel=1, **kw)
    assert a["cache_hits"] == b["cache_hits"] > 0
    # (a later probe of the same worker may run into the leaf an earlier one is holding: that ends the collection and is counted
    # as a conflict -- deterministically)
    assert np.array_equal(a["record"], b["record"]) and a["evals"] == b["evals"] and a["conflicts"] == b["conflicts"]
    for record in a["record"]:
        replay = api.Board(7.5)
        for m, index in enumerate(record):
            assert replay.is_valid(m % 2, int(index % 19), int(index // 19))
            replay.place(m % 2, int(index % 19), int(index // 19))


def test_to_pretty_summarises_a_search_like_the_reference(api_full):
    """tree::to_pretty (src/libdg_mcts/tree.rs:1683-1731): "Nodes: N, Win: W%, PV: path" and a line per child, best first, with the
    child's visits, value, lower confidence bound, prior and its own greedy path; vertices in GTP style (no i column)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    prior = np.full(362, 1e-3, dtype=np.float32)
    prior[3 * 19 + 3], prior[15 * 19 + 8], prior[361] = 0.5, 0.3, 0.01
    root = api.Node(api.BLACK, 0.5, prior / prior.sum())
    api.search_run(root, api.SyntheticBoard(7), num_rollouts=64, num_threads=1, seed=3)
    text = root.to_pretty()
    lines = text.strip("\n").split("\n")
    counts = np.array([root.with_(i).count() for i in range(362)])
    assert re.fullmatch(r"Nodes: %d, Win: \d+\.\d%%, PV: ([a-hj-t]\d+|pass)( ([a-hj-t]\d+|pass))*" % counts.sum(), lines[0]), lines[0]
    assert len(lines) == 1 + min(10, int((counts > 0).sum()))
    best_value, best = root.best(0.0)
    first = re.fullmatch(r" *([a-hj-t]\d+|pass) -> +(\d+) \(W: +[\d.]+% / +-?[\d.]+%\) \(N: +[\d.]+%\) PV: ([a-hj-t]\d+|pass) ?(.*)", lines[1])
    assert first, lines[1]
    vertex = "pass" if best == 361 else "abcdefghjklmnopqrst"[best % 19] + str(best // 19 + 1)
    assert first.group(1) == vertex == first.group(3) and lines[0].split("PV: ")[1].split(" ")[0] == vertex
    assert "i" not in "".join(l.split("PV:")[1] for l in lines).replace("pass", "")


def test_sparse_children_answer_like_a_dense_node(api_full):
    """tree.cpp keeps a slot per child that was ever selected (eight in the node, chunks of 16 .. 256 beyond, an index map from
    the sixteenth on); a child without a slot must read as Child::with_value (tree.rs:276-285).  A root is driven across every
    one of those boundaries -- probes that each pick a new child (the "expanding" flag keeps them apart), inserts in a shuffled
    order, undos -- and after every step all 362 children, the totals, softmax, best and the early-termination lead are compared
    with a dense restatement kept in numpy."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    rng = np.random.default_rng(5)
    vloss = api.config.VLOSS_CNT
    prior = _prior_distribution(rng)
    root = api.Node(api.BLACK, 0.5, prior)
    count, vcount, value = np.zeros(362, dtype=np.int64), np.zeros(362, dtype=np.int64), np.full(362, np.float32(0.5))

    def check():
        for i in range(362):
            child = root.with_(i)
            assert (child.count(), child.vcount()) == (count[i], vcount[i]), i
            assert child.value() == value[i], i
        assert root.total_count == count.sum() and root.vtotal_count == vcount.sum()
        visits = count.astype(np.float32)
        expected = np.where(count != 0, visits / np.float32(visits.sum()), 0.0) if count.sum() else np.zeros(362)
        assert np.array_equal(root.softmax(), expected.astype(np.float32))
        top = np.sort(count)[::-1]
        assert root.min_promote_rollouts() == max(int(top[0] - top[1]), 0)

    check()
    touched, steps = set(), iter((3, 5, 1, 7, 9, 31, 64) + (130,) * 40)
    while len(touched) < 260:                            # slot counts around 8, 16, 24, 56, 120, 248: every chunk and the map
        traces = []
        for _ in range(next(steps)):
            trace = api.probe(root)
            if not isinstance(trace, api.Trace):
                assert trace == api.PROBE_CONFLICT       # (the best child is still being expanded: nothing was changed)
                continue
            i = trace[0][2]                              # (a visited child may be chosen again and the probe go on below it)
            touched.add(i)
            vcount[i] += vloss
            traces.append(trace)
        check()
        order = rng.permutation(len(traces))
        for k in order[: len(order) // 4]:               # a quarter is undone: the slot stays, the child reads as untouched
            vcount[traces[k][0][2]] -= vloss
            api.undo(traces[k], True)
        for k in order[len(order) // 4:]:
            i, v = traces[k][0][2], np.float32(0.4 * rng.random())   # (below the 0.5 of an unvisited child: the search keeps widening)
            api.insert(traces[k], api.BLACK, float(v), _prior_distribution(rng))
            vcount[i] -= vloss
            count[i] += 1
            value[i] = value[i] + (v - value[i]) / np.float32(count[i])
        check()
    assert int((count > 0).sum()) > 150
    _, best = root.best(0.0)
    assert count[best] > 0 and root.with_(best).has_node()
    # a child that never got a slot has no sub-tree and reads as Child::with_value
    untouched = int(np.flatnonzero((count == 0) & (vcount == 0))[-1])
    assert not root.with_(untouched).has_node() and root.with_(untouched).value() == np.float32(0.5)


def test_many_threads_in_one_root_cross_the_slot_map_boundary_safely(api_full):
    """a node keeps its first 15 child slots in the node and in one 16-slot chunk, and an index -> slot map from the
    sixteenth on.  A reader that saw "no map yet" and THEN a slot count past 23 (other threads adding slots in between) scanned
    past the chunk.  Now the map is published before the count that needs it and the reader switches to it.  Sixteen search
    threads probing fresh roots with a flat prior (every probe's virtual loss sends the next one to a new child) cross that
    boundary in every attempt; the tree must come out consistent every time (ASan / TSan run the same stress in tools/sanitize_host.sh)."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import threading
    prior = np.full(362, 1.0 / 362, dtype=np.float32)
    for attempt in range(20):
        root = api.Node(api.BLACK, 0.5, prior)
        start = threading.Barrier(16)
        inserted = []

        def work():
            start.wait()
            mine = 0
            for _ in range(6):                                             # (ctypes releases the GIL: the probes run side by side)
                trace = api.probe(root)
                if isinstance(trace, int):
                    continue
                api.insert(trace, 1 - trace[len(trace) - 1][1], 0.5, prior)
                mine += 1
            inserted.append(mine)

        threads = [threading.Thread(target=work) for _ in range(16)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert root.total_count == sum(inserted) and root.vtotal_count == 0
        counts = _visit_counts(root)
        assert counts.sum() == root.total_count and int((counts > 0).sum()) > 24        # well past the first chunk: the map is in use


def test_nodes_freed_on_another_thread_than_the_one_that_made_them(api_full):
    """tree nodes and leaf boards come from per-thread free lists (dgnn_internal.h BlockPool); the sanitizer builds bypass
    them.  The shipped path: trees built on one thread and freed on another (what pool workers and migrating fibers do), more nodes
    than a list keeps, and a thread that ends with blocks in its list."""
    api = api_full          # (entry points of include/dgnn_test.h: the test build of the library)
    import threading
    trees = []

    def build():
        for seed in range(6):
            root = _fresh_root(api)
            api.search_run(root, api.SyntheticBoard(7 + seed), num_rollouts=900, num_threads=1, seed=seed)
            trees.append(root)

    t = threading.Thread(target=build)
    t.start()
    t.join()
    assert all(r.total_count > 0 and r.is_done(900) for r in trees)
    trees.clear()                                                         # their nodes go to THIS thread's list (it keeps 2 048, the rest goes to the heap)
    root = _fresh_root(api)                                               # ... and are handed out again here
    evals, _ = api.search_run(root, api.SyntheticBoard(3), num_rollouts=3000, num_threads=4, seed=9)
    assert root.total_count == evals and root.is_done(3000) and root.vtotal_count == 0
