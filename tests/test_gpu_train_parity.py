"""-m gpu: the train path AS THE ENGINE RUNS IT against the oracle, at the sizes the engine trains.

`Network.train` on one GPU runs the fused optimizer update (RMSProp applied by the kernels that complete each gradient
element), does not store dn1, and reads uint8 states where the transport delivers them (ga3c_engine.hip: train_grads /
launch_backward).  ThreadTrainer assembles TRAINING_MIN_BATCH_SIZE + 1 .. + TIME_MAX + 1 rows (ThreadTrainer.py:49-59): with
the "batch = 128" setting of BASELINE configs[2] that is 128 .. 132 rows.  Round 2 held these kernel variants to the oracle
at B <= 24 only; here: weights and the RMSProp `ms` slot after two steps at B = 128 / 129 / 132,
A = 6 / 18, both input formats, through the C ABI, against oracle.train_step (NetworkVP_discrate.py:99-105,130).

Tolerance: 1e-5 absolute on weights after two steps of lr = 3e-4 (the test of round 1 at B = 24 uses the same bar): a
weight moves by ~lr / sqrt(ms + eps) ~ 3e-4 per step, so 1e-5 is ~2 % of one step's movement; `ms` to 1e-5 relative to
its largest entry.
"""
import os

import numpy as np
import pytest

import ga3c_oracle as o

pytestmark = pytest.mark.gpu

LR, BETA = 3e-4, 0.01


def _flat(d):
    return np.concatenate([np.asarray(d[k]).reshape(-1) for k in o.PARAM_ORDER])


def _batch(bsz, num_actions, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    xk = rng.integers(0, 256, size=(bsz, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128.0) - np.float32(1.0)
    act = rng.integers(0, num_actions, size=bsz)
    y = rng.uniform(-1, 1, size=bsz)
    return xk, x, np.eye(num_actions, dtype=np.float32)[act], y


@pytest.fixture(scope="module")
def nets():
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    made = {}

    def get(num_actions):
        if num_actions not in made:
            made[num_actions] = Network("gpu:0", "train_parity", num_actions, (84, 84, 4), max_batch=148, predict_lanes=1)
        return made[num_actions]
    yield get
    for n in made.values():
        n.close()


_ORACLE = {}


def _oracle_two_steps(num_actions, bsz):
    """(weights, ms, losses of step 1) after two oracle steps from the seeded initial weights; cached per (A, B)."""
    key = (num_actions, bsz)
    if key not in _ORACLE:
        _, x, a, y = _batch(bsz, num_actions, 7000 + 10 * bsz + num_actions)
        params = o.init_params(num_actions)
        ms = {k: np.ones_like(v) for k, v in params.items()}
        first = None
        for _ in range(2):
            losses, _ = o.train_step(params, ms, x.astype(np.float64), y, a.astype(np.float64), LR, BETA)
            first = first or losses
        _ORACLE[key] = (_flat(params), _flat(ms), np.array([first["cost_p_1_agg"], first["cost_p_2_agg"], first["cost_v"]]))
    return _ORACLE[key]


def _reset(net, num_actions):
    net.set_arena(0, _flat(o.init_params(num_actions)).astype(np.float32))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    net.set_arena(2, np.zeros(net.param_count, np.float32))
    net.learning_rate, net.beta = LR, BETA


# 133: the last size whose rows past the first 128 wait in dense1_bwd_tile's tail area (D1B_TAIL_ROWS = 5); 134 / 145: a second
# chunk, with its dn2 tile cut over the waves (<= 16 rows) and without; beyond 128 rows dense1/w is stepped by conv2_dx_wd
@pytest.mark.parametrize("fmt", ["f32", "u8"])
@pytest.mark.parametrize("bsz", [128, 129, 132, 133, 134, 145])
@pytest.mark.parametrize("num_actions", [6, 18])
def test_production_train_step_matches_oracle(nets, num_actions, bsz, fmt):
    net = nets(num_actions)
    xk, x, a, y = _batch(bsz, num_actions, 7000 + 10 * bsz + num_actions)
    want_w, want_ms, want_l = _oracle_two_steps(num_actions, bsz)
    _reset(net, num_actions)
    step0 = net.get_global_step()
    net.train(xk if fmt == "u8" else x, y, a)
    first = np.array(net.last_losses, np.float64)
    net.train(xk if fmt == "u8" else x, y, a)
    assert net.get_global_step() == step0 + 2
    assert np.allclose(first, want_l, rtol=1e-4, atol=1e-4), (first, want_l)
    got_w, got_ms = net.get_arena(0), net.get_arena(1)
    assert np.max(np.abs(got_w - want_w)) < 1e-5, np.max(np.abs(got_w - want_w))
    assert np.max(np.abs(got_ms - want_ms)) < 1e-5 * max(1.0, np.max(np.abs(want_ms)))
    # the step is real and tensor-wise right: per tensor the update has the oracle's norm within 1e-3
    init = _flat(o.init_params(num_actions))
    off = 0
    for name in o.PARAM_ORDER:
        size = int(np.prod(o.param_shapes(num_actions)[name]))
        dw = np.linalg.norm(want_w[off:off + size] - init[off:off + size])
        dg = np.linalg.norm(got_w[off:off + size].astype(np.float64) - init[off:off + size])
        assert dw > 0 and abs(dg - dw) < 1e-3 * dw + 1e-7, (name, dg, dw)
        off += size


@pytest.mark.parametrize("bsz", [128, 132, 134])
def test_u8_and_f32_production_steps_give_the_same_bits(nets, bsz):
    net = nets(6)
    xk, x, a, y = _batch(bsz, 6, 4242 + bsz)
    outs = []
    for xin in (x, xk):
        _reset(net, 6)
        net.train(xin, y, a)
        net.train(xin, y, a)
        outs.append((net.get_arena(0), net.get_arena(1)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_dense1_weight_step_inside_conv_bwd_gives_the_bits_of_the_epilogue_step(nets, monkeypatch):
    """The fused update steps dense1/w inside conv_bwd (beside its MFMA phases) instead of in dense1_bwd_tile's epilogue
    (GA3C_WD_STEP_IN_CONV_BWD=0 keeps the epilogue, 2 defers at every batch size, the default from 121 rows on).  Same arithmetic per element: weights, ms and -- through the
    fragment-ordered copy of dense1/w that predictions read -- the predictions after the steps must be bit-identical, at a
    full grid (128 rows: one 16-row group per workgroup), at grids below the 242 groups (8, 40, 120 rows: the leftover
    loop) and with momentum."""
    import ga3c_amd  # noqa: F401
    import Config
    from NetworkVP import Network
    # GA3C_CONV_BWD_MIN=1: the fused conv_bwd at every size up to 128 rows (by default the split launches take batches below
    # 97 rows since round 4, so the default network is compared from there on)
    monkeypatch.setenv("GA3C_CONV_BWD_MIN", "1")
    monkeypatch.setenv("GA3C_WD_STEP_IN_CONV_BWD", "2")
    ref = Network("gpu:0", "wd_conv_bwd", 6, (84, 84, 4), max_batch=136, predict_lanes=1)
    monkeypatch.setenv("GA3C_WD_STEP_IN_CONV_BWD", "0")
    net = Network("gpu:0", "wd_epilogue", 6, (84, 84, 4), max_batch=136, predict_lanes=1)
    monkeypatch.delenv("GA3C_WD_STEP_IN_CONV_BWD")
    monkeypatch.delenv("GA3C_CONV_BWD_MIN")
    dflt = nets(6)
    try:
        for bsz in (8, 40, 120, 121, 128):
            xk, x, a, y = _batch(bsz, 6, 9100 + bsz)
            outs = []
            for n in (ref, net, dflt):
                _reset(n, 6)
                for xin in (x, xk, x):
                    n.train(xin, y, a)
                outs.append((n.get_arena(0), n.get_arena(1), n.predict_p_v_logits(x)))
            for other in outs[1:] if bsz >= 97 else outs[1:2]:
                assert np.array_equal(outs[0][0], other[0]) and np.array_equal(outs[0][1], other[1]), bsz
                assert all(np.array_equal(g, w) for g, w in zip(outs[0][2], other[2])), bsz
            assert not np.array_equal(outs[0][0], _flat(o.init_params(6)).astype(np.float32))
    finally:
        net.close()
        ref.close()
    # momentum: a Network reads the optimizer's constants from Config when it is created
    monkeypatch.setattr(Config.Config, "RMSPROP_MOMENTUM", 0.5)
    pair = []
    monkeypatch.setenv("GA3C_CONV_BWD_MIN", "1")
    for flag in ("2", "0"):
        monkeypatch.setenv("GA3C_WD_STEP_IN_CONV_BWD", flag)
        pair.append(Network("gpu:0", "wd_mom" + flag, 6, (84, 84, 4), max_batch=136, predict_lanes=1))
    monkeypatch.delenv("GA3C_WD_STEP_IN_CONV_BWD")
    monkeypatch.delenv("GA3C_CONV_BWD_MIN")
    try:
        xk, x, a, y = _batch(64, 6, 9164)
        outs = []
        for n in pair:
            _reset(n, 6)
            for _ in range(3):
                n.train(x, y, a)
            outs.append((n.get_arena(0), n.get_arena(1), n.get_arena(2)))
        assert all(np.array_equal(g, w) for g, w in zip(outs[0], outs[1]))
        assert np.any(outs[0][2] != 0)
    finally:
        for n in pair:
            n.close()


def test_split_path_scheduling_switches_leave_the_bits_alone(monkeypatch):
    """Beyond the fused conv kernels' 128 rows (round 4): dense1/w stepped by workgroups of their own in conv2_dx's launch
    (GA3C_WD_STEP_IN_CONV2_DX, at the back or the front of the grid: GA3C_WD_BLOCKS_FIRST), conv2_dw cut for three
    workgroups per CU (GA3C_C2DW_OCC), dense1_bwd_tile's rows past 128 worked on out of the tail area beside the first chunk
    (GA3C_D1B_TAIL; off: a second chunk whose dn2 tile is cut over the waves the same way), conv2_dw and conv1_dw side by
    side in one launch behind conv2_dx (GA3C_DW_PAIR; the number of conv1_dw workgroups = partial slabs is a function of the
    batch size alone).  Each moves work, none changes an
    element's arithmetic or a sum's order: weights, `ms`, momentum and -- through the fragment-ordered copy of dense1/w --
    the predictions after three steps are bit-identical with all of them off, at 129 / 132 / 133 (tail area), 134 / 140
    (second chunk) and 100 rows (split path below 128: GA3C_CONV_BWD=0)."""
    import ga3c_amd  # noqa: F401
    import Config
    from NetworkVP import Network
    monkeypatch.setattr(Config.Config, "RMSPROP_MOMENTUM", 0.5)
    monkeypatch.setenv("GA3C_CONV_BWD", "0")
    settings = [{}, {"GA3C_WD_STEP_IN_CONV2_DX": "0", "GA3C_C2DW_OCC": "2", "GA3C_D1B_TAIL": "0", "GA3C_DW_PAIR": "0", "GA3C_C2F_QUARTER": "0"},
                {"GA3C_WD_BLOCKS_FIRST": "1"}]
    made = []
    try:
        for i, env in enumerate(settings):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            made.append(Network("gpu:0", "split_sw%d" % i, 6, (84, 84, 4), max_batch=140, predict_lanes=1))
            for k in env:
                monkeypatch.delenv(k)
        for bsz in (8, 40, 100, 129, 132, 133, 134, 140):
            xk, x, a, y = _batch(bsz, 6, 9300 + bsz)
            outs = []
            for n in made:
                _reset(n, 6)
                for xin in (x, xk, x):
                    n.train(xin, y, a)
                outs.append((n.get_arena(0), n.get_arena(1), n.get_arena(2)) + tuple(n.predict_p_v_logits(x[:min(bsz, 128)])))
            for other in outs[1:]:
                assert all(np.array_equal(g, w) for g, w in zip(outs[0], other)), bsz
            assert np.any(outs[0][2] != 0) and not np.array_equal(outs[0][0], _flat(o.init_params(6)).astype(np.float32))
    finally:
        for n in made:
            n.close()


@pytest.mark.parametrize("num_actions", [6, 18])
def test_train_offsets_on_132_transport_rows_matches_oracle(num_actions):
    """The zero-copy trainer path at the engine's own batch: 132 rows lying in 22 rollout slots of the registered
    transport (6 rows each, the TIME_MAX + 1 rows of ProcessAgent.py:145-178), trained through ga3c_net_train_gather."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    import Transport as tp
    bsz = 132
    t = tp.Transport.create(tp.unique_name("t_132"), 4, num_actions, 84 * 84 * 4, 24, 6)
    net = Network("gpu:0", "rows132", num_actions, (84, 84, 4), max_batch=136, predict_lanes=1)
    try:
        net.register_transport(t)
        xk, x, a, y = _batch(bsz, num_actions, 7000 + 10 * bsz + num_actions)
        want_w, want_ms, _ = _oracle_two_steps(num_actions, bsz)
        offs = []
        rows = xk.reshape(bsz, -1)
        for k, slot in enumerate(reversed(range(22))):          # slots in an order of their own: the offsets are scattered
            states, _, _ = t.rollout_views(slot)
            states[:6] = rows[6 * k:6 * k + 6]
            offs.append(t.rollout_row_offsets(slot, 6))
        offs = np.concatenate(offs)
        _reset(net, num_actions)
        net.train_offsets(offs, y, a)
        net.train_offsets(offs, y, a)
        got_w, got_ms = net.get_arena(0), net.get_arena(1)
        assert np.max(np.abs(got_w - want_w)) < 1e-5
        assert np.max(np.abs(got_ms - want_ms)) < 1e-5 * max(1.0, np.max(np.abs(want_ms)))
        # and bit-identical to the same rows handed over as a host batch
        _reset(net, num_actions)
        net.train(xk, y, a)
        net.train(xk, y, a)
        assert np.array_equal(got_w, net.get_arena(0)) and np.array_equal(got_ms, net.get_arena(1))
    finally:
        net.close()
        t.shutdown()
        t.close()


def test_golden_fixture_losses_gradients_and_update_norms(nets, golden_dir):
    """tests/golden/nn_small.npz beyond its forward entries: losses, dz, dv, every gradient tensor (dense1/w as the stored
    strided sample) and, through the production train step, the norm of each tensor's update."""
    z = np.load(os.path.join(golden_dir, "nn_small.npz"))
    for num_actions in (6, 4, 18):
        import ga3c_amd  # noqa: F401
        from NetworkVP import Network
        net = nets(num_actions) if num_actions in (6, 18) else Network("gpu:0", "golden4", 4, (84, 84, 4), max_batch=8,
                                                                       predict_lanes=1)
        try:
            t = "A%d_" % num_actions
            xk = z[t + "x_u8"]
            x = xk.astype(np.float32) / np.float32(128.0) - np.float32(1.0)
            a = np.eye(num_actions, dtype=np.float32)[z[t + "actions"]]
            y = z[t + "y_r"]
            _reset(net, num_actions)
            losses = net.compute_grads(x, y, a)
            assert np.allclose(losses, z[t + "losses"], rtol=1e-4, atol=1e-4)
            for name in ("dz", "dv"):
                want = z[t + name]
                got = net.fetch(name, want.size).reshape(want.shape)
                assert np.max(np.abs(got - want)) < 1e-4 * max(1.0, np.max(np.abs(want))), name
            grad = net.get_arena(3)
            off = 0
            for name in o.PARAM_ORDER:
                shape = o.param_shapes(num_actions)[name]
                size = int(np.prod(shape))
                g = grad[off:off + size].astype(np.float64)
                off += size
                key = name.replace("/", ".")
                want = z[t + "grad." + key]
                got = g.reshape(shape) if want.size == size else g[::997]
                scale = max(1.0, np.max(np.abs(want)))
                assert np.max(np.abs(got - want)) < 1e-4 * scale, name
                gn = float(z[t + "gnorm." + key])
                assert abs(np.linalg.norm(g) - gn) < 1e-4 * max(1.0, gn), name
            # one production step (fused update): the movement of every tensor has the fixture's norm
            init = net.get_arena(0).astype(np.float64)
            net.train(xk, y, a)
            moved = net.get_arena(0).astype(np.float64) - init
            off = 0
            for name in o.PARAM_ORDER:
                size = int(np.prod(o.param_shapes(num_actions)[name]))
                want = float(z[t + "delta_norm." + name.replace("/", ".")])
                got = np.linalg.norm(moved[off:off + size])
                off += size
                # f32 weights: the stored delta is rounded to the weight's ulp, a relative 1e-3 of the norm covers it
                assert abs(got - want) < 1e-3 * want + 1e-7, (name, got, want)
        finally:
            if num_actions == 4:
                net.close()


def test_two_trainer_threads_pipeline_and_stay_reproducible():
    """Config.TRAINERS = 2 (Server.py:132-134): two threads call train_offsets concurrently.  Each thread stages its rows into an
    intake of its own on the staging stream while the other thread's step is in flight; the steps themselves are taken one
    after the other (the lane's mutex), so with both threads training the SAME rows the result must equal the same number of
    sequential steps from one thread bit for bit -- whatever the interleaving was -- and the engine's counters must show
    both calls and rows."""
    import threading
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    import Transport as tp
    bsz, per_thread = 132, 6
    t = tp.Transport.create(tp.unique_name("t_pipe"), 4, 6, 84 * 84 * 4, 24, 6)
    net = Network("gpu:0", "pipe", 6, (84, 84, 4), max_batch=136, predict_lanes=1)
    try:
        net.register_transport(t)
        xk, _, a, y = _batch(bsz, 6, 999)
        rows = xk.reshape(bsz, -1)
        offs = []
        for k in range(22):
            states, _, _ = t.rollout_views(k)
            states[:6] = rows[6 * k:6 * k + 6]
            offs.append(t.rollout_row_offsets(k, 6))
        offs = np.concatenate(offs)
        _reset(net, 6)
        for _ in range(2 * per_thread):
            net.train_offsets(offs, y, a)
        want_w, want_ms = net.get_arena(0), net.get_arena(1)
        _reset(net, 6)
        net.stats(reset=True)
        errors = []

        def trainer():
            try:
                for _ in range(per_thread):
                    net.train_offsets(offs, y, a)
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
        ths = [threading.Thread(target=trainer) for _ in range(2)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(60)
        assert not errors, errors
        assert np.array_equal(net.get_arena(0), want_w) and np.array_equal(net.get_arena(1), want_ms)
        st = net.stats()
        assert st["train_calls"] == 2 * per_thread and st["train_rows"] == 2 * per_thread * bsz
        assert st["train_sync_ns"] > 0 and st["train_launch_ns"] > 0 and st["train_stage_ns"] > 0
    finally:
        net.close()
        t.shutdown()
        t.close()


def test_four_prediction_lanes_on_two_streams_agree_with_one_lane():
    """Lanes beyond two share the two prediction streams (ga3c_net_create: the engine keeps to four busy streams).  Four
    threads predict concurrently through four lanes; every answer must equal the single-lane answer bit for bit, and a lane
    must never return before ITS step is done although another lane's later work sits on the same stream."""
    import threading
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    one = Network("gpu:0", "one", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    four = Network("gpu:0", "four", 6, (84, 84, 4), max_batch=64, predict_lanes=4)
    try:
        batches = [_batch(7 + 9 * k, 6, 50 + k)[0] for k in range(4)]
        want = [one.predict_p_and_v(b) for b in batches]
        errors = []

        def worker(k):
            try:
                for _ in range(40):
                    p, v = four.predict_p_and_v(batches[k])
                    if not (np.array_equal(p, want[k][0]) and np.array_equal(v, want[k][1])):
                        errors.append("lane thread %d: answer differs from the single-lane answer" % k)
                        return
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(60)
        assert not errors, errors
        st = four.stats()
        assert st["predict_calls"] == 160 and st["predict_rows"] == 40 * sum(b.shape[0] for b in batches)
    finally:
        one.close()
        four.close()


def test_comm_info_reports_what_the_communicator_says():
    """bench.py's rccl_ranks comes from ncclCommCount / ncclCommUserRank of the attached communicator (ga3c_net_comm_info),
    not from the launcher's environment: (0, -1, -1) without one, (1, 0, device) for a one-rank communicator."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    net = Network("gpu:0", "ci", 6, (84, 84, 4), max_batch=8, predict_lanes=1)
    try:
        assert net.comm_info() == (0, -1, -1)
        net.comm_init(Network.make_comm_id(), 0, 1)
        assert net.comm_info() == (1, 0, 0)
    finally:
        net.close()
