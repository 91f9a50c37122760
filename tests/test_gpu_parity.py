"""-m gpu: parity of the HIP path (through the C ABI) against the oracle.

Tolerances: BASELINE.json asks for policy/value within 1e-4 in fp32; gradients and the optimizer
step are held to 1e-4 relative to the largest entry of each tensor (f32 summation order differs
from the f64 oracle, nothing else does).
"""
import os

import numpy as np
import pytest

import ga3c_oracle as o

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def nets():
    import ga3c_amd  # noqa: F401  (puts the flat modules on sys.path)
    from NetworkVP import Network
    made = {}

    def get(num_actions, max_batch=160):
        key = (num_actions, max_batch)
        if key not in made:
            made[key] = Network("gpu:0", "test", num_actions, (84, 84, 4), max_batch=max_batch, predict_lanes=2)
        return made[key]
    yield get
    for n in made.values():
        n.close()


def _oracle_params(net):
    arena = net.get_arena(0).astype(np.float64)
    out, off = {}, 0
    for name in o.PARAM_ORDER:
        shape = o.param_shapes(net.num_actions)[name]
        size = int(np.prod(shape))
        out[name] = arena[off:off + size].reshape(shape)
        off += size
    return out


def _flat(d, num_actions):
    return np.concatenate([np.asarray(d[k]).reshape(-1) for k in o.PARAM_ORDER])


def _batch(bsz, num_actions, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    xk = rng.integers(0, 256, size=(bsz, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128.0) - np.float32(1.0)
    act = rng.integers(0, num_actions, size=bsz)
    y = rng.uniform(-1, 1, size=bsz)
    return xk, x, np.eye(num_actions, dtype=np.float32)[act], y


def test_init_matches_oracle_init(nets):
    net = nets(6)
    want = _flat(o.init_params(6, seed=12345), 6)
    got = net.get_arena(0)
    assert got.shape == (1005623,)
    assert np.array_equal(got, want.astype(np.float32))


@pytest.mark.parametrize("num_actions,bsz", [(6, 1), (6, 5), (6, 37), (6, 128), (6, 130), (6, 145), (4, 19), (18, 33)])
def test_forward_matches_oracle(nets, num_actions, bsz):
    net = nets(num_actions)
    _, x, _, _ = _batch(bsz, num_actions, 100 + bsz)
    p, v, z = net.predict_p_v_logits(x)
    ref = o.forward(_oracle_params(net), x.astype(np.float64))
    assert np.max(np.abs(p - ref["p"])) < TOL
    assert np.max(np.abs(v - ref["v"])) < TOL
    assert np.max(np.abs(z - ref["z"])) < TOL
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-5)


def test_forward_golden_fixture(nets, golden_dir):
    z = np.load(os.path.join(golden_dir, "nn_small.npz"))
    for num_actions in (6, 4, 18):
        net = nets(num_actions)
        net.set_arena(0, _flat(o.init_params(num_actions), num_actions))
        t = "A%d_" % num_actions
        x = z[t + "x_u8"].astype(np.float32) / np.float32(128.0) - np.float32(1.0)
        p, v, logits = net.predict_p_v_logits(x)
        assert np.max(np.abs(p - z[t + "p"])) < TOL
        assert np.max(np.abs(v - z[t + "v"])) < TOL
        assert np.max(np.abs(logits - z[t + "z"])) < TOL


def test_u8_path_is_bit_identical_to_f32_path(nets):
    net = nets(6)
    xk, x, _, _ = _batch(9, 6, 77)
    p1, v1, z1 = net.predict_p_v_logits(x)
    p2, v2, z2 = net.predict_p_v_logits(xk)
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2) and np.array_equal(z1, z2)


def test_activations_match_oracle(nets):
    net = nets(6)
    _, x, a, y = _batch(7, 6, 5)
    net.compute_grads(x, y, a)
    ref = o.forward(_oracle_params(net), x.astype(np.float64), keep=True)
    for name, want in (("n1", ref["n1"]), ("n2", ref["n2"]), ("d1", ref["d1"])):
        got = net.fetch(name, want.size).reshape(want.shape)
        assert np.max(np.abs(got - want)) < TOL, name


# 96 / 97: the split conv backward below, the fused conv_bwd from 97 rows on; 131 / 143 / 150 / 160: conv2_dw + conv1_dw in one
# launch with 768 - 4 B conv1_dw workgroups (to 142 rows) and with 256; 144: a second chunk of exactly 16 rows in dense1_bwd_tile
@pytest.mark.parametrize("num_actions,bsz,flags", [(6, 1, {}), (6, 2, {}), (6, 5, {}), (6, 37, {}), (6, 96, {}), (6, 97, {}), (6, 128, {}),
                                                   (6, 131, {}), (6, 143, {}), (6, 144, {}), (6, 150, {}), (6, 160, {}),
                                                   (4, 16, {}), (18, 21, {}), (1, 9, {}), (25, 7, {}), (64, 6, {})])
def test_gradients_match_oracle(nets, num_actions, bsz, flags):
    net = nets(num_actions)
    _, x, a, y = _batch(bsz, num_actions, 300 + bsz)
    net.beta = 0.01
    losses = net.compute_grads(x, y, a)
    params = _oracle_params(net)
    ref_l, ref_g = o.loss_and_grads(params, x.astype(np.float64), y, a.astype(np.float64), 0.01)
    want_l = np.array([ref_l["cost_p_1_agg"], ref_l["cost_p_2_agg"], ref_l["cost_v"]])
    assert np.allclose(losses, want_l, rtol=1e-4, atol=1e-4)
    for name, want in (("dz", ref_g["dz"]), ("dv", ref_g["dv"]), ("dd1", ref_g["dd1"]), ("dn2", ref_g["dn2"]),
                       ("dn1", ref_g["dn1"])):
        got = net.fetch(name, want.size).reshape(want.shape)
        scale = max(np.max(np.abs(want)), 1e-6)
        assert np.max(np.abs(got - want)) < TOL * max(scale, 1.0), name
    got = net.get_arena(3)
    off = 0
    for name in o.PARAM_ORDER:
        want = np.asarray(ref_g[name]).reshape(-1)
        g = got[off:off + want.size]
        off += want.size
        scale = max(np.max(np.abs(want)), 1.0)
        assert np.max(np.abs(g - want)) < TOL * scale, (name, np.max(np.abs(g - want)), scale)


def test_train_step_matches_oracle_rmsprop(nets):
    net = nets(6)
    net.set_arena(0, _flat(o.init_params(6), 6))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    _, x, a, y = _batch(24, 6, 9)
    params = _oracle_params(net)
    ms = {k: np.ones_like(v) for k, v in params.items()}
    net.learning_rate, net.beta = 3e-4, 0.01
    step0 = net.get_global_step()
    for it in range(3):
        net.train(x, y, a, None, None, 0)
        o.train_step(params, ms, x.astype(np.float64), y, a.astype(np.float64), 3e-4, 0.01)
    assert net.get_global_step() == step0 + 3
    got, want = net.get_arena(0), _flat(params, 6)
    assert np.max(np.abs(got - want)) < 1e-5
    got_ms, want_ms = net.get_arena(1), _flat(ms, 6)
    assert np.max(np.abs(got_ms - want_ms)) < 1e-5 * max(1.0, np.max(np.abs(want_ms)))
    # the update is real: weights moved by about lr/sqrt(1.1) per nonzero gradient entry
    assert np.max(np.abs(got - _flat(o.init_params(6), 6))) > 1e-4


def test_train_is_reproducible_bit_for_bit(nets):
    net = nets(6)
    _, x, a, y = _batch(40, 6, 21)
    outs = []
    for _ in range(2):
        net.set_arena(0, _flat(o.init_params(6), 6))
        net.set_arena(1, np.ones(net.param_count, np.float32))
        net.train(x, y, a, None, None, 0)
        outs.append(net.get_arena(0))
    assert np.array_equal(outs[0], outs[1])


def test_bad_shapes_are_rejected(nets):
    net = nets(6, 160)
    _, x, a, y = _batch(2, 6, 1)
    with pytest.raises(RuntimeError):
        net.predict_p_and_v(np.zeros((161, 84, 84, 4), np.float32))
    with pytest.raises(RuntimeError):
        net.predict_p_and_v(np.zeros((0, 84, 84, 4), np.float32))


def test_rccl_single_rank_communicator_allreduce_is_identity():
    """The multi-GPU exchange step cannot be exercised on a one-GPU box; this checks that the RCCL path
    links, initialises and runs (a 1-rank sum all-reduce leaves the gradient arena unchanged)."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    net = Network("gpu:0", "dp", 6, (84, 84, 4), max_batch=16, predict_lanes=1)
    try:
        _, x, a, y = _batch(8, 6, 3)
        net.compute_grads(x, y, a)
        before = net.get_arena(3)
        net.comm_init(Network.make_comm_id(), 0, 1)
        import _native as nat
        nat.check(net._lib.ga3c_net_allreduce_grads(net._h), "allreduce")
        assert np.array_equal(net.get_arena(3), before)
        # the train path with a communicator attached: the exchange is OVERLAPPED with the backward pass (dense1/w and
        # the head gradients are all-reduced on a second stream while the conv gradients are still being computed); with
        # one rank the sum is the identity, so the step must equal the step of a net without a communicator bit for bit
        plain = Network("gpu:0", "dp_plain", 6, (84, 84, 4), max_batch=16, predict_lanes=1)
        try:
            for n in (net, plain):
                n.set_arena(0, _flat(o.init_params(6), 6))
                n.set_arena(1, np.ones(n.param_count, np.float32))
                n.learning_rate, n.beta = 3e-4, 0.01
                for _ in range(3):
                    n.train(x, y, a)
            assert np.array_equal(net.get_arena(0), plain.get_arena(0))
            assert np.array_equal(net.get_arena(1), plain.get_arena(1))
            assert np.array_equal(net.get_arena(3), plain.get_arena(3))
        finally:
            plain.close()
    finally:
        net.close()


def test_rccl_blocking_exchange_equals_overlapped(monkeypatch):
    """GA3C_COMM_OVERLAP=0 keeps the round-1 form (one all-reduce of the whole arena behind the backward pass)."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    _, x, a, y = _batch(12, 6, 31)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GA3C_COMM_OVERLAP", flag)
        net = Network("gpu:0", "dp%s" % flag, 6, (84, 84, 4), max_batch=16, predict_lanes=1)
        try:
            net.comm_init(Network.make_comm_id(), 0, 1)
            net.learning_rate, net.beta = 3e-4, 0.01
            for _ in range(2):
                net.train(x, y, a)
            outs.append(net.get_arena(0))
        finally:
            net.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("cfg", [dict(USE_LOG_SOFTMAX=True), dict(MIN_POLICY=0.01), dict(USE_GRAD_CLIP=True, GRAD_CLIP_NORM=0.002),
                                 dict(RMSPROP_MOMENTUM=0.5), dict(LOG_EPSILON=0.3)])
def test_config_branches_match_oracle(cfg):
    """Config.USE_LOG_SOFTMAX (NetworkVP_discrate.py:64-71), MIN_POLICY (:73-74), USE_GRAD_CLIP with
    tf.clip_by_average_norm (:120-123), RMSProp momentum (:101-105) and the LOG_EPSILON gate of tf.maximum."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    from NetworkVP import Network
    saved = {k: getattr(Config, k) for k in cfg}
    for k, v in cfg.items():
        setattr(Config, k, v)
    net = None
    try:
        net = Network("gpu:0", "branch", 6, (84, 84, 4), max_batch=32, predict_lanes=1)
        _, x, a, y = _batch(17, 6, 55)
        params = _oracle_params(net)
        kw = dict(log_eps=Config.LOG_EPSILON, min_policy=Config.MIN_POLICY, use_log_softmax=Config.USE_LOG_SOFTMAX)
        net.learning_rate, net.beta = 1e-3, 0.02
        p, v = net.predict_p_and_v(x)
        ref = o.forward(params, x.astype(np.float64), Config.MIN_POLICY, Config.USE_LOG_SOFTMAX)
        assert np.max(np.abs(p - ref["p"])) < TOL and np.max(np.abs(v - ref["v"])) < TOL
        ms = {k: np.ones_like(t) for k, t in params.items()}
        mom = {k: np.zeros_like(t) for k, t in params.items()}
        for _ in range(2):
            net.train(x, y, a)
            _, g = o.loss_and_grads(params, x.astype(np.float64), y, a.astype(np.float64), 0.02, **kw)
            if Config.USE_GRAD_CLIP:
                g = {k: (o.clip_by_average_norm(np.asarray(g[k]), Config.GRAD_CLIP_NORM) if k in o.PARAM_ORDER else g[k])
                     for k in g}
            o.rmsprop_update(params, ms, g, 1e-3, decay=Config.RMSPROP_DECAY, eps=Config.RMSPROP_EPSILON,
                             momentum=Config.RMSPROP_MOMENTUM, mom=mom)
        got, want = net.get_arena(0), _flat(params, 6)
        assert np.max(np.abs(got - want)) < 2e-5, np.max(np.abs(got - want))
        if Config.USE_GRAD_CLIP:     # the clip must actually bite for this test to mean anything: at least one tensor scaled
            _, raw = o.loss_and_grads(params, x.astype(np.float64), y, a.astype(np.float64), 0.02, **kw)
            scales = {k: Config.GRAD_CLIP_NORM / max(np.sqrt(np.sum(np.asarray(raw[k]) ** 2)) / np.asarray(raw[k]).size,
                                                     Config.GRAD_CLIP_NORM) for k in o.PARAM_ORDER}
            assert min(scales.values()) < 0.9, scales
            assert max(scales.values()) == 1.0, scales     # ... and at least one left alone: both sides of the max()
        if Config.RMSPROP_MOMENTUM:
            assert np.max(np.abs(net.get_arena(2) - _flat(mom, 6))) < 2e-5
    finally:
        if net is not None:
            net.close()
        for k, v in saved.items():
            setattr(Config, k, v)


def test_large_batch_and_u8_train_path(nets):
    net = nets(6, 600)
    xk, x, a, y = _batch(513, 6, 91)
    net.set_arena(0, _flat(o.init_params(6), 6))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    p, v, _ = net.predict_p_v_logits(x)
    ref = o.forward(_oracle_params(net), x[:64].astype(np.float64))
    assert np.max(np.abs(p[:64] - ref["p"])) < TOL and np.max(np.abs(v[:64] - ref["v"])) < TOL
    # forward is row-independent; only the split-K slice count of dense1 depends on the batch size (f32 rounding)
    p64, v64, _ = net.predict_p_v_logits(x[:64])
    assert np.max(np.abs(p[:64] - p64)) < 1e-6 and np.max(np.abs(v[:64] - v64)) < 1e-6
    net.learning_rate, net.beta = 3e-4, 0.01
    net.train(xk, y, a)                      # uint8 frames straight into train
    after_u8 = net.get_arena(0)
    net.set_arena(0, _flat(o.init_params(6), 6))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    net.train(x, y, a)
    assert np.array_equal(after_u8, net.get_arena(0))
    # linearity of the sum-reduced gradient: g(rows 0..512) = g(rows 0..255) + g(rows 256..512)
    net.compute_grads(x, y, a)
    g_all = net.get_arena(3)
    net.compute_grads(x[:256], y[:256], a[:256])
    g_lo = net.get_arena(3)
    net.compute_grads(x[256:], y[256:], a[256:])
    g_hi = net.get_arena(3)
    assert np.max(np.abs(g_all - (g_lo + g_hi))) < 1e-4 * max(1.0, np.max(np.abs(g_all)))


def test_dense1_fragment_and_tile_kernels_give_the_same_bits(nets, monkeypatch):
    """The engine picks dense1's register-fragment kernel (no LDS) for prediction steps while two or more lanes are at work,
    the LDS-tiled one otherwise: the choice depends on timing, so the two must agree bit for bit (same split-K slices, same
    two-accumulator summation order)."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    ref = nets(6)
    monkeypatch.setenv("GA3C_D1F_TILE", "0")
    net = Network("gpu:0", "test_d1frag", 6, (84, 84, 4), max_batch=160, predict_lanes=2)
    monkeypatch.delenv("GA3C_D1F_TILE")
    try:
        for n in (net, ref):
            n.set_arena(0, _flat(o.init_params(6), 6))
        for B in (1, 17, 64, 100, 128, 131, 160):
            _, x, a, y = _batch(B, 6, 900 + B)
            got, want = net.predict_p_v_logits(x), ref.predict_p_v_logits(x)
            assert all(np.array_equal(g, w) for g, w in zip(got, want)), B
    finally:
        net.close()


def test_batch_row_permutation_properties(nets):
    """Full predictor batch (128, BASELINE configs[1]): rows are independent, so permuting the batch permutes the
    outputs bit for bit; the sum-reduced gradient is permutation invariant up to f32 summation order."""
    net = nets(6)
    _, x, a, y = _batch(128, 6, 404)
    perm = np.random.default_rng(1).permutation(128)
    p, v, z = net.predict_p_v_logits(x)
    pp, vp, zp = net.predict_p_v_logits(x[perm])
    assert np.array_equal(p[perm], pp) and np.array_equal(v[perm], vp) and np.array_equal(z[perm], zp)
    net.beta = 0.01
    l1 = net.compute_grads(x, y, a)
    g1 = net.get_arena(3)
    l2 = net.compute_grads(x[perm], y[perm], a[perm])
    g2 = net.get_arena(3)
    assert np.allclose(l1, l2, rtol=1e-5, atol=1e-5)
    assert np.max(np.abs(g1 - g2)) < 1e-4 * max(1.0, np.max(np.abs(g1)))


def test_network_helper_surface_matches_reference(nets, tmp_path, monkeypatch):
    """The rest of the reference's Network surface (NetworkVP.py:233-246,259-288): predict_single / predict_p /
    predict_v, get_global_step, get_variables_names / get_variable_value with the TF variable names, log, and the
    checkpoint naming + episode-from-filename rule."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    net = nets(6)
    _, x, a, y = _batch(3, 6, 8)
    p, v = net.predict_p_and_v(x)
    assert np.array_equal(net.predict_p(x), p) and np.array_equal(net.predict_v(x), v)
    assert np.array_equal(net.predict_single(x[1]), p[1])
    names = net.get_variables_names()
    assert names == [n + ":0" for n in o.PARAM_ORDER]
    shapes = o.param_shapes(6)
    for n in names:
        assert net.get_variable_value(n).shape == shapes[n[:-2]]
    w = net.get_variable_value("logits_v/b:0")
    net.set_variable_value("logits_v/b:0", w + 1.0)
    assert np.allclose(net.predict_v(x), v + 1.0, atol=1e-6)        # the value head's bias moved by exactly 1
    net.set_variable_value("logits_v/b:0", w)
    step = net.get_global_step()
    net.learning_rate, net.beta = 3e-4, 0.01
    net.train(x, y, a, x, np.zeros(3, bool), 0)
    assert net.get_global_step() == step + 1
    net.log(x, y, a, 17)
    row = open("logs/test/scalars.csv").read().strip().split(",")
    assert int(row[0]) == 17 and len(row) == 7
    net.save(123)
    assert sorted(f for f in __import__("os").listdir("checkpoints")) == ["test_00000123.npz"]
    with np.load("checkpoints/test_00000123.npz") as z:
        assert "dense1/w:0" in z.files and "dense1/w/RMSProp:0" in z.files and "step" in z.files
        assert z["conv11/w:0"].shape == (8, 8, 4, 16) and int(z["step"]) == step + 1
    theta = net.get_arena(0)
    net.set_arena(0, np.zeros_like(theta))
    monkeypatch.setattr(Config, "LOAD_EPISODE", 0)
    assert net.load() == 123 and np.array_equal(net.get_arena(0), theta)


def test_graph_replayed_prediction_steps_are_bit_identical(nets, monkeypatch):
    """GA3C_GRAPHS=1: gather + forward captured once per (batch, weight buffer, intake) and replayed with one launch
    (BASELINE configs[4] names a hipGraph-captured predictor step).  Same kernels, same arguments -> same bits, also
    after the optimizer flipped the weight buffer and after the transport was re-registered."""
    import Transport as tp
    from NetworkVP import Network
    plain = nets(6)
    monkeypatch.setenv("GA3C_GRAPHS", "1")
    g = Network("gpu:0", "test_graphs", 6, (84, 84, 4), max_batch=160, predict_lanes=2)
    monkeypatch.delenv("GA3C_GRAPHS")
    try:
        for net in (plain, g):
            net.set_arena(0, _flat(o.init_params(6), 6))
            net.set_arena(1, np.ones(net.param_count, np.float32))
        for seed, bsz in ((1, 1), (2, 7), (3, 64), (4, 7), (5, 64)):            # repeats replay the cached graphs
            xk, x, a, y = _batch(bsz, 6, seed)
            for inp in (xk, x):
                p0, v0 = plain.predict_p_and_v(inp)
                p1, v1 = g.predict_p_and_v(inp)
                assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
        xk, x, a, y = _batch(16, 6, 9)
        for net in (plain, g):
            net.learning_rate, net.beta = 3e-4, 0.01
            net.train(x, y, a, None, None, 0)                                   # flips theta[cur]
        assert np.array_equal(plain.get_arena(0), g.get_arena(0))
        p0, v0 = plain.predict_p_and_v(xk)
        p1, v1 = g.predict_p_and_v(xk)
        assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
        for round_ in range(2):                                                 # second round: a NEW segment address
            t = tp.Transport.create(tp.unique_name("t_graph"), 8, 6, 84 * 84 * 4, 2, 6)
            try:
                t.agent_states[:] = np.random.default_rng(round_).integers(0, 256, size=t.agent_states.shape, dtype=np.uint8)
                ids = np.array([5, 1, 6], np.uint32)
                want = plain.predict_p_and_v(np.ascontiguousarray(t.agent_states[ids]).reshape(3, 84, 84, 4))
                g.register_transport(t)
                for _ in range(2):
                    got = g.predict_offsets(t.state_offsets(ids))
                    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
                g.unregister_transport()
            finally:
                t.shutdown()
                t.close()
    finally:
        g.close()
        plain.set_arena(0, _flat(o.init_params(6), 6))
        plain.set_arena(1, np.ones(plain.param_count, np.float32))


def test_log_evaluates_the_batch_it_is_given_and_writes_histograms(nets, tmp_path, monkeypatch):
    """Network.log (NetworkVP.py:259-265): forward + loss on ITS arguments with the current weights -- not the last train
    step's numbers -- plus the histograms of NetworkVP_discrate.py:140-146, held to the oracle."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import histogram_proto
    monkeypatch.chdir(tmp_path)
    net = nets(6)
    net.set_arena(0, _flat(o.init_params(6), 6))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    xk, x, a, y = _batch(21, 6, 808)
    xk2, x2, a2, y2 = _batch(33, 6, 809)
    net.learning_rate, net.beta = 3e-4, 0.01
    net.train(x, y, a)                                   # some OTHER batch was trained last
    step = net.get_global_step()
    theta = net.get_arena(0)
    losses = net.log(xk2, y2, a2, 41)                    # uint8 input
    losses_f32 = net.log(x2, y2, a2, 42)                 # the same rows as f32
    assert np.array_equal(losses, losses_f32)
    assert net.get_global_step() == step and np.array_equal(net.get_arena(0), theta)     # log trains nothing
    params = _oracle_params(net)
    ref_l, _ = o.loss_and_grads(params, x2.astype(np.float64), y2, a2.astype(np.float64), 0.01)
    want = np.array([ref_l["cost_p_1_agg"], ref_l["cost_p_2_agg"], ref_l["cost_v"]])
    assert np.allclose(losses, want, rtol=1e-4, atol=1e-4)
    rows = [r.split(",") for r in open("logs/test/scalars.csv").read().strip().splitlines()]
    assert [int(r[0]) for r in rows] == [41, 42] and all(len(r) == 7 for r in rows)
    got = np.array([float(t) for t in rows[0][1:]])
    assert np.allclose(got, [want[0], want[1], -(want[0] + want[1]), want[2], 3e-4, 0.01], rtol=1e-4, atol=1e-4)
    ref = o.forward(params, x2.astype(np.float64), keep=True)
    with np.load("logs/test/histograms_00000041.npz") as z:
        tags = {k.rsplit("/", 1)[0] for k in z.files}
        assert tags == {"weights_%s:0" % n for n in o.PARAM_ORDER} | {"activation_lastdense", "activation_v", "activation_p"}
        for tag, values in (("activation_lastdense", ref["d1"]), ("activation_v", ref["v"]), ("activation_p", ref["p"]),
                            ("weights_dense1/w:0", params["dense1/w"]), ("weights_logits_p/b:0", params["logits_p/b"])):
            h = histogram_proto(values)
            assert z[tag + "/num"] == h["num"] == np.asarray(values).size
            assert abs(z[tag + "/min"] - h["min"]) < TOL and abs(z[tag + "/max"] - h["max"]) < TOL
            assert abs(z[tag + "/sum"] - h["sum"]) < TOL * max(1.0, abs(h["sum"])) * 10
            assert abs(z[tag + "/sum_squares"] - h["sum_squares"]) < TOL * max(1.0, h["sum_squares"]) * 10
            assert z[tag + "/bucket"].sum() == h["num"] and np.all(np.diff(z[tag + "/bucket_limit"]) > 0)
            # same buckets up to values that sit within f32 rounding of a bucket limit
            lim = np.union1d(z[tag + "/bucket_limit"], h["bucket_limit"])
            cg = np.zeros(lim.size)
            cw = np.zeros(lim.size)
            cg[np.searchsorted(lim, z[tag + "/bucket_limit"])] = z[tag + "/bucket"]
            cw[np.searchsorted(lim, h["bucket_limit"])] = h["bucket"]
            assert np.abs(cg - cw).sum() <= max(4, 0.002 * h["num"]), tag
