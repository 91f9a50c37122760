"""-m gpu: (1) the variables by NAME and save / load through the C ABI (ga3c_net_param_name / _param_info / _get_param /
_set_param / _save / _load: NetworkVP.py:267-288 of the reference); (2) a SATURATED policy -- logits_p/w scaled so that at the
default LOG_EPSILON min(p) < 1e-6 < max(p) ~ 1: the clamp log(max(p, eps)), TensorFlow's `maximum` gradient mask and a
max-subtracted softmax with a wide spread, which init-scale weights never reach (NetworkVP_discrate.py:73-85) -- forward,
gradients and two production train steps against the oracle."""
import os

import numpy as np
import pytest

import ga3c_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets():
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    made = {}

    def get(num_actions, tag="p"):
        key = (num_actions, tag)
        if key not in made:
            made[key] = Network("gpu:0", "params_%s" % tag, num_actions, (84, 84, 4), max_batch=40, predict_lanes=1)
        return made[key]
    yield get
    for n in made.values():
        n.close()


def _flat(d):
    return np.concatenate([np.asarray(d[k]).reshape(-1) for k in o.PARAM_ORDER])


@pytest.mark.parametrize("num_actions", [6, 18])
def test_names_table_and_one_variable_at_a_time(nets, num_actions):
    net = nets(num_actions)
    assert net.get_variables_names() == [n + ":0" for n in o.PARAM_ORDER]          # arena order = TensorFlow variable order
    off = 0
    for name in o.PARAM_ORDER:
        shape = tuple(o.param_shapes(num_actions)[name])
        got_off, count, got_shape = net._param_info(name)
        assert (got_off, count, got_shape) == (off, int(np.prod(shape)), shape), name
        assert net._param_info(name + ":0") == (got_off, count, got_shape)          # TensorFlow's suffix is accepted
        off += count
    assert off == net.param_count
    net.set_arena(0, _flat(o.init_params(num_actions)).astype(np.float32))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    rng = np.random.default_rng(5)
    new = rng.normal(size=(256, num_actions)).astype(np.float32)
    before = net.get_arena(0)
    net.set_variable_value("logits_p/w:0", new)
    after = net.get_arena(0)
    o_, c_, _ = net._param_info("logits_p/w")
    assert np.array_equal(after[o_:o_ + c_], new.ravel())
    assert np.array_equal(np.delete(after, np.s_[o_:o_ + c_]), np.delete(before, np.s_[o_:o_ + c_]))   # nothing else moved
    assert np.array_equal(net.get_variable_value("logits_p/w"), new)
    assert np.array_equal(net.get_variable_value("conv12/b", which=1), np.ones(32, np.float32))         # its `ms` slot
    # dense1/w by name reaches the fragment-ordered copy the forward pass reads: predictions follow the new weights
    wd = (net.get_variable_value("dense1/w") * np.float32(0.5)).astype(np.float32)
    net.set_variable_value("dense1/w", wd)
    xk = rng.integers(0, 256, size=(4, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128) - np.float32(1)
    params, k = {}, 0
    arena = net.get_arena(0).astype(np.float64)
    for name in o.PARAM_ORDER:
        shape = o.param_shapes(num_actions)[name]
        params[name] = arena[k:k + int(np.prod(shape))].reshape(shape)
        k += int(np.prod(shape))
    p, v = net.predict_p_and_v(x)
    ref = o.forward(params, x.astype(np.float64))
    assert np.max(np.abs(p - ref["p"])) < 1e-4 and np.max(np.abs(v - ref["v"])) < 1e-4
    with pytest.raises(RuntimeError):
        net.get_variable_value("dense2/w")
    with pytest.raises(RuntimeError):
        net.set_variable_value("conv11/b", np.zeros(15, np.float32))                # wrong element count


def test_save_and_load_through_the_c_abi(nets, tmp_path):
    import ga3c_amd  # noqa: F401
    import _native as nat
    net = nets(6)
    rng = np.random.default_rng(11)
    arenas = [rng.normal(size=net.param_count).astype(np.float32), rng.uniform(0.5, 2.0, net.param_count).astype(np.float32),
              rng.normal(size=net.param_count).astype(np.float32) * np.float32(1e-3)]
    for w, a in enumerate(arenas):
        net.set_arena(w, a)
    nat.check(net._lib.ga3c_net_set_step(net._h, 4321))
    path = str(tmp_path / "ckpt_00000007.npz")
    nat.check(net._lib.ga3c_net_save(net._h, path.encode()), "save")
    with np.load(path, allow_pickle=False) as z:                                    # numpy reads it, keyed by the TF names
        assert int(z["step"]) == 4321 and len(z.files) == 31
        off = 0
        for name in o.PARAM_ORDER:
            shape = tuple(o.param_shapes(6)[name])
            size = int(np.prod(shape))
            for w, suffix in enumerate((":0", "/RMSProp:0", "/RMSProp_1:0")):
                assert z[name + suffix].shape == shape and z[name + suffix].dtype == np.float32
                assert np.array_equal(z[name + suffix].ravel(), arenas[w][off:off + size])
            off += size
    other = nets(6, "fresh")
    nat.check(other._lib.ga3c_net_load(other._h, path.encode()), "load")
    for w in range(3):
        assert np.array_equal(other.get_arena(w), arenas[w])
    assert other.get_global_step() == 4321
    # a checkpoint numpy wrote (savez: zip64 member headers) loads too
    np_path = str(tmp_path / "from_numpy.npz")
    with np.load(path, allow_pickle=False) as z:
        np.savez(np_path, **{k: z[k] for k in z.files})
    nat.check(other._lib.ga3c_net_set_step(other._h, 0))
    nat.check(other._lib.ga3c_net_load(other._h, np_path.encode()), "load of numpy's file")
    assert other.get_global_step() == 4321 and np.array_equal(other.get_arena(0), arenas[0])
    # another action count: refused, and the network is left as it was
    wrong = nets(18)
    keep = wrong.get_arena(0)
    assert wrong._lib.ga3c_net_load(wrong._h, path.encode()) != 0
    assert b"logits_p" in nat.hip_lib().ga3c_last_error()
    assert np.array_equal(wrong.get_arena(0), keep)
    assert wrong._lib.ga3c_net_load(wrong._h, str(tmp_path / "missing.npz").encode()) != 0


SCALE = 200.0


def _saturated(num_actions, seed):
    params = o.init_params(num_actions)
    params["logits_p/w"] = params["logits_p/w"] * SCALE
    params["dense1/w"] = params["dense1/w"] * 2.0
    rng = np.random.Generator(np.random.PCG64(seed))
    xk = rng.integers(0, 256, size=(32, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128) - np.float32(1)
    act = rng.integers(0, num_actions, size=32)
    y = rng.uniform(-1, 1, size=32)
    return params, xk, x, np.eye(num_actions, dtype=np.float32)[act], y


@pytest.mark.parametrize("num_actions,seed", [(6, 9206), (18, 9118)])
def test_saturated_policy_forward_gradients_and_train_steps(nets, num_actions, seed):
    net = nets(num_actions)
    params, xk, x, a, y = _saturated(num_actions, seed)
    eps = 1e-6
    ref = o.forward(params, x.astype(np.float64))
    p64 = ref["p"]
    sel = (p64 * a).sum(axis=1)
    # the case is what it claims to be: both sides of the clamp are populated, rows with a clamped SELECTED action exist, and
    # no probability sits so close to eps that f32 and f64 could disagree about the side
    assert p64.min() < 1e-9 and p64.max() > 0.9999 and 0.1 < (p64 < eps).mean() < 0.7
    assert (sel < eps).sum() >= 4 and (sel >= eps).sum() >= 4
    assert np.min(np.abs(p64 / eps - 1.0)) > 0.01
    net.set_arena(0, _flat(params).astype(np.float32))
    net.set_arena(1, np.ones(net.param_count, np.float32))
    net.set_arena(2, np.zeros(net.param_count, np.float32))
    net.learning_rate, net.beta = 3e-4, 0.01
    p, v, z = net.predict_p_v_logits(x)
    assert np.max(np.abs(p - p64)) < 1e-4 and np.max(np.abs(v - ref["v"])) < 1e-4
    assert np.max(np.abs(z - ref["z"])) < 1e-4 * max(1.0, np.max(np.abs(ref["z"])))      # logits of +-50: 1e-4 is relative here
    # small probabilities to RELATIVE precision (they decide the clamp and carry the entropy term)
    mask = p64 > 1e-12
    assert np.max(np.abs(p[mask] / p64[mask] - 1.0)) < 2e-3
    losses = net.compute_grads(x, y, a)
    ref_l, ref_g = o.loss_and_grads(params, x.astype(np.float64), y, a.astype(np.float64), 0.01)
    want_l = np.array([ref_l["cost_p_1_agg"], ref_l["cost_p_2_agg"], ref_l["cost_v"]])
    assert np.allclose(losses, want_l, rtol=1e-4, atol=1e-4), (losses, want_l)
    dz = net.fetch("dz", ref_g["dz"].size).reshape(ref_g["dz"].shape)
    assert np.max(np.abs(dz - ref_g["dz"])) < 1e-4 * max(1.0, np.max(np.abs(ref_g["dz"])))
    got, off = net.get_arena(3), 0
    for name in o.PARAM_ORDER:
        want = np.asarray(ref_g[name]).reshape(-1)
        g = got[off:off + want.size]
        off += want.size
        assert np.max(np.abs(g - want)) < 1e-4 * max(np.max(np.abs(want)), 1.0), name
    # two production steps (uint8 states, fused update) from the saturated weights
    ms = {k: np.ones_like(t) for k, t in params.items()}
    for _ in range(2):
        o.train_step(params, ms, x.astype(np.float64), y, a.astype(np.float64), 3e-4, 0.01)
        net.train(xk, y, a)
    assert np.max(np.abs(net.get_arena(0) - _flat(params))) < 2e-5
    want_ms = _flat(ms)
    assert np.max(np.abs(net.get_arena(1) - want_ms) / np.maximum(1.0, np.abs(want_ms))) < 1e-4
