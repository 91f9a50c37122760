"""Self-checks of the numpy oracle (NN numerics are 'parity unpinned': TensorFlow is absent,
so the restatement is validated by float64 finite differences and hand-computed cases)."""
import numpy as np
import pytest

import ga3c_oracle as o


def _case(bsz, num_actions, seed):
    p = o.init_params(num_actions)
    x = o.synthetic_states(bsz, seed=seed).astype(np.float64)
    rng = np.random.default_rng(seed)
    y = rng.normal(size=bsz)
    a = np.eye(num_actions)[rng.integers(0, num_actions, bsz)]
    return p, x, y, a


def test_param_count_matches_survey():
    assert o.param_count(6) == 1_005_623
    assert o.param_count(4) == 1_005_623 - 2 * 257
    assert o.param_count(18) == 1_005_623 + 12 * 257


def test_same_padding_geometry():
    # TF SAME: out=ceil(in/s); pad=max((out-1)s+k-in,0); before=pad//2
    for cfg, n_in in ((o.CONV1, 84), (o.CONV2, 21)):
        out = -(-n_in // cfg["s"])
        total = max((out - 1) * cfg["s"] + cfg["k"] - n_in, 0)
        assert out == cfg["out"] and total // 2 == cfg["pad"]


def test_conv_matches_direct_loops():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 21, 21, 16))
    w = rng.normal(size=(4, 4, 16, 32))
    b = rng.normal(size=32)
    y, _ = o._conv_fwd(x, w, b, o.CONV2)
    for (i, j, oc) in ((0, 0, 0), (10, 10, 31), (5, 0, 7), (0, 9, 3), (10, 3, 12)):
        acc = b[oc]
        for u in range(4):
            for v in range(4):
                yy, xx = 2 * i + u - 1, 2 * j + v - 1
                if 0 <= yy < 21 and 0 <= xx < 21:
                    acc += x[0, yy, xx, :] @ w[u, v, :, oc]
        assert abs(acc - y[0, i, j, oc]) < 1e-10


@pytest.mark.parametrize("use_log_softmax,min_policy", [(False, 0.0), (False, 0.01), (True, 0.0)])
def test_gradients_finite_difference(use_log_softmax, min_policy):
    p, x, y, a = _case(2, 6, 7)
    kw = dict(use_log_softmax=use_log_softmax, min_policy=min_policy)
    adv0 = y - o.forward(p, x, min_policy, use_log_softmax)["v"]
    _, g = o.loss_and_grads(p, x, y, a, 0.01, **kw)
    rng = np.random.default_rng(11)
    h = 1e-6
    for name in o.PARAM_ORDER:
        gn = g[name].reshape(p[name].shape)
        # probe entries with non-trivial gradient so the relative check means something
        flat_idx = np.argsort(-np.abs(gn).ravel())[:64]
        for fi in rng.choice(flat_idx, size=min(2, flat_idx.size), replace=False):
            idx = np.unravel_index(fi, gn.shape)
            old = p[name][idx]
            p[name][idx] = old + h
            lp, _ = o.loss_and_grads(p, x, y, a, 0.01, adv_const=adv0, **kw)
            p[name][idx] = old - h
            lm, _ = o.loss_and_grads(p, x, y, a, 0.01, adv_const=adv0, **kw)
            p[name][idx] = old
            num = (lp["cost_all"] - lm["cost_all"]) / (2 * h)
            assert abs(num - gn[idx]) <= 1e-5 * max(1.0, abs(gn[idx])), (name, idx, num, gn[idx])


def test_loss_hand_case():
    # one sample, uniform policy: p=1/A, entropy term = -beta*A*(1/A)*log(1/A) = beta*log(A)
    num_actions = 4
    p = {k: np.zeros(s) for k, s in o.param_shapes(num_actions).items()}
    x = np.zeros((1, 84, 84, 4))
    a = np.eye(num_actions)[[2]]
    losses, g = o.loss_and_grads(p, x, np.array([1.5]), a, beta=0.01)
    assert np.isclose(losses["cost_p_1_agg"], np.log(0.25) * 1.5)
    assert np.isclose(losses["cost_p_2_agg"], 0.01 * np.log(4.0))
    assert np.isclose(losses["cost_v"], 0.5 * 1.5 ** 2)
    assert np.isclose(g["logits_v/b"][0], -1.5)


def test_log_epsilon_gate():
    # tf.maximum passes no gradient to x when x < eps: drive one probability below eps
    num_actions = 3
    p = {k: np.zeros(s) for k, s in o.param_shapes(num_actions).items()}
    p["logits_p/b"] = np.array([0.0, 0.0, -40.0])
    x = np.zeros((1, 84, 84, 4))
    a = np.eye(3)[[2]]
    _, g = o.loss_and_grads(p, x, np.array([1.0]), a, beta=0.0)
    assert np.all(g["dz"] == 0.0)   # selected prob < eps and beta = 0 -> no policy gradient


def test_rmsprop_one_step_by_hand():
    params = {k: np.full(s, 0.5) for k, s in o.param_shapes(4).items()}
    ms = {k: np.ones(s) for k, s in o.param_shapes(4).items()}
    grads = {k: np.full(s, 2.0) for k, s in o.param_shapes(4).items()}
    o.rmsprop_update(params, ms, grads, lr=0.1)
    ms_expect = 0.99 * 1.0 + 0.01 * 4.0
    assert np.allclose(ms["dense1/w"], ms_expect)
    assert np.allclose(params["conv11/b"], 0.5 - 0.1 * 2.0 / np.sqrt(ms_expect + 0.1))


def test_clip_by_average_norm():
    g = np.full(10, 1000.0)
    # ||g||/n = 1000*sqrt(10)/10 = 316.2 > 40 -> scale by 40/316.2
    assert np.allclose(o.clip_by_average_norm(g, 40.0), g * 40.0 / (1000.0 * np.sqrt(10) / 10))
    small = np.full(10, 1.0)
    assert np.allclose(o.clip_by_average_norm(small, 40.0), small)


def test_f32_restatement_close_to_f64():
    p64, x, y, a = _case(4, 6, 5)
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    f64 = o.forward(p64, x)
    f32 = o.forward(p32, x.astype(np.float32))
    assert f32["p"].dtype == np.float32
    assert np.max(np.abs(f64["p"] - f32["p"])) < 1e-5
    assert np.max(np.abs(f64["v"] - f32["v"])) < 1e-5
