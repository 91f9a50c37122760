"""The f32 C port (bench.py's cpu_baseline) against the numpy f64 oracle."""
import numpy as np

import ga3c_oracle as o
import ga3c_oracle_cport as oc


def _flat(d):
    return np.concatenate([np.asarray(d[k]).reshape(-1) for k in o.PARAM_ORDER])


def _case(bsz, num_actions, seed):
    p = o.init_params(num_actions)
    x = o.synthetic_states(bsz, seed=seed)
    rng = np.random.default_rng(seed)
    return p, x, rng.uniform(-1, 1, bsz), np.eye(num_actions, dtype=np.float32)[rng.integers(0, num_actions, bsz)]


def test_cport_forward_matches_oracle():
    for num_actions, bsz in ((6, 11), (18, 3)):
        p, x, _, _ = _case(bsz, num_actions, 3)
        theta = _flat(p).astype(np.float32)
        assert theta.size == oc.lib().ga3c_oc_param_count(num_actions)
        got_p, got_v = oc.predict(theta, num_actions, x)
        ref = o.forward(p, x.astype(np.float64))
        assert np.max(np.abs(got_p - ref["p"])) < 1e-5
        assert np.max(np.abs(got_v - ref["v"])) < 1e-5


def test_cport_train_step_matches_oracle():
    p, x, y, a = _case(13, 6, 4)
    theta = _flat(p).astype(np.float32)
    ms = np.ones_like(theta)
    losses, grad = oc.train(theta, ms, 6, x, y, a, lr=3e-4, beta=0.01)
    ref_ms = {k: np.ones_like(v) for k, v in p.items()}
    ref_l, ref_g = o.train_step(p, ref_ms, x.astype(np.float64), y, a.astype(np.float64), 3e-4, 0.01)
    assert np.allclose(losses, [ref_l["cost_p_1_agg"], ref_l["cost_p_2_agg"], ref_l["cost_v"]], rtol=1e-4, atol=1e-5)
    off = 0
    for k in o.PARAM_ORDER:
        want = np.asarray(ref_g[k]).reshape(-1)
        got = grad[off:off + want.size]
        off += want.size
        assert np.max(np.abs(got - want)) < 1e-4 * max(1.0, np.max(np.abs(want))), k
    assert np.max(np.abs(theta - _flat(p))) < 1e-6
    assert np.max(np.abs(ms - _flat(ref_ms))) < 1e-5
