"""The numpy oracle's forward and its hand-derived backward against an independent implementation: the same graph written
with torch (CPU, float64: F.conv2d on explicitly padded input, matmul, softmax) and differentiated by autograd with the
advantage detached as tf.stop_gradient does (NetworkVP_discrate.py:78).  Torch is not the reference -- TensorFlow is absent
here, the NN numerics stay 'parity unpinned' -- but it is a second, machine-differentiated statement of the cited lines
(NetworkVP.py:212-228, NetworkDNav.py:80-90, NetworkVP_discrate.py:58-85,100), so an algebra slip in the oracle's gradient
formulas or its im2col cannot hide behind its own finite-difference self-check."""
import numpy as np
import pytest

import ga3c_oracle as o

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _same_pad(n, k, s):
    """TF 'SAME': total = max((ceil(n/s) - 1) s + k - n, 0), the extra cell on the bottom / right."""
    total = max((-(-n // s) - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _torch_graph(params, x, y_r, a, beta, log_eps, min_policy, use_log_softmax):
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)                  # NHWC -> NCHW

    def conv(inp, w, b, k, s):
        lo, hi = _same_pad(inp.shape[2], k, s)
        inp = F.pad(inp, (lo, hi, lo, hi))
        return F.conv2d(inp, w.permute(3, 2, 0, 1), b, stride=s)                   # HWIO -> OIHW, cross-correlation

    n1 = torch.relu(conv(xt, t["conv11/w"], t["conv11/b"], 8, 4))
    n2 = torch.relu(conv(n1, t["conv12/w"], t["conv12/b"], 4, 2))
    flat = n2.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                          # index (h*11+w)*32+c
    d1 = torch.relu(flat @ t["dense1/w"] + t["dense1/b"])
    v = (d1 @ t["logits_v/w"] + t["logits_v/b"])[:, 0]
    z = d1 @ t["logits_p/w"] + t["logits_p/b"]
    yt, at = torch.tensor(y_r, dtype=torch.float64), torch.tensor(a, dtype=torch.float64)
    adv = yt - v.detach()
    if use_log_softmax:
        ls = F.log_softmax(z, dim=1)
        s = F.softmax(z, dim=1)
        p = s
        c1 = ((ls * at).sum(1) * adv).sum()
        c2 = (-beta * (ls * s).sum(1)).sum()
    else:
        p = (F.softmax(z, dim=1) + min_policy) / (1.0 + min_policy * z.shape[1])
        eps = torch.tensor(log_eps, dtype=torch.float64)
        c1 = (torch.log(torch.maximum((p * at).sum(1), eps)) * adv).sum()
        c2 = (-beta * (torch.log(torch.maximum(p, eps)) * p).sum(1)).sum()
    cost_v = 0.5 * ((yt - v) ** 2).sum()
    cost_all = -(c1 + c2) + cost_v
    cost_all.backward()
    return dict(p=p, v=v, z=z, n1=n1.permute(0, 2, 3, 1), n2=n2.permute(0, 2, 3, 1), d1=d1), \
        dict(cost_p_1_agg=c1, cost_p_2_agg=c2, cost_v=cost_v, cost_all=cost_all), {k: t[k].grad for k in t}


@pytest.mark.parametrize("num_actions,bsz,use_log_softmax,min_policy", [(6, 3, False, 0.0), (4, 2, False, 0.01), (18, 2, True, 0.0),
                                                                        (1, 2, False, 0.0)])
def test_forward_loss_and_gradients_match_autograd(num_actions, bsz, use_log_softmax, min_policy):
    params = o.init_params(num_actions)
    x = o.synthetic_states(bsz, seed=90 + num_actions).astype(np.float64).reshape(bsz, 84, 84, 4)
    rng = np.random.default_rng(num_actions)
    y = rng.normal(size=bsz)
    a = np.eye(num_actions)[rng.integers(0, num_actions, bsz)]
    beta = 0.01
    f = o.forward(params, x, min_policy, use_log_softmax, keep=True)
    losses, g = o.loss_and_grads(params, x, y, a, beta, min_policy=min_policy, use_log_softmax=use_log_softmax)
    tf_, tl, tg = _torch_graph(params, x, y, a, beta, 1e-6, min_policy, use_log_softmax)
    for name in ("p", "v", "z", "n1", "n2", "d1"):
        assert np.max(np.abs(f[name] - tf_[name].detach().numpy())) < 1e-12, name
    for name in ("cost_p_1_agg", "cost_p_2_agg", "cost_v", "cost_all"):
        assert abs(losses[name] - float(tl[name].detach())) < 1e-10 * max(1.0, abs(losses[name])), name
    for name in o.PARAM_ORDER:
        want = tg[name].numpy().reshape(-1)
        got = np.asarray(g[name]).reshape(-1)
        assert np.max(np.abs(got - want)) < 1e-10 * max(1.0, np.max(np.abs(want))), name


def test_same_padding_rule_gives_the_surveyed_splits():
    assert _same_pad(84, 8, 4) == (2, 2) and _same_pad(21, 4, 2) == (1, 2)
