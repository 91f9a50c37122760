"""CPU oracle for the GA3C actor-learner hot path (TEST INFRASTRUCTURE ONLY).

This module is a numpy restatement of the reference's algorithm for the path
named in BASELINE.json.  It is imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, always as the
checker.  The product (ga3c_amd/) never imports it and has no CPU fallback.

PARITY PINNING
  * Plumbing (returns, rollout arrays, action draws, batching): pinned to the
    reference RUN in the build container (tests/golden/make_golden.py imports the
    reference's own ProcessAgent / ThreadPredictor / ThreadTrainer and records
    what they return): `accumulate_rewards_fork` against 100 recorded cases
    (25 reward vectors x 4 flag settings, SURVEY.md §8-a3's hex vector among
    them), `convert_data` and `select_action_index` against recorded arrays and
    draws, the batchers against recorded traces.
  * NN numerics (forward, loss, gradients, RMSProp): PARITY UNPINNED.  The
    reference computes them inside TensorFlow 1.x (un-vendored, unpinned:
    "TensorFlow 1.0", /root/reference/README.md:8), which is absent from this
    image and ships no golden vectors or tests.  The restatement below follows
    the reference's graph-building call sites and TF-1.x's published op
    semantics; it is self-checked by float64 finite differences
    (tests/test_oracle_numerics.py).

Reference lines followed (paths relative to /root/reference/ga3c):
  conv layer        NetworkVP.py:212-228   (HWIO filter, SAME, +b, ReLU; init 1/sqrt(fan_in))
  topology          NetworkDNav.py:80-90   (conv 8x8x16 s4 -> conv 4x4x32 s2 -> flatten -> dense 256)
  dense layer       NetworkDNav.py:256-269 (ReLU, init 1/sqrt(in_dim)); NetworkVP.py:194-210
  heads             NetworkVP_discrate.py:60,63,73-74
  loss              NetworkVP_discrate.py:61,64-85
  optimizer         NetworkVP_discrate.py:99-105,120-123,130 + Config.py:111-122
  returns           ProcessAgent.py:69-84 (call site :148-149)
  rollout arrays    ProcessAgent.py:86-100
  action draw       ProcessAgent.py:109-115
  batching          ThreadPredictor.py:45-66, ThreadTrainer.py:42-62
  lr/beta schedule  Server.py:168-175
"""
import numpy as np

H = W = 84
C = 4
CONV1 = dict(k=8, s=4, cin=4, cout=16, out=21, pad=2)    # SAME: total pad 4 -> 2 before
CONV2 = dict(k=4, s=2, cin=16, cout=32, out=11, pad=1)   # SAME: total pad 3 -> 1 before, 2 after
FLAT = 11 * 11 * 32
HID = 256

PARAM_ORDER = ("conv11/w", "conv11/b", "conv12/w", "conv12/b", "dense1/w", "dense1/b",
               "logits_v/w", "logits_v/b", "logits_p/w", "logits_p/b")


def param_shapes(num_actions):
    return {
        "conv11/w": (8, 8, 4, 16), "conv11/b": (16,),
        "conv12/w": (4, 4, 16, 32), "conv12/b": (32,),
        "dense1/w": (FLAT, HID), "dense1/b": (HID,),
        "logits_v/w": (HID, 1), "logits_v/b": (1,),
        "logits_p/w": (HID, num_actions), "logits_p/b": (num_actions,),
    }


def param_count(num_actions):
    return sum(int(np.prod(s)) for s in param_shapes(num_actions).values())


def _fan_in(name, shape):
    # NetworkVP.py:214 (conv: k*k*cin), NetworkDNav.py:258 (dense: in_dim);
    # biases use the same bound as their weight.
    return int(np.prod(shape[:-1]))


def init_params(num_actions, seed=12345, dtype=np.float64):
    """U(-d, d), d = 1/sqrt(fan_in); PCG64(seed) drawn in PARAM_ORDER."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = param_shapes(num_actions)
    out = {}
    for base in ("conv11", "conv12", "dense1", "logits_v", "logits_p"):
        wshape = shapes[base + "/w"]
        d = 1.0 / np.sqrt(_fan_in(base, wshape))
        out[base + "/w"] = rng.uniform(-d, d, size=wshape).astype(np.float32).astype(dtype)
        out[base + "/b"] = rng.uniform(-d, d, size=shapes[base + "/b"]).astype(np.float32).astype(dtype)
    return out


def synthetic_states(batch, seed=12345, dtype=np.float32):
    """k/128 - 1, k ~ U{0..255} (value set of Environment.py:56-61)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k = rng.integers(0, 256, size=(batch, H, W, C), dtype=np.uint8)
    return (k.astype(np.float32) / np.float32(128.0) - np.float32(1.0)).astype(dtype)


# ---------------------------------------------------------------- conv helpers
def _im2col(x, k, s, out, pad_before):
    """x [B,Hin,Win,Cin] -> cols [B,out,out,k,k,Cin] with TF SAME zero padding."""
    b, hin, win, cin = x.shape
    total = max((out - 1) * s + k - hin, 0)
    after = total - pad_before
    xp = np.pad(x, ((0, 0), (pad_before, after), (pad_before, after), (0, 0)))
    win_view = np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(1, 2))  # [B,H',W',Cin,k,k]
    win_view = win_view[:, ::s, ::s][:, :out, :out]
    return np.ascontiguousarray(np.transpose(win_view, (0, 1, 2, 4, 5, 3)))


def _conv_fwd(x, w, b, cfg):
    cols = _im2col(x, cfg["k"], cfg["s"], cfg["out"], cfg["pad"])
    bsz = x.shape[0]
    kk = cfg["k"] * cfg["k"] * cfg["cin"]
    y = cols.reshape(bsz * cfg["out"] * cfg["out"], kk) @ w.reshape(kk, cfg["cout"]) + b
    return y.reshape(bsz, cfg["out"], cfg["out"], cfg["cout"]), cols


def _conv_bwd(dy, cols, w, cfg, in_hw, need_dx):
    bsz = dy.shape[0]
    k, s, out, cin, cout = cfg["k"], cfg["s"], cfg["out"], cfg["cin"], cfg["cout"]
    kk = k * k * cin
    dy2 = dy.reshape(bsz * out * out, cout)
    dw = (cols.reshape(bsz * out * out, kk).T @ dy2).reshape(w.shape)
    db = dy2.sum(axis=0)
    dx = None
    if need_dx:
        dcols = (dy2 @ w.reshape(kk, cout).T).reshape(bsz, out, out, k, k, cin)
        total = max((out - 1) * s + k - in_hw, 0)
        pb = cfg["pad"]
        dxp = np.zeros((bsz, in_hw + total, in_hw + total, cin), dtype=dy.dtype)
        for u in range(k):
            for v in range(k):
                dxp[:, u:u + s * out:s, v:v + s * out:s, :] += dcols[:, :, :, u, v, :]
        dx = dxp[:, pb:pb + in_hw, pb:pb + in_hw, :]
    return dw, db, dx


# ---------------------------------------------------------------- forward
def forward(params, x, min_policy=0.0, use_log_softmax=False, keep=False):
    """Returns dict with z (logits), p, v (+ activations when keep=True).

    Dtype follows `x`/`params` (run in float64 for ground truth, float32 for the
    like-for-like restatement).
    """
    x = np.asarray(x)
    bsz = x.shape[0]
    x = x.reshape(bsz, H, W, C)
    n1pre, cols1 = _conv_fwd(x, params["conv11/w"], params["conv11/b"], CONV1)
    n1 = np.maximum(n1pre, 0)
    n2pre, cols2 = _conv_fwd(n1, params["conv12/w"], params["conv12/b"], CONV2)
    n2 = np.maximum(n2pre, 0)
    flat = n2.reshape(bsz, FLAT)                 # index (h*11+w)*32+c  (NetworkDNav.py:86-89)
    d1 = np.maximum(flat @ params["dense1/w"] + params["dense1/b"], 0)
    v = (d1 @ params["logits_v/w"] + params["logits_v/b"])[:, 0]
    z = d1 @ params["logits_p/w"] + params["logits_p/b"]
    zs = z - z.max(axis=1, keepdims=True)
    e = np.exp(zs)
    s = e / e.sum(axis=1, keepdims=True)
    num_actions = z.shape[1]
    if use_log_softmax:
        p = s                                     # NetworkVP_discrate.py:65
    else:
        p = (s + min_policy) / (1.0 + min_policy * num_actions)   # :73-74
    out = dict(z=z, p=p, v=v)
    if keep:
        out.update(x=x, cols1=cols1, n1=n1, cols2=cols2, n2=n2, flat=flat, d1=d1, s=s, zs=zs, e=e)
    return out


# ---------------------------------------------------------------- loss + grads
def loss_and_grads(params, x, y_r, a, beta, log_eps=1e-6, min_policy=0.0, use_log_softmax=False,
                   adv_const=None):
    """Sum-over-batch A3C loss of NetworkVP_discrate.py:61-85,100 and its gradient.

    Returns (losses, grads): losses = dict(cost_p_1_agg, cost_p_2_agg, cost_v, cost_all);
    grads = dict keyed like params, plus 'dz','dv' per-sample head gradients.
    `adv_const` replaces y_r - v by a constant so that a numerical derivative of
    cost_all sees what tf.stop_gradient makes autodiff see.
    """
    f = forward(params, x, min_policy, use_log_softmax, keep=True)
    dt = f["z"].dtype
    y_r = np.asarray(y_r).astype(dt)
    a = np.asarray(a).astype(dt)
    z, p, v, s = f["z"], f["p"], f["v"], f["s"]
    num_actions = z.shape[1]
    adv = y_r - v                                  # stop_gradient(v) (:78)
    if adv_const is not None:                      # finite-difference tests freeze the advantage
        adv = np.asarray(adv_const).astype(dt)
    if use_log_softmax:
        ls = f["zs"] - np.log(f["e"].sum(axis=1, keepdims=True))
        lsel = (ls * a).sum(axis=1)
        cost_p_1 = lsel * adv
        cost_p_2 = -beta * (ls * s).sum(axis=1)
        ent = (s * ls).sum(axis=1, keepdims=True)
        dz = -adv[:, None] * (a - s * a.sum(axis=1, keepdims=True)) + beta * s * (ls - ent)
    else:
        sel = (p * a).sum(axis=1)
        cost_p_1 = np.log(np.maximum(sel, log_eps)) * adv
        logp = np.log(np.maximum(p, log_eps))
        cost_p_2 = -beta * (logp * p).sum(axis=1)
        # tf.maximum routes the gradient to x when x >= eps
        g_sel = np.where(sel >= log_eps, 1.0 / np.maximum(sel, log_eps), 0.0)
        g_p = -(adv * g_sel)[:, None] * a + beta * (logp + np.where(p >= log_eps, 1.0, 0.0))
        g_s = g_p / (1.0 + min_policy * num_actions)
        dz = s * (g_s - (g_s * s).sum(axis=1, keepdims=True))
    dv = v - y_r
    cost_v = 0.5 * np.sum((y_r - v) ** 2)
    c1, c2 = cost_p_1.sum(), cost_p_2.sum()
    losses = dict(cost_p_1_agg=c1, cost_p_2_agg=c2, cost_v=cost_v, cost_all=-(c1 + c2) + cost_v)

    d1, flat, n2, n1 = f["d1"], f["flat"], f["n2"], f["n1"]
    g = {}
    g["logits_p/w"] = d1.T @ dz
    g["logits_p/b"] = dz.sum(axis=0)
    g["logits_v/w"] = d1.T @ dv[:, None]
    g["logits_v/b"] = dv.sum(keepdims=True)
    dd1 = (dz @ params["logits_p/w"].T + dv[:, None] @ params["logits_v/w"].T) * (d1 > 0)
    g["dense1/w"] = flat.T @ dd1
    g["dense1/b"] = dd1.sum(axis=0)
    dn2 = (dd1 @ params["dense1/w"].T).reshape(n2.shape) * (n2 > 0)
    g["conv12/w"], g["conv12/b"], dn1 = _conv_bwd(dn2, f["cols2"], params["conv12/w"], CONV2, 21, True)
    dn1 = dn1 * (n1 > 0)
    g["conv11/w"], g["conv11/b"], _ = _conv_bwd(dn1, f["cols1"], params["conv11/w"], CONV1, 84, False)
    g["dz"], g["dv"], g["dd1"], g["dn2"], g["dn1"] = dz, dv, dd1, dn2, dn1
    return losses, g


def clip_by_average_norm(g, clip):
    """tf.clip_by_average_norm (NetworkVP_discrate.py:121): g*clip/max(||g||2/n, clip)."""
    n = g.size
    return g * clip / max(np.sqrt(np.sum(g * g)) / n, clip)


def rmsprop_update(params, ms, grads, lr, decay=0.99, eps=0.1, momentum=0.0, mom=None):
    """TF-1.x ApplyRMSProp: ms<-rho*ms+(1-rho)g^2; mom<-mu*mom+lr*g/sqrt(ms+eps); theta-=mom.

    The `ms` slot starts at ones (TF RMSPropOptimizer._create_slots), eps inside the sqrt.
    Updates dicts in place and returns them.
    """
    for k in PARAM_ORDER:
        dt = params[k].dtype.type
        g = grads[k].reshape(params[k].shape).astype(params[k].dtype)
        ms[k] = dt(decay) * ms[k] + dt(1.0 - decay) * g * g
        step = dt(lr) * g / np.sqrt(ms[k] + dt(eps))
        if momentum != 0.0:
            mom[k] = dt(momentum) * mom[k] + step
            step = mom[k]
        params[k] = params[k] - step
    return params, ms


def train_step(params, ms, x, y_r, a, lr, beta, **kw):
    losses, g = loss_and_grads(params, x, y_r, a, beta, **kw)
    rmsprop_update(params, ms, g, lr)
    return losses, g


# ---------------------------------------------------------------- plumbing
def accumulate_rewards_fork(rewards, gamma, terminal_reward, discounting=True,
                            use_intermediate_reward=False):
    """ProcessAgent.py:69-84, bit-exact: python-float (IEEE f64) sequential products.

    Returns the list of T per-step training targets (ALL T rows are kept, :84).
    Only the branch `DISCOUNTING and not USE_INTERMEDIATE_REWARD` writes anything
    back (:76-82); REWARD_CLIPPING never reaches the output (:73-74).
    """
    out = [float(r) for r in rewards]
    reward_sum = float(terminal_reward)
    for t in reversed(range(0, len(out) - 1)):
        if discounting:
            reward_sum = float(gamma) * reward_sum
            if use_intermediate_reward:
                # reference does `reward_sum = gamma*reward_sum + r` here and stores nothing
                reward_sum = float(gamma) * reward_sum + float(np.clip(out[t], -1, 1))
            else:
                out[t] = reward_sum
    return out


def returns_nstep(rewards, gamma, bootstrap_value, rmin=-1.0, rmax=1.0):
    """Upstream-GA3C n-step return (the lines the fork commented out, ProcessAgent.py:83,146):
    R <- clip(r_t) + gamma*R from t=T-2 down to 0, seeded with the bootstrap value; last row dropped."""
    out = [float(r) for r in rewards]
    reward_sum = float(bootstrap_value)
    for t in reversed(range(0, len(out) - 1)):
        r = min(max(out[t], rmin), rmax)
        reward_sum = float(gamma) * reward_sum + r
        out[t] = reward_sum
    return out[:-1]


def convert_data(states, actions, rewards, next_states, dones, num_actions):
    """ProcessAgent.py:86-100 (discrete branch :93-98): a rollout as the five arrays the trainer queue carries --
    states stacked as they are (f32), one-hot actions in f32 made from a float64 identity, returns left in f64."""
    x_ = np.array(list(states))
    x2_ = np.array(list(next_states))
    done_ = np.array(list(dones))
    a_ = np.eye(num_actions)[np.array(list(actions))].astype(np.float32)
    r_ = np.array([r for r in rewards])
    return x_, r_, a_, x2_, done_


def select_action_index(prediction, u, play_mode=False):
    """ProcessAgent.py:109-115: argmax in PLAY_MODE, else np.random.choice(actions, p=prediction).  `u` is the one
    uniform that call draws from the global RandomState (random_sample()); the rest is choice()'s own arithmetic:
    float64 cumulative sum of p, divided by its last entry, searchsorted(side='right')."""
    if play_mode:
        return int(np.argmax(prediction))
    cdf = np.cumsum(np.asarray(prediction, dtype=np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side='right'))


def predictor_batches(n_queued, batch_max):
    """ThreadPredictor.py:50-55: block for 1, then drain greedily up to batch_max."""
    sizes = []
    left = n_queued
    while left > 0:
        take = min(left, batch_max)
        sizes.append(take)
        left -= take
    return sizes


def trainer_batches(rollout_rows, min_batch):
    """ThreadTrainer.py:48-59: concatenate rollouts while batch_size <= TRAINING_MIN_BATCH_SIZE."""
    sizes, cur = [], 0
    for rows in rollout_rows:
        cur += rows
        if cur > min_batch:
            sizes.append(cur)
            cur = 0
    return sizes


def anneal(start, end, anneal_episodes, episode):
    """Server.py:168-175."""
    mult = (end - start) / anneal_episodes
    return start + mult * min(episode, anneal_episodes - 1)
