"""-m gpu: the BASELINE.json configurations that round 1 left untested on hardware, and the drop-in scripts.

  configs[3]  Breakout (A = 4), batch 512: ALL rows of the forward pass and the full gradient against the f64 oracle
              (the oracle is evaluated in row chunks: the forward pass is row-independent and the loss is a sum over rows,
              NetworkVP_discrate.py:61,83-85), then one optimizer step.  (Its 4-rank RCCL part needs 4 GPUs.)
  configs[4]  Boxing (A = 18) + Pong (A = 6) mixed: two engines at once on one device, DYNAMIC_SETTINGS on,
              GA3C_GRAPHS=1 (hipGraph-replayed predictor steps), each engine's checkpoint checked against the oracle.
  scripts     sh _train.sh KEY=VALUE ... / sh _play.sh / sh _clean.sh as the reference is driven (_train.sh:1-3,
              GA3C.py:38-59): results.txt wire format, status line, checkpoint naming, play mode leaves the weights alone.
"""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import ga3c_oracle as o

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ga3c_amd")
TOL = 1e-4
STATUS = re.compile(r"^\[Time: +\d+\] \[Episode: +\d+ Score: +-?\d+\.\d{4}\] \[RScore: +-?\d+\.\d{4} RPPS: +\d+\] "
                    r"\[PPS: +\d+ TPS: +\d+\] \[NT: +(\d+) NP: +(\d+) NA: +(\d+)\]\[RSize: +\d+\]$")
RESULT = re.compile(r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d, -?\d+, \d+$")


def _params_of(arena, num_actions):
    out, off = {}, 0
    for name in o.PARAM_ORDER:
        shape = o.param_shapes(num_actions)[name]
        size = int(np.prod(shape))
        out[name] = arena[off:off + size].astype(np.float64).reshape(shape)
        off += size
    return out


def _flat(d):
    return np.concatenate([np.asarray(d[k]).reshape(-1) for k in o.PARAM_ORDER])


def _batch(bsz, num_actions, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    xk = rng.integers(0, 256, size=(bsz, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128.0) - np.float32(1.0)
    return xk, x, np.eye(num_actions, dtype=np.float32)[rng.integers(0, num_actions, size=bsz)], rng.uniform(-1, 1, size=bsz)


def test_config3_breakout_batch512_all_rows_and_full_gradient():
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    A, B, CH = 4, 512, 64
    net = Network("gpu:0", "breakout", A, (84, 84, 4), max_batch=B, predict_lanes=2)
    try:
        _, x, a, y = _batch(B, A, 3512)
        params = _params_of(net.get_arena(0), A)
        p, v, z = net.predict_p_v_logits(x)
        net.beta = 0.01
        losses = net.compute_grads(x, y, a)
        got_g = net.get_arena(3)
        got_rows = {k: net.fetch(k, n * B).reshape(B, n) for k, n in (("dz", A), ("dv", 1), ("dd1", 256))}
        want_l = np.zeros(3)
        want_g = {k: 0.0 for k in o.PARAM_ORDER}
        for lo in range(0, B, CH):
            xs = x[lo:lo + CH].astype(np.float64)
            ref = o.forward(params, xs)
            assert np.max(np.abs(p[lo:lo + CH] - ref["p"])) < TOL                      # every row, not a sample of them
            assert np.max(np.abs(v[lo:lo + CH] - ref["v"])) < TOL
            assert np.max(np.abs(z[lo:lo + CH] - ref["z"])) < TOL
            l, g = o.loss_and_grads(params, xs, y[lo:lo + CH], a[lo:lo + CH].astype(np.float64), 0.01)
            want_l += [l["cost_p_1_agg"], l["cost_p_2_agg"], l["cost_v"]]
            for k in o.PARAM_ORDER:
                want_g[k] = want_g[k] + np.asarray(g[k])
            for k in ("dz", "dv", "dd1"):
                want = np.asarray(g[k]).reshape(CH, -1)
                assert np.max(np.abs(got_rows[k][lo:lo + CH] - want)) < TOL * max(1.0, np.max(np.abs(want))), k
        assert np.allclose(losses, want_l, rtol=1e-4, atol=1e-3)
        off = 0
        for name in o.PARAM_ORDER:
            want = np.asarray(want_g[name]).reshape(-1)
            g = got_g[off:off + want.size]
            off += want.size
            scale = max(np.max(np.abs(want)), 1.0)
            assert np.max(np.abs(g - want)) < TOL * scale, (name, np.max(np.abs(g - want)), scale)
        # one optimizer step on those gradients (TF-1.x RMSProp, ms slot = ones)
        net.learning_rate = 3e-4
        net.apply_grads()
        ms = {k: np.ones_like(t) for k, t in params.items()}
        o.rmsprop_update(params, ms, want_g, 3e-4)
        assert np.max(np.abs(net.get_arena(0) - _flat(params))) < 1e-5
        assert np.max(np.abs(net.get_arena(1) - _flat(ms))) < 1e-5 * max(1.0, np.max(np.abs(_flat(ms))))
    finally:
        net.close()


def _check_checkpoint_against_oracle(path, num_actions, seed):
    """The weights an engine saved drive the HIP path and the oracle to the same outputs, and one more train step from
    the saved optimizer state lands on the same weights."""
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network, PARAM_ORDER
    with np.load(path, allow_pickle=False) as z:
        assert z["logits_p/w:0"].shape == (256, num_actions) and int(z["step"]) > 0
        theta = np.concatenate([z[n + ":0"].ravel() for n in PARAM_ORDER])
        ms_flat = np.concatenate([z[n + "/RMSProp:0"].ravel() for n in PARAM_ORDER])
    assert np.all(np.isfinite(theta)) and np.all(ms_flat > 0)
    fresh = _flat(o.init_params(num_actions)).astype(np.float32)
    assert np.max(np.abs(theta - fresh)) > 1e-5                       # the engine trained
    net = Network("gpu:0", "check", num_actions, (84, 84, 4), max_batch=32, predict_lanes=1)
    try:
        net.set_arena(0, theta)
        net.set_arena(1, ms_flat)
        xk, x, a, y = _batch(24, num_actions, seed)
        params = _params_of(theta, num_actions)
        p, v = net.predict_p_and_v(xk)
        ref = o.forward(params, x.astype(np.float64))
        assert np.max(np.abs(p - ref["p"])) < TOL and np.max(np.abs(v - ref["v"])) < TOL
        ms = _params_of(ms_flat, num_actions)
        net.learning_rate, net.beta = 3e-4, 0.01
        net.train(xk, y, a)
        o.train_step(params, ms, x.astype(np.float64), y, a.astype(np.float64), 3e-4, 0.01)
        assert np.max(np.abs(net.get_arena(0) - _flat(params))) < 2e-5
    finally:
        net.close()


@pytest.mark.timeout(300)
def test_config4_boxing_and_pong_mixed_dynamic_workers_graph_predictor(tmp_path):
    env = dict(os.environ, GA3C_GRAPHS="1", PYTHONUNBUFFERED="1")
    args = ["GAMES=Boxing:18:0,Pong:6:0", "AGENTS=6", "PREDICTORS=2", "TRAINERS=2", "DYNAMIC_SETTINGS=True",
            "DYNAMIC_SETTINGS_INITIAL_WAIT=2", "DYNAMIC_SETTINGS_STEP_WAIT=1", "EPISODES=100000000", "MAX_SECONDS=14",
            "SYNTHETIC_EPISODE_LENGTH=60", "SAVE_FREQUENCY=25", "TRAINING_MIN_BATCH_SIZE=31", "PREDICTION_BATCH_SIZE=64",
            "PRINT_STATS_FREQUENCY=5"]
    run = subprocess.run(["sh", os.path.join(PKG, "_train_mixed.sh")] + args, cwd=str(tmp_path), env=env, capture_output=True,
                         text=True, timeout=240)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "engine Boxing ended with status 0" in run.stdout and "engine Pong ended with status 0" in run.stdout
    shapes = set()
    for line in run.stdout.splitlines():
        m = STATUS.match(line)
        if m:
            shapes.add(tuple(int(g) for g in m.groups()))
    assert len(shapes) >= 2, shapes                                    # the random walk moved NT / NP / NA while both engines ran
    for name, A in (("Boxing", 18), ("Pong", 6)):
        lines = open(os.path.join(str(tmp_path), "results_%s.txt" % name)).read().strip().splitlines()
        assert len(lines) >= 25 and all(RESULT.match(ln) for ln in lines)
        found = sorted(glob.glob(os.path.join(str(tmp_path), "checkpoints", "%s_????????.npz" % name)))
        assert found, os.listdir(os.path.join(str(tmp_path), "checkpoints"))
        _check_checkpoint_against_oracle(found[-1], A, 4000 + A)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("where", ["package_dir", "elsewhere"])
def test_train_play_clean_scripts(tmp_path, where):
    """`sh _train.sh KEY=VALUE ...` then `sh _play.sh` the way the reference's README drives them.  'package_dir': from
    inside ga3c_amd/ (the reference's own usage; skipped when that directory already holds run outputs, so that
    _clean.sh never deletes somebody's results); 'elsewhere': the scripts also work from any directory."""
    if where == "package_dir":
        cwd = PKG
        if any(os.path.exists(os.path.join(PKG, n)) for n in ("results.txt", "checkpoints", "logs")):
            pytest.skip("ga3c_amd/ holds run outputs")
        call = lambda script, *a: ["sh", script] + list(a)                  # noqa: E731
    else:
        cwd = str(tmp_path)
        call = lambda script, *a: ["sh", os.path.join(PKG, script)] + list(a)   # noqa: E731
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    common = ["AGENTS=8", "PREDICTORS=2", "TRAINERS=1", "SYNTHETIC_EPISODE_LENGTH=30", "TRAINING_MIN_BATCH_SIZE=15",
              "PREDICTION_BATCH_SIZE=32", "PRINT_STATS_FREQUENCY=1", "TENSORBOARD=True", "TENSORBOARD_UPDATE_FREQUENCY=20"]
    try:
        run = subprocess.run(call("_train.sh", "EPISODES=60", "SAVE_FREQUENCY=20", *common), cwd=cwd, env=env,
                             capture_output=True, text=True, timeout=200)
        assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
        status = [ln for ln in run.stdout.splitlines() if STATUS.match(ln)]
        assert len(status) >= 55                                               # one line per episode (PRINT_STATS_FREQUENCY=1)
        assert {STATUS.match(ln).groups() for ln in status} == {("1", "2", "8")}
        lines = open(os.path.join(cwd, "results.txt")).read().strip().splitlines()
        assert len(lines) >= 60 and all(RESULT.match(ln) for ln in lines)
        found = sorted(glob.glob(os.path.join(cwd, "checkpoints", "network_????????.npz")))
        assert found
        episode = int(re.search(r"network_(\d{8})\.npz$", found[-1]).group(1))
        assert 20 <= episode <= 80
        scalars = open(os.path.join(cwd, "logs", "network", "scalars.csv")).read().strip().splitlines()
        assert scalars and all(len(r.split(",")) == 7 for r in scalars)
        assert glob.glob(os.path.join(cwd, "logs", "network", "histograms_*.npz"))
        before = open(found[-1], "rb").read()
        n_results = len(lines)
        # play mode: one agent, greedy, checkpoint loaded, nothing trained or saved (GA3C.py:46-54)
        play = subprocess.run(call("_play.sh", "EPISODES=%d" % (episode + 6), "SYNTHETIC_EPISODE_LENGTH=30",
                                   "PRINT_STATS_FREQUENCY=1"), cwd=cwd, env=env, capture_output=True, text=True, timeout=200)
        assert play.returncode == 0, play.stdout[-3000:] + play.stderr[-3000:]
        pstatus = [STATUS.match(ln) for ln in play.stdout.splitlines() if STATUS.match(ln)]
        assert len(pstatus) >= 5 and {m.groups() for m in pstatus} == {("1", "1", "1")}
        assert all("TPS:     0]" in m.group(0) for m in pstatus)             # nothing is trained in play mode
        assert len(open(os.path.join(cwd, "results.txt")).read().strip().splitlines()) >= n_results + 5
        assert sorted(glob.glob(os.path.join(cwd, "checkpoints", "network_????????.npz"))) == found
        assert open(found[-1], "rb").read() == before
    finally:
        if where == "package_dir":
            subprocess.run(["sh", "_clean.sh"], cwd=cwd)
            assert not os.path.exists(os.path.join(cwd, "results.txt"))
            assert not glob.glob(os.path.join(cwd, "checkpoints", "*")) and not glob.glob(os.path.join(cwd, "logs", "*", "*"))
            for d in (os.path.join(cwd, "logs", "network"), os.path.join(cwd, "logs"), os.path.join(cwd, "checkpoints")):
                if os.path.isdir(d):
                    os.rmdir(d)
