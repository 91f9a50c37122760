"""Regenerates the fixtures in tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_golden.py`); /root/reference does not exist on the GPU box.

What is produced, and from what:
  batcher_traces.json   by importing the REFERENCE's own ThreadPredictor / ThreadTrainer / Config
                        (they import unchanged with numpy) and driving them with a fake server;
  returns_fork.json     the vectors the survey recorded from the reference's
                        ProcessAgent._accumulate_rewards.  ProcessAgent
                        itself needs gym/skimage, which are absent and stay absent, so it is not
                        imported here;
  nn_small.npz          f64-oracle outputs on seeded inputs (the NN path has no reference fixture:
                        "parity unpinned", see oracle/ga3c_oracle.py);
  frontend.npz          RGB frames and the uint8 84x84 planes Environment._preprocess (Environment.py:52-60) makes
                        of them, computed with the reference's own ingredients where this image has them:
                        numpy.dot for the gray product and Pillow's Image.resize(BILINEAR) -- the resampler
                        scipy.misc.imresize called; scipy.misc.bytescale (SciPy <= 1.2, absent) is the one step
                        taken from the restatement in oracle/frame_frontend.py.
Fixtures are data only: inputs and expected outputs.
"""
import json
import os
import queue
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference/ga3c"


def batcher_traces():
    sys.path.insert(0, REF)
    from Config import Config                       # reference module
    from ThreadPredictor import ThreadPredictor     # reference module
    from ThreadTrainer import ThreadTrainer         # reference module

    out = {}
    # ---- predictor: 300 queued requests from 64 agents, batch max 128
    state_dim, n_agents, n_req, n_act = 32, 64, 300, 6

    class WaitQ:
        def __init__(self):
            self.items = []

        def put(self, item):
            self.items.append(item)

    class Agent:
        def __init__(self):
            self.wait_q = WaitQ()

    class Model:
        def __init__(self):
            self.batch_sizes = []

        def predict_p_and_v(self, batch):
            self.batch_sizes.append(int(batch.shape[0]))
            return batch[:, :n_act].copy(), batch.sum(axis=1)

    class Server:
        pass

    for batch_max in (128, 32):
        Config.PREDICTION_BATCH_SIZE = batch_max
        srv = Server()
        srv.model = Model()
        srv.agents = [Agent() for _ in range(n_agents)]
        q = queue.Queue()
        rng = np.random.default_rng(5)
        states = rng.integers(0, 256, size=(n_req, state_dim)).astype(np.float32)
        ids = [int(i % n_agents) for i in range(n_req)]
        for i in range(n_req):
            q.put((ids[i], states[i]))
        th = ThreadPredictor(srv, 0, state_dim, q)
        th.start()
        t0 = time.time()
        while sum(srv.model.batch_sizes) < n_req and time.time() - t0 < 10:
            time.sleep(0.01)
        time.sleep(0.05)
        th.exit_flag = True           # thread stays parked in q.get(); it is a daemon
        routed = [[float(v) for (_, v) in srv.agents[a].wait_q.items] for a in range(n_agents)]
        out["predictor_%d" % batch_max] = dict(
            n_requests=n_req, n_agents=n_agents, batch_max=batch_max, state_dim=state_dim,
            seed=5, batch_sizes=srv.model.batch_sizes, value_routed_per_agent=routed)

    # ---- trainer: rollouts of given row counts, TRAINING_MIN_BATCH_SIZE in {0, 8, 127}
    Config.USE_REPLAY_MEMORY = False
    Config.TRAIN_MODELS = True
    for min_batch, rows in ((0, [5, 6, 6, 3]), (8, [5] * 6), (127, [6] * 30 + [3, 6, 6])):
        Config.TRAINING_MIN_BATCH_SIZE = min_batch
        calls = []

        class TServer:
            def __init__(self):
                self.training_q = queue.Queue()

            def train_model(self, x, r, a, x2, done, tid):
                calls.append(dict(rows=int(x.shape[0]), r_sum=float(np.sum(r)), first=float(x[0, 0]),
                                  last=float(x[-1, 0])))

        srv = TServer()
        base = 0
        for n in rows:
            x = (np.arange(n, dtype=np.float32) + base).reshape(n, 1)
            base += n
            srv.training_q.put((x, x[:, 0].astype(np.float64), np.eye(4, dtype=np.float32)[np.zeros(n, int)],
                                x.copy(), np.zeros(n, bool)))
        th = ThreadTrainer(srv, 0)
        th.start()
        t0 = time.time()
        while not srv.training_q.empty() and time.time() - t0 < 10:
            time.sleep(0.01)
        time.sleep(0.05)
        th.exit_flag = True
        out["trainer_min%d" % min_batch] = dict(min_batch=min_batch, rollout_rows=rows, calls=calls)
    with open(os.path.join(HERE, "batcher_traces.json"), "w") as f:
        json.dump(out, f, indent=1)


def returns_fork():
    # Recorded by the survey from the reference's ProcessAgent._accumulate_rewards
    # (SURVEY.md §8-a3 golden vector, Appendix C flag table).  Data only.
    data = dict(
        source="SURVEY.md section 8-a3 + Appendix C (reference ProcessAgent.py:69-84 run at survey time)",
        rewards=[0.0, 0.5, -3.0, 2.0, 1.0], gamma=0.99, terminal_reward=1.0,
        cases=[
            dict(reward_clipping=True, discounting=True, use_intermediate_reward=False,
                 out_hex=["0x1.ebd33d7f3c762p-1", "0x1.f0cb07d0aed99p-1", "0x1.f5cfaacd9e83ep-1",
                          "0x1.fae147ae147aep-1", "0x1.0p+0"],
                 out_repr=[0.96059601, 0.9702989999999999, 0.9801, 0.99, 1.0]),
            dict(reward_clipping=False, discounting=True, use_intermediate_reward=False,
                 out_repr=[0.96059601, 0.9702989999999999, 0.9801, 0.99, 1.0]),
            dict(reward_clipping=True, discounting=True, use_intermediate_reward=True,
                 out_repr=[0.0, 0.5, -3.0, 2.0, 1.0]),
            dict(reward_clipping=True, discounting=False, use_intermediate_reward=False,
                 out_repr=[0.0, 0.5, -3.0, 2.0, 1.0]),
        ],
        convert_data_dtypes=dict(x_="float32", r_="float64", a_="float32", x2_="float32", done_="bool"))
    # ORACLE-DERIVED cases (oracle/ga3c_oracle.py:accumulate_rewards_fork, NOT recorded from the reference): they pin the C
    # ABI and the agents' rollout code to the restatement on the shapes the recorded vector does not cover -- T = 1, T = 2,
    # zero / negative terminal rewards and a TIME_MAX + 1 = 6 row first rollout (ProcessAgent.py:157-162).  Kept under
    # their own key so the reference-recorded vector above stays distinguishable.
    import ga3c_oracle as o
    rng = np.random.default_rng(606)
    derived = []
    for rewards, gamma in (([2.5], 0.99), ([7.0, -1.0], 0.5), ([0.0, 0.0, 0.0], 0.99), ([1.0, -1.0, 0.0, 0.0, 1.0, -1.0], 0.99),
                           ([0.25, 0.0, -0.75, 3.0, 0.0, 0.0], 0.99), (list(rng.normal(size=6)), 0.99),
                           (list(rng.normal(size=33)), 0.97), ([0.0] * 5 + [-1.0], 0.99)):
        rewards = [float(r) for r in rewards]
        for disc, inter in ((True, False), (True, True), (False, False)):
            out = o.accumulate_rewards_fork(rewards, gamma, rewards[-1], discounting=disc, use_intermediate_reward=inter)
            derived.append(dict(rewards_hex=[r.hex() for r in rewards], gamma=gamma, terminal_reward_hex=rewards[-1].hex(),
                                discounting=disc, use_intermediate_reward=inter, out_hex=[float(v).hex() for v in out]))
    data["oracle_derived_note"] = ("computed by oracle/ga3c_oracle.py, not by the reference: regression vectors for the C ABI "
                                   "and ProcessAgent (T=1, T=2, zero / negative terminal, TIME_MAX+1 rows)")
    data["oracle_derived_cases"] = derived
    with open(os.path.join(HERE, "returns_fork.json"), "w") as f:
        json.dump(data, f, indent=1)


def nn_small():
    import ga3c_oracle as o
    out = {}
    for num_actions, bsz in ((6, 5), (4, 3), (18, 2)):
        rng = np.random.Generator(np.random.PCG64(1000 + num_actions))
        xk = rng.integers(0, 256, size=(bsz, 84, 84, 4), dtype=np.uint8)
        act = rng.integers(0, num_actions, size=bsz)
        y_r = rng.uniform(-1.0, 1.0, size=bsz)
        x = xk.astype(np.float64) / 128.0 - 1.0
        a = np.eye(num_actions)[act]
        params = o.init_params(num_actions, seed=12345)
        fwd = o.forward(params, x, keep=True)
        losses, g = o.loss_and_grads(params, x, y_r, a, beta=0.01)
        ms = {k: np.ones_like(v) for k, v in params.items()}
        new = {k: v.copy() for k, v in params.items()}
        o.rmsprop_update(new, ms, g, lr=3e-4)
        t = "A%d_" % num_actions
        out[t + "x_u8"] = xk
        out[t + "actions"] = act.astype(np.int32)
        out[t + "y_r"] = y_r
        out[t + "z"], out[t + "p"], out[t + "v"] = fwd["z"], fwd["p"], fwd["v"]
        out[t + "n1_sum"] = fwd["n1"].sum(axis=(1, 2, 3))
        out[t + "n2_sum"] = fwd["n2"].sum(axis=(1, 2, 3))
        out[t + "d1"] = fwd["d1"]
        out[t + "losses"] = np.array([losses["cost_p_1_agg"], losses["cost_p_2_agg"], losses["cost_v"]])
        out[t + "dz"], out[t + "dv"] = g["dz"], g["dv"]
        for k in o.PARAM_ORDER:
            gk = g[k].reshape(params[k].shape)
            key = k.replace("/", ".")
            out[t + "gnorm." + key] = np.sqrt(np.sum(gk * gk))
            if gk.size <= 8192:
                out[t + "grad." + key] = gk
            else:                       # dense1/w: keep a strided sample
                out[t + "grad." + key] = gk.ravel()[::997].copy()
            out[t + "delta_norm." + key] = np.sqrt(np.sum((new[k] - params[k]) ** 2))
    np.savez_compressed(os.path.join(HERE, "nn_small.npz"), **out)


def frontend():
    from PIL import Image
    import frame_frontend as ff
    rng = np.random.default_rng(2024)
    frames = {}
    pong = np.empty((210, 160, 3), np.uint8)
    pong[:] = (144, 72, 17)                                  # playfield
    pong[:34] = (109, 118, 43)                               # score band
    pong[24:34] = (236, 236, 236)
    pong[194:] = (236, 236, 236)
    pong[90:106, 16:20] = (213, 130, 74)                     # paddles and ball
    pong[120:136, 140:144] = (92, 186, 92)
    pong[101:105, 79:81] = (236, 236, 236)
    pong[4:20, 36:44] = (213, 130, 74)
    frames["pong_like"] = pong
    frames["random"] = rng.integers(0, 256, size=(210, 160, 3), dtype=np.uint8)
    tall = (np.add.outer(np.arange(250) * 3, np.arange(160) * 5)[..., None] // np.array([1, 2, 3]) % 256).astype(np.uint8)
    frames["tall_gradient"] = tall
    frames["constant"] = np.full((210, 160, 3), 77, np.uint8)
    frames["small_random"] = rng.integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
    frames["low_contrast"] = (100 + rng.integers(0, 3, size=(210, 160, 3))).astype(np.uint8)
    out = {}
    for name, rgb in frames.items():
        gray = np.dot(rgb[..., :3], [0.299, 0.587, 0.114])                       # Environment.py:54, verbatim call
        img = Image.frombytes('L', (gray.shape[1], gray.shape[0]), ff.bytescale(gray).tobytes())   # toimage()
        plane = np.asarray(img.resize((84, 84), resample=Image.BILINEAR))       # imresize(..., 'bilinear')
        out["rgb_" + name] = rgb
        out["plane_" + name] = plane.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **out)


if __name__ == "__main__":
    frontend()
    returns_fork()
    nn_small()
    batcher_traces()
    print("golden fixtures written to", HERE)
    os._exit(0)     # reference batcher threads are parked in blocking get()s
