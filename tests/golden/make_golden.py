"""Regenerates the fixtures in tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_golden.py`); /root/reference does not exist on the GPU box.

What is produced, and from what:
  batcher_traces.json   by importing the REFERENCE's own ThreadPredictor / ThreadTrainer / Config
                        (they import unchanged with numpy) and driving them with a fake server;
  returns_fork.json     ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84) RUN here on 25 reward vectors x the
                        four flag settings of SURVEY.md Appendix C (the survey's own hex vector is case 0);
  process_agent.json    ProcessAgent.convert_data (:86-100) and select_action (:109-115) RUN here: array dtypes /
                        shapes / values, and the draws under fixed np.random.seed values for A = 4, 6, 18 and in
                        PLAY_MODE.  ProcessAgent is imported as SURVEY.md Appendix C records (empty modules named
                        skimage / gym so that its import statements pass: see _import_reference_process_agent);
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


def _import_reference_process_agent():
    """The reference's own ProcessAgent module (ProcessAgent.py:44-178), imported the way SURVEY.md Appendix C records:
    its import chain (ProcessAgent.py:36-39 -> EnvironmentPend.py:35-40 -> PyperEnvironment.py:1 -> pyper_env.py:19-20)
    names `skimage` and `gym`, which this image does not have.  EMPTY modules of those names are registered so that the
    import statements pass; they hold nothing but the two names the chain imports `from` them (bound to None), nothing of
    either library is restated, and none of the three functions recorded below touches them (they are numpy + Config
    only).  The reference's files stay where they are; only the recorded inputs / outputs leave this container."""
    import types
    import warnings
    for name, names in (("skimage", ()), ("skimage.morphology", ("disk",)), ("skimage.color", ("rgb2gray",)),
                        ("gym", ()), ("gym.wrappers", ())):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            for n in names:
                setattr(mod, n, None)
            sys.modules[name] = mod
    sys.modules["gym"].wrappers = sys.modules["gym.wrappers"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import ProcessAgent as ref_pa               # reference module
    from Config import Config as RefConfig          # reference module
    from Experience import Experience as RefExperience
    return ref_pa.ProcessAgent, RefConfig, RefExperience


SOURCE = "reference run by make_golden.py"


def returns_fork():
    """ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84) RUN on 25 reward vectors x the four flag settings of
    SURVEY.md Appendix C; every number in the fixture is what the reference's function returned here."""
    RefAgent, RefConfig, RefExperience = _import_reference_process_agent()
    rng = np.random.default_rng(606)
    inputs = [([0.0, 0.5, -3.0, 2.0, 1.0], 0.99, 1.0)]        # the survey's vector (terminal given separately there)
    for rewards, gamma in (([2.5], 0.99), ([7.0, -1.0], 0.5), ([0.0, 0.0, 0.0], 0.99), ([1.0, -1.0, 0.0, 0.0, 1.0, -1.0], 0.99),
                           ([0.25, 0.0, -0.75, 3.0, 0.0, 0.0], 0.99), (list(rng.normal(size=6)), 0.99),
                           (list(rng.normal(size=33)), 0.97), ([0.0] * 5 + [-1.0], 0.99)):
        rewards = [float(r) for r in rewards]
        inputs.append((rewards, gamma, rewards[-1]))          # call site :148-149: terminal = the last raw reward
        inputs.append((rewards, gamma, 0.0))
        inputs.append((rewards, 1.0, -rewards[-1]))
    cases = []
    keep = {k: getattr(RefConfig, k) for k in ("REWARD_CLIPPING", "DISCOUNTING", "USE_INTERMEDIATE_REWARD")}
    try:
        for rewards, gamma, terminal in inputs:
            for clip, disc, inter in ((True, True, False), (False, True, False), (True, True, True), (True, False, False)):
                RefConfig.REWARD_CLIPPING, RefConfig.DISCOUNTING, RefConfig.USE_INTERMEDIATE_REWARD = clip, disc, inter
                exps = [RefExperience(None, 0, None, r, None, False) for r in rewards]
                out = RefAgent._accumulate_rewards(exps, gamma, terminal)
                assert out is exps
                cases.append(dict(rewards_hex=[float(r).hex() for r in rewards], gamma=gamma,
                                  terminal_reward_hex=float(terminal).hex(), reward_clipping=clip, discounting=disc,
                                  use_intermediate_reward=inter, rows_out=len(out),
                                  out_hex=[float(e.reward).hex() for e in out]))
        # the one setting that does not return: clipping off + intermediate rewards on reads an unset name (:73-80)
        RefConfig.REWARD_CLIPPING, RefConfig.DISCOUNTING, RefConfig.USE_INTERMEDIATE_REWARD = False, True, True
        try:
            RefAgent._accumulate_rewards([RefExperience(None, 0, None, r, None, False) for r in (1.0, 2.0)], 0.99, 2.0)
            raised = None
        except Exception as e:                      # noqa: BLE001 - the type is what is recorded
            raised = type(e).__name__
    finally:
        for k, v in keep.items():
            setattr(RefConfig, k, v)
    data = dict(source=SOURCE, function="ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84)", cases=cases,
                clip_off_intermediate_on_raises=raised)
    with open(os.path.join(HERE, "returns_fork.json"), "w") as f:
        json.dump(data, f, indent=1)


def process_agent():
    """ProcessAgent.convert_data (ProcessAgent.py:86-100) and ProcessAgent.select_action (:109-115) RUN here: dtypes,
    shapes and values of the five arrays a rollout becomes, and the actions drawn under fixed np.random.seed values."""
    RefAgent, RefConfig, RefExperience = _import_reference_process_agent()
    rng = np.random.default_rng(707)
    out = dict(source=SOURCE, convert_data=[], select_action=[])
    keep = {k: getattr(RefConfig, k) for k in ("CONTINUOUS_INPUT", "PLAY_MODE")}
    try:
        RefConfig.CONTINUOUS_INPUT = False

        class Self:                                 # convert_data reads self.num_actions only
            pass

        for num_actions, rows, state_shape in ((4, 1, (3, 3, 2)), (6, 6, (3, 3, 2)), (18, 5, (2, 2, 4)), (6, 2, (84, 84, 4))):
            me = Self()
            me.num_actions = num_actions
            big = int(np.prod(state_shape)) > 64
            states = [(rng.integers(0, 256, size=state_shape).astype(np.float32) / np.float32(128) - np.float32(1))
                      for _ in range(rows + 1)]
            actions = [int(a) for a in rng.integers(0, num_actions, size=rows)]
            rewards = [float(r) for r in rng.normal(size=rows)]
            dones = [False] * (rows - 1) + [True]
            exps = [RefExperience(states[t], actions[t], None, rewards[t], states[t + 1], dones[t]) for t in range(rows)]
            x_, r_, a_, x2_, done_ = RefAgent.convert_data(me, exps)
            rec = dict(num_actions=num_actions, state_shape=list(state_shape), actions=actions,
                       rewards_hex=[r.hex() for r in rewards], dones=dones,
                       dtypes=dict(x_=str(x_.dtype), r_=str(r_.dtype), a_=str(a_.dtype), x2_=str(x2_.dtype),
                                   done_=str(done_.dtype)),
                       shapes=dict(x_=list(x_.shape), r_=list(r_.shape), a_=list(a_.shape), x2_=list(x2_.shape),
                                   done_=list(done_.shape)),
                       a_=a_.tolist(), r_hex=[float(v).hex() for v in r_], done_=[bool(v) for v in done_],
                       x_is_stack_of_states=bool(all(np.array_equal(x_[t], states[t]) for t in range(rows))),
                       x2_is_stack_of_next_states=bool(all(np.array_equal(x2_[t], states[t + 1]) for t in range(rows))))
            if not big:                             # small states travel whole (f32 values are k/128 - 1: exact in JSON)
                rec["states"] = [s.tolist() for s in states]
                rec["x_"] = x_.tolist()
                rec["x2_"] = x2_.tolist()
            out["convert_data"].append(rec)

        for num_actions in (4, 6, 18):
            actions = np.arange(num_actions)
            for kind in ("softmax", "peaked", "zeros", "onehot", "uniform"):
                z = rng.normal(size=num_actions) * (8.0 if kind == "peaked" else 1.5)
                if kind == "uniform":
                    z[:] = 0.0
                p = np.exp(z - z.max())
                if kind == "zeros":
                    p[rng.permutation(num_actions)[:num_actions // 2]] = 0.0
                if kind == "onehot":
                    p[:] = 0.0
                    p[int(rng.integers(0, num_actions))] = 1.0
                p = (p / p.sum()).astype(np.float32)        # a prediction is a float32 row of the policy head
                for seed in (0, 1, 12345):
                    RefConfig.PLAY_MODE = False
                    np.random.seed(seed)
                    draws = [int(RefAgent.select_action(actions, p)) for _ in range(24)]
                    RefConfig.PLAY_MODE = True
                    play = int(RefAgent.select_action(actions, p))
                    out["select_action"].append(dict(num_actions=num_actions, kind=kind, seed=seed,
                                                     prediction_f32_hex=[float(v).hex() for v in p],
                                                     draws=draws, play_mode_action=play))
    finally:
        for k, v in keep.items():
            setattr(RefConfig, k, v)
    with open(os.path.join(HERE, "process_agent.json"), "w") as f:
        json.dump(out, f, indent=1)


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
    # `python tests/golden/make_golden.py [name ...]`: all fixtures, or only the named ones
    makers = dict(frontend=frontend, returns_fork=returns_fork, process_agent=process_agent, nn_small=nn_small,
                  batcher_traces=batcher_traces)
    for name in (sys.argv[1:] or list(makers)):
        makers[name]()
    print("golden fixtures written to", HERE)
    os._exit(0)     # reference batcher threads are parked in blocking get()s
