"""CPU tests of the pieces around the hot path: ThreadDynamicAdjustment (reference
ThreadDynamicAdjustment.py:53-144), the two return modes of ProcessAgent, convert_data dtypes."""
import numpy as np
import pytest


@pytest.fixture()
def cfg():
    import ga3c_amd  # noqa: F401
    from Config import Config
    saved = {k: v for k, v in vars(Config).items() if k.isupper()}
    yield Config
    for k, v in saved.items():
        setattr(Config, k, v)


class _Stats:
    class _V:
        value = 0

    def __init__(self):
        self.trainer_count, self.predictor_count, self.agent_count = self._V(), self._V(), self._V()


class _Server:
    def __init__(self):
        self.trainers, self.predictors, self.agents = [], [], []
        self.stats = _Stats()
        self.max_agents = 64
        self.log = []

    def add_trainer(self): self.trainers.append(0); self.log.append("+T")
    def remove_trainer(self): self.trainers.pop(); self.log.append("-T")
    def add_predictor(self): self.predictors.append(0); self.log.append("+P")
    def remove_predictor(self): self.predictors.pop(); self.log.append("-P")
    def add_agent(self): self.agents.append(0); self.log.append("+A")
    def remove_agent(self): self.agents.pop(); self.log.append("-A")


def test_dynamic_adjustment_creates_and_resizes_workers(cfg):
    from ThreadDynamicAdjustment import ThreadDynamicAdjustment
    cfg.TRAINERS, cfg.PREDICTORS, cfg.AGENTS, cfg.DYNAMIC_SETTINGS = 2, 3, 5, False
    srv = _Server()
    da = ThreadDynamicAdjustment(srv)
    da.run()                                           # not enabled: create the initial workers and return
    assert (len(srv.trainers), len(srv.predictors), len(srv.agents)) == (2, 3, 5)
    assert (srv.stats.trainer_count.value, srv.stats.predictor_count.value, srv.stats.agent_count.value) == (2, 3, 5)
    da.trainer_count, da.predictor_count, da.agent_count = 1, 4, 3
    da.enable_disable_components()
    assert (len(srv.trainers), len(srv.predictors), len(srv.agents)) == (1, 4, 3)
    # agents ARE removed when the target drops (the reference's branch at :79 can never run)
    assert srv.log.count("-A") == 2


def test_dynamic_adjustment_random_walk_bounds(cfg):
    from ThreadDynamicAdjustment import ThreadDynamicAdjustment
    cfg.TRAINERS, cfg.PREDICTORS, cfg.AGENTS = 1, 1, 1
    da = ThreadDynamicAdjustment(_Server())
    np.random.seed(0)
    seen = set()
    for _ in range(200):
        before = (da.trainer_count, da.predictor_count, da.agent_count)
        da.random_walk()
        after = (da.trainer_count, da.predictor_count, da.agent_count)
        assert all(a >= 1 for a in after) and all(abs(a - b) <= 1 for a, b in zip(after, before))
        assert da.agent_count <= 64
        seen.add(after)
    assert len(seen) > 10


def _experiences(rewards):
    from Experience import Experience
    s = np.zeros((84, 84, 4), np.uint8)
    return [Experience(s, i % 3, np.full(3, 1 / 3, np.float32), r, s, False) for i, r in enumerate(rewards)]


def test_return_modes(cfg):
    from ProcessAgent import ProcessAgent
    import ga3c_oracle as o
    rewards = [0.0, 0.5, -3.0, 2.0, 1.0]
    cfg.RETURN_MODE = 'fork'
    out = ProcessAgent._accumulate_rewards(_experiences(rewards), 0.99, 1.0)
    assert [e.reward for e in out] == o.accumulate_rewards_fork(rewards, 0.99, 1.0) and len(out) == 5
    cfg.USE_INTERMEDIATE_REWARD = True                  # the fork then writes nothing back (ProcessAgent.py:79-82)
    out = ProcessAgent._accumulate_rewards(_experiences(rewards), 0.99, 1.0)
    assert [e.reward for e in out] == rewards
    cfg.USE_INTERMEDIATE_REWARD = False
    cfg.RETURN_MODE = 'nstep'                           # upstream GA3C: clip, bootstrap, drop the last row
    out = ProcessAgent._accumulate_rewards(_experiences(rewards), 0.99, 0.25)
    assert [e.reward for e in out] == o.returns_nstep(rewards, 0.99, 0.25) and len(out) == 4
    r = 0.25
    for t in (3, 2, 1, 0):
        r = 0.99 * r + min(max(rewards[t], -1), 1)
        assert out[t].reward == r


def test_convert_data_matches_reference_recording(cfg, golden_dir):
    """The product's ProcessAgent.convert_data against what the reference's convert_data (ProcessAgent.py:86-100)
    returned in tests/golden/make_golden.py: dtypes, shapes, one-hot rows, returns, done flags, stacked states."""
    import json, os
    from Experience import Experience
    from ProcessAgent import ProcessAgent
    g = json.load(open(os.path.join(golden_dir, "process_agent.json")))
    assert g["source"] == "reference run by make_golden.py"
    for rec in g["convert_data"]:
        rows, shape = len(rec["actions"]), tuple(rec["state_shape"])
        agent = ProcessAgent.__new__(ProcessAgent)
        agent.num_actions = rec["num_actions"]
        states = ([np.array(st, dtype=np.float32) for st in rec["states"]] if "states" in rec
                  else [np.full(shape, t / 128.0 - 1.0, np.float32) for t in range(rows + 1)])
        rewards = [float.fromhex(h) for h in rec["rewards_hex"]]
        exps = [Experience(states[t], rec["actions"][t], None, rewards[t], states[t + 1], rec["dones"][t])
                for t in range(rows)]
        got = dict(zip(("x_", "r_", "a_", "x2_", "done_"), agent.convert_data(exps)))
        assert {k: str(v.dtype) for k, v in got.items()} == rec["dtypes"]
        assert {k: list(v.shape) for k, v in got.items()} == rec["shapes"]
        assert got["a_"].tolist() == rec["a_"] and [float(v).hex() for v in got["r_"]] == rec["r_hex"]
        assert [bool(v) for v in got["done_"]] == rec["done_"]
        if "x_" in rec:
            assert got["x_"].tolist() == rec["x_"] and got["x2_"].tolist() == rec["x2_"]
    # uint8 states (STATE_TRANSPORT = 'u8') become the same float32 values k / 128 - 1 (Environment.py:59-60)
    agent = ProcessAgent.__new__(ProcessAgent)
    agent.num_actions = 3
    x_, r_, a_, x2_, done_ = agent.convert_data(_experiences([0.0, 1.0, -1.0]))
    assert x_.dtype == np.float32 and x_.shape == (3, 84, 84, 4) and np.all(x_ == -1.0)


def test_accumulate_rewards_and_select_action_match_reference_recording(cfg, golden_dir):
    """ProcessAgent._accumulate_rewards / select_action of the product, class-level, against the reference's recorded
    outputs (same fixtures the C ABI and the oracle are held to)."""
    import json, os
    from ProcessAgent import ProcessAgent
    g = json.load(open(os.path.join(golden_dir, "returns_fork.json")))
    cfg.RETURN_MODE = 'fork'
    for case in g["cases"]:
        cfg.REWARD_CLIPPING, cfg.DISCOUNTING = case["reward_clipping"], case["discounting"]
        cfg.USE_INTERMEDIATE_REWARD = case["use_intermediate_reward"]
        rewards = [float.fromhex(h) for h in case["rewards_hex"]]
        out = ProcessAgent._accumulate_rewards(_experiences(rewards), case["gamma"],
                                               float.fromhex(case["terminal_reward_hex"]))
        assert [float(e.reward).hex() for e in out] == case["out_hex"] and len(out) == case["rows_out"]
    # the one flag setting under which the reference does not return (recorded exception type)
    cfg.REWARD_CLIPPING, cfg.DISCOUNTING, cfg.USE_INTERMEDIATE_REWARD = False, True, True
    assert g["clip_off_intermediate_on_raises"] == "UnboundLocalError"
    with pytest.raises(UnboundLocalError):
        ProcessAgent._accumulate_rewards(_experiences([1.0, 2.0]), 0.99, 2.0)
    g = json.load(open(os.path.join(golden_dir, "process_agent.json")))
    for case in g["select_action"]:
        p = np.array([float.fromhex(h) for h in case["prediction_f32_hex"]], dtype=np.float32)
        actions = np.arange(case["num_actions"])
        cfg.PLAY_MODE = False
        np.random.seed(case["seed"])
        assert [ProcessAgent.select_action(actions, p) for _ in case["draws"]] == case["draws"]
        np.random.seed(case["seed"])        # float64 copy of the row: the numpy path of select_action
        assert [ProcessAgent.select_action(actions, p.astype(np.float64)) for _ in case["draws"]] == case["draws"]
        cfg.PLAY_MODE = True
        assert ProcessAgent.select_action(actions, p) == case["play_mode_action"]


def test_select_action_draws_equal_numpy_choice(cfg):
    from ProcessAgent import ProcessAgent
    rng = np.random.default_rng(0)
    actions = np.arange(6)
    for trial in range(200):
        z = rng.normal(size=6).astype(np.float32) * 3
        p = np.exp(z - z.max())
        p = (p / p.sum()).astype(np.float32)
        np.random.seed(trial)
        want = [int(np.random.choice(actions, p=p)) for _ in range(5)]
        np.random.seed(trial)
        got = [ProcessAgent.select_action(actions, p) for _ in range(5)]
        assert got == want
    # the C sampler (float32 vectors) and the numpy fallback (anything else) against np.random.choice: other action
    # counts, vectors with exact zeros and with one dominant entry, many draws each
    for n_act in (1, 4, 18, 64):
        actions_n = np.arange(n_act)
        for trial in range(40):
            z = rng.normal(size=n_act) * (1 + trial % 7)
            p64 = np.exp(z - z.max())
            if n_act > 2 and trial % 3 == 0:
                p64[rng.integers(0, n_act, size=n_act // 2)] = 0.0
                p64[0] += 1e-3
            p32 = (p64 / p64.sum()).astype(np.float32)
            for p in (p32, p32.astype(np.float64)):
                np.random.seed(1000 + trial)
                want = [int(np.random.choice(actions_n, p=p / p.sum() if p.dtype == np.float64 else p)) for _ in range(20)] \
                    if p.dtype == np.float32 else None
                if want is None:       # float64 input: choice() insists on an exactly normalised vector; compare the two paths
                    np.random.seed(1000 + trial)
                    want = [ProcessAgent.select_action(actions_n, p32) for _ in range(20)]
                np.random.seed(1000 + trial)
                got = [ProcessAgent.select_action(actions_n, p) for _ in range(20)]
                assert got == want, (n_act, trial, p.dtype)
    cfg.PLAY_MODE = True
    assert ProcessAgent.select_action(actions, np.array([0.1, 0.5, 0.2, 0.1, 0.05, 0.05], np.float32)) == 1


def test_environment_frames_equal_stacked_integer_planes(cfg):
    """The uint32-shift FIFO + Generator.bytes() must be the documented source: planes from
    integers(0, 256, uint8) of PCG64(RANDOM_SEED + id), stacked oldest-first along the last axis."""
    from Environment import Environment
    cfg.SYNTHETIC_EPISODE_LENGTH = 6
    env = Environment(3)
    ref = np.random.Generator(np.random.PCG64(cfg.RANDOM_SEED + 3))
    frames = [ref.integers(0, 256, size=(84, 84), dtype=np.uint8)]
    assert env.current_u8 is None
    for t in range(1, 10):
        reward, done = env.step(0)
        ref.random()
        frames.append(ref.integers(0, 256, size=(84, 84), dtype=np.uint8))
        if t < 3:
            assert env.current_u8 is None
            continue
        want = np.stack(frames[-4:], axis=-1)
        assert env.current_u8.shape == (84, 84, 4) and env.current_u8.dtype == np.uint8
        assert np.array_equal(env.current_u8, want)
        if t > 3:
            assert np.array_equal(env.previous_u8, np.stack(frames[-5:-1], axis=-1))
        assert done == (t >= 6 + 3)


def test_gym_frame_source_fails_loudly_without_gym(monkeypatch):
    """Config.FRAME_SOURCE = 'gym' (the reference's real emulator, Environment.py:41-50) is an optional hook: gym / ALE are
    absent offline, and the hook must say so instead of falling back to synthetic frames."""
    import importlib.util
    if importlib.util.find_spec("gym") is not None:
        pytest.skip("gym is installed")
    from Config import Config
    from Environment import Environment
    monkeypatch.setattr(Config, "FRAME_SOURCE", "gym")
    with pytest.raises(ImportError, match="FRAME_SOURCE = 'gym'"):
        Environment(0)


def test_mixed_launcher_builds_one_engine_per_game():
    import ga3c_amd  # noqa: F401
    import GA3C_mixed as gm
    games = gm.parse_games("Boxing:18:0+1+2+3,Pong:6:4")
    assert games == [("Boxing", 18, [0, 1, 2, 3]), ("Pong", 6, [4])]
    (n0, c0, e0), (n1, c1, e1) = gm.commands(games, ["AGENTS=256"])
    assert "torch.distributed.run" in c0 and "--nproc-per-node" in c0 and e0["HIP_VISIBLE_DEVICES"] == "0,1,2,3"
    assert "NUM_ACTIONS=18" in c0 and "NETWORK_NAME=Boxing" in c0 and "AGENTS=256" in c0
    assert "DEVICE=gpu:4" in c1 and "RESULTS_FILENAME=results_Pong.txt" in c1
    with pytest.raises(ValueError):
        gm.parse_games("Pong:6,Pong:6")
