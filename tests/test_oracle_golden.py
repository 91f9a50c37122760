"""Pins the oracle to what the reference itself produced (plumbing) and to the committed
NN fixtures (own f64 goldens; NN parity is unpinned at the TensorFlow boundary)."""
import json
import os

import numpy as np

import ga3c_oracle as o


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_returns_fork_bit_exact(golden_dir):
    """Every case was RUN through the reference's ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84) by
    tests/golden/make_golden.py: 25 reward vectors x 4 flag settings, compared bit for bit."""
    g = _load(golden_dir, "returns_fork.json")
    assert g["source"] == "reference run by make_golden.py" and "oracle_derived_cases" not in g
    assert len(g["cases"]) == 100
    for case in g["cases"]:
        rewards = [float.fromhex(h) for h in case["rewards_hex"]]
        got = o.accumulate_rewards_fork(rewards, case["gamma"], float.fromhex(case["terminal_reward_hex"]),
                                        discounting=case["discounting"],
                                        use_intermediate_reward=case["use_intermediate_reward"])
        assert len(got) == case["rows_out"] == len(rewards)
        assert [float(v).hex() for v in got] == case["out_hex"], case
    # the survey's own vector is case 0 of the recording
    assert g["cases"][0]["out_hex"][:2] == ["0x1.ebd33d7f3c762p-1", "0x1.f0cb07d0aed99p-1"]


def test_convert_data_equals_reference_recording(golden_dir):
    g = _load(golden_dir, "process_agent.json")
    assert g["source"] == "reference run by make_golden.py"
    for rec in g["convert_data"]:
        rows, shape = len(rec["actions"]), tuple(rec["state_shape"])
        if "states" in rec:
            states = [np.array(s, dtype=np.float32) for s in rec["states"]]
        else:
            states = [np.zeros(shape, np.float32) for _ in range(rows + 1)]
        rewards = [float.fromhex(h) for h in rec["rewards_hex"]]
        x_, r_, a_, x2_, done_ = o.convert_data(states[:-1], rec["actions"], rewards, states[1:], rec["dones"],
                                                rec["num_actions"])
        got = dict(x_=x_, r_=r_, a_=a_, x2_=x2_, done_=done_)
        assert {k: str(v.dtype) for k, v in got.items()} == rec["dtypes"]
        assert {k: list(v.shape) for k, v in got.items()} == rec["shapes"]
        assert a_.tolist() == rec["a_"] and [float(v).hex() for v in r_] == rec["r_hex"]
        assert [bool(v) for v in done_] == rec["done_"]
        if "x_" in rec:
            assert x_.tolist() == rec["x_"] and x2_.tolist() == rec["x2_"]


def test_select_action_equals_reference_draws(golden_dir):
    g = _load(golden_dir, "process_agent.json")
    assert {c["num_actions"] for c in g["select_action"]} == {4, 6, 18}
    for case in g["select_action"]:
        p = np.array([float.fromhex(h) for h in case["prediction_f32_hex"]], dtype=np.float32)
        np.random.seed(case["seed"])
        us = [np.random.random_sample() for _ in case["draws"]]
        assert [o.select_action_index(p, u) for u in us] == case["draws"], case
        assert o.select_action_index(p, 0.5, play_mode=True) == case["play_mode_action"]


def test_returns_is_sequential_product_not_pow():
    # contract is gamma*(gamma*(...)) in f64, which differs from gamma**k in the last bit for some k
    out = o.accumulate_rewards_fork([0.0] * 40, 0.99, 1.0)
    seq = 1.0
    for k in range(1, 40):
        seq = 0.99 * seq
        assert out[39 - k] == seq
    assert any(out[39 - k] != 0.99 ** k for k in range(1, 40))


def test_returns_edge_cases():
    assert o.accumulate_rewards_fork([], 0.99, 1.0) == []
    assert o.accumulate_rewards_fork([2.5], 0.99, 2.5) == [2.5]         # single row keeps raw reward
    assert o.accumulate_rewards_fork([7.0, -1.0], 0.5, -1.0) == [-0.5, -1.0]


def test_predictor_batching_matches_reference_trace(golden_dir):
    g = _load(golden_dir, "batcher_traces.json")
    for key in ("predictor_128", "predictor_32"):
        t = g[key]
        assert o.predictor_batches(t["n_requests"], t["batch_max"]) == t["batch_sizes"]
        # routing: request i of agent a gets v = sum(state_i); regenerate the states and compare
        rng = np.random.default_rng(t["seed"])
        states = rng.integers(0, 256, size=(t["n_requests"], t["state_dim"])).astype(np.float32)
        for agent in range(t["n_agents"]):
            mine = [float(states[i].sum()) for i in range(t["n_requests"]) if i % t["n_agents"] == agent]
            assert mine == t["value_routed_per_agent"][agent]


def test_trainer_batching_matches_reference_trace(golden_dir):
    g = _load(golden_dir, "batcher_traces.json")
    for key in ("trainer_min0", "trainer_min8", "trainer_min127"):
        t = g[key]
        assert o.trainer_batches(t["rollout_rows"], t["min_batch"]) == [c["rows"] for c in t["calls"]]


def test_nn_small_fixture_reproduces(golden_dir):
    z = np.load(os.path.join(golden_dir, "nn_small.npz"))
    for num_actions in (6, 4, 18):
        t = "A%d_" % num_actions
        x = z[t + "x_u8"].astype(np.float64) / 128.0 - 1.0
        params = o.init_params(num_actions)
        f = o.forward(params, x)
        assert np.allclose(f["p"], z[t + "p"], rtol=0, atol=1e-12)
        assert np.allclose(f["v"], z[t + "v"], rtol=0, atol=1e-12)
        a = np.eye(num_actions)[z[t + "actions"]]
        losses, g = o.loss_and_grads(params, x, z[t + "y_r"], a, beta=0.01)
        assert np.allclose([losses["cost_p_1_agg"], losses["cost_p_2_agg"], losses["cost_v"]],
                           z[t + "losses"], rtol=1e-12)
        assert np.allclose(g["conv11/w"], z[t + "grad.conv11.w"], rtol=0, atol=1e-12)


def test_anneal_schedule():
    assert o.anneal(3e-4, 3e-4, 40000, 10) == 3e-4
    assert o.anneal(1.0, 0.0, 10, 5) == 0.5
    assert o.anneal(1.0, 0.0, 10, 50) == 1.0 - 0.1 * 9      # clamped at ANNEAL-1 (Server.py:173)
