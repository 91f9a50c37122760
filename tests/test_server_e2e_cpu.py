"""End-to-end plumbing on CPU: real agent processes (forkserver), the shared-memory transport, the
predictor / trainer threads, the stats process and the Server loop -- with a stand-in model so that no
GPU is needed.  Checks the rollout shapes the reference produces (first rollout TIME_MAX+1 rows, later
ones TIME_MAX + 1 carried) and the results.txt wire format."""
import os
import re

import numpy as np
import pytest


class _StandInModel:
    """Uniform policy, zero value; records what the batching threads hand over."""
    def __init__(self, n_act):
        self.n_act = n_act
        self.learning_rate = self.beta = 0.0
        self.pred_batches, self.train_rows, self.train_dtypes = [], [], set()

    def predict_p_and_v(self, x):
        self.pred_batches.append(x.shape)
        b = x.shape[0]
        return np.full((b, self.n_act), 1.0 / self.n_act, np.float32), np.zeros(b, np.float32)

    def train(self, x, y_r, a, x2, done, tid):
        assert x.shape[1:] == (84, 84, 4) and a.shape == (x.shape[0], self.n_act) and y_r.shape == (x.shape[0],)
        assert np.all(a.sum(axis=1) == 1.0)
        self.train_rows.append(x.shape[0])
        self.train_dtypes.add(str(x.dtype))

    def save(self, episode):
        pass

    def log(self, *a, **k):
        pass


@pytest.mark.timeout(120)
def test_server_runs_agents_predictor_trainer_stats(tmp_path, monkeypatch):
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    saved = {k: getattr(Config, k) for k in ("AGENTS", "PREDICTORS", "TRAINERS", "SYNTHETIC_EPISODE_LENGTH", "TIME_MAX",
                                             "DYNAMIC_SETTINGS", "SAVE_MODELS", "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS")}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 3, 1, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 23, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, False
    Config.TRAINING_MIN_BATCH_SIZE, Config.NUM_ACTIONS = 0, 6
    try:
        from Server import Server
        model = _StandInModel(6)
        srv = Server(model=model, max_agents=8)
        srv.main(max_seconds=6)
        assert srv.predictions_served > 50
        assert model.train_rows and model.train_dtypes == {"uint8"}
        # 23-step episodes cut at TIME_MAX=5: rollouts of 6 rows (first: TIME_MAX+1; later: 1 carried + 5), last one shorter
        assert max(model.train_rows) == 6 and set(model.train_rows) <= {1, 2, 3, 4, 5, 6}
        assert all(s[1:] == (84, 84, 4) for s in model.pred_batches)
        assert srv.training_step == len(model.train_rows) == srv.stats.training_count.value
        lines = open("results.txt").read().strip().splitlines()
        assert lines and all(re.match(r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d, -?\d+, \d+$", ln) for ln in lines)
        # frame accounting of ProcessAgent.py:174: each rollout contributes len(r_)+1 and carries one row over:
        # 23 steps -> rollouts of 6,6,6,6,3 rows -> 7+7+7+7+4 = 32
        lengths = [int(ln.split(", ")[2]) for ln in lines]
        assert set(lengths) == {32}
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)


def test_status_line_format_is_the_references():
    import ga3c_amd  # noqa: F401
    from ProcessStats import ProcessStats
    line = ProcessStats.status_line(35, 30, -20.0, -20.4, 899, 900, 186, 2, 2, 32, 0)
    assert line == ("[Time:       35] [Episode:       30 Score:   -20.0000] [RScore:   -20.4000 RPPS:   899] "
                    "[PPS:   900 TPS:   186] [NT:  2 NP:  2 NA: 32][RSize:        0]")


class _DeviceFrontEndStandIn:
    """Stand-in for the HIP Network in FRONTEND = 'device' mode: frames_* entry points backed by the oracle's front-end,
    so that the whole raw-frame protocol (agent -> slot -> predictor -> queue; rollouts as (agent, plane)) runs on CPU."""

    def __init__(self, n_act):
        import frame_frontend as ff
        self.ff, self.n_act = ff, n_act
        self.learning_rate = self.beta = 0.0
        self.transport = None
        self.queues, self.pushed, self.has_state = {}, {}, set()
        self.problems, self.train_rows, self.pred_sizes, self.resets = [], [], [], 0
        self.history = None

    def register_transport(self, transport):
        self.transport = transport

    def unregister_transport(self):
        self.transport = None

    def frames_config(self, max_agents, height, width, channels, history=0):
        self.shape, self.history = (height, width, channels), history

    def push_frame_offsets(self, offsets, agents, reset=None):
        n = int(np.prod(self.shape))
        for i, (off, a) in enumerate(zip(offsets, agents)):
            a = int(a)
            q = self.queues.setdefault(a, self.ff.FrameQueue())
            if reset is not None and reset[i]:
                q.clear()
                self.resets += 1
            frame = self.transport._raw[off: off + n].reshape(self.shape)
            q.push(frame[:84, :84, 0])          # any plane will do here: the arithmetic is held to the oracle on the GPU
            seq = self.pushed.get(a, 0)
            self.pushed[a] = seq + 1
            if q.state_u8() is not None:
                self.has_state.add((a, seq))

    def predict_frames(self, agents):
        for a in agents:
            if self.queues[int(a)].state_u8() is None:
                self.problems.append("prediction asked for agent %d before its queue was full" % a)
        self.pred_sizes.append(len(agents))
        b = len(agents)
        return np.full((b, self.n_act), 1.0 / self.n_act, np.float32), np.zeros(b, np.float32)

    def train_frames(self, agents, seqs, y_r, a):
        for ag, s in zip(agents, seqs):
            if (int(ag), int(s)) not in self.has_state:
                self.problems.append("row (%d, %d) names no state" % (ag, s))
            if self.pushed[int(ag)] - (int(s) - 3) > self.history:
                self.problems.append("row (%d, %d) left the history" % (ag, s))
        assert a.shape == (len(agents), self.n_act) and y_r.shape == (len(agents),)
        self.train_rows.append(len(agents))

    def save(self, episode):
        pass

    def log(self, *a, **k):
        pass


@pytest.mark.timeout(120)
def test_device_front_end_protocol_on_cpu(tmp_path, monkeypatch):
    """FRAME_SOURCE = 'rgb', FRONTEND = 'device': agents ship raw frames, the predictor pushes them in order and predicts
    only on full queues, rollouts arrive as (agent, plane sequence) rows that name existing states."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    keys = ("AGENTS", "PREDICTORS", "TRAINERS", "SYNTHETIC_EPISODE_LENGTH", "TIME_MAX", "DYNAMIC_SETTINGS", "SAVE_MODELS",
            "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS", "FRAME_SOURCE", "FRONTEND")
    saved = {k: getattr(Config, k) for k in keys}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 3, 2, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 23, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, False
    Config.TRAINING_MIN_BATCH_SIZE, Config.NUM_ACTIONS = 11, 6
    Config.FRAME_SOURCE, Config.FRONTEND = 'rgb', 'device'
    try:
        from Server import Server
        model = _DeviceFrontEndStandIn(6)
        srv = Server(model=model, max_agents=8)
        assert srv.device_frontend and srv.transport.row_bytes == 16 and srv.transport.state_bytes >= 210 * 160 * 3
        # every rollout in flight + the one being filled, plus the rows the trainers stage after giving the slots back
        assert model.history == (Config.MAX_QUEUE_SIZE + 2) * 6 + 8 + 2 * (11 + 6)
        srv.main(max_seconds=6)
        assert model.problems == []
        assert srv.predictions_served > 50 and sum(model.pred_sizes) == srv.predictions_served
        assert model.train_rows and min(model.train_rows) > 11 and srv.training_step == len(model.train_rows)
        assert model.resets >= 3                                    # every episode start clears its queue
        lengths = [int(ln.split(", ")[2]) for ln in open("results.txt").read().strip().splitlines()]
        assert lengths and set(lengths) == {32}                     # same frame accounting as the state-shipping path
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)


def test_rgb_source_with_host_front_end_feeds_reference_planes(monkeypatch):
    """FRAME_SOURCE = 'rgb', FRONTEND = 'host': Environment runs the reference's _preprocess in the agent; its states must
    be the oracle's planes of the same emulator frames, stacked oldest first."""
    import ga3c_amd  # noqa: F401
    import frame_frontend as ff
    from Config import Config
    from Environment import Environment
    monkeypatch.setattr(Config, "FRAME_SOURCE", "rgb")
    monkeypatch.setattr(Config, "FRONTEND", "host")
    monkeypatch.setattr(Config, "SYNTHETIC_EPISODE_LENGTH", 5)
    env = Environment(1)
    planes = [ff.preprocess_u8(env.frame)]
    assert env.frame.shape == (210, 160, 3) and env.current_u8 is None
    for t in range(1, 7):
        env.step(0)
        planes.append(ff.preprocess_u8(env.frame))
        if t >= 3:
            assert np.array_equal(env.current_u8, np.stack(planes[-4:], axis=-1))
    raw = Environment(1)
    monkeypatch.setattr(Config, "FRONTEND", "device")
    dev = Environment(1)
    assert dev.on_device and np.array_equal(dev.frame, raw.frame) and dev.frames_queued == 1
    dev.step(0)
    assert dev.current_u8 is None and dev.frames_queued == 2


class _ZeroCopyStandIn(_StandInModel):
    """A stand-in with the zero-copy entry points, so that trainers keep rollout slots while they fill a batch."""

    def register_transport(self, transport):
        self.transport = transport
        self.offset_batches = []

    def unregister_transport(self):
        self.transport = None

    def predict_offsets(self, offsets):
        b = len(offsets)
        return np.full((b, self.n_act), 1.0 / self.n_act, np.float32), np.zeros(b, np.float32)

    def train_offsets(self, offsets, y_r, a):
        self.offset_batches.append(len(offsets))


@pytest.mark.timeout(120)
def test_trainers_cannot_starve_the_agents_of_rollout_slots(tmp_path, monkeypatch):
    """TRAINING_MIN_BATCH_SIZE larger than what the rollout slots hold, two trainers, zero-copy intake: without the
    trainers' turn-taking and spill rule every slot ends up held by a trainer waiting for more rows and the engine
    stops dead (found on the GPU box with batch 512, MAX_QUEUE_SIZE 100).  Batches must keep coming, whole."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=4, PREDICTORS=1, TRAINERS=2, SYNTHETIC_EPISODE_LENGTH=50, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=59, NUM_ACTIONS=6, ROLLOUT_SLOTS=6, ZERO_COPY=True).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    model = _ZeroCopyStandIn(6)
    srv = Server(model=model, max_agents=8)
    assert srv.zero_copy and srv.transport.train_slots == 6            # 6 slots x 6 rows < 60 rows per batch
    trainers = []
    real_add = srv.add_trainer
    monkeypatch.setattr(srv, "add_trainer", lambda: (real_add(), trainers.append(srv.trainers[-1])))
    srv.main(max_seconds=6)
    batches = model.train_rows + model.offset_batches
    assert len(batches) >= 5 and min(batches) >= 60 and srv.training_step == len(batches)
    assert model.train_rows and model.train_dtypes == {"uint8"}         # spilled batches went through the host path
    spills = sum(t.spills for t in trainers)                           # a batch spilled at shutdown may never be trained
    assert len(model.train_rows) <= spills <= len(model.train_rows) + 2


class _StateCacheStandIn(_ZeroCopyStandIn):
    """A stand-in with the state-cache entry points (Config.STATE_CACHE): its `begin` keeps the bytes of every state it is
    handed under the row's name, its train_frames looks the named rows up -- and fails on a name it never saw or one that has
    been overwritten in its ring of `depth` states per agent."""

    def __init__(self, n_act):
        super().__init__(n_act)
        import ctypes as C
        self.kept, self.depth, self.trained, self.errors, self.held = {}, None, [], [], {}
        self.lock = __import__("threading").Lock()

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.c_int32, C.c_int32,
                     C.POINTER(C.c_int32))
        def begin(net, offsets, agents, seqs, batch, u8, ticket):
            raw = self.transport._raw
            with self.lock:
                for i in range(batch):
                    ag, sq, off = int(agents[i]), int(seqs[i]), int(offsets[i])
                    self.kept[(ag, sq % self.depth)] = (sq, bytes(raw[off:off + 64]))
                self.held[id(ticket)] = batch
            ticket[0] = 0
            return 0

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))
        def end(net, ticket, batch, p, v):
            for i in range(batch):
                v[i] = 0.0
                for a in range(self.n_act):
                    p[i * self.n_act + a] = 1.0 / self.n_act
            return 0
        self._begin, self._end, self._C = begin, end, C

    def state_cache_config(self, max_agents, depth):
        self.depth = int(depth)

    def gather_entry(self):
        return None, None, 1

    def gather_entries_pipelined(self):
        return None

    def gather_entries_pipelined_cached(self):
        C = self._C
        return C.cast(self._begin, C.c_void_p).value, C.cast(self._end, C.c_void_p).value, None, 1

    def train_frames(self, agents, seqs, y_r, a):
        with self.lock:
            for ag, sq in zip(agents.tolist(), seqs.tolist()):
                got = self.kept.get((ag, sq % self.depth))
                if got is None or got[0] != sq:
                    self.errors.append((ag, sq, None if got is None else got[0]))
            self.trained.append(len(agents))


@pytest.mark.timeout(120)
def test_state_cache_protocol_on_cpu(tmp_path, monkeypatch):
    """Config.STATE_CACHE with a stand-in engine: the agents name the states of their experiences by request number, the native
    predictor loop hands every row's name to the engine, the native batch assembly collects names instead of states, and every
    name a trainer asks for is one the engine was given -- and still holds in a ring of the depth the Server computed."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=5, PREDICTORS=2, TRAINERS=2, SYNTHETIC_EPISODE_LENGTH=40, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=17, NUM_ACTIONS=6, ZERO_COPY=True, STATE_CACHE=True,
                     PREDICTION_BATCH_SIZE=32).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    model = _StateCacheStandIn(6)
    srv = Server(model=model, max_agents=8)
    assert srv.state_cache and srv.zero_copy and model.depth and srv.transport.row_bytes == 16
    srv.main(max_seconds=5)
    monkeypatch.setattr(Config, "STATE_CACHE_ACTIVE", False)
    assert len(model.trained) >= 5 and min(model.trained) >= 18 and srv.training_step == len(model.trained)
    assert not model.errors, model.errors[:5]
    assert len(model.kept) > 50 and not model.train_rows and not model.offset_batches    # no state ever travelled in a rollout


@pytest.mark.timeout(60)
def test_a_run_that_ends_at_once_shuts_down_cleanly(tmp_path, monkeypatch):
    """EPISODES already reached (e.g. _play.sh on a checkpoint from a later episode): the main loop ends while the
    adjustment thread is still starting workers; shutdown must wait for it instead of joining unstarted threads."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=3, PREDICTORS=2, TRAINERS=2, DYNAMIC_SETTINGS=False, SAVE_MODELS=False, EPISODES=0,
                     NUM_ACTIONS=6).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    srv = Server(model=_StandInModel(6), max_agents=8)
    srv.main()
    assert not srv.agents and not srv.predictors and not srv.trainers


@pytest.mark.timeout(180)
def test_agent_ids_are_recycled_and_counts_stay_real(tmp_path, monkeypatch):
    """with DYNAMIC_SETTINGS every +1 of the random walk used to consume an agent id for good.  Here: far
    more add / remove cycles than there are slots; every add must start a live agent, ids stay inside the transport,
    and the dynamic adjustment's targets follow what is really running when an add is refused."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=2, PREDICTORS=1, TRAINERS=1, SYNTHETIC_EPISODE_LENGTH=15, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=0, NUM_ACTIONS=6, PRINT_STATS_FREQUENCY=10 ** 9).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    import threading
    model = _StandInModel(6)
    srv = Server(model=model, max_agents=3)
    th = threading.Thread(target=srv.main, kwargs=dict(max_seconds=8), daemon=True)
    th.start()
    import time
    t0 = time.time()
    while len(srv.agents) < 2 and time.time() - t0 < 20:
        time.sleep(0.05)
    seen = set()
    for cycle in range(7):                                   # 7 removals + 7 adds on 3 slots
        srv.remove_agent()
        assert srv.add_agent() is True
        seen.add(srv.agents[-1].id)
        assert len(srv.agents) == 2 and all(a.is_alive() for a in srv.agents)
        assert len({a.id for a in srv.agents}) == 2 and all(0 <= a.id < 3 for a in srv.agents)
    assert seen <= {0, 1, 2}
    assert srv.add_agent() is True and srv.add_agent() is False           # third slot, then none left
    da = srv.dynamic_adjustment
    da.agent_count = 5                                       # the walk asks for more than fits
    da.enable_disable_components()
    assert da.agent_count == len(srv.agents) == 3
    served = srv.predictions_served
    time.sleep(0.5)
    assert srv.predictions_served > served                   # the recycled agents are being served
    th.join(60)
    assert not th.is_alive() and srv.failure is None


class _FailingModel(_StandInModel):
    def train(self, x, y_r, a, x2, done, tid):
        raise RuntimeError("device lost")


@pytest.mark.timeout(120)
def test_server_stops_and_raises_when_a_worker_thread_dies(tmp_path, monkeypatch):
    """A trainer (or predictor) thread that dies used to leave the agents waiting forever; the server now stops and main()
    re-raises, so `python GA3C.py` ends with a non-zero status."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=2, PREDICTORS=1, TRAINERS=1, SYNTHETIC_EPISODE_LENGTH=15, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=0, NUM_ACTIONS=6, PRINT_STATS_FREQUENCY=10 ** 9).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    srv = Server(model=_FailingModel(6), max_agents=4)
    with pytest.raises(RuntimeError, match="ThreadTrainer 0 died"):
        srv.main(max_seconds=60)
    assert not srv.agents and not srv.trainers and not srv.predictors
