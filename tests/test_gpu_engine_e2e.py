"""-m gpu: the whole actor-learner engine on the real HIP Network -- agent processes, shared-memory
transport, predictor and trainer threads -- for a few seconds; then the weights must have moved and
every prediction must be a probability vector."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(180)
@pytest.mark.filterwarnings("error::pytest.PytestUnhandledThreadExceptionWarning")
@pytest.mark.parametrize("zero_copy,state_cache", [(True, True), (True, False), (False, False)])
def test_engine_trains_on_gpu(tmp_path, monkeypatch, zero_copy, state_cache):
    """state_cache: rollouts name their states (agent, request number) and the engine trains on the copies its
    predictions kept in HBM (Config.STATE_CACHE, the default); off: the rollouts carry the states, read out of the
    transport by the GPU (zero_copy) or copied by the trainer threads."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(Config, "ZERO_COPY", zero_copy)
    monkeypatch.setattr(Config, "STATE_CACHE", state_cache)
    saved = {k: getattr(Config, k) for k in ("AGENTS", "PREDICTORS", "TRAINERS", "SYNTHETIC_EPISODE_LENGTH", "TIME_MAX",
                                             "DYNAMIC_SETTINGS", "SAVE_MODELS", "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS",
                                             "PREDICTION_BATCH_SIZE")}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 6, 2, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 40, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, True
    Config.TRAINING_MIN_BATCH_SIZE, Config.NUM_ACTIONS, Config.PREDICTION_BATCH_SIZE = 11, 6, 32
    try:
        from Server import Server
        srv = Server(max_agents=8)
        assert srv.zero_copy == zero_copy and srv.state_cache == state_cache
        before = srv.model.get_arena(0)
        srv.main(max_seconds=5)
        after = srv.model.get_arena(0)
        assert srv.predictions_served > 100 and srv.training_step > 5
        assert srv.model.get_global_step() == srv.training_step
        assert np.all(np.isfinite(after)) and np.max(np.abs(after - before)) > 1e-5
        # checkpoint round trip through the reference's naming scheme (checkpoints/<name>_%08d)
        srv.model.save(7)
        srv.model.set_arena(0, np.zeros_like(after))
        Config.LOAD_EPISODE = 7          # the run itself may have saved later episodes (SAVE_FREQUENCY)
        assert srv.model.load() == 7
        Config.LOAD_EPISODE = 0
        assert np.array_equal(srv.model.get_arena(0), after)
        p, v = srv.model.predict_p_and_v(np.zeros((3, 84, 84, 4), np.float32))
        assert np.allclose(p.sum(axis=1), 1.0, atol=1e-5)
        srv.model.close()
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)


def test_gather_from_registered_transport_is_bit_identical_to_host_path():
    """ga3c_net_predict_gather / train_gather read the states out of the pinned shm segment themselves; the
    result must equal feeding the same bytes through the host-buffer entry points."""
    import ga3c_amd  # noqa: F401
    import Transport as tp
    from NetworkVP import Network
    t = tp.Transport.create(tp.unique_name("t_zc"), 40, 6, 84 * 84 * 4, 8, 6)
    net = Network("gpu:0", "zc", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    try:
        net.register_transport(t)
        rng = np.random.default_rng(12)
        frames = rng.integers(0, 256, size=(40, 84 * 84 * 4), dtype=np.uint8)
        t.agent_states[:] = frames
        ids = rng.permutation(40)[:33].astype(np.uint32)
        p1, v1 = net.predict_offsets(t.state_offsets(ids))
        p2, v2 = net.predict_p_and_v(frames[ids].reshape(-1, 84, 84, 4))
        assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
        # the split call of the pipelined predictor loop (begin: enqueue; end: wait + copy out): the same bits, and an end
        # without a begin is refused instead of unlocking a lane nobody holds
        import _native as nat
        offs64 = np.ascontiguousarray(t.state_offsets(ids), dtype=np.int64)
        ticket = nat.C.c_int32(-1)
        nat.check(net._lib.ga3c_net_predict_gather_begin(net._h, nat.ptr(offs64, nat.i64p), ids.size, 1, nat.C.byref(ticket)), "begin")
        p3, v3 = np.empty_like(p1), np.empty_like(v1)
        nat.check(net._lib.ga3c_net_predict_gather_end(net._h, ticket.value, ids.size, nat.ptr(p3), nat.ptr(v3)), "end")
        assert np.array_equal(p1, p3) and np.array_equal(v1, v3)
        assert net._lib.ga3c_net_predict_gather_end(net._h, ticket.value, ids.size, nat.ptr(p3), nat.ptr(v3)) < 0
        # training rows out of two rollout slots
        offs, xs = [], []
        for slot, rows in ((3, 6), (5, 4)):
            states, _, _ = t.rollout_views(slot)
            states[:rows] = rng.integers(0, 256, size=(rows, 84 * 84 * 4), dtype=np.uint8)
            offs.append(t.rollout_row_offsets(slot, rows))
            xs.append(states[:rows].copy())
        offs, xs = np.concatenate(offs), np.concatenate(xs).reshape(-1, 84, 84, 4)
        y = rng.uniform(-1, 1, 10)
        a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, 10)]
        theta0 = net.get_arena(0)
        net.learning_rate, net.beta = 3e-4, 0.01
        # Network.log on rows still lying in the transport: the same evaluation as on the copied-out states
        ev_rows = net.evaluate(None, y, a, offsets=offs)
        ev_host = net.evaluate(xs, y, a)
        assert all(np.array_equal(p, q) for p, q in zip(ev_rows, ev_host)) and np.array_equal(net.get_arena(0), theta0)
        net.train_offsets(offs, y, a)
        got = net.get_arena(0)
        net.set_arena(0, theta0)
        net.set_arena(1, np.ones_like(theta0))
        net.train(xs, y, a)
        assert np.array_equal(got, net.get_arena(0))
        with pytest.raises(RuntimeError):
            net.predict_offsets(np.array([t.nbytes], dtype=np.int64))       # outside the registered segment
    finally:
        net.close()
        t.shutdown()
        t.close()


@pytest.mark.parametrize("switch", ["GA3C_OFFSETS_IN_ARGS", "GA3C_STOP_EVENTS"])
def test_alternate_launch_paths_of_a_gathered_step_give_the_same_bits(monkeypatch, switch):
    """The offsets of a scattered batch travel in the kernel arguments (GA3C_OFFSETS_IN_ARGS=0: read from pinned memory) and
    a step's completion event is the stop event of its last launch (GA3C_STOP_EVENTS=0: a record of its own): the
    fallbacks stay in the library (hipGraph replays use the first), so they are held to the default path bit for bit --
    predictions on transport rows from two threads, and a train call on rollout rows."""
    import threading
    import ga3c_amd  # noqa: F401
    import Transport as tp
    from NetworkVP import Network
    t = tp.Transport.create(tp.unique_name("t_alt"), 48, 6, 84 * 84 * 4, 8, 6)
    rng = np.random.default_rng(21)
    t.agent_states[:] = rng.integers(0, 256, size=(48, 84 * 84 * 4), dtype=np.uint8)
    states, _, _ = t.rollout_views(2)
    states[:6] = rng.integers(0, 256, size=(6, 84 * 84 * 4), dtype=np.uint8)
    y = rng.uniform(-1, 1, 6)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, 6)]
    outs = []
    try:
        for flag in ("1", "0"):
            monkeypatch.setenv(switch, flag)
            net = Network("gpu:0", "alt" + flag, 6, (84, 84, 4), max_batch=64, predict_lanes=2)
            monkeypatch.delenv(switch)
            try:
                net.register_transport(t)
                net.learning_rate, net.beta = 3e-4, 0.01
                got = {}

                def work(k):
                    ids = np.arange(k, 48, 2, dtype=np.uint32)
                    res = None
                    for _ in range(20):
                        res = net.predict_offsets(t.state_offsets(ids))
                    got[k] = res
                th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                net.train_offsets(t.rollout_row_offsets(2, 6), y, a)
                outs.append((got[0][0], got[0][1], got[1][0], got[1][1], net.get_arena(0)))
            finally:
                net.close()
        assert all(np.array_equal(p, q) for p, q in zip(outs[0], outs[1]))
    finally:
        t.shutdown()
        t.close()


def test_state_cache_keeps_what_predictions_read_and_trains_on_it_by_name():
    """ga3c_net_predict_gather_begin_cached stores the uint8 state of every row it reads in HBM (ring of `depth` per agent,
    slot = request number % depth); ga3c_net_train_cached / evaluate_cached on rows NAMED (agent, request number) must be
    ga3c_net_train_gather / evaluate on the same bytes, bit for bit; a name that was never stored (a skipped request number:
    the slot's tag says another request), has been overwritten, or is within 4 requests of falling out of its agent's window
    is refused with GA3C_ELOST (nat.StateLost) and nothing is trained; ga3c_net_stats reports the cache's bytes and the rows
    refused."""
    import ga3c_amd  # noqa: F401
    import _native as nat
    import Transport as tp
    from NetworkVP import Network
    C = nat.C
    t = tp.Transport.create(tp.unique_name("t_cache"), 40, 6, 84 * 84 * 4, 8, 8)
    net = Network("gpu:0", "cache", 6, (84, 84, 4), max_batch=64, predict_lanes=2)
    ref = Network("gpu:0", "cache_ref", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    depth = 8
    try:
        net.register_transport(t)                          # (ref gets the same bytes through the host-buffer entry points:
        for n in (net, ref):                                # bit-identical to the zero-copy ones, tested above)
            n.learning_rate, n.beta = 3e-4, 0.01
        net.state_cache_config(40, depth)
        rng = np.random.default_rng(31)
        kept = {}                                           # (agent, request number) -> the bytes that were predicted on

        def predict_named(ids, seqs):
            offs = np.ascontiguousarray(t.state_offsets(ids), dtype=np.int64)
            ag = np.ascontiguousarray(ids, dtype=np.int32)
            sq = np.ascontiguousarray(seqs, dtype=np.int64)
            ticket = C.c_int32(-1)
            nat.check(net._lib.ga3c_net_predict_gather_begin_cached(net._h, nat.ptr(offs, nat.i64p), nat.ptr(ag, nat.i32p),
                                                                    nat.ptr(sq, nat.i64p), ids.size, 1, C.byref(ticket)), "begin_cached")
            p, v = np.empty((ids.size, 6), np.float32), np.empty(ids.size, np.float32)
            nat.check(net._lib.ga3c_net_predict_gather_end(net._h, ticket.value, ids.size, nat.ptr(p), nat.ptr(v)), "end")
            return p, v

        for step in range(3):                               # three requests per agent, ragged batches
            t.agent_states[:] = rng.integers(0, 256, size=(40, 84 * 84 * 4), dtype=np.uint8)
            for ids in (np.arange(0, 17, dtype=np.uint32), np.arange(17, 40, dtype=np.uint32)):
                seqs = 10 + step + ids.astype(np.int64) * 3
                p, v = predict_named(ids, seqs)
                want = ref.predict_p_and_v(t.agent_states[ids].reshape(-1, 84, 84, 4))
                assert np.array_equal(p, want[0]) and np.array_equal(v, want[1])
                for i, s_ in zip(ids, seqs):
                    kept[(int(i), int(s_))] = t.agent_states[int(i)].copy()
        # a train batch of 26 named rows against the same bytes put into rollout slots of the transport
        names = [(a_, 10 + st + a_ * 3) for a_ in (3, 39, 17, 0, 21) for st in (0, 1, 2)] + [(a_, 12 + a_ * 3) for a_ in range(5, 16)]
        agents = np.array([n_[0] for n_ in names], np.int32)
        seqs = np.array([n_[1] for n_ in names], np.int64)
        rows = len(names)
        y = rng.uniform(-1, 1, rows)
        a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, rows)]
        xs = np.stack([kept[key] for key in names]).reshape(-1, 84, 84, 4)
        ev_named = net.evaluate(None, y, a, frames=(agents, seqs))
        ev_rows = ref.evaluate(xs, y, a)
        assert all(np.array_equal(p_, q_) for p_, q_ in zip(ev_named, ev_rows))
        net.train_frames(agents, seqs, y, a)
        ref.train(xs, y, a)
        assert np.array_equal(net.get_arena(0), ref.get_arena(0)) and np.array_equal(net.get_arena(1), ref.get_arena(1))
        # names the cache does not hold: refused as LOST, and the weights stay as they are
        assert net.stats()["state_cache_bytes"] == 40 * depth * 84 * 84 * 4 and net.stats()["state_cache_lost_rows"] == 0
        before = net.get_arena(0)
        with pytest.raises(nat.StateLost):
            net.train_frames(np.array([3], np.int32), np.array([22], np.int64), y[:1], a[:1])        # newer than the newest stored (21)
        with pytest.raises(nat.StateLost):
            net.train_frames(np.array([3, 3], np.int32), np.array([19, 17], np.int64), y[:2], a[:2])  # 17 was never stored: the
        #                                                                                             slot's tag is -1, not 17
        with pytest.raises(RuntimeError):
            net.train_frames(np.array([40], np.int32), np.array([10], np.int64), y[:1], a[:1])       # no such agent
        assert net.stats()["state_cache_lost_rows"] == 3 and np.array_equal(net.get_arena(0), before)
        # agent 7 goes on predicting: request 31 of it is still held while fewer than depth - 4 newer ones exist, then lost --
        # BEFORE its slot (31 % 8) is really overwritten by request 39
        one = np.array([7], np.uint32)                      # (its requests so far: 31, 32, 33)
        for sq, expect_held in ((34, True), (35, False)):
            predict_named(one, np.array([sq], np.int64))
            if expect_held:
                net.evaluate(None, y[:1], a[:1], frames=(np.array([7], np.int32), np.array([31], np.int64)))
            else:
                with pytest.raises(nat.StateLost):
                    net.evaluate(None, y[:1], a[:1], frames=(np.array([7], np.int32), np.array([31], np.int64)))
        t.agent_states[:] = rng.integers(0, 256, size=(40, 84 * 84 * 4), dtype=np.uint8)
        # a batch beyond the 192 offsets that travel in the kernel arguments (they are then read out of the pinned array)
        big = Network("gpu:0", "cache_big", 6, (84, 84, 4), max_batch=256, predict_lanes=1)
        bref = Network("gpu:0", "cache_bigref", 6, (84, 84, 4), max_batch=256, predict_lanes=1)
        net.unregister_transport()                          # (one registration of the segment at a time)
        try:
            big.register_transport(t)
            big.state_cache_config(40, depth)
            for n in (big, bref):
                n.learning_rate, n.beta = 3e-4, 0.01
            ids40 = np.arange(40, dtype=np.uint32)
            offs40 = np.ascontiguousarray(t.state_offsets(ids40), dtype=np.int64)
            tk = C.c_int32()
            p40, v40 = np.empty((40, 6), np.float32), np.empty(40, np.float32)
            nat.check(big._lib.ga3c_net_predict_gather_begin_cached(big._h, nat.ptr(offs40, nat.i64p), nat.ptr(ids40.astype(np.int32), nat.i32p),
                                                                    nat.ptr(np.full(40, 7, np.int64), nat.i64p), 40, 1, C.byref(tk)), "begin_cached")
            nat.check(big._lib.ga3c_net_predict_gather_end(big._h, tk.value, 40, nat.ptr(p40), nat.ptr(v40)), "end")
            ag200 = (np.arange(200) % 40).astype(np.int32)
            y200 = rng.uniform(-1, 1, 200)
            a200 = np.eye(6, dtype=np.float32)[rng.integers(0, 6, 200)]
            big.train_frames(ag200, np.full(200, 7, np.int64), y200, a200)
            bref.train(t.agent_states[ag200].reshape(-1, 84, 84, 4), y200, a200)
            assert np.array_equal(big.get_arena(0), bref.get_arena(0))
            # ... and a PREDICTION batch beyond the fused conv stack's 128 rows: the gathered batch is filed into the cache by
            # a copy kernel (here 160 rows: every agent four times, under four request numbers)
            t.agent_states[:] = rng.integers(0, 256, size=(40, 84 * 84 * 4), dtype=np.uint8)
            ids160 = (np.arange(160) % 40).astype(np.uint32)
            sq160 = (8 + np.arange(160) // 40).astype(np.int64)
            offs160 = np.ascontiguousarray(t.state_offsets(ids160), dtype=np.int64)
            p160, v160 = np.empty((160, 6), np.float32), np.empty(160, np.float32)
            nat.check(big._lib.ga3c_net_predict_gather_begin_cached(big._h, nat.ptr(offs160, nat.i64p), nat.ptr(ids160.astype(np.int32), nat.i32p),
                                                                    nat.ptr(sq160, nat.i64p), 160, 1, C.byref(tk)), "begin_cached")
            nat.check(big._lib.ga3c_net_predict_gather_end(big._h, tk.value, 160, nat.ptr(p160), nat.ptr(v160)), "end")
            want160 = bref.predict_p_and_v(t.agent_states[ids160].reshape(-1, 84, 84, 4))
            assert np.array_equal(p160, want160[0]) and np.array_equal(v160, want160[1])
            big.train_frames(ids160[:100].astype(np.int32), np.full(100, 10, np.int64), y200[:100], a200[:100])   # every agent's request 10
            bref.train(t.agent_states[ids160[:100]].reshape(-1, 84, 84, 4), y200[:100], a200[:100])
            assert np.array_equal(big.get_arena(0), bref.get_arena(0))
        finally:
            big.unregister_transport()
            big.close()
            bref.close()
            net.register_transport(t)
        net.train_frames(np.array([3], np.int32), np.array([19], np.int64), y[:1], a[:1])            # agent 3 holds 19, 20, 21
        predict_named(np.array([3], np.uint32), np.array([19 + depth], np.int64))                     # ... until request 23 takes 19's slot
        with pytest.raises(RuntimeError):
            net.train_frames(np.array([3], np.int32), np.array([19], np.int64), y[:1], a[:1])
        net.train_frames(np.array([3], np.int32), np.array([19 + depth], np.int64), y[:1], a[:1])
    finally:
        net.close()
        ref.close()
        t.shutdown()
        t.close()


def test_gather_of_f32_states_from_registered_transport():
    """STATE_TRANSPORT = 'f32': the slots hold 28,224 floats; the fused conv stack then reads them out of the HIP-registered
    host segment by LDS-DMA (global_load_lds over PCIe).  Must equal the host-buffer path bit for bit."""
    import ga3c_amd  # noqa: F401
    import Transport as tp
    from NetworkVP import Network
    t = tp.Transport.create(tp.unique_name("t_zc32"), 24, 6, 4 * 84 * 84 * 4, 4, 6)
    net = Network("gpu:0", "zc32", 6, (84, 84, 4), max_batch=32, predict_lanes=1)
    try:
        net.register_transport(t)
        rng = np.random.default_rng(13)
        states = (rng.integers(0, 256, size=(24, 84 * 84 * 4), dtype=np.uint8).astype(np.float32) / np.float32(128) - np.float32(1))
        t.agent_states.view(np.float32)[:] = states
        ids = rng.permutation(24)[:17].astype(np.uint32)
        p1, v1 = net.predict_offsets(t.state_offsets(ids))
        p2, v2 = net.predict_p_and_v(states[ids].reshape(-1, 84, 84, 4))
        assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
    finally:
        net.close()
        t.shutdown()
        t.close()


@pytest.mark.timeout(120)
def test_concurrent_predictors_and_trainer_on_one_network():
    """The reference calls predict from NP threads and train from NT threads on one object without locks
    (Server.py:123-134,141-153).  Here: 3 predictor threads (host-buffer and zero-copy entry points) and 2 trainer
    threads hammer one Network for 2 s.  Every prediction must be a valid distribution and must equal what the
    weights before OR after some train step produce -- checked at the end against a quiescent re-evaluation."""
    import threading
    import time
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    net = Network("gpu:0", "conc", 6, (84, 84, 4), max_batch=64, predict_lanes=3)
    rng = np.random.default_rng(3)
    xs = [rng.integers(0, 256, size=(n, 84, 84, 4), dtype=np.uint8) for n in (1, 17, 64)]
    xt = rng.integers(0, 256, size=(48, 84, 84, 4), dtype=np.uint8)
    yt = rng.uniform(-1, 1, 48)
    at = np.eye(6, dtype=np.float32)[rng.integers(0, 6, 48)]
    net.learning_rate, net.beta = 1e-4, 0.01
    stop = time.time() + 2.0
    errors, counts = [], {"p": 0, "t": 0}

    def predictor(k):
        try:
            while time.time() < stop:
                p, v = net.predict_p_and_v(xs[k])
                if not (np.all(np.isfinite(p)) and np.all(np.isfinite(v)) and np.allclose(p.sum(axis=1), 1.0, atol=1e-5)):
                    errors.append("bad prediction")
                counts["p"] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    def trainer():
        try:
            while time.time() < stop:
                net.train(xt, yt, at)
                counts["t"] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=predictor, args=(k,)) for k in range(3)] + [threading.Thread(target=trainer) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    try:
        assert not errors, errors[:3]
        assert counts["p"] > 50 and counts["t"] > 50
        assert net.get_global_step() == counts["t"]
        # quiescent: the same call twice gives the same bits, and the weights are finite
        p1, v1 = net.predict_p_and_v(xs[1])
        p2, v2 = net.predict_p_and_v(xs[1])
        assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
        assert np.all(np.isfinite(net.get_arena(0)))
    finally:
        net.close()


@pytest.mark.timeout(120)
def test_hogwild_train_lanes():
    """train_lanes >= 2 = the reference's unlocked NT trainer threads.  Called from ONE thread the lanes take turns and
    the result is bit-identical to the synchronous net; called from two threads the steps overlap on the GPU."""
    import threading
    import time
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    rng = np.random.default_rng(5)
    xk = rng.integers(0, 256, size=(32, 84, 84, 4), dtype=np.uint8)
    y = rng.uniform(-1, 1, 32)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, 32)]
    sync = Network("gpu:0", "s", 6, (84, 84, 4), max_batch=32, predict_lanes=1, train_lanes=1)
    hog = Network("gpu:0", "h", 6, (84, 84, 4), max_batch=32, predict_lanes=1, train_lanes=2)
    try:
        for net in (sync, hog):
            net.learning_rate, net.beta = 3e-4, 0.01
            for _ in range(5):
                net.train(xk, y, a)
        assert np.array_equal(sync.get_arena(0), hog.get_arena(0)) and np.array_equal(sync.get_arena(1), hog.get_arena(1))
        p1, v1 = sync.predict_p_and_v(xk)
        p2, v2 = hog.predict_p_and_v(xk)
        assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
        stop = time.time() + 1.5
        counts, errors = [0, 0], []

        def trainer(i):
            try:
                while time.time() < stop:
                    hog.train(xk, y, a)
                    counts[i] += 1
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
        ths = [threading.Thread(target=trainer, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(60)
        assert not errors and min(counts) > 20
        assert hog.get_global_step() == 5 + sum(counts)
        assert np.all(np.isfinite(hog.get_arena(0)))
    finally:
        sync.close()
        hog.close()


@pytest.mark.timeout(180)
@pytest.mark.filterwarnings("error::pytest.PytestUnhandledThreadExceptionWarning")     # a dying batcher thread fails the test
@pytest.mark.parametrize("frontend", ["device", "host"])
def test_engine_on_raw_frames(tmp_path, monkeypatch, frontend):
    """FRAME_SOURCE = 'rgb': agents produce 210x160x3 emulator frames.  FRONTEND = 'device': they ship the raw frame, the
    HIP front-end queues planes in HBM, predictions read the queues, rollouts name their states by (agent, plane) and are
    re-assembled from the plane history.  FRONTEND = 'host': the agent runs ga3c_frame_preprocess and ships states."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    for k, v in dict(AGENTS=6, PREDICTORS=2, TRAINERS=1, SYNTHETIC_EPISODE_LENGTH=40, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=11, NUM_ACTIONS=6, PREDICTION_BATCH_SIZE=32,
                     FRAME_SOURCE='rgb', FRONTEND=frontend).items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    srv = Server(max_agents=8)
    try:
        assert srv.device_frontend == (frontend == "device")
        before = srv.model.get_arena(0)
        srv.main(max_seconds=5)
        after = srv.model.get_arena(0)
        assert srv.predictions_served > 100 and srv.training_step > 5
        assert srv.model.get_global_step() == srv.training_step
        assert np.all(np.isfinite(after)) and np.max(np.abs(after - before)) > 1e-5
        assert all(not th.is_alive() for th in srv.trainers + srv.predictors)
        if frontend == "device":
            seen = [srv.model.frame_state(a) for a in range(6)]       # an agent may just have started an episode
            assert all(1 <= depth <= 4 for _, depth in seen) and any(depth == 4 for _, depth in seen)
            assert all(s.shape == (84, 84, 4) and int(s.max()) - int(s.min()) > 50 for s, _ in seen if s is not None)
    finally:
        srv.model.close()


@pytest.mark.parametrize("helper", ["1", "0", "pipelined"])
def test_engine_on_planes_with_the_frame_queue_on_the_device(tmp_path, monkeypatch, helper):
    """FRAME_SOURCE = 'planes', FRONTEND = 'device': agents ship their newest 84x84 plane (7,056 B per step), the frame queues
    and the plane history live in HBM, rollouts name their states by (agent, plane).  helper: the answers of a batch are
    given by the serve loop's helper thread (round 3's default), by the loop itself (GA3C_RESPONDER=0, the default), or by
    the loop between the two halves of the next batch (Config.PIPELINED_FRAMES, overlap from two queued requests on)."""
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    if helper == "pipelined":
        monkeypatch.setenv("GA3C_RESPONDER", "0")
        monkeypatch.setenv("GA3C_PIPELINE_MIN_QUEUED", "2")
        monkeypatch.setattr(Config, "PIPELINED_FRAMES", True)
    else:
        monkeypatch.setenv("GA3C_RESPONDER", helper)
    for k, v in dict(AGENTS=6, PREDICTORS=2, TRAINERS=1, SYNTHETIC_EPISODE_LENGTH=40, TIME_MAX=5, DYNAMIC_SETTINGS=False,
                     SAVE_MODELS=False, TRAINING_MIN_BATCH_SIZE=11, NUM_ACTIONS=6, PREDICTION_BATCH_SIZE=32,
                     FRAME_SOURCE='planes', FRONTEND='device').items():
        monkeypatch.setattr(Config, k, v)
    from Server import Server
    srv = Server(max_agents=8)
    try:
        assert srv.device_frontend and srv.frame_shape == (84, 84, 1) and srv.transport.state_bytes == 84 * 84
        before = srv.model.get_arena(0)
        srv.main(max_seconds=5)
        after = srv.model.get_arena(0)
        assert srv.predictions_served > 100 and srv.training_step > 5
        assert srv.model.get_global_step() == srv.training_step
        assert np.all(np.isfinite(after)) and np.max(np.abs(after - before)) > 1e-5
        assert all(not th.is_alive() for th in srv.trainers + srv.predictors)
        seen = [srv.model.frame_state(a) for a in range(6)]
        assert all(1 <= depth <= 4 for _, depth in seen) and any(depth == 4 for _, depth in seen)
    finally:
        srv.model.close()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("agents,predictors", [(256, 2), (512, 4)])
def test_engine_at_the_agent_counts_of_baseline_configs_3_and_4(tmp_path, monkeypatch, agents, predictors):
    """BASELINE configs[3] / [4] name 256 and 512 agents.  That many Python agent processes do not fit a one-GPU box's
    16-core quota, so the agents here are native threads speaking the agent side of the C ABI (tests/native/native_agents.cpp:
    state into the slot, submit, futex wait, sample, ship a rollout every TIME_MAX steps) -- everything on the server side is
    the product: transport, native predictor loops, trainer threads with pipelined intakes, the HIP network.  The shape that
    collapsed twice before (a herd of hundreds of actors: the CAS-loop request ring in round 2; five busy hardware queues
    with 4 predictor threads) must serve predictions in real batches, train, and lose no worker."""
    import os
    import subprocess
    import threading
    import time
    import ga3c_amd  # noqa: F401
    from Config import Config
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "native_agents")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "native", "native_agents.cpp"), "-L", os.path.join(root, "ga3c_amd"),
                           "-lga3c_host", "-Wl,-rpath," + os.path.join(root, "ga3c_amd")])
    monkeypatch.chdir(tmp_path)
    keys = ("AGENTS", "PREDICTORS", "TRAINERS", "DYNAMIC_SETTINGS", "SAVE_MODELS", "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS",
            "PREDICTION_BATCH_SIZE", "TRAIN_MODELS", "LOAD_CHECKPOINT", "EPISODES", "PRINT_STATS_FREQUENCY")
    saved = {k: getattr(Config, k) for k in keys}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 0, predictors, 2
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS, Config.LOAD_CHECKPOINT, Config.TRAIN_MODELS = False, False, False, True
    Config.TRAINING_MIN_BATCH_SIZE, Config.PREDICTION_BATCH_SIZE, Config.NUM_ACTIONS = 127, 128, 6
    Config.EPISODES, Config.PRINT_STATS_FREQUENCY = 10 ** 9, 10 ** 9
    try:
        from Server import Server
        srv = Server(max_agents=agents)
        seconds, marks, out = 8.0, [], {}

        def driver():
            time.sleep(1.0)
            proc = subprocess.Popen([exe, srv.transport.name, str(agents), str(seconds - 2.5), "1"] + (["cache"] if srv.state_cache else []),
                                    stdout=subprocess.PIPE, text=True)
            for at in (3.0, seconds - 2.0):
                time.sleep(max(0.0, at - (time.perf_counter() - t0)))
                marks.append((time.perf_counter(), srv.predictions_served, srv.training_step, sum(p.batches for p in srv.predictors)))
            out["tool"] = proc.communicate(timeout=60)[0]
        t0 = time.perf_counter()
        th = threading.Thread(target=driver, daemon=True)
        th.start()
        srv.main(max_seconds=seconds)
        th.join(90)
        assert len(marks) == 2, "the sampler did not finish"
        (ta, pa, sa, ba), (tb, pb, sb_, bb) = marks
        pps, tps, batch = (pb - pa) / (tb - ta), (sb_ - sa) / (tb - ta), (pb - pa) / max(bb - ba, 1)
        stats = srv.model.stats()
        srv.model.close()

        def cgroup():
            try:
                return dict(line.split()[:2] for line in open("/sys/fs/cgroup/cpu.stat"))
            except OSError:
                return {}
        # Round 3 saw this test at 240 k predictions/s inside the suite against 495-550 k alone and lowered its floors to
        # 100 k "right behind other engine tests".  The cause, found in round 4 (profiles/r04_engine_matrix.md): hundreds of
        # agent threads free to roam the host's 256 CPUs -- every wake on a cold idle core, 38 us of CPU per prediction, the
        # cgroup's 16-CPU quota spent and the whole process throttled in every period: 120-400 k from run to run of the SAME
        # command.  The server now places itself on L3 domains next to the GPU (Placement.py): 7.5 us per prediction, no
        # throttling, 0.74-1.03 M predictions/s and 5.7-7.9 k train steps/s at 256 / 512 agents, run after run.  The floors are
        # back at half of that; the message carries what would explain a miss.
        where = srv.placement["why"] if srv.placement else "unplaced"
        assert pps > 350e3 and tps > 2500 and batch > 8, (pps, tps, batch, where, cgroup(), out.get("tool"), srv.lost_train_batches)
        assert srv.lost_train_batches == 0                        # the state cache held every named row
        assert stats["predict_weight_waits"] == 0                 # predictions never wait for a step in flight
        assert stats["train_rows"] / max(stats["train_calls"], 1) > 127
        assert all(t.is_alive() is False for t in srv.predictors + srv.trainers) and srv.failure is None
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)
