"""-m gpu: the device-side frame front-end (ga3c_net_frames_*, through the C ABI) against oracle/frame_frontend.py and
the golden planes.  Everything here is byte work: the bar is bit-exact."""
import os

import numpy as np
import pytest

import frame_frontend as ff

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def net():
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    n = Network("gpu:0", "test_frontend", 6, (84, 84, 4), max_batch=64, predict_lanes=2)
    yield n
    n.close()


def golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "frontend.npz"))
    return {k[4:]: (g[k], g["plane_" + k[4:]]) for k in g.files if k.startswith("rgb_")}


def atari_like(rng, n, shape=(210, 160, 3)):
    """Frames with the structure that matters to bytescale / the resampler: flat fields, sprites, noise, narrow range."""
    out = np.empty((n,) + shape, np.uint8)
    for i in range(n):
        kind = i % 4
        if kind == 0:
            out[i] = rng.integers(0, 256, size=shape, dtype=np.uint8)
        elif kind == 1:
            out[i] = rng.integers(0, 256, size=(1, 1, shape[2]), dtype=np.uint8)
            for _ in range(6):
                y, x = rng.integers(0, shape[0] - 16), rng.integers(0, shape[1] - 8)
                out[i, y:y + rng.integers(1, 16), x:x + rng.integers(1, 8)] = rng.integers(0, 256, size=shape[2], dtype=np.uint8)
        elif kind == 2:
            out[i] = (100 + rng.integers(0, 4, size=shape)).astype(np.uint8)
        else:
            out[i] = rng.integers(0, 256, size=shape[2], dtype=np.uint8)          # constant frame: max == min
    return out


def test_golden_planes(net):
    g = golden()
    for name, (rgb, plane) in g.items():
        net.frames_config(4, *rgb.shape)
        got = net.preprocess_frames(rgb)
        assert got.shape == (1, 84, 84) and got.dtype == np.uint8
        assert np.array_equal(got[0], plane), name


@pytest.mark.parametrize("shape", [(210, 160, 3), (210, 160, 4), (250, 160, 3), (96, 96, 3), (84, 84, 3), (60, 200, 3)])
def test_preprocess_equals_oracle(net, shape):
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + shape[2])
    n = 21
    rgb = atari_like(rng, n, shape)
    net.frames_config(32, *shape)
    got = net.preprocess_frames(rgb)
    for i in range(n):
        assert np.array_equal(got[i], ff.preprocess_u8(rgb[i])), "frame %d of %s" % (i, shape)


def test_frame_queue_and_prediction_from_device_states(net):
    rng = np.random.default_rng(11)
    n_agents, steps = 9, 7
    net.frames_config(16, 210, 160, 3)
    queues = [ff.FrameQueue() for _ in range(n_agents)]
    ids = np.array([3, 0, 7, 12, 5, 9, 15, 1, 8], np.int32)           # queue rows are not the batch rows
    for t in range(steps):
        rgb = atari_like(rng, n_agents)
        reset = np.zeros(n_agents, np.uint8)
        if t == 0:
            reset[:] = 1
        if t == 5:
            reset[2] = 1                                                  # one episode ends: its queue starts over
        net.push_frames(rgb, ids, reset)
        for k in range(n_agents):
            if reset[k]:
                queues[k].clear()
            queues[k].push(ff.preprocess_u8(rgb[k]))
            state, depth = net.frame_state(ids[k])
            want = queues[k].state_u8()
            assert depth == len(queues[k].q)
            assert (state is None) == (want is None)
            if want is not None:
                assert np.array_equal(state, want)
    ready = [k for k in range(n_agents) if queues[k].state_u8() is not None]
    assert 2 not in ready and len(ready) == n_agents - 1
    states = np.stack([queues[k].state_u8() for k in ready])
    p, v = net.predict_frames(ids[ready])
    p2, v2 = net.predict_p_and_v(states)                                  # the same bytes through the host-buffer path
    assert np.array_equal(p, p2) and np.array_equal(v, v2)
    with pytest.raises(RuntimeError, match="no state yet"):
        net.predict_frames(ids[[2]])
    with pytest.raises(RuntimeError, match="twice"):
        net.push_frames(rgb[:2], np.array([4, 4], np.int32))
    with pytest.raises(RuntimeError, match="outside"):
        net.push_frames(rgb[:1], np.array([16], np.int32))


def test_frames_read_in_place_from_pinned_and_registered_memory(net):
    import Transport as tp
    rng = np.random.default_rng(5)
    net.frames_config(8, 210, 160, 3)
    rgb = atari_like(rng, 5)
    want = net.preprocess_frames(rgb)                                     # pageable -> staged
    pinned = net.pinned_array(rgb.shape, np.uint8)
    pinned[:] = rgb
    assert np.array_equal(net.preprocess_frames(pinned), want)
    # frames lying in the shared-memory transport (an agent's slot is big enough for one 100,800-byte frame)
    t = tp.Transport.create(tp.unique_name("t_fr"), 4, 6, 4 * 84 * 84 * 4, 2, 6)
    try:
        net.register_transport(t)
        slot = t.state_view(1)[:210 * 160 * 3].reshape(1, 210, 160, 3)
        slot[:] = rgb[3]
        assert np.array_equal(net.preprocess_frames(slot)[0], want[3])
    finally:
        net.unregister_transport()
        t.shutdown()
        t.close()


def test_frames_need_config(net):
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    fresh = Network("gpu:0", "test_frontend2", 4, (84, 84, 4), max_batch=8, predict_lanes=1)
    try:
        fresh._frame_shape = (210, 160, 3)
        with pytest.raises(RuntimeError, match="frames_config first"):
            fresh.preprocess_frames(np.zeros((210, 160, 3), np.uint8))
        with pytest.raises(RuntimeError):
            fresh.frames_config(4, 210, 160, 2)
    finally:
        fresh.close()


def test_training_rows_from_the_plane_history(net):
    """Rollouts as (agent, plane sequence number): train_frames re-assembles each row's [84,84,4] state from the
    device-side plane history; the step must equal, bit for bit, a train() on the states the oracle's frame queues held."""
    import Transport as tp
    from NetworkVP import Network
    rng = np.random.default_rng(21)
    n_agents, steps, hist = 5, 14, 12
    other = Network("gpu:0", "test_frontend_ref", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    t = tp.Transport.create(tp.unique_name("t_hist"), 8, 6, 4 * 84 * 84 * 4, 2, 6)     # slots hold one raw frame
    try:
        theta = net.get_arena(0)
        for m in (net, other):
            m.set_arena(0, theta)
            m.set_arena(1, np.ones_like(theta))
            m.learning_rate, m.beta = 3e-4, 0.01
        net.frames_config(8, 210, 160, 3, history=hist)
        net.register_transport(t)
        ids = np.array([6, 2, 0, 7, 3], np.int32)
        queues = [ff.FrameQueue() for _ in range(n_agents)]
        states = {}                                                  # (agent, seq) -> state the queue held after that push
        for step in range(steps):
            rgb = atari_like(rng, n_agents)
            reset = np.zeros(n_agents, np.uint8)
            if step == 0:
                reset[:] = 1
            if step == 6:
                reset[1] = 1
            if step % 2:                                             # odd steps: frames handed over in the transport slots
                for k, a in enumerate(ids):
                    t.state_view(a)[:210 * 160 * 3] = rgb[k].reshape(-1)
                seq = net.push_frame_offsets(t.state_offsets(ids.astype(np.uint32)), ids, reset)
            else:
                seq = net.push_frames(rgb, ids, reset)
            assert seq.tolist() == [step] * n_agents
            for k in range(n_agents):
                if reset[k]:
                    queues[k].clear()
                queues[k].push(ff.preprocess_u8(rgb[k]))
                if queues[k].state_u8() is not None:
                    states[(int(ids[k]), step)] = queues[k].state_u8()
        live = [(a, s) for (a, s) in states if steps - (s - 3) <= hist]
        assert len(live) >= 30 and any(s % hist < 3 for _, s in live)           # rows that wrap around the ring
        pick = [live[i] for i in rng.permutation(len(live))[:24]]
        x = np.stack([states[k] for k in pick])
        y = rng.uniform(-1, 1, len(pick))
        act = np.eye(6, dtype=np.float32)[rng.integers(0, 6, len(pick))]
        # Network.log on rows named by (agent, plane): the same evaluation as on the states themselves, nothing trained
        ev_frames = net.evaluate(None, y, act, frames=([a for a, _ in pick], [s for _, s in pick]))
        ev_states = other.evaluate(x, y, act)
        assert all(np.array_equal(p, q) for p, q in zip(ev_frames, ev_states))
        assert np.array_equal(net.get_arena(0), theta)
        assert net.frames_pushed(int(ids[0])) == steps and net.frames_pushed(1) == 0
        net.train_frames([a for a, _ in pick], [s for _, s in pick], y, act)
        other.train(x, y, act, None, None, 0)
        assert np.array_equal(net.get_arena(0), other.get_arena(0))
        assert np.array_equal(net.last_losses, other.last_losses)
        gone = [(a, s) for (a, s) in states if steps - (s - 3) > hist][0]
        with pytest.raises(RuntimeError, match="has left the"):
            net.train_frames([gone[0]], [gone[1]], y[:1], act[:1])
        with pytest.raises(RuntimeError, match="no state at plane"):
            net.train_frames([int(ids[0])], [2], y[:1], act[:1])
        with pytest.raises(RuntimeError, match="no state at plane"):
            net.train_frames([int(ids[0])], [steps], y[:1], act[:1])
    finally:
        net.unregister_transport()
        t.shutdown()
        t.close()
        other.close()


def test_ready_made_planes_keep_only_the_frame_queue_on_the_device(net):
    """frames_config(..., 84, 84, 1): the "frames" are final planes (FRAME_SOURCE = 'planes', FRONTEND = 'device') -- no
    gray / bytescale / resize, only the 4-deep queue and the plane history in HBM.  Queue contents, predictions and
    training rows must equal what the host-side stacking of the same planes gives."""
    from NetworkVP import Network
    rng = np.random.default_rng(31)
    n_agents, steps, hist = 6, 11, 12
    other = Network("gpu:0", "test_planes_ref", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    try:
        theta = net.get_arena(0)
        for m in (net, other):
            m.set_arena(0, theta)
            m.set_arena(1, np.ones_like(theta))
            m.learning_rate, m.beta = 3e-4, 0.01
        net.frames_config(8, 84, 84, 1, history=hist)
        ids = np.array([5, 1, 7, 0, 3, 6], np.int32)
        queues = [ff.FrameQueue() for _ in range(n_agents)]
        states = {}
        for step in range(steps):
            planes = rng.integers(0, 256, size=(n_agents, 84, 84, 1), dtype=np.uint8)
            reset = np.zeros(n_agents, np.uint8)
            if step == 0:
                reset[:] = 1
            if step == 4:
                reset[3] = 1
            seq = net.push_frames(planes, ids, reset)
            assert seq.tolist() == [step] * n_agents
            for k in range(n_agents):
                if reset[k]:
                    queues[k].clear()
                queues[k].push(planes[k, :, :, 0])
                state, depth = net.frame_state(ids[k])
                want = queues[k].state_u8()
                assert depth == len(queues[k].q) and (state is None) == (want is None)
                if want is not None:
                    assert np.array_equal(state, want)
                    states[(int(ids[k]), step)] = want
        ready = list(range(n_agents))
        p, v = net.predict_frames(ids[ready])
        p2, v2 = other.predict_p_and_v(np.stack([queues[k].state_u8() for k in ready]))
        assert np.array_equal(p, p2) and np.array_equal(v, v2)
        live = [(a, s) for (a, s) in states if steps - (s - 3) <= hist]
        pick = [live[i] for i in rng.permutation(len(live))[:20]]
        x = np.stack([states[k] for k in pick])
        y = rng.uniform(-1, 1, len(pick))
        act = np.eye(6, dtype=np.float32)[rng.integers(0, 6, len(pick))]
        net.train_frames([a for a, _ in pick], [s for _, s in pick], y, act)
        other.train(x, y, act, None, None, 0)
        assert np.array_equal(net.get_arena(0), other.get_arena(0))
        with pytest.raises(RuntimeError, match="ready-made"):
            net.frames_config(8, 80, 84, 1)
    finally:
        other.close()


def test_serve_frames_is_push_plus_predict_in_one_round_trip(net):
    """ga3c_net_serve_frames (the native raw-frame predictor loop's callback) against the two-call path."""
    import Transport as tp
    from NetworkVP import Network
    rng = np.random.default_rng(33)
    other = Network("gpu:0", "test_frontend_two_call", 6, (84, 84, 4), max_batch=64, predict_lanes=1)
    t = tp.Transport.create(tp.unique_name("t_srvfr"), 8, 6, 4 * 84 * 84 * 4, 2, 6)
    t2 = tp.Transport.create(tp.unique_name("t_srvfr2"), 8, 6, 4 * 84 * 84 * 4, 2, 6)   # one segment per engine
    try:
        other.set_arena(0, net.get_arena(0))
        ids = np.array([1, 4, 6, 3], np.int32)
        for m, seg in ((net, t), (other, t2)):
            m.frames_config(8, 210, 160, 3, history=16)
            m.register_transport(seg)
        offs = t.state_offsets(ids.astype(np.uint32))
        assert np.array_equal(offs, t2.state_offsets(ids.astype(np.uint32)))
        for step in range(7):
            rgb = atari_like(rng, 4)
            for k, a in enumerate(ids):
                t.state_view(a)[:210 * 160 * 3] = rgb[k].reshape(-1)
                t2.state_view(a)[:210 * 160 * 3] = rgb[k].reshape(-1)
            flags = np.zeros(4, np.uint32)
            if step == 0:
                flags |= tp.REQ_RESET
            if step == 4:
                flags[2] |= tp.REQ_RESET                              # agent 6 starts a new episode
            depth_after = np.array([min(step + 1, 4)] * 4)
            depth_after[2] = min(step + 1, 4) if step < 4 else min(step - 3, 4)
            flags[depth_after < 4] |= tp.REQ_NO_PREDICT
            p, v = net.serve_frames(offs, ids, flags)
            other.push_frame_offsets(offs, ids, (flags & tp.REQ_RESET) != 0)
            want = np.nonzero((flags & tp.REQ_NO_PREDICT) == 0)[0]
            if want.size:
                p2, v2 = other.predict_frames(ids[want])
                assert np.array_equal(p[want], p2) and np.array_equal(v[want], v2)
            assert not p[(flags & tp.REQ_NO_PREDICT) != 0].any()
            for a in ids:
                s1, d1 = net.frame_state(a)
                s2, d2 = other.frame_state(a)
                assert d1 == d2 and ((s1 is None and s2 is None) or np.array_equal(s1, s2))
        # the call in two halves, two batches in flight on the net's two lanes (ga3c_pq_serve_frames_pipelined answers one
        # batch between the halves of the next): agents 1, 4 | 6, 3, ended in the other order; same bits as one call
        rgb = atari_like(rng, 4)
        for k, a in enumerate(ids):
            t.state_view(a)[:210 * 160 * 3] = rgb[k].reshape(-1)
            t2.state_view(a)[:210 * 160 * 3] = rgb[k].reshape(-1)
        flags = np.array([0, tp.REQ_NO_PREDICT, 0, 0], np.uint32)
        ta = net.serve_frames_begin(offs[:2], ids[:2], flags[:2])
        tb = net.serve_frames_begin(offs[2:], ids[2:], flags[2:])
        assert ta != tb
        pb, vb = net.serve_frames_end(tb, flags[2:])
        pa, va = net.serve_frames_end(ta, flags[:2])
        p2, v2 = other.serve_frames(offs, ids, flags)
        assert np.array_equal(np.concatenate([pa, pb]), p2) and np.array_equal(np.concatenate([va, vb]), v2)
        assert not pa[1].any() and p2[0].any()
        with pytest.raises(RuntimeError, match="no batch was begun"):
            net.serve_frames_end(ta, flags[:2])
        with pytest.raises(RuntimeError, match="asks for a prediction with"):
            net.serve_frames(offs[:1], np.array([7], np.int32), np.array([tp.REQ_RESET], np.uint32))
        p, v = net.serve_frames(offs, ids, np.zeros(4, np.uint32))             # the lanes were given back: the net still serves
        assert p.shape == (4, 6) and np.isfinite(p).all()
    finally:
        for m in (net, other):
            m.unregister_transport()
        for seg in (t, t2):
            seg.shutdown()
            seg.close()
        other.close()
