"""Latency of one synchronous train call through the C ABI (development aid): rows gathered out of the registered transport
segment (the zero-copy trainer path) against the same batch resident in HBM -- from ONE calling thread, and from TWO
(Config.TRAINERS = 2, Server.py:132-134): the second thread's gather runs on the lane's staging stream while the first
thread's step is in flight, so the time per call is what the slower of the two phases takes.

    python tools/train_latency.py [batch ...]
"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ga3c_amd, Transport as tp
from NetworkVP import Network
import _native as nat

batches = [int(b) for b in sys.argv[1:]] or [32, 128, 130]
t = tp.Transport.create(tp.unique_name("tlat"), 160, 6, 84 * 84 * 4, 32, 6)
net = Network("gpu:0", "tlat", 6, (84, 84, 4), max_batch=160, predict_lanes=2)
net.register_transport(t)
rng = np.random.default_rng(0)
t.agent_states[:] = rng.integers(0, 256, size=(160, 84 * 84 * 4), dtype=np.uint8)
net.learning_rate, net.beta = 3e-4, 0.01
for n in batches:
    offs = t.state_offsets(np.arange(n, dtype=np.uint32))
    if os.environ.get("TL_ROLLOUT_ROWS"):      # rows as they lie in rollout slots: runs of 6 contiguous rows
        offs = np.concatenate([t.rollout_row_offsets(k, 6) for k in range((n + 5) // 6)])[:n]
    y = rng.uniform(-1, 1, n).astype(np.float32)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, n)]
    for _ in range(30):
        net.train_offsets(offs, y, a)
    t0 = time.perf_counter()
    for _ in range(300):
        net.train_offsets(offs, y, a)
    call = (time.perf_counter() - t0) / 300 * 1e6
    # two trainer threads, each with rows of its own (the GIL is released inside the ctypes call)
    offs2 = t.state_offsets(np.arange(160 - n, 160, dtype=np.uint32)) if n <= 80 else offs
    go = threading.Barrier(3)

    def trainer(o_):
        go.wait()
        for _ in range(300):
            net.train_offsets(o_, y, a)
        go.wait()
    ths = [threading.Thread(target=trainer, args=(o_,)) for o_ in (offs, offs2)]
    for th in ths:
        th.start()
    net.stats(reset=True)
    go.wait()
    t0 = time.perf_counter()
    go.wait()
    call2 = (time.perf_counter() - t0) / 600 * 1e6
    st = net.stats()
    for th in ths:
        th.join()
    split = {k: round(st[k] / max(st["train_calls"], 1) / 1e3, 1) for k in ("train_stage_ns", "train_lane_wait_ns", "train_launch_ns", "train_sync_ns")}
    xk = np.ascontiguousarray(t.agent_states[:n]).reshape(n, 84, 84, 4)
    nat.check(net._lib.ga3c_net_upload_u8(net._h, nat.ptr(xk, nat.u8p), nat.ptr(y), nat.ptr(a), n))
    ms = nat.C.c_float()
    nat.check(net._lib.ga3c_net_time_resident(net._h, 1, n, 300, 3e-4, 0.01, nat.C.byref(ms)))
    print("batch %3d: synchronous train_offsets call %6.1f us from one thread, %6.1f us per call from two threads "
          "(us per call: %s); resident, back to back %5.1f us per step" % (n, call, call2, split, ms.value / 300 * 1e3))
# rows named (agent, request number) out of the engine's state cache (Config.STATE_CACHE): kept by the predictions that read them
cnet = Network("gpu:0", "tlat_cache", 6, (84, 84, 4), max_batch=160, predict_lanes=2)
net.unregister_transport()
cnet.register_transport(t)
cnet.learning_rate, cnet.beta = 3e-4, 0.01
cnet.state_cache_config(160, 8)
for lo in (0, 80):
    ids_ = np.arange(lo, lo + 80, dtype=np.uint32)
    offs_ = np.ascontiguousarray(t.state_offsets(ids_), dtype=np.int64)
    ag_, sq_ = ids_.astype(np.int32), np.full(80, 5, np.int64)
    tk = nat.C.c_int32()
    nat.check(cnet._lib.ga3c_net_predict_gather_begin_cached(cnet._h, nat.ptr(offs_, nat.i64p), nat.ptr(ag_, nat.i32p), nat.ptr(sq_, nat.i64p), 80, 1, nat.C.byref(tk)))
    p_, v_ = np.empty((80, 6), np.float32), np.empty(80, np.float32)
    nat.check(cnet._lib.ga3c_net_predict_gather_end(cnet._h, tk.value, 80, nat.ptr(p_), nat.ptr(v_)))
for n in batches:
    ag, sq = np.arange(n, dtype=np.int32), np.full(n, 5, np.int64)
    y = rng.uniform(-1, 1, n).astype(np.float32)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, n)]
    for _ in range(30):
        cnet.train_frames(ag, sq, y, a)
    cnet.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(300):
        cnet.train_frames(ag, sq, y, a)
    one = (time.perf_counter() - t0) / 300 * 1e6
    st = cnet.stats()
    print("   one thread, us per call:", {k: round(st[k] / max(st["train_calls"], 1) / 1e3, 1) for k in ("train_stage_ns", "train_lane_wait_ns", "train_launch_ns", "train_sync_ns")})
    go = threading.Barrier(3)

    def ctrainer():
        go.wait()
        for _ in range(300):
            cnet.train_frames(ag, sq, y, a)
        go.wait()
    ths = [threading.Thread(target=ctrainer) for _ in range(2)]
    for th in ths:
        th.start()
    go.wait()
    t0 = time.perf_counter()
    go.wait()
    two = (time.perf_counter() - t0) / 600 * 1e6
    for th in ths:
        th.join()
    print("batch %3d: synchronous train call on rows NAMED in the state cache %6.1f us from one thread, %6.1f us per call from two" % (n, one, two))
cnet.unregister_transport()
cnet.close()
# rows named by (agent, plane) out of the device-side plane history (frame queue on the device)
net.frames_config(160, 84, 84, 1, history=16)
ids = np.arange(160, dtype=np.int32)
for step in range(8):
    net.push_frames(rng.integers(0, 256, size=(160, 84, 84, 1), dtype=np.uint8), ids, np.full(160, step == 0, np.uint8))
for n in batches:
    ag = ids[:n].copy()
    sq = np.full(n, 7, np.int64)
    y = rng.uniform(-1, 1, n).astype(np.float32)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, n)]
    for _ in range(30):
        net.train_frames(ag, sq, y, a)
    t0 = time.perf_counter()
    for _ in range(300):
        net.train_frames(ag, sq, y, a)
    print("batch %3d: synchronous train_frames call %6.1f us" % (n, (time.perf_counter() - t0) / 300 * 1e6))
t.shutdown(); t.close(); net.close()
