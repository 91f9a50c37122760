"""Latency of one predictor round trip through the C ABI at small batch (development aid).

    python tools/predict_latency.py            # plain launches (default)
    GA3C_GRAPHS=1 python tools/predict_latency.py   # prediction steps replayed from captured hipGraphs
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ga3c_amd, Transport as tp
from NetworkVP import Network

t = tp.Transport.create(tp.unique_name("lat"), 128, 6, 84 * 84 * 4, 8, 6)
net = Network("gpu:0", "lat", 6, (84, 84, 4), max_batch=128, predict_lanes=2)
net.register_transport(t)
rng = np.random.default_rng(0)
t.agent_states[:] = rng.integers(0, 256, size=(128, 84 * 84 * 4), dtype=np.uint8)
print("graphs:", "on" if os.environ.get("GA3C_GRAPHS") else "off")
for n in (1, 8, 16, 32, 64, 128):
    ids = np.arange(n, dtype=np.uint32)
    offs = t.state_offsets(ids)
    x = np.ascontiguousarray(t.agent_states[:n]).reshape(n, 84, 84, 4)
    for _ in range(50):
        net.predict_offsets(offs)
    net.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(500):
        net.predict_offsets(offs)
    a = (time.perf_counter() - t0) / 500 * 1e6
    st = net.stats(reset=True)
    c = max(st["predict_calls"], 1)
    split = "launch %.1f sync %.1f gpu span %.1f" % (st["predict_launch_ns"] / c / 1e3, st["predict_sync_ns"] / c / 1e3, st["predict_gpu_ns"] / c / 1e3)
    for _ in range(50):
        net.predict_p_and_v(x)
    t0 = time.perf_counter()
    for _ in range(500):
        net.predict_p_and_v(x)
    b = (time.perf_counter() - t0) / 500 * 1e6
    print("batch %3d: zero-copy gather %.1f us per call (%s; the span only with GA3C_TIME_PREDICTIONS=1), host-buffer u8 %.1f us per call" % (n, a, split, b))
net.unregister_transport(); net.close(); t.shutdown(); t.close()
