"""Development aid: Hogwild train-lane throughput (ga3c_net_time_train_lanes) against the number of lanes.
    python tools/train_lanes.py [batch] [max lanes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ga3c_amd, _native as nat
from NetworkVP import Network
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 4
net = Network("gpu:0", "tl", 6, (84, 84, 4), max_batch=B, predict_lanes=1, train_lanes=NL)
rng = np.random.default_rng(0)
x = rng.integers(0, 256, size=(B, 84, 84, 4), dtype=np.uint8).astype(np.float32) / 128 - 1
y = rng.uniform(-1, 1, B).astype(np.float32)
a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, B)]
nat.check(net._lib.ga3c_net_upload(net._h, nat.ptr(x), nat.ptr(y), nat.ptr(a), B))
ms = nat.C.c_float()
for nl in range(1, NL + 1):
    nat.check(net._lib.ga3c_net_time_train_lanes(net._h, B, 50, nl, 3e-4, 0.01, nat.C.byref(ms)))
    nat.check(net._lib.ga3c_net_time_train_lanes(net._h, B, 400, nl, 3e-4, 0.01, nat.C.byref(ms)))
    print("train lanes %d: %.2f us per step -> %.1f k steps/s" % (nl, ms.value / 400 * 1e3, 400 / ms.value))
net.close()
