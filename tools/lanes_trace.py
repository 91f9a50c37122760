"""Development aid: N resident prediction steps on exactly NL lanes, for a rocprofv3 --kernel-trace (tools/timeline.py reads it).
usage: python tools/lanes_trace.py [B] [NL] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ga3c_amd, _native as nat   # noqa: E401,F401
from NetworkVP import Network
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
net = Network("gpu:0", "l", 6, (84, 84, 4), max_batch=B, predict_lanes=NL)
x = np.random.default_rng(0).integers(0, 256, size=(B, 84, 84, 4), dtype=np.uint8).astype(np.float32) / 128 - 1
nat.check(net._lib.ga3c_net_upload(net._h, nat.ptr(x), None, None, B))
ms = nat.C.c_float()
nat.check(net._lib.ga3c_net_time_predict_lanes(net._h, B, 50, NL, nat.C.byref(ms)))
nat.check(net._lib.ga3c_net_time_predict_lanes(net._h, B, K, NL, nat.C.byref(ms)))
print("lanes %d: %.2f us per step -> %.2f M pred/s" % (NL, ms.value / K * 1e3, K * B / ms.value / 1e3))
net.close()
