"""Development aid: 30 train steps at 128 rows on a network with a ONE-RANK RCCL communicator attached (the N-GPU code path),
then 30 without (the fused single-GPU step), for a rocprofv3 --kernel-trace timeline (tools/timeline.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ga3c_amd  # noqa: E402,F401
from NetworkVP import Network  # noqa: E402
import _native as nat  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.Generator(np.random.PCG64(1))
x = rng.integers(0, 256, size=(B, 84, 84, 4), dtype=np.uint8).astype(np.float32) / np.float32(128) - np.float32(1)
a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, B)]
y = rng.uniform(-1, 1, B).astype(np.float32)
ms = nat.C.c_float()
for dp in (True, False):
    net = Network("gpu:0", "dp_step", 6, (84, 84, 4), max_batch=B, predict_lanes=1)
    nat.check(net._lib.ga3c_net_upload(net._h, nat.ptr(x), nat.ptr(y), nat.ptr(a), B))
    if dp:
        net.comm_init(Network.make_comm_id(), 0, 1)
    nat.check(net._lib.ga3c_net_time_resident(net._h, 1, B, 30, 3e-4, 0.01, nat.C.byref(ms)))
    print("dp" if dp else "fused", "%.2f us per step" % (ms.value / 30 * 1e3), file=sys.stderr)
    net.close()
