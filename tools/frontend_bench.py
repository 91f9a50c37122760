"""Times the frame front-end kernel alone (resident frames) -- the command profiles/r01_l_* were collected with.

    python tools/frontend_bench.py [frames] [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import ga3c_amd  # noqa: E402,F401
import _native as nat  # noqa: E402
from NetworkVP import Network  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
net = Network("gpu:0", "frontend_bench", 6, (84, 84, 4), max_batch=8, predict_lanes=1)
rng = np.random.Generator(np.random.PCG64(12345 + 99))
for shape in ((210, 160, 3), (250, 160, 3)):
    frames = rng.integers(0, 256, size=(nf,) + shape, dtype=np.uint8)
    net.frames_config(nf, *shape)
    nat.check(net._lib.ga3c_net_frames_upload(net._h, nat.ptr(frames, nat.u8p), nf), "upload")
    ms = nat.C.c_float()
    nat.check(net._lib.ga3c_net_time_frames(net._h, nf, 5, nat.C.byref(ms)), "time")
    nat.check(net._lib.ga3c_net_time_frames(net._h, nf, iters, nat.C.byref(ms)), "time")
    us = ms.value * 1e3 / iters
    byts = shape[0] * shape[1] * shape[2] + 2 * 28224
    print("%s: %d frames per launch, %.1f us per launch, %.2f M frames/s, %.0f GB/s algorithmic"
          % (shape, nf, us, nf / us, nf * byts / us / 1e3))
net.close()
