#!/usr/bin/env python3
"""Development aid: sha256 of the weights and the RMSProp slots after a few production train steps at the given row counts,
uint8 and f32 states.  Run it under two settings of an engine switch (section 8b of DESIGN.md) and compare the lines: kernel
variants that only change scheduling (GA3C_C2DW_OCC, GA3C_D1B_TAIL) must print the same digests.
usage: python tools/ab_bits.py [rows ...]      (default 129 132 133)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    rows = [int(a) for a in sys.argv[1:]] or [129, 132, 133]
    for B in rows:
        for u8 in (False, True):
            net = Network("gpu:0", "ab_bits", 6, (84, 84, 4), max_batch=B, predict_lanes=1)
            rng = np.random.Generator(np.random.PCG64(B))
            xk = rng.integers(0, 256, size=(B, 84, 84, 4), dtype=np.uint8)
            x = xk if u8 else xk.astype(np.float32) / np.float32(128) - np.float32(1)
            a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, B)]
            y = rng.uniform(-1, 1, B).astype(np.float32)
            net.learning_rate, net.beta = 3e-4, 0.01
            for _ in range(3):
                net.train(x, y, a)
            h = hashlib.sha256()
            h.update(np.ascontiguousarray(net.get_arena(0)).tobytes())
            h.update(np.ascontiguousarray(net.get_arena(1)).tobytes())
            print("rows %d %s %s" % (B, "u8 " if u8 else "f32", h.hexdigest()[:32]), flush=True)
            net.close()


if __name__ == "__main__":
    main()
