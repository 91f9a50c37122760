#!/usr/bin/env python3
"""Development aid: per-launch time of named kernels (ga3c_net_time_kernel), interleaved rounds, median and min.
usage: python tools/ktime.py [--batch B] name [name ...]      (@predict / @train: whole steps, one lane)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--u8", action="store_true", help="@train / @predict on uint8-resident states (what the transport delivers)")
    ap.add_argument("names", nargs="+")
    args = ap.parse_args()
    import ga3c_amd  # noqa: F401
    from NetworkVP import Network
    import _native as nat
    B = args.batch
    net = Network("gpu:0", "ktime", 6, (84, 84, 4), max_batch=B, predict_lanes=1)
    rng = np.random.Generator(np.random.PCG64(1))
    xk = rng.integers(0, 256, size=(B, 84, 84, 4), dtype=np.uint8)
    x = xk.astype(np.float32) / np.float32(128) - np.float32(1)
    a = np.eye(6, dtype=np.float32)[rng.integers(0, 6, B)]
    y = rng.uniform(-1, 1, B).astype(np.float32)
    net.learning_rate, net.beta = 3e-4, 0.01
    net.train(x, y, a)                       # every workspace buffer holds real data
    lib, h = net._lib, net._h
    nat.check(lib.ga3c_net_upload(h, nat.ptr(x), nat.ptr(y), nat.ptr(a), B))
    nat.check(lib.ga3c_net_upload_u8(h, nat.ptr(xk, nat.u8p), nat.ptr(y), nat.ptr(a), B))
    nat.check(lib.ga3c_net_upload(h, nat.ptr(x), nat.ptr(y), nat.ptr(a), B))
    if args.u8:
        nat.check(lib.ga3c_net_upload_u8(h, nat.ptr(xk, nat.u8p), nat.ptr(y), nat.ptr(a), B))
    ms = nat.C.c_float()
    res = {n: [] for n in args.names}
    for _ in range(args.rounds):
        for n in args.names:
            if n in ("@predict", "@train"):                # whole steps on one lane, inputs resident (ga3c_net_time_resident)
                nat.check(lib.ga3c_net_time_resident(h, 0 if n == "@predict" else 1, B, 200, 3e-4, 0.01, nat.C.byref(ms)), n)
                res[n].append(ms.value / 200 * 1e3)
                continue
            nat.check(lib.ga3c_net_time_kernel(h, n.encode(), B, 30, nat.C.byref(ms)), n)
            res[n].append(ms.value / 30 * 1e3)
    for n in args.names:
        v = sorted(res[n])
        print("%-24s median %7.2f us   min %7.2f us" % (n, v[len(v) // 2], v[0]))
    net.close()


if __name__ == "__main__":
    main()
