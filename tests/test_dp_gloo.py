"""world_size-2 CPU test of the data-parallel contract (gloo): gradients of disjoint row shards, summed by
all-reduce, followed by the same RMSProp step on every rank, equal the single-process step on the whole
batch.  The per-shard gradients come from the oracle's C port (no GPU here); the sharding and the id
exchange are the product's (ga3c_amd/DataParallel.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import ga3c_amd  # noqa: F401
    import DataParallel as dp
    import ga3c_oracle as o
    import ga3c_oracle_cport as oc
    oc.lib().ga3c_oc_set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        token = dp.exchange_comm_id(lambda: np.arange(128, dtype=np.uint8) * 2 % 251, rank)
        num_actions, rows = 6, 11                      # odd row count: shards of 6 and 5
        params = o.init_params(num_actions)
        theta = np.concatenate([params[k].reshape(-1) for k in o.PARAM_ORDER]).astype(np.float32)
        x = o.synthetic_states(rows, seed=4)
        rng = np.random.default_rng(4)
        y = rng.uniform(-1, 1, rows).astype(np.float32)
        a = np.eye(num_actions, dtype=np.float32)[rng.integers(0, num_actions, rows)]
        lo, hi = dp.shard_bounds(rows, rank, world)
        _, g = oc.train(theta.copy(), np.ones_like(theta), num_actions, x[lo:hi], y[lo:hi], a[lo:hi], lr=-1.0, beta=0.01)
        gt = torch.from_numpy(g)
        dist.all_reduce(gt, op=dist.ReduceOp.SUM)       # sum, not mean: the loss is a sum over rows
        ms = np.ones_like(theta)
        ms += (g * g - ms) * np.float32(0.01)
        new = theta - (g * np.float32(3e-4)) / np.sqrt(np.float32(0.1) + ms)
        # single-process reference on the whole batch
        th_full, ms_full = theta.copy(), np.ones_like(theta)
        _, g_full = oc.train(th_full, ms_full, num_actions, x, y, a, lr=3e-4, beta=0.01)
        out_q.put((rank, (lo, hi), token.tolist(), float(np.max(np.abs(g - g_full))), float(np.max(np.abs(new - th_full))),
                   float(np.max(np.abs(g_full)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sum_allreduce_equals_single_process_step():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [(0, 6), (6, 11)]
    assert results[0][2] == results[1][2] == [int(v) for v in (np.arange(128, dtype=np.uint8) * 2 % 251)]
    for _, _, _, gdiff, tdiff, gmax in results:
        assert gdiff < 1e-5 * max(1.0, gmax)            # f32 summation order only
        assert tdiff < 1e-6


def test_shard_bounds_cover_rows_exactly():
    import ga3c_amd  # noqa: F401
    import DataParallel as dp
    for rows in (1, 7, 128, 511):
        for world in (1, 2, 3, 4, 8):
            spans = [dp.shard_bounds(rows, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rows
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
