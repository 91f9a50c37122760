"""world_size-2 CPU tests of the data-parallel contract (gloo): gradients of disjoint row shards, summed by
all-reduce, followed by the same RMSProp step on every rank, equal the single-process step on the whole
batch.  The per-shard gradients come from the oracle's C port (no GPU here).  First test: the product's sharding and id
exchange (ga3c_amd/DataParallel.py) around hand-made shards; second test: the shards themselves come out of the product's
Server / ThreadTrainer / EngineGroup path, one Server per rank."""
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


def _worker(rank, world, port, out_q, tmp):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GA3C_DP_DIR=tmp)
    import torch
    import torch.distributed as dist
    import ga3c_amd  # noqa: F401
    import DataParallel as dp
    import ga3c_oracle as o
    import ga3c_oracle_cport as oc
    oc.lib().ga3c_oc_set_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert "torch" not in dp.__dict__                  # the product's control plane is torch-free (file + TCP rendezvous)
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
def test_two_rank_sum_allreduce_equals_single_process_step(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
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


class _ShardModel:
    """What libga3c_hip.so does in ga3c_net_train with a communicator attached, on CPU: gradient of this rank's rows (the
    oracle's C port), all-reduce with op = sum (gloo standing in for RCCL), the same RMSProp step on every rank.  The
    batches come from the PRODUCT's Server / ThreadTrainer / DataParallel path; every step's rows are kept for the check."""

    def __init__(self, oc, theta, group, log):
        self.oc, self.theta, self.ms, self.group, self.log = oc, theta.copy(), np.ones_like(theta), group, log
        self.learning_rate = self.beta = None

    def predict_p_and_v(self, x):
        b = x.shape[0]
        return np.full((b, 6), 1.0 / 6, np.float32), np.zeros(b, np.float32)

    def train(self, x, y_r, a, x2, done, tid):
        import time
        import torch
        import torch.distributed as dist
        time.sleep(0.03)                                  # keeps the run (and the rows the check replays) short
        xf = x.astype(np.float32) / np.float32(128.0) - np.float32(1.0)
        _, g = self.oc.train(self.theta.copy(), self.ms.copy(), 6, xf.reshape(x.shape[0], -1), y_r, a, lr=-1.0, beta=self.beta)
        gt = torch.from_numpy(g)
        dist.all_reduce(gt, op=dist.ReduceOp.SUM, group=self.group)
        self.ms += (g * g - self.ms) * np.float32(0.01)
        self.theta -= (g * np.float32(self.learning_rate)) / np.sqrt(np.float32(0.1) + self.ms)
        self.log.append((x.copy(), np.asarray(y_r, np.float32).copy(), a.copy(), float(self.learning_rate), float(self.beta)))

    def save(self, episode):
        pass



def _engine_worker(rank, world, port, tmp):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.chdir(tmp)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), GA3C_DP_DIR=tmp)
    import torch.distributed as dist
    import ga3c_amd  # noqa: F401
    from Config import Config
    import DataParallel
    import ga3c_oracle as o
    import ga3c_oracle_cport as oc
    oc.lib().ga3c_oc_set_threads(2)
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 2, 1, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX, Config.TRAINING_MIN_BATCH_SIZE = 12, 5, 7
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS, Config.ZERO_COPY = False, False, False
    Config.PRINT_STATS_FREQUENCY = 10 ** 9
    Config.RESULTS_FILENAME = "results_rank%d.txt" % rank
    Config.RANDOM_SEED += 1000 * rank                             # different rollouts on every rank (GA3C.py does the same)
    Config.EPISODES, Config.ANNEALING_EPISODE_COUNT = 10 ** 6, 300        # rank 0 stops on the clock (max_seconds)
    Config.LEARNING_RATE_START, Config.LEARNING_RATE_END = 3e-4, 1e-4      # annealed by rank 0's episode count
    group = DataParallel.EngineGroup.from_env()
    group.WINDOW = 2                                              # a short run: keep rank 0 from granting it all at once
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from Server import Server
    params = o.init_params(6)
    theta = np.concatenate([params[k].reshape(-1) for k in o.PARAM_ORDER]).astype(np.float32)
    log = []
    model = _ShardModel(oc, theta, dist.new_group(backend="gloo"), log)
    srv = Server(model=model, max_agents=4, engine_group=group)
    srv.main(max_seconds=1.5)
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), theta=model.theta, steps=len(log),
             **{"%s%d" % (k, i): v for i, row in enumerate(log) for k, v in zip(("x", "y", "a", "lr", "beta"), row)})
    group.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_product_server_path_shards_steps_and_replicas_stay_equal(tmp_path):
    """Two ranks, each the product's Server + ThreadTrainer + EngineGroup (credit, per-step lr) with its own agents; the model
    is the CPU stand-in above.  Afterwards: both ranks took the same steps with the same lr, their weights are equal bit
    for bit, and they equal a single process that trains each step on the concatenation of the two ranks' rows."""
    import multiprocessing as mp
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ga3c_oracle as o
    import ga3c_oracle_cport as oc
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    z = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    steps = int(z[0]["steps"])
    assert steps == int(z[1]["steps"]) and steps >= 3
    assert np.array_equal(z[0]["theta"], z[1]["theta"])                                   # replicas
    lrs = [float(z[0]["lr%d" % i]) for i in range(steps)]
    assert lrs == [float(z[1]["lr%d" % i]) for i in range(steps)]                          # same lr on the same step
    assert len(set(lrs)) > 1                                                               # ... and the anneal really moved
    params = o.init_params(6)
    theta = np.concatenate([params[k].reshape(-1) for k in o.PARAM_ORDER]).astype(np.float32)
    ms = np.ones_like(theta)
    oc.lib().ga3c_oc_set_threads(4)
    for i in range(steps):
        x = np.concatenate([z[r]["x%d" % i] for r in range(2)])
        assert min(z[r]["x%d" % i].shape[0] for r in range(2)) > 7                         # ThreadTrainer's rule: MORE than MIN rows
        xf = (x.astype(np.float32) / np.float32(128.0) - np.float32(1.0)).reshape(x.shape[0], -1)
        y = np.concatenate([z[r]["y%d" % i] for r in range(2)])
        a = np.concatenate([z[r]["a%d" % i] for r in range(2)])
        oc.train(theta, ms, 6, xf, y, a, lr=lrs[i], beta=float(z[0]["beta%d" % i]))
    assert np.max(np.abs(theta - z[0]["theta"])) < 2e-6
