"""Two Servers in lock step (one per rank, as under torch.distributed.run), CPU only: stand-in model whose train()
blocks in a gloo barrier the way the real one blocks in the RCCL all-reduce.  Both ranks must take the same number
of train steps and shut down without leaving the other inside a collective."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _CollectiveStandIn:
    def __init__(self, n_act, group):
        self.n_act, self.group = n_act, group
        self.learning_rate = self.beta = 0.0
        self.steps, self.lrs = 0, []

    def predict_p_and_v(self, x):
        b = x.shape[0]
        return np.full((b, self.n_act), 1.0 / self.n_act, np.float32), np.zeros(b, np.float32)

    def train(self, x, y_r, a, x2, done, tid):
        import torch.distributed as dist
        dist.barrier(group=self.group)          # stands for ncclAllReduce inside ga3c_net_train
        self.steps += 1
        self.lrs.append(self.learning_rate)

    def save(self, episode):
        pass

    def log(self, *a, **k):
        pass


def _rank_main(rank, world, port, tmp, out_q):
    sys.path.insert(0, ROOT)
    os.chdir(tmp)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    import ga3c_amd  # noqa: F401
    from Config import Config
    import DataParallel
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 2, 1, 2
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 30, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, False
    Config.PRINT_STATS_FREQUENCY = 10 ** 9
    Config.RESULTS_FILENAME = "results_rank%d.txt" % rank
    Config.LEARNING_RATE_START = Config.LEARNING_RATE_END = 1e-3 * (rank + 1)   # deliberately different: rank 0's must win
    group = DataParallel.EngineGroup.from_env()
    data_group = dist.new_group(backend="gloo")
    from Server import Server
    model = _CollectiveStandIn(6, data_group)
    srv = Server(model=model, max_agents=4, engine_group=group)
    srv.main(max_seconds=2.0 + rank)                      # ranks want to stop at different times
    out_q.put((rank, srv.training_step, model.steps, sorted(set(model.lrs[-5:]))))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_servers_stop_on_the_same_train_step(tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, steps0, m0, lr0), (r1, steps1, m1, lr1) = res
    assert steps0 == steps1 == m0 == m1 and steps0 > 10
    assert lr0 == lr1 == [1e-3]                           # rank 0's learning rate reached rank 1
