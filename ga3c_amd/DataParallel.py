"""Data-parallel layout of the train path over the GPUs of one node (no counterpart in the
reference, which drives a single device: ga3c/Config.py:62, NetworkVP.py:50).

One process per GPU.  The reference's loss is a SUM over rows (NetworkVP_discrate.py:61,83-85,100), so
per-rank gradients of disjoint row shards add up exactly to the single-GPU gradient of the whole batch:
each rank runs forward/backward on its shard, the flat gradient arena (4.02 MB, f32) is all-reduced with
op = sum by RCCL inside ga3c_net_train / ga3c_net_apply_grads, and every rank applies the identical
RMSProp step, keeping weights and optimizer slots replicated.  Predictions need no collective: requests
are sharded across ranks and answered from the local replica.

torch.distributed (gloo) is used only as the launcher-side control plane: it carries the 128-byte RCCL
id from rank 0 to the others and provides barriers; the data path never touches it.
"""
import os

import numpy as np


def shard_bounds(rows, rank, world):
    """Contiguous [lo, hi) of `rows` owned by `rank`; the first rows % world ranks get one extra row."""
    base, extra = divmod(int(rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def exchange_comm_id(make_id, rank, nbytes=128):
    """Rank 0 calls make_id() (Network.make_comm_id); every rank returns the same uint8[nbytes] token."""
    import torch
    import torch.distributed as dist
    token = torch.from_numpy(np.ascontiguousarray(make_id(), np.uint8) if rank == 0 else np.zeros(nbytes, np.uint8))
    dist.broadcast(token, src=0)
    return token.numpy()


def attach(net, rank, world):
    """Give `net` an RCCL communicator spanning the process group (call after init_process_group)."""
    if world > 1:
        net.comm_init(exchange_comm_id(type(net).make_comm_id, rank), rank, world)
    return net


class EngineGroup:
    """Lock-step control plane for one Server per GPU (launched with torch.distributed.run).

    Every train step contains an RCCL all-reduce, so all ranks must take exactly the same number of steps.
    Rank 0 decides when to stop; the decision travels as "stop after global step S" over the gloo group, polled
    by every rank's main loop (Server.main, 100 Hz).  Trainer threads take a step only while the model's
    global step is below S, checked under Server.dp_lock, so all ranks end on the same step and none is left
    waiting inside a collective.
    """
    MARGIN = 32      # steps between the decision and the stop, so that every rank hears of it in time

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.stop_step = None

    @classmethod
    def from_env(cls):
        import torch.distributed as dist
        rank, local_rank, world = env_rank_world()
        if world <= 1:
            return None
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return cls(rank, world)

    def poll(self, want_stop, current_step, lr, beta):
        """Collective: call once per main-loop turn on every rank.
        Returns (agreed stop step or None, learning rate, beta) -- the last two are rank 0's, so that every rank
        applies the same optimizer step (the anneal of Server.py:168-175 follows rank 0's episode count)."""
        import torch
        import torch.distributed as dist
        msg = torch.zeros(4, dtype=torch.float64)
        if self.rank == 0:
            if self.stop_step is None and want_stop:
                self.stop_step = int(current_step) + self.MARGIN
            msg[0] = 1.0 if self.stop_step is not None else 0.0
            msg[1] = float(self.stop_step or 0)
            msg[2], msg[3] = float(lr), float(beta)
        dist.broadcast(msg, src=0, group=self.group)
        if msg[0] == 1.0:
            self.stop_step = int(msg[1])
        return self.stop_step, float(msg[2]), float(msg[3])
