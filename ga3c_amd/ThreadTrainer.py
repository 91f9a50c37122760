"""Training batcher (reference: ga3c/ThreadTrainer.py:33-62): take rollouts from the training
queue until the batch holds MORE than TRAINING_MIN_BATCH_SIZE rows, then one server.train_model().

Rollouts arrive as slots of the shared-memory transport (states, f32 returns, int32 actions);
rows are copied once into a staging batch instead of the reference's repeated np.concatenate.
x2_ and done_ are not transported (unused by the A3C nets, NetworkVP.py:254); train_model gets None.
"""
from threading import Thread

import numpy as np

from Config import Config


class ThreadTrainer(Thread):
    def __init__(self, server, id, transport=None):
        super(ThreadTrainer, self).__init__()
        self.daemon = True
        self.id = id
        self.server = server
        self.transport = transport if transport is not None else server.transport
        self.exit_flag = False

    def run(self):
        t = self.transport
        cap = Config.TRAINING_MIN_BATCH_SIZE + t.train_rows
        state_dim = tuple(self.server.state_dim)
        u8 = t.state_bytes == int(np.prod(state_dim))
        alloc = getattr(self.server.model, "pinned_array", None)
        shape = (cap, t.state_bytes)
        zero_copy = getattr(self.server, "zero_copy", False)
        on_device = getattr(self.server, "device_frontend", False)     # rows name states kept in HBM: (plane seq, agent)
        seq_stage = np.zeros(cap, np.int64)
        agent_stage = np.zeros(cap, np.int32)
        x_stage = None
        if not zero_copy and not on_device:
            x_stage = alloc(shape, np.uint8) if alloc else np.zeros(shape, np.uint8)
        off_stage = np.zeros(cap, np.int64)
        r_stage = np.zeros(cap, np.float32)
        a_stage = np.zeros(cap, np.int32)
        eye = np.eye(t.num_actions, dtype=np.float32)
        while not self.exit_flag:
            batch_size = 0
            held = []                                   # zero-copy: slots stay ours until the GPU has read them
            while batch_size <= Config.TRAINING_MIN_BATCH_SIZE and not self.exit_flag:
                slot = t.pop_rollout(Config.QUEUE_TIMEOUT_MS)
                if slot == -3:
                    continue                            # timeout: look at exit_flag again
                if slot < 0:
                    return                              # transport shut down
                rows = t.rows(slot)
                states, returns, actions = t.rollout_views(slot)
                r_stage[batch_size:batch_size + rows] = returns[:rows]
                a_stage[batch_size:batch_size + rows] = actions[:rows]
                if on_device:
                    seq_stage[batch_size:batch_size + rows] = states[:rows, :8].view(np.int64).ravel()
                    agent_stage[batch_size:batch_size + rows] = states[:rows, 8:12].view(np.int32).ravel()
                    t.release(slot)
                elif zero_copy:
                    off_stage[batch_size:batch_size + rows] = t.rollout_row_offsets(slot, rows)
                    held.append(slot)
                else:
                    x_stage[batch_size:batch_size + rows] = states[:rows]
                    t.release(slot)
                batch_size += rows
            if batch_size and Config.TRAIN_MODELS and not self.exit_flag:
                if on_device:
                    self.server.train_model_frames(agent_stage[:batch_size], seq_stage[:batch_size], r_stage[:batch_size],
                                                   eye[a_stage[:batch_size]], self.id)
                elif zero_copy:
                    self.server.train_model_rows(off_stage[:batch_size], r_stage[:batch_size],
                                                 eye[a_stage[:batch_size]], self.id)
                else:
                    xb = x_stage[:batch_size] if u8 else x_stage[:batch_size].view(np.float32)
                    self.server.train_model(xb.reshape((batch_size,) + state_dim), r_stage[:batch_size],
                                            eye[a_stage[:batch_size]], None, None, self.id)
            for slot in held:
                t.release(slot)
            if self.exit_flag or batch_size == 0:
                break
