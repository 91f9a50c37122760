"""Shared-memory transport between agent processes and the server's batching threads.

Stands where the reference has three pickling multiprocessing.Queues (Server.py:73-75 prediction_q /
training_q, ProcessAgent.py:64 wait_q).  A thin numpy/ctypes face over libga3c_host.so
(include/ga3c_host.h): agents write a state straight into their slot of one POSIX shm segment and
sleep on a futex; predictors pop ids from a lock-free ring, read the slots, answer in place.
"""
import ctypes as C
import os

import numpy as np

import _native as nat

TIMEOUT, CLOSED = -3, -4
REQ_RESET, REQ_NO_PREDICT = 1, 2          # request flags (include/ga3c_host.h: GA3C_REQ_*)


class Transport:
    def __init__(self, handle, owner):
        self._lib = nat.host_lib()
        self._h = handle
        self.owner = owner
        cfg = nat.ShmConfig()
        nat.check_host(self._lib.ga3c_shm_get_config(self._h, C.byref(cfg)))
        self.max_agents, self.num_actions = cfg.max_agents, cfg.num_actions
        self.state_bytes, self.train_slots, self.train_rows = cfg.state_bytes, cfg.train_slots, cfg.train_rows
        self.row_bytes = cfg.rollout_row_bytes or cfg.state_bytes      # bytes of one rollout row
        self.nbytes = self._lib.ga3c_shm_bytes(self._h)
        self.base = self._lib.ga3c_shm_base(self._h)
        self._raw = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.base), dtype=np.uint8)
        off0 = self.state_off0 = self._lib.ga3c_shm_state_offset(self._h, 0)
        stride = self.agent_stride = self._lib.ga3c_shm_agent_stride(self._h)
        # [max_agents, state_bytes] strided view over every agent's state
        self.agent_states = np.lib.stride_tricks.as_strided(self._raw[off0:], shape=(self.max_agents, self.state_bytes),
                                                            strides=(stride, 1), writeable=True)
        self._views = {}
        self._vcells = {}
        self._rcells = {}
        self._round_trip = self._lib.ga3c_pq_round_trip
        self._submit = self._lib.ga3c_pq_submit_flags
        # the same entry point with untyped pointer arguments: an address or a byref() goes through without a POINTER object
        self._wait = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32)(("ga3c_pq_wait", self._lib))
        self._ro_off0 = self._lib.ga3c_shm_rollout_offset(self._h, 0)
        self._ro_stride = self._lib.ga3c_shm_rollout_stride(self._h)

    # ---- lifecycle
    @classmethod
    def create(cls, name, max_agents, num_actions, state_bytes, train_slots, train_rows, rollout_row_bytes=0):
        cfg = nat.ShmConfig(max_agents, num_actions, state_bytes, train_slots, train_rows, rollout_row_bytes)
        h = C.c_void_p()
        nat.check_host(nat.host_lib().ga3c_shm_create(name.encode(), C.byref(cfg), C.byref(h)), "ga3c_shm_create")
        t = cls(h, True)
        t.name = name
        return t

    @classmethod
    def attach(cls, name):
        h = C.c_void_p()
        nat.check_host(nat.host_lib().ga3c_shm_attach(name.encode(), C.byref(h)), "ga3c_shm_attach")
        t = cls(h, False)
        t.name = name
        return t

    def shutdown(self):
        self._lib.ga3c_shm_shutdown(self._h)

    def unlink(self):
        """Remove the segment's name and keep it mapped (ga3c_shm_unlink): a server that ends without close()."""
        if self._h:
            self._lib.ga3c_shm_unlink(self._h)

    def close(self):
        if self._h:
            self._raw = self.agent_states = None
            self._views = {}
            self._lib.ga3c_shm_close(self._h, 1 if self.owner else 0)
            self._h = None

    # ---- agent side of predict (once per agent step: views, the value cell and its reference are kept, the pointers go to
    # C as plain addresses -- 9.3 -> 6.3 us of interpreter / ctypes time per round trip)
    def state_view(self, agent, dtype=np.uint8):
        key = (agent, np.dtype(dtype).char)
        view = self._views.get(key)
        if view is None:
            view = self._views[key] = self.agent_states[agent].view(dtype)
        return view

    def submit(self, agent, flags=0):
        rc = self._submit(self._h, agent, flags)
        return rc if rc in (0, TIMEOUT, CLOSED) else nat.check_host(rc, "ga3c_pq_submit_flags")

    def round_trip(self, agent, state, flags, timeout_ms, u, submit=True):
        """One agent step's conversation with the predictor in one foreign call (ga3c_pq_round_trip): `state` (a contiguous
        array, or None when the slot was filled in place) into the slot, submit, wait, draw the action for uniform `u`
        (u < 0: no draw).  -> (rc, p, v, action index); rc = TIMEOUT leaves the request in flight: call again with
        submit=False."""
        p = np.empty(self.num_actions, np.float32)
        cell = self._rcells.get(agent)
        if cell is None:
            v, a = C.c_float(), C.c_int32()
            cell = self._rcells[agent] = (v, C.addressof(v), a, C.addressof(a))
        rc = self._round_trip(self._h, agent, state.ctypes.data if state is not None else None,
                              state.nbytes if state is not None else 0, flags, 1 if submit else 0, timeout_ms, u,
                              p.ctypes.data, cell[1], cell[3])
        if rc not in (0, TIMEOUT, CLOSED):
            nat.check_host(rc, "ga3c_pq_round_trip")
        return rc, p, cell[0].value, cell[2].value

    def request_flags(self, ids):
        out = np.empty(ids.size, np.uint32)
        nat.check_host(self._lib.ga3c_pq_request_flags(self._h, nat.ptr(ids, nat.u32p), ids.size, nat.ptr(out, nat.u32p)),
                       "ga3c_pq_request_flags")
        return out

    def wait(self, agent, timeout_ms=-1):
        p = np.empty(self.num_actions, np.float32)
        cell = self._vcells.get(agent)
        if cell is None:                       # one value cell per agent id (an id has one request in flight: one waiter)
            v = C.c_float()
            cell = self._vcells[agent] = (v, C.byref(v))
        rc = self._wait(self._h, agent, p.ctypes.data, cell[1], timeout_ms)
        if rc not in (0, TIMEOUT, CLOSED):
            nat.check_host(rc, "ga3c_pq_wait")
        return rc, p, cell[0].value

    def agent_idle(self, agent):
        """True when every request of `agent` has been answered (its id may then be handed to a new agent)."""
        return nat.check_host(self._lib.ga3c_pq_agent_idle(self._h, int(agent)), "ga3c_pq_agent_idle") == 1

    # ---- predictor side
    def pop_batch(self, ids, timeout_ms):
        return nat.check_host(self._lib.ga3c_pq_pop_batch(self._h, nat.ptr(ids, nat.u32p), ids.size, timeout_ms),
                              "ga3c_pq_pop_batch")

    def respond(self, ids, n, p, v):
        nat.check_host(self._lib.ga3c_pq_respond(self._h, nat.ptr(ids, nat.u32p), n, nat.ptr(p), nat.ptr(v)),
                       "ga3c_pq_respond")

    def serve(self, entry, net_handle, u8, max_batch, slice_ms, stats):
        """One time slice of the native predictor loop (ga3c_pq_serve).  `entry` is the address of a function with
        ga3c_net_predict_gather's signature, `stats` a _native.ServeStats that keeps accumulating."""
        rc = self._lib.ga3c_pq_serve(self._h, entry, net_handle, int(u8), int(max_batch), int(slice_ms), C.addressof(stats))
        return nat.check_host(rc, "ga3c_pq_serve")

    def serve_pipelined(self, begin, end, net_handle, u8, max_batch, slice_ms, stats):
        """One time slice of the native predictor loop that answers batch k beside the GPU's work on batch k+1
        (ga3c_pq_serve_pipelined); `begin` / `end` are the addresses of ga3c_net_predict_gather_begin / _end."""
        rc = self._lib.ga3c_pq_serve_pipelined(self._h, begin, end, net_handle, int(u8), int(max_batch), int(slice_ms),
                                               C.addressof(stats))
        return nat.check_host(rc, "ga3c_pq_serve_pipelined")

    def serve_pipelined_cached(self, begin, end, net_handle, u8, max_batch, slice_ms, stats):
        """serve_pipelined for an engine that keeps the states it reads (`begin`: ga3c_net_predict_gather_begin_cached)."""
        rc = self._lib.ga3c_pq_serve_pipelined_cached(self._h, begin, end, net_handle, int(u8), int(max_batch), int(slice_ms),
                                                      C.addressof(stats))
        return nat.check_host(rc, "ga3c_pq_serve_pipelined_cached")

    def request_seq(self, agent):
        """Number of `agent`'s newest request (in flight or answered last): the name of the state it carried."""
        n = C.c_int64()
        nat.check_host(self._lib.ga3c_pq_request_seq(self._h, int(agent), C.byref(n)), "ga3c_pq_request_seq")
        return n.value

    def set_spin(self, spin_us):
        nat.check_host(self._lib.ga3c_pq_set_spin(self._h, int(spin_us)), "ga3c_pq_set_spin")

    def wake_latency(self):
        """{'ready': (answers, mean us, max us), 'slept': (...)}: answer published -> agent back from its wait (ga3c_pq_wake_latency)."""
        out = np.zeros(6, np.int64)
        nat.check_host(self._lib.ga3c_pq_wake_latency(self._h, out.ctypes.data), "ga3c_pq_wake_latency")
        return {k: (int(out[3 * i]), float(out[3 * i + 1]) / max(int(out[3 * i]), 1) / 1e3, float(out[3 * i + 2]) / 1e3)
                for i, k in enumerate(("ready", "slept"))}

    def set_linger(self, linger_us, min_batch):
        nat.check_host(self._lib.ga3c_pq_set_linger(self._h, int(linger_us), int(min_batch)), "ga3c_pq_set_linger")

    def serve_frames(self, entry, net_handle, max_batch, slice_ms, stats):
        """One time slice of the native predictor loop for raw-frame requests (ga3c_pq_serve_frames)."""
        rc = self._lib.ga3c_pq_serve_frames(self._h, entry, net_handle, int(max_batch), int(slice_ms), C.addressof(stats))
        return nat.check_host(rc, "ga3c_pq_serve_frames")

    def serve_frames_pipelined(self, begin, end, net_handle, max_batch, slice_ms, stats):
        """The same loop answering batch k beside the GPU's work on batch k+1 (ga3c_pq_serve_frames_pipelined); `begin` /
        `end` are the addresses of ga3c_net_serve_frames_begin / _end."""
        rc = self._lib.ga3c_pq_serve_frames_pipelined(self._h, begin, end, net_handle, int(max_batch), int(slice_ms),
                                                      C.addressof(stats))
        return nat.check_host(rc, "ga3c_pq_serve_frames_pipelined")

    # ---- training queue
    def rollout_views(self, slot):
        base = self._ro_off0 + slot * self._ro_stride
        states = self._raw[base: base + self.train_rows * self.row_bytes].reshape(self.train_rows, self.row_bytes)
        rp = self._lib.ga3c_tq_returns(self._h, slot)
        ap = self._lib.ga3c_tq_actions(self._h, slot)
        returns = np.frombuffer((C.c_float * self.train_rows).from_address(rp), dtype=np.float32)
        actions = np.frombuffer((C.c_int32 * self.train_rows).from_address(ap), dtype=np.int32)
        return states, returns, actions

    def acquire(self, timeout_ms):
        return nat.check_host(self._lib.ga3c_tq_acquire(self._h, timeout_ms), "ga3c_tq_acquire")

    def commit(self, slot, rows):
        nat.check_host(self._lib.ga3c_tq_commit(self._h, slot, rows), "ga3c_tq_commit")

    def pop_rollout(self, timeout_ms):
        return nat.check_host(self._lib.ga3c_tq_pop(self._h, timeout_ms), "ga3c_tq_pop")

    def rows(self, slot):
        return nat.check_host(self._lib.ga3c_tq_rows(self._h, slot), "ga3c_tq_rows")

    def release(self, slot):
        nat.check_host(self._lib.ga3c_tq_release(self._h, slot), "ga3c_tq_release")

    def state_offsets(self, ids):
        """Byte offsets (into the segment) of the states of agents `ids` -- what the GPU gather consumes."""
        return self.state_off0 + ids.astype(np.int64) * self.agent_stride

    def rollout_row_offsets(self, slot, rows):
        return self._ro_off0 + slot * self._ro_stride + np.arange(rows, dtype=np.int64) * self.row_bytes

    def collect(self, min_rows, timeout_ms, hold_timeout_ms, state, slots, offsets, returns, actions, seqs=None, agents=None):
        """ThreadTrainer's batch assembly in native code (ga3c_tq_collect).  `state` = int32[2]: rows and slots of the batch
        in progress, kept across calls.  Returns 0 (batch complete), -3 (timeout), -4 (closed) or 1 (starved: give slots back).
        seqs / agents: the rows name states kept on the device; they are decoded there and the slots released at once."""
        return nat.check_host(self._lib.ga3c_tq_collect(self._h, min_rows, timeout_ms, hold_timeout_ms, state.ctypes.data,
                                                        state.ctypes.data + 4, slots.ctypes.data, offsets.ctypes.data,
                                                        returns.ctypes.data, actions.ctypes.data, len(returns), len(slots),
                                                        seqs.ctypes.data if seqs is not None else None,
                                                        agents.ctypes.data if agents is not None else None),
                              "ga3c_tq_collect")

    def release_many(self, slots, n):
        nat.check_host(self._lib.ga3c_tq_release_many(self._h, slots.ctypes.data, int(n)), "ga3c_tq_release_many")

    def ready_count(self):
        return self._lib.ga3c_tq_ready_count(self._h)

    def free_count(self):
        return self._lib.ga3c_tq_free_count(self._h)


def unique_name(tag="ga3c"):
    return "/%s_%d_%d" % (tag, os.getpid(), int.from_bytes(os.urandom(3), "little"))


def select_action_index(prediction, u):
    """Index np.random.choice(n, p=prediction) returns when its uniform draw is `u` (ga3c_select_action: the call's own
    float64 cdf / searchsorted arithmetic in C, a third of the numpy calls' time)."""
    return nat.host_lib().ga3c_select_action(prediction.ctypes.data_as(nat.f32p), prediction.size, u)


def accumulate_rewards_fork(rewards, gamma, terminal_reward, discounting=True, use_intermediate_reward=False):
    """ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84) through the C ABI, f64, bit-exact."""
    r = np.ascontiguousarray(rewards, dtype=np.float64)
    out = np.empty_like(r)
    nat.check_host(nat.host_lib().ga3c_returns_fork(nat.ptr(r, nat.f64p), r.size, float(gamma), float(terminal_reward),
                                                    int(bool(discounting)), int(bool(use_intermediate_reward)),
                                                    nat.ptr(out, nat.f64p)), "ga3c_returns_fork")
    return out


def returns_nstep(rewards, gamma, bootstrap_value, rmin=-1.0, rmax=1.0):
    r = np.ascontiguousarray(rewards, dtype=np.float64)
    out = np.empty(max(r.size - 1, 0), np.float64)
    if r.size > 1:
        nat.check_host(nat.host_lib().ga3c_returns_nstep(nat.ptr(r, nat.f64p), r.size, float(gamma), float(bootstrap_value),
                                                         float(rmin), float(rmax), nat.ptr(out, nat.f64p)),
                       "ga3c_returns_nstep")
    return out
