"""Data-parallel layout of the train path over the GPUs of one node (no counterpart in the
reference, which drives a single device: ga3c/Config.py:62, NetworkVP.py:50).

One process per GPU.  The reference's loss is a SUM over rows (NetworkVP_discrate.py:61,83-85,100), so
per-rank gradients of disjoint row shards add up exactly to the single-GPU gradient of the whole batch:
each rank runs forward/backward on its shard, the flat gradient arena (4.02 MB, f32) is all-reduced with
op = sum by RCCL inside ga3c_net_train / ga3c_net_apply_grads, and every rank applies the identical
RMSProp step, keeping weights and optimizer slots replicated.  Predictions need no collective: requests
are sharded across ranks and answered from the local replica.

Control plane (this file; no torch, no MPI): the launcher -- `python -m torch.distributed.run`, or anything else that
sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT -- only starts the processes.  Rank 0 listens on an
ephemeral TCP port of MASTER_ADDR and publishes it in a small rendezvous file named after MASTER_ADDR:MASTER_PORT;
the other ranks read the file and connect.  Over those sockets travel the 128-byte RCCL id and, while the engine
runs, rank 0's step credits; the data path (RCCL over xGMI) never touches them.
"""
import atexit
import json
import os
import select
import socket
import struct
import tempfile
import time

import numpy as np

RENDEZVOUS_TIMEOUT_S = 300.0


def _private_dir():
    """Where the rendezvous files live unless GA3C_DP_DIR says otherwise: a directory of this user's own (0700) under the
    temp dir.  All ranks must see the same directory -- one node, or GA3C_DP_DIR on storage the nodes share."""
    d = os.path.join(tempfile.gettempdir(), "ga3c_dp_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("%s is not a private directory of this user; set GA3C_DP_DIR" % d)
    return d


def shard_bounds(rows, rank, world):
    """Contiguous [lo, hi) of `rows` owned by `rank`; the first rows % world ranks get one extra row."""
    base, extra = divmod(int(rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control-plane peer closed the connection")
        buf += chunk
    return bytes(buf)


class Rendezvous:
    """Star of TCP connections, rank 0 in the middle.  `tag` separates several groups that share one launcher (the RCCL
    id exchange of bench.py and the engine group of GA3C.py use different tags)."""

    def __init__(self, rank, world, tag="dp", addr=None, port=None, directory=None):
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = port or os.environ.get("MASTER_PORT", "29500")
        directory = directory or os.environ.get("GA3C_DP_DIR") or _private_dir()
        self.path = os.path.join(directory, "ga3c_rendezvous_%s_%s_%s.json" % (str(addr).replace(":", "_"), port, tag))
        self.peers = []          # rank 0: sockets of ranks 1..world-1, in rank order
        self.up = None           # other ranks: the socket to rank 0
        if self.world <= 1:
            return
        if self.rank == 0:
            self._serve(addr)
        else:
            self._join()

    def _serve(self, addr):
        token = os.urandom(16).hex()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((addr, 0))
        except OSError:
            srv.bind(("127.0.0.1", 0))
        srv.listen(self.world)
        # the file carries the token that admits a rank: readable by its owner only, never written through a link somebody
        # else planted (O_EXCL | O_NOFOLLOW on a fresh name, then an atomic rename over a stale file of an earlier run)
        tmp = "%s.%d.%s.tmp" % (self.path, os.getpid(), os.urandom(4).hex())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            json.dump({"host": srv.getsockname()[0], "port": srv.getsockname()[1], "token": token, "world": self.world}, f)
        os.replace(tmp, self.path)
        atexit.register(self._unlink)
        got = {}
        srv.settimeout(RENDEZVOUS_TIMEOUT_S)
        while len(got) < self.world - 1:
            conn, _ = srv.accept()
            conn.settimeout(30.0)
            try:
                hello = _recv_exact(conn, 36)
                peer_token, peer_rank = hello[:32].decode(), struct.unpack("<i", hello[32:])[0]
                if peer_token != token or not (0 < peer_rank < self.world) or peer_rank in got:
                    conn.close()
                    continue
                conn.sendall(b"OK")
            except (OSError, ConnectionError, UnicodeDecodeError):
                conn.close()
                continue
            conn.settimeout(None)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            got[peer_rank] = conn
        srv.close()
        self.peers = [got[r] for r in range(1, self.world)]

    def _join(self):
        deadline = time.time() + RENDEZVOUS_TIMEOUT_S
        while True:
            try:
                with open(self.path) as f:
                    info = json.load(f)
                if info.get("world") != self.world:
                    raise ValueError("rendezvous file of another job")
                s = socket.create_connection((info["host"], info["port"]), timeout=5.0)
                s.sendall(info["token"].encode() + struct.pack("<i", self.rank))
                if _recv_exact(s, 2) != b"OK":
                    raise ConnectionError("rendezvous refused")
                s.settimeout(None)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.up = s
                return
            except (OSError, ValueError, ConnectionError, KeyError):
                # no file yet, or the file of an earlier run whose rank 0 is gone: rank 0 will (re)write it
                if time.time() > deadline:
                    raise TimeoutError("no rendezvous with rank 0 through %s" % self.path)
                time.sleep(0.05)

    def _unlink(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass

    # ---- rank 0 -> everyone
    def broadcast(self, payload=None, nbytes=0):
        """Rank 0 passes `payload` (bytes); every rank returns it."""
        if self.world <= 1:
            return payload
        if self.rank == 0:
            for s in self.peers:
                s.sendall(payload)
            return payload
        return _recv_exact(self.up, nbytes)

    def barrier(self):
        """Everyone -> rank 0 -> everyone."""
        if self.world <= 1:
            return
        if self.rank == 0:
            for s in self.peers:
                _recv_exact(s, 1)
            for s in self.peers:
                s.sendall(b"B")
        else:
            self.up.sendall(b"b")
            _recv_exact(self.up, 1)

    def reduce(self, values, op="max"):
        """Element-wise max / sum of a short list of floats over the ranks (everyone -> rank 0 -> everyone); the control
        plane's own little all-reduce: bench.py's max-over-ranks timing and its error flags need nothing bigger."""
        vals = [float(v) for v in values]
        if self.world <= 1:
            return vals
        fmt = struct.Struct("<%dd" % len(vals))
        if self.rank == 0:
            rows = [vals] + [list(fmt.unpack(_recv_exact(s, fmt.size))) for s in self.peers]
            out = [max(c) if op == "max" else sum(c) for c in zip(*rows)]
            for s in self.peers:
                s.sendall(fmt.pack(*out))
            return out
        self.up.sendall(fmt.pack(*vals))
        return list(fmt.unpack(_recv_exact(self.up, fmt.size)))

    def close(self):
        for s in self.peers + ([self.up] if self.up else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.up = [], None
        if self.rank == 0 and self.world > 1:
            self._unlink()


def exchange_comm_id(make_id, rank, nbytes=128, world=None, rendezvous=None):
    """Rank 0 calls make_id() (Network.make_comm_id); every rank returns the same uint8[nbytes] token."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    rv = rendezvous or Rendezvous(rank, world, tag="rcclid")
    try:
        payload = np.ascontiguousarray(make_id(), np.uint8).tobytes() if rank == 0 else None
        return np.frombuffer(rv.broadcast(payload, nbytes), dtype=np.uint8).copy()
    finally:
        if rendezvous is None:
            rv.barrier()         # rank 0 keeps the file until everybody has the id
            rv.close()


def attach(net, rank, world, rendezvous=None):
    """Give `net` an RCCL communicator spanning the ranks of the launcher."""
    if world > 1:
        net.comm_init(exchange_comm_id(type(net).make_comm_id, rank, world=world, rendezvous=rendezvous), rank, world)
    return net


_MSG = struct.Struct("<qqddi4x")          # credit, from_step, lr, beta, stop (2: abort, from_step = the late rank)
_UP = struct.Struct("<qq")                # rank r -> rank 0: steps finished, steps started


class GroupStalled(RuntimeError):
    """A rank has kept the others waiting inside a collective (or for credit) for longer than EngineGroup.STALL_S."""


class EngineGroup:
    """Lock-step control plane for one Server per GPU.

    Every train step contains an RCCL all-reduce, so all ranks must take exactly the same number of steps, with the
    same learning rate and beta.  Rank 0 hands out CREDIT: "steps up to G may be taken, and steps after F use (lr,
    beta)".  A rank never starts a step beyond the credit it has received, so it can never enter a collective that the
    others will not join; a rank that hears late (its main loop was delayed) stalls until the message arrives instead
    of hanging.  Stopping = rank 0 no longer extends the credit and says so; every rank leaves when it has taken step
    G.  The learning rate of a step is a function of the step alone (the messages arrive in order on every rank), so
    the replicas apply identical updates.
    """
    WINDOW = 64          # least number of steps of credit ahead of rank 0's own step count
    STALL_S = 120.0      # a rank that keeps the others inside a collective this long is named and the group stops

    def __init__(self, rank, world, rendezvous=None):
        self.rank, self.world = rank, world
        self.rv = rendezvous if rendezvous is not None else Rendezvous(rank, world, tag="engine")
        self.credit = 0                  # steps 1..credit may be taken
        self.stopping = False            # the credit is final
        self.schedule = []               # (from_step, lr, beta): steps > from_step use lr, beta (until the next entry)
        self._last_step, self._pace = 0, 0
        # progress as each rank reports it: a train step that contains an all-reduce completes on every rank or on none,
        # so "finished" counts never differ -- the rank that is late is the one that has not STARTED the step the others
        # are waiting in.  Server bumps `started` (under dp_lock) right before it calls the model.
        self.STALL_S = float(os.environ.get("GA3C_DP_STALL_S", self.STALL_S))
        self.started = 0
        self.failure = None              # GroupStalled once the group has been given up (every rank gets the same text)
        now = time.time()
        self._peer = [[0, 0, None] for _ in range(world)]    # rank 0: [finished, started, since when behind the others]
        self._heard = now                                    # other ranks: when rank 0 last spoke

    @classmethod
    def from_env(cls):
        rank, _, world = env_rank_world()
        if world <= 1:
            return None
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return cls(rank, world)

    # ---- what Server asks
    def may_step(self, training_step):
        return training_step < self.credit

    def finished(self, training_step):
        return self.stopping and training_step >= self.credit

    def rates_for(self, step):
        """(lr, beta) of the `step`-th train step (1-based), or None before the first credit arrived."""
        out = None
        for from_step, lr, beta in self.schedule:
            if from_step < step:
                out = (lr, beta)
            else:
                break
        return out

    def _apply(self, credit, from_step, lr, beta, stop):
        if not self.schedule or self.schedule[-1][1:] != (lr, beta):
            self.schedule.append((from_step, lr, beta))
            if len(self.schedule) > 4096:        # steps before the oldest entries are long past
                del self.schedule[:2048]
        self.credit = max(self.credit, credit)
        self.stopping = self.stopping or bool(stop)

    def note_started(self, step):
        """Server: this rank is about to enter train step `step` (1-based)."""
        self.started = max(self.started, step)

    def _late_rank(self, now):
        """Rank 0: the rank that has not started the step the others have been waiting in for STALL_S, or None."""
        ahead = max(p[1] for p in self._peer)
        late = []
        for r, p in enumerate(self._peer):
            if p[1] >= ahead:
                p[2] = None
            elif p[2] is None:
                p[2] = now                       # from now on somebody waits for this rank
            elif now - p[2] > self.STALL_S:
                late.append((p[1], r))
        return min(late)[1] if late else None

    def _give_up(self, late, now):
        p = self._peer[late]
        self.failure = GroupStalled("rank %d has not started train step %d for %.0f s while the other ranks wait for it in "
                                    "that step's all-reduce (it has finished %d steps)"
                                    % (late, max(q[1] for q in self._peer), now - p[2], p[0]))
        self.stopping = True
        msg = _MSG.pack(self.credit, late, 0.0, 0.0, 2)
        for s in self.rv.peers:
            try:
                s.sendall(msg)
            except OSError:
                pass
        raise self.failure

    def poll(self, want_stop, current_step, lr, beta):
        """Call once per main-loop turn (Server.main).  Rank 0 extends the credit and announces it; the other ranks take
        in whatever has arrived and report their progress.  Never blocks.  Raises GroupStalled (on every rank, naming the
        same late rank) when a rank has kept the group waiting for STALL_S."""
        if self.failure is not None:
            raise self.failure
        now = time.time()
        if self.rank == 0:
            self._peer[0][0], self._peer[0][1] = current_step, self.started
            for r, s in enumerate(self.rv.peers, start=1):
                while True:
                    ready, _, _ = select.select([s], [], [], 0)
                    if not ready:
                        break
                    try:
                        fin, sta = _UP.unpack(_recv_exact(s, _UP.size))
                    except (ConnectionError, OSError):
                        break                    # that rank has left (after the stop, or it died: the stall rule names it)
                    self._peer[r][0], self._peer[r][1] = fin, sta
            late = self._late_rank(now)
            if late is not None:
                self._give_up(late, now)
            if not self.stopping:
                self._pace = max(self._pace // 2, current_step - self._last_step)      # steps per poll, recent peak
                self._last_step = current_step
                credit = max(self.credit, current_step + max(self.WINDOW, 4 * self._pace))
                if want_stop:
                    credit = max(self.credit, current_step)     # what is granted stays granted; nothing is added
                msg = _MSG.pack(credit, self.credit, float(lr), float(beta), 1 if want_stop else 0)
                self._apply(credit, self.credit, float(lr), float(beta), want_stop)
                for s in self.rv.peers:
                    s.sendall(msg)
        else:
            if self.rv.up is not None:
                try:
                    self.rv.up.sendall(_UP.pack(current_step, self.started))
                except OSError:
                    pass                         # rank 0 has left: the receive below says whether that was in order
            while self.rv.up is not None:
                ready, _, _ = select.select([self.rv.up], [], [], 0)
                if not ready:
                    break
                try:
                    raw = _recv_exact(self.rv.up, _MSG.size)
                except ConnectionError:
                    if not self.stopping:
                        raise                    # rank 0 is gone without having said stop
                    self.rv.close()              # rank 0 has left after its last message: nothing more will come
                    break
                self._heard = now
                credit, from_step, mlr, mbeta, stop = _MSG.unpack(raw)
                if stop == 2:
                    self.stopping = True
                    self.failure = GroupStalled("rank 0 stopped the group: rank %d did not start a train step the other "
                                                "ranks were waiting in for %.0f s" % (from_step, self.STALL_S))
                    raise self.failure
                self._apply(credit, from_step, mlr, mbeta, stop)
            if self.rv.up is not None and not self.stopping and now - self._heard > self.STALL_S:
                self.failure = GroupStalled("rank %d has heard nothing from rank 0 for %.0f s (no step credit)"
                                            % (self.rank, now - self._heard))
                raise self.failure
        return self.credit

    def close(self):
        self.rv.close()
