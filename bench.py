#!/usr/bin/env python3
"""bench.py -- the reference's headline metric (predictions/s + training-steps/s, 84x84x4 stacked
frames, NetworkVP) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch that is already resident in HBM:
  predict leg (the headline `value`): one ThreadPredictor batch (BASELINE.json configs[1]:
      batch = 128 states) through the HIP NetworkVP forward -> p, v.  The K steps are dealt round-robin to
      NP prediction lanes = predictor threads, each lane with its own HIP stream and workspace exactly as
      ThreadPredictor uses them.  NP = 2 (--predictors): the reference's and this package's default
      Config.PREDICTORS = 2 (Config.py:57, README.md:28-32 "NP: 2"), on the HIP runtime's default number of
      hardware queues -- the configuration a Server with Config defaults runs.  The 1-, 3- and 4-lane figures of
      the same K steps are reported beside it under "predict_lanes", and the 8-hardware-queue variant (a process-wide
      runtime setting the package does not make) under "predict_lanes_8_hw_queues", measured in a child process;
  train leg (reported under "train"): one ThreadTrainer batch (configs[2]: 128 rows) through forward,
      loss, backward, RCCL all-reduce of the gradient arena when N > 1, RMSProp.
Per-GPU work is fixed as N grows (weak scaling); predictions need no collective.
No torch in this process: the device sync is ga3c_net_sync (hipStreamSynchronize on every stream of the
network), the barrier and the max over ranks travel over the package's own control plane (DataParallel.Rendezvous:
TCP sockets found through MASTER_ADDR / MASTER_PORT); the product path is libga3c_hip.so via ctypes.
`value` / `ms_per_step` are HOST WALL-CLOCK time of the bracketed K-step block (rounds 1 and 2's definition; round 3
quoted the GPU event span, which is now the extra `gpu_span_ms_per_step` / `value_gpu_span`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this stack
# GPU_MAX_HW_QUEUES is NOT touched: every leg of this process runs on the runtime's default, like the package itself

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_PREDICTION = {4: 7_580_160, 6: 7_581_184, 18: 7_587_328}
FLOP_CONV1_PER_SAMPLE = 3_612_672                                       # 441 * 256 * 16 * 2
FLOP_CONV2_PER_SAMPLE = 1_982_464                                       # 121 * 256 * 32 * 2
HBM_PEAK_GBS = 8000.0                                                   # MI355X_MICROARCH.md, HBM3E
MFMA_F32_PEAK_TFLOPS = 157.3                                            # MI355X_MICROARCH.md, f32-input MFMA


def run_engine(model, seconds, agents, B, A, frames="planes", predictors=2, trainers=2, min_batch=None, engine_group=None):
    """The whole engine for `seconds`: synthetic agents -> shm transport -> ThreadPredictor / ThreadTrainer -> `model`
    (None = the HIP Network).  Returns rates over the steady window of the run."""
    import threading
    from Config import Config
    from Server import Server
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = agents, predictors, trainers
    Config.TRAINING_MIN_BATCH_SIZE = B - 1 if min_batch is None else min_batch
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS, Config.TENSORBOARD = False, False, False
    Config.PRINT_STATS_FREQUENCY = 10 ** 9
    Config.RESULTS_FILENAME = os.devnull
    Config.NUM_ACTIONS = A
    if frames in ("planes", "planes-device"):
        Config.FRAME_SOURCE, Config.FRONTEND = "planes", ("device" if frames == "planes-device" else "host")
    else:
        Config.FRAME_SOURCE, Config.FRONTEND = "rgb", frames.split("-")[1]
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    try:
        srv = Server(model=model, max_agents=agents, engine_group=engine_group)
        marks, native = [], []

        def sampler():      # steady window: from 40% of the run (agents forked, queues warm) to just before the stop
            for at in (0.4 * seconds, seconds - 0.3):
                time.sleep(max(0.0, at - (time.perf_counter() - t0)))
                marks.append((time.perf_counter(), srv.predictions_served, srv.training_step, srv.frame_counter,
                              sum(p.batches for p in srv.predictors)))
                native.append(any(p.native for p in srv.predictors))

        t0 = time.perf_counter()
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        srv.main(max_seconds=seconds)
        dt = time.perf_counter() - t0
        th.join(timeout=5)
        whole = {"predictions_per_sec": srv.predictions_served / dt, "training_steps_per_sec": srv.training_step / dt,
                 "seconds": dt}
        if len(marks) == 2 and marks[1][0] > marks[0][0]:
            (ta, pa, sa, fa, ba), (tb, pb, sb, fb, bb) = marks
            w = tb - ta
            steady = {"predictions_per_sec": (pb - pa) / w, "training_steps_per_sec": (sb - sa) / w, "seconds": w,
                      "train_rows_per_step": (fb - fa) / max(sb - sa, 1), "mean_predict_batch": (pb - pa) / max(bb - ba, 1)}
        else:
            steady = dict(whole, train_rows_per_step=srv.frame_counter / max(srv.training_step, 1), mean_predict_batch=None)
        res = dict(steady, agents=agents, predictors=predictors, trainers=trainers, whole_run=whole,
                   native_predictor_loop=bool(native and native[0]),
                   rollouts_name_their_states=bool(getattr(srv, "state_cache", False) or getattr(srv, "device_frontend", False)),
                   lost_train_batches=getattr(srv, "lost_train_batches", 0),
                   cpu_placement=(getattr(srv, "placement", None) or {}).get("why"))
        if model is None:
            srv.model.close()
        return res
    finally:
        sys.stdout = real_stdout


class CPortModel:
    """oracle/ga3c_oracle_c.c behind the Network surface the Server drives (cpu_baseline leg only)."""

    def __init__(self, oc, theta, A):
        self.oc, self.theta, self.ms, self.A = oc, theta.copy(), np.ones_like(theta), A
        self.learning_rate, self.beta = 3e-4, 0.01
        self.lock = __import__("threading").Lock()

    @staticmethod
    def _f32(x):
        return x.astype(np.float32) / np.float32(128.0) - np.float32(1.0) if x.dtype == np.uint8 else x

    def predict_p_and_v(self, x):
        return list(self.oc.predict(self.theta, self.A, self._f32(x).reshape(x.shape[0], -1)))

    def train(self, x, y_r, a, x2=None, done=None, trainer_id=0):
        with self.lock:
            self.oc.train(self.theta, self.ms, self.A, self._f32(x).reshape(x.shape[0], -1), y_r, a, self.learning_rate, self.beta)

    def get_global_step(self):
        return 0

    def log(self, *a, **k):
        pass

    def save(self, episode):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--actions", type=int, default=6)
    ap.add_argument("--predictors", type=int, default=2, help="prediction lanes the K predict steps are dealt to (predictor threads; "
                                                               "Config.PREDICTORS = 2 is the reference's and the package's default)")
    ap.add_argument("--lanes-only", action="store_true",
                    help="child mode of the 8-hardware-queue extra: print only {lanes: predictions/s} for 1..4 lanes and exit")
    ap.add_argument("--no-lane-sweep", action="store_true",
                    help="skip the extra 1- and 3-lane legs (use with --predictors 1 under rocprofv3 so that kernels never overlap)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample length; 0 disables it")
    ap.add_argument("--e2e-seconds", type=float, default=8.0,
                    help="also run the whole engine (agent processes -> transport -> predictor/trainer threads) this long; 0 disables")
    ap.add_argument("--e2e-agents", type=int, default=32)
    ap.add_argument("--device-override", type=int, default=-1,
                    help="rehearsal only: put every rank on this device (RCCL then refuses duplicate GPUs and the "
                         "train leg is reported as null)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        world, rank, local_rank = 1, 0, 0

    import ga3c_amd  # noqa: F401
    from Config import Config
    from NetworkVP import Network
    import _native as nat
    import DataParallel
    import Placement
    ndev = nat.C.c_int32()
    if nat.hip_lib().ga3c_device_count(nat.C.byref(ndev)) != 0 or ndev.value < 1:
        sys.exit("bench.py needs an MI355X: no HIP device")
    if args.device_override >= 0:
        local_rank = args.device_override
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rv = DataParallel.Rendezvous(rank, world, tag="bench")      # barrier + max over ranks (no-ops at N = 1)
    # the product's CPU placement (Server.__init__ does the same): this process's threads next to its GPU
    placement = Placement.place(os.environ.get("GA3C_CPU_AFFINITY") or Config.CPU_AFFINITY, local_rank)

    B, A, K, W = args.batch, args.actions, args.steps, args.warmup
    Config.PREDICTION_BATCH_SIZE = B
    NP = max(1, args.predictors)
    # ThreadTrainer assembles TRAINING_MIN_BATCH_SIZE + 1 .. + TIME_MAX + 1 rows (ThreadTrainer.py:49-59): with the
    # "batch = 128" setting (MIN = 127) that is 128 .. 132 + 1; the extra train leg below times the 132-row step
    TB = B + Config.TIME_MAX - 1
    net = Network("gpu:%d" % local_rank, "bench", A, (84, 84, 4), max_batch=TB, predict_lanes=max(NP, 4))
    lib, h = net._lib, net._h

    dp_error = None
    if world > 1:
        try:
            DataParallel.attach(net, rank, world)      # RCCL communicator; the 128-byte id travels over the control plane
        except RuntimeError as e:                       # predictions need no collective: keep the headline leg alive
            dp_error = str(e)
        if rv.reduce([0 if dp_error is None else 1])[0] and dp_error is None:
            dp_error = "RCCL communicator failed on another rank"

    # synthetic inputs of the reference's shape and value set
    rng = np.random.Generator(np.random.PCG64(12345 + rank))
    x_tb = rng.integers(0, 256, size=(TB, 84, 84, 4), dtype=np.uint8).astype(np.float32) / np.float32(128) - np.float32(1)
    act_tb = np.eye(A, dtype=np.float32)[rng.integers(0, A, TB)]
    y_r_tb = rng.uniform(-1, 1, TB).astype(np.float32)
    x, act, y_r = x_tb[:B], act_tb[:B], y_r_tb[:B]
    nat.check(lib.ga3c_net_upload(h, nat.ptr(x_tb), nat.ptr(y_r_tb), nat.ptr(act_tb), TB), "upload")
    lr, beta = float(Config.LEARNING_RATE_START), float(Config.BETA_START)

    def device_sync(handle=None):
        nat.check(lib.ga3c_net_sync(handle or h), "sync")

    def barrier_sync():
        rv.barrier()
        device_sync()

    def timed_block(mode, steps, lanes=0, rows=None):
        """EXACTLY `steps` steps between barrier + synchronize on both sides.  Returns (GPU span, host wall), both in
        seconds and both the max over ranks.  Host wall = the clock around the block, from behind the first barrier + sync to
        behind the closing sync: what `value` is computed from.  GPU span = HIP events on the streams the steps run on: first
        start event to last end event of the block (ga3c_net_last_lanes_gpu_ms; the train stream's own event pair for the
        train leg).  The wall clock also contains the launch latency of the block's first kernel, the wake-up of the
        synchronising host threads and the Python / ctypes calls: tens of microseconds, a seventh of a 20-step block of this
        path and nothing of a 300-step one."""
        ev_ms, gpu_ms = nat.C.c_float(), nat.C.c_float()
        rows = B if rows is None else rows
        barrier_sync()
        t0 = time.perf_counter()
        if lanes:
            nat.check(lib.ga3c_net_time_predict_lanes(h, rows, steps, lanes, nat.C.byref(ev_ms)), "time_predict_lanes")
        else:
            nat.check(lib.ga3c_net_time_resident(h, mode, rows, steps, lr, beta, nat.C.byref(gpu_ms)), "time_resident")
        device_sync()
        t1 = time.perf_counter()
        if lanes:                                          # (the events' arithmetic: after the clock has been read)
            nat.check(lib.ga3c_net_last_lanes_gpu_ms(h, nat.C.byref(gpu_ms)), "last_lanes_gpu_ms")
        gpu_s, wall_s = gpu_ms.value * 1e-3, t1 - t0
        # (the lanes' copies of the resident batch are made once per uploaded batch, by the warm-up call: nothing but the K
        # steps sits between the two synchronisations.  ga3c_net_time_predict_lanes also reads the host clock itself, from
        # the release of the lanes to the last lane's wait: the `inner` figure, a few microseconds of call overhead shorter)
        inner_s = ev_ms.value * 1e-3 if lanes else wall_s
        if world > 1:
            rv.barrier()
            gpu_s, wall_s, inner_s = rv.reduce([gpu_s, wall_s, inner_s])
        inner_walls.append(inner_s)
        return gpu_s, wall_s

    MIN_TIMED_S, MAX_BLOCKS = 0.05, 400
    blocks_used = {}
    inner_walls = []

    def timed(mode, steps, lanes=0, tag=None, rows=None):
        """A K-step block of this path lasts well under a millisecond at the driver's K = 20: the bracketed K-step block is
        repeated until at least 50 ms have been timed and the MEDIAN block is reported (every rank runs the same number
        of blocks: the count is decided on rank 0's clock).  Returns (median GPU span, median host wall) in seconds."""
        gpus, walls, total = [], [], 0.0
        while True:
            g, w = timed_block(mode, steps, lanes, rows)
            gpus.append(g)
            walls.append(w)
            total += w
            more = 1 if (total < MIN_TIMED_S and len(walls) < MAX_BLOCKS) else 0
            if world > 1:                                  # the count is decided on rank 0's clock
                more = int(rv.reduce([more if rank == 0 else 0])[0])
            if not more:
                break
        if tag:
            blocks_used[tag] = len(walls)
        return float(np.median(gpus)), float(np.median(walls))

    ev_ms = nat.C.c_float()
    for mode in (0, 1):
        if W > 0:
            nat.check(lib.ga3c_net_time_resident(h, mode, B, W, lr, beta, nat.C.byref(ev_ms)), "warmup")
    if W > 0:
        nat.check(lib.ga3c_net_time_predict_lanes(h, B, W, NP, nat.C.byref(ev_ms)), "warmup")
    if args.lanes_only:      # child of the 8-hardware-queue extra (see below): the lane sweep and nothing else
        res = {str(nl): K * B / timed(0, K, lanes=nl)[1] for nl in (1, 2, 3, 4)}
        net.close()
        print(json.dumps(dict(res, hardware_queues=os.environ.get("GPU_MAX_HW_QUEUES"))))
        return
    del inner_walls[:]
    pred_gpu_s, pred_s = timed(0, K, lanes=NP, tag="predict")      # pred_s: host wall-clock of the median bracketed block
    pred_inner_s = float(np.median(inner_walls))
    one_s = timed(0, K, lanes=1)[1]
    sweep = {}
    if not args.no_lane_sweep:
        for nl in (2, 3, 4):
            if nl != NP:
                sweep[nl] = timed(0, K, lanes=nl)[1]
    dp_guard = None
    if world > 1:
        # the collective legs below have never run with N > 1 on hardware in this build's development (one-GPU box): if a
        # rank never returns from one, the job must still end and leave the driver its line, with the prediction leg measured
        import threading

        def give_up_dp():
            if rank == 0:
                print(json.dumps({
                    "metric": "predictions_per_sec", "value": world * K * B / pred_s, "unit": "predictions/s", "n_gpus": world,
                    "steps": K, "warmup": W, "ms_per_step": pred_s / K * 1e3, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "PongDeterministic-v4 84x84x4 stacked frames, NetworkVP forward, predictor batch=%d, "
                                           "NP=%d predictor lanes per GPU, A=%d (BASELINE configs[1])" % (B, NP, A),
                               "global_batch": world * B, "parallelism": "dp%d" % world},
                    "train": None, "roofline": None, "cpu_baseline": None,
                    "error": "a data-parallel (RCCL) leg did not finish within 300 s; predictions need no collective and are reported"}),
                    flush=True)
            os._exit(3)      # a process that gives up on a GPU collective must not look like a clean run (every rank exits non-zero)
        dp_guard = threading.Timer(300.0, give_up_dp)
        dp_guard.daemon = True
        dp_guard.start()
    if dp_error is None:
        train_gpu_s, train_s = timed(1, K, tag="train")
        train_tb_gpu_s, train_tb_s = timed(1, K, rows=TB)   # what the engine's trainers really assemble at MIN = B - 1
    else:                                               # no communicator: the data-parallel train leg is not measured
        train_s, train_gpu_s, train_tb_s, train_tb_gpu_s = None, None, None, None
    allreduce_us = None
    if world > 1 and dp_error is None:      # the exchange step alone: 4.02 MB f32 sum all-reduce, events on the train stream
        barrier_sync()
        nat.check(lib.ga3c_net_time_allreduce(h, 50, nat.C.byref(ev_ms)), "time_allreduce")
        allreduce_us = ev_ms.value / 50 * 1e3
    # the engine's own intake format: uint8 frames resident in HBM, converted inside the conv kernels (extra figure)
    xk = np.ascontiguousarray(((x_tb + np.float32(1)) * np.float32(128)).astype(np.uint8))
    nat.check(lib.ga3c_net_upload_u8(h, nat.ptr(xk, nat.u8p), nat.ptr(y_r_tb), nat.ptr(act_tb), TB), "upload_u8")
    nat.check(lib.ga3c_net_time_predict_lanes(h, B, max(W, 1), NP, nat.C.byref(ev_ms)), "warmup")
    u8_s = timed(0, K, lanes=NP)[1]
    u8_train_s = timed(1, K)[1] if dp_error is None else None
    u8_train_tb_s = timed(1, K, rows=TB)[1] if dp_error is None else None
    nat.check(lib.ga3c_net_upload(h, nat.ptr(x_tb), nat.ptr(y_r_tb), nat.ptr(act_tb), TB), "upload")
    comm_ranks, comm_rank, comm_dev = net.comm_info()
    if dp_guard is not None:
        dp_guard.cancel()

    out = None
    if rank == 0:
        pps = world * K * B / pred_s
        tps = K / train_s if train_s else None  # synchronous data-parallel: one global step per K
        out = {
            "metric": "predictions_per_sec", "value": pps, "unit": "predictions/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": pred_s / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PongDeterministic-v4 84x84x4 stacked frames, NetworkVP forward, "
                                   "predictor batch=%d, NP=%d predictor lanes per GPU, A=%d (BASELINE configs[1])" % (B, NP, A),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "inputs": "resident in HBM (f32 NHWC), outputs p,v left in HBM"},
            "predict_lanes": dict({"1": world * K * B / one_s, str(NP): pps},
                                  **{str(nl): world * K * B / s_ for nl, s_ in sweep.items()},
                                  unit="predictions/s", hardware_queues=os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                                  note="same K steps dealt to 1 .. 4 prediction lanes (one persistent host thread per lane); "
                                       "`value` is the NP = %d figure, the default of the reference and of this package; lanes "
                                       "beyond three share the three prediction streams (ga3c_net_create: the engine keeps to "
                                       "four normal-priority streams)" % NP),
            "gpu_span_ms_per_step": pred_gpu_s / K * 1e3, "value_gpu_span": world * K * B / pred_gpu_s,
            "inner_wall_ms_per_step": pred_inner_s / K * 1e3,
            "timing": "ms_per_step / value: HOST WALL-CLOCK (time.perf_counter) of the median bracketed K-step block, max over "
                      "ranks: barrier + device synchronisation, the K steps, device synchronisation -- rounds 1 and 2's "
                      "definition; launch latency of the first kernel, the wake-up of the lanes' host threads and of the "
                      "waiting caller are in it.  Each lane's copy of the resident batch is made once per uploaded batch (by "
                      "the warm-up call), not inside the bracket (until round 4 it was re-made by every timed call: +2 us per "
                      "step at K = 20).  gpu_span_ms_per_step / value_gpu_span: HIP events on the lanes' streams, first start to "
                      "last end (round 3's headline).  inner_wall_ms_per_step: the host clock read inside "
                      "ga3c_net_time_predict_lanes, from the release of the lanes to the last lane's wait",
            "cpu_placement": placement, "torch_imported": "torch" in sys.modules,
            "train": {"metric": "training_steps_per_sec", "value": tps, "unit": "steps/s",
                      "ms_per_step": train_s / K * 1e3 if train_s else None, "rows_per_step": world * B,
                      "gpu_span_ms_per_step": train_gpu_s / K * 1e3 if train_gpu_s else None,
                      "trained_samples_per_sec": world * K * B / train_s if train_s else None,
                      "workload": "forward+loss+backward%s+RMSProp, %d rows per GPU (BASELINE configs[2])"
                                  % ("+RCCL all-reduce(sum) of 4.02 MB grads" if world > 1 else "", B),
                      "train_%d" % TB: {"rows_per_step": world * TB, "ms_per_step": train_tb_s / K * 1e3 if train_tb_s else None,
                                         "gpu_span_ms_per_step": train_tb_gpu_s / K * 1e3 if train_tb_gpu_s else None,
                                         "steps_per_sec": K / train_tb_s if train_tb_s else None,
                                         "note": "the largest batch ThreadTrainer assembles at TRAINING_MIN_BATCH_SIZE = %d "
                                                 "(ThreadTrainer.py:49-59): MIN + TIME_MAX rows" % (B - 1)}},
            "timed_blocks": dict(blocks_used, min_timed_ms=MIN_TIMED_S * 1e3,
                                 note="the bracketed K-step block is repeated until >= 50 ms are timed; ms_per_step and "
                                      "value are those of the MEDIAN block"),
            "data_parallel_error": dp_error,
            "rccl_ranks": comm_ranks if world > 1 else 1,
            "rccl_comm": {"ranks": comm_ranks, "rank": comm_rank, "device": comm_dev,
                          "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the attached communicator "
                                    "(ga3c_net_comm_info); 0 / -1 / -1 = no communicator (N = 1 attaches none)"},
            "allreduce_us": allreduce_us,
            "allreduce_note": "average of 50 back-to-back ncclAllReduce(sum, f32, 1,005,623 elements = 4.02 MB) on the train "
                              "stream, HIP events around them; inside a train step the dense1/w part overlaps with the conv "
                              "backward kernels" if world > 1 else None,
            "uint8_resident": {"predictions_per_sec": world * K * B / u8_s,
                               "training_steps_per_sec": K / u8_train_s if u8_train_s else None,
                               "training_steps_per_sec_%d_rows" % TB: K / u8_train_tb_s if u8_train_tb_s else None,
                               "note": "same legs with the batch resident as uint8 frames (28,224 B per state), the "
                                       "format the shared-memory transport delivers; bit-identical results"},
        }

    # ---- roofline of the dominant kernel (conv1 forward: 48% of the forward FLOPs), rank 0 only
    if rank == 0:
        kernels = {}
        iters = 50
        for name in ("conv_stack_fwd", "conv1_fwd", "conv2_fwd", "dense1_fwd", "heads", "dense1_dw", "dense1_dx", "dense1_bwd", "dense1_bwd_tile", "conv2_dw",
                     "conv2_dx", "conv1_dw", "slab_reduce", "rmsprop"):
            nat.check(lib.ga3c_net_time_kernel(h, name.encode(), B, 5, nat.C.byref(ev_ms)), name)
            nat.check(lib.ga3c_net_time_kernel(h, name.encode(), B, iters, nat.C.byref(ev_ms)), name)
            kernels[name] = ev_ms.value / iters * 1e3          # microseconds per launch
        # dominant kernel of the prediction step: the fused conv stack (conv1 + conv2 = 74 % of the forward FLOPs)
        flop_launch = (FLOP_CONV1_PER_SAMPLE + FLOP_CONV2_PER_SAMPLE) * B
        t_conv = kernels["conv_stack_fwd"] * 1e-6
        achieved = flop_launch / t_conv / 1e12
        traffic, traffic_commit, traffic_src, prof_us = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("conv_stack_fwd_B%d" % B)
            traffic_commit, traffic_src = tj.get("commit"), tj.get("source")
            prof_us = tj.get("conv_stack_fwd_B%d_rocprof_avg_us" % B)
        out["roofline"] = {"kernel": "conv_stack_fwd_kernel<false, false>", "bound": "mfma", "achieved": achieved,
                           "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                           "traffic": traffic, "traffic_commit": traffic_commit, "traffic_source": traffic_src,
                           "avg_launch_us": kernels["conv_stack_fwd"], "algorithmic_flop_per_launch": flop_launch,
                           "algorithmic_bytes_per_launch": B * (84 * 84 * 4 * 4 + 11 * 11 * 32 * 4) + 4 * (4112 + 8224)}
        if prof_us:      # rocprofv3's begin-to-end time of each dispatch in the committed profile (includes the dispatch ramp that
            #              back-to-back launches overlap with the previous kernel's drain; the live figure above is launch-to-launch)
            out["roofline"]["profile_avg_launch_us"] = prof_us
            out["roofline"]["profile_frac"] = flop_launch / (prof_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS
        out["kernel_us"] = kernels
        flop = FLOP_PER_PREDICTION.get(A, 7_581_184)
        out["end_to_end_mfma_frac"] = out["value"] * flop / (world * MFMA_F32_PEAK_TFLOPS * 1e12)

    # ---- CPU baseline: the oracle's C port on this box's host cores, kernel-only, forward and
    # forward-backward + RMSProp at B in {1, 32, 128, 512} (rank 0, N = 1 only).  The headline sample is the bench batch.
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import ga3c_oracle_cport as oc
        oc.lib().ga3c_oc_set_threads(min(os.cpu_count() or 1, 16))   # a one-GPU box's CPU share is 16 cores
        theta = net.get_arena(0)
        crng = np.random.Generator(np.random.PCG64(12345))            # Config.py:187

        def cpu_leg(bsz, seconds):
            xb = crng.integers(0, 256, size=(bsz, 84, 84, 4), dtype=np.uint8).astype(np.float32) / np.float32(128) - np.float32(1)
            ab = np.eye(A, dtype=np.float32)[crng.integers(0, A, bsz)]
            yb = crng.uniform(-1, 1, bsz).astype(np.float32)
            if bsz == B:
                xb, ab, yb = x, act, y_r
            oc.predict(theta, A, xb)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                oc.predict(theta, A, xb)
                n += 1
            dt = time.perf_counter() - t0
            ms, th2 = np.ones_like(theta), theta.copy()
            oc.train(th2, ms, A, xb, yb, ab, lr, beta)
            m, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 0.6 * seconds:
                oc.train(th2, ms, A, xb, yb, ab, lr, beta)
                m += 1
            dt2 = time.perf_counter() - t1
            return {"predictions_per_sec": n * bsz / dt, "train_steps_per_sec": m / dt2, "forward_passes": n,
                    "train_steps": m, "seconds": dt + dt2}

        head = cpu_leg(B, 0.4 * args.cpu_seconds)
        sizes = {str(B): head}
        for bsz in (1, 32, 128, 512):
            if bsz != B:
                sizes[str(bsz)] = cpu_leg(bsz, 0.1 * args.cpu_seconds)
        out["cpu_baseline"] = {"value": head["predictions_per_sec"], "unit": "predictions/s", "cores": oc.threads(), "kind": "port",
                               "sample": "%d forward passes of the same %d-state batch (%.1f s) by oracle/ga3c_oracle_c.c, "
                                         "OpenMP over %d threads of %d host cores"
                                         % (head["forward_passes"], B, head["seconds"], oc.threads(), os.cpu_count()),
                               "train_steps_per_sec": head["train_steps_per_sec"],
                               "kernel_only_by_batch": dict(sizes, note="BASELINE.md section 4 item 1: forward and forward-backward + "
                                                            "RMSProp at B in {1, 32, 128, 512}, same C port, same threads")}

    cpu_theta = net.get_arena(0) if (rank == 0 and world == 1 and args.cpu_seconds > 0) else None
    net.close()

    # ---- extra: the same lane sweep with 8 hardware queues.  GPU_MAX_HW_QUEUES is read once, when HIP initialises, and is
    # process-wide, so this runs in a child process; the package itself leaves the runtime's default (ga3c_amd/__init__.py)
    if rank == 0 and world == 1 and not args.no_lane_sweep:
        import subprocess
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--lanes-only", "--steps", str(max(K, 100)),
                                    "--warmup", str(max(W, 10)), "--batch", str(B), "--actions", str(A)],
                                   env=dict(os.environ, GPU_MAX_HW_QUEUES="8", GA3C_LANE_STREAMS="4"), capture_output=True, text=True, timeout=240)
            line = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
            out["predict_lanes_8_hw_queues"] = dict(json.loads(line[-1]), unit="predictions/s",
                                                    note="child process with GPU_MAX_HW_QUEUES=8 and GA3C_LANE_STREAMS=4 (a stream "
                                                         "and a hardware queue per lane, nothing else running): an extra, not "
                                                         "the product's configuration") if line else \
                {"error": "child printed no line (rc %d): %s" % (child.returncode, child.stderr[-300:])}
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            out["predict_lanes_8_hw_queues"] = {"error": repr(e)}

    # ---- Hogwild trainers: the reference's default NT = 2 trainer threads update shared weights unlocked (Server.py:132-134)
    if rank == 0 and world == 1:
        hog = Network("gpu:%d" % local_rank, "bench_hogwild", A, (84, 84, 4), max_batch=B, predict_lanes=1, train_lanes=4)
        nat.check(hog._lib.ga3c_net_upload(hog._h, nat.ptr(x), nat.ptr(y_r), nat.ptr(act), B), "upload")
        ms = nat.C.c_float()
        res = {}
        for nl in (1, 2, 4):
            nat.check(hog._lib.ga3c_net_time_train_lanes(hog._h, B, max(W, 1), nl, lr, beta, nat.C.byref(ms)), "warmup")
            device_sync(hog._h)
            nat.check(hog._lib.ga3c_net_time_train_lanes(hog._h, B, K, nl, lr, beta, nat.C.byref(ms)), "time_train_lanes")
            res[str(nl)] = K / (ms.value * 1e-3)
        out["train"]["hogwild_lanes"] = dict(res, unit="steps/s", note="Config.HOGWILD: NT train lanes update the weights "
                                             "concurrently and unlocked like the reference's trainer threads; the headline "
                                             "train figure above is the synchronous single-lane mode")
        # ---- the data-parallel CODE PATH on one GPU: a one-rank RCCL communicator attached, so a step
        # runs what every rank of an N-GPU job runs -- gradients to the arena, ncclAllReduce on the comm stream overlapped with
        # the conv backward (two calls, events between the streams), the separate rmsprop kernel instead of the fused updates.
        # The per-rank step cost an 8-GPU run starts from; the exchange itself (one rank: a copy) is not a scaling figure.
        dpn = Network("gpu:%d" % local_rank, "bench_dp1", A, (84, 84, 4), max_batch=TB, predict_lanes=1)
        nat.check(dpn._lib.ga3c_net_upload(dpn._h, nat.ptr(x_tb), nat.ptr(y_r_tb), nat.ptr(act_tb), TB), "upload")
        dp1 = {}
        sys.stdout.flush()
        saved_stdout = os.dup(1)          # RCCL prints its version banner on stdout when a communicator is made: this
        os.dup2(2, 1)                     # process's stdout carries the ONE JSON line and nothing else
        try:
            dpn.comm_init(Network.make_comm_id(), 0, 1)
            for rows in (B, TB):
                nat.check(dpn._lib.ga3c_net_time_resident(dpn._h, 1, rows, max(W, 5), lr, beta, nat.C.byref(ms)), "warmup")
                walls, spans = [], []
                while sum(walls) < MIN_TIMED_S and len(walls) < MAX_BLOCKS:
                    device_sync(dpn._h)
                    t0 = time.perf_counter()
                    nat.check(dpn._lib.ga3c_net_time_resident(dpn._h, 1, rows, K, lr, beta, nat.C.byref(ms)), "time_resident")
                    device_sync(dpn._h)
                    walls.append(time.perf_counter() - t0)
                    spans.append(ms.value * 1e-3)
                dp1["rows_%d" % rows] = {"ms_per_step": float(np.median(walls)) / K * 1e3, "gpu_span_ms_per_step": float(np.median(spans)) / K * 1e3,
                                         "steps_per_sec": K / float(np.median(walls))}
            nat.check(dpn._lib.ga3c_net_time_allreduce(dpn._h, 50, nat.C.byref(ms)), "time_allreduce")
            dp1["allreduce_us_one_rank"] = ms.value / 50 * 1e3
            dp1["rccl_comm"] = dict(zip(("ranks", "rank", "device"), dpn.comm_info()))
            dp1["note"] = ("one-rank RCCL communicator (ga3c_net_comm_init) on this GPU: the train step of the N-GPU code path -- "
                           "non-fused rmsprop_kernel, slab_reduce without the update, dense1_bwd_tile without it, both "
                           "ncclAllReduce calls and the comm-stream events -- beside the fused single-GPU step above")
        except RuntimeError as e:
            dp1["error"] = str(e)
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        out["train"]["train_dp_1rank"] = dp1
        dpn.close()
        # ---- frame front-end: raw 210x160x3 frames -> uint8 planes -> device frame queues
        nf = 256                                            # one workgroup per frame, one frame per CU
        frng = np.random.Generator(np.random.PCG64(Config.RANDOM_SEED + 99))
        frames = frng.integers(0, 256, size=(nf, 210, 160, 3), dtype=np.uint8)
        hog.frames_config(nf, 210, 160, 3)
        nat.check(hog._lib.ga3c_net_frames_upload(hog._h, nat.ptr(frames, nat.u8p), nf), "frames_upload")
        nat.check(hog._lib.ga3c_net_time_frames(hog._h, nf, 5, nat.C.byref(ms)), "time_frames")
        nat.check(hog._lib.ga3c_net_time_frames(hog._h, nf, 50, nat.C.byref(ms)), "time_frames")
        us = ms.value * 1e3 / 50
        fbytes = 210 * 160 * 3 + 2 * 28224                   # frame in; the agent's queue read and written back
        fe_traffic = None                                     # PMC-derived HBM bytes per launch (profiles/r01_l_frontend_pmc.md)
        if os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")):
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                fe_traffic = json.load(f).get("frame_frontend_210x160x3_n%d" % nf)
        fe = {"frames_per_sec": nf / (us * 1e-6), "frames_per_launch": nf, "launch_us": us,
              "roofline": {"kernel": "frame_frontend_kernel<3>", "bound": "hbm", "achieved": nf * fbytes / (us * 1e-6) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nf * fbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_bytes_per_frame": fbytes, "traffic": fe_traffic},
              "note": "frames resident in HBM; gray (f64) + per-frame min/max bytescale + Pillow-exact bilinear 84x84 + push "
                      "into the device-side 4-deep queues; bit-exact with oracle/frame_frontend.py"}
        if args.cpu_seconds > 0:
            import frame_frontend as ff
            host = nat.host_lib()
            plane = np.empty((84, 84), np.uint8)
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < min(args.cpu_seconds, 3.0):
                ff.preprocess_u8(frames[k % nf])
                k += 1
            dt_o = time.perf_counter() - t0
            j, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                host.ga3c_frame_preprocess(nat.ptr(frames[j % nf], nat.u8p), 210, 160, 3, 84, 84, nat.ptr(plane, nat.u8p))
                j += 1
            dt_h = time.perf_counter() - t0
            fe["cpu_baseline"] = {"value": k / dt_o, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": "%d frames through oracle/frame_frontend.py (numpy) in %.1f s" % (k, dt_o),
                                  "host_c_frames_per_sec": j / dt_h,
                                  "host_c_note": "ga3c_frame_preprocess (libga3c_host.so), one thread: what an actor "
                                                 "process would otherwise spend per emulator frame"}
        out["frontend"] = fe
        hog.close()

    # ---- whole engine, BASELINE configs[1]/[2] shape: agents -> shm transport -> ThreadPredictor / ThreadTrainer -> HIP
    if rank == 0 and world == 1 and args.e2e_seconds > 0:
        out["e2e"] = dict(run_engine(None, args.e2e_seconds, args.e2e_agents, B, A),
                          note="synthetic Python agents (PCG64 frames) on this box's host cores; predictions served and "
                               "train steps taken by the engine over the steady window of the run (whole_run includes "
                               "forking the agents)")
        # BASELINE configs[2] names 64 agents for the predictor + trainer engine
        r64 = run_engine(None, max(4.0, args.e2e_seconds * 0.75), 2 * args.e2e_agents, B, A)
        out["e2e"]["agents_x2"] = {k: r64[k] for k in ("predictions_per_sec", "training_steps_per_sec", "mean_predict_batch",
                                                        "seconds", "agents", "predictors", "trainers")}
        # the same 64 agents shipping only their newest 84x84 plane (Config.FRONTEND = 'device' with FRAME_SOURCE = 'planes'):
        # the 4-deep frame queue and the plane history live in HBM, rollouts name their states -- 7 KB per step over PCIe
        # instead of 62 KB, and no frame stacking in the agents
        rpd = run_engine(None, max(4.0, args.e2e_seconds * 0.75), 2 * args.e2e_agents, B, A, frames="planes-device")
        out["e2e"]["agents_x2_frame_queue_on_device"] = {k: rpd[k] for k in ("predictions_per_sec", "training_steps_per_sec",
                                                                              "mean_predict_batch", "seconds", "agents", "predictors", "trainers")}
        # the same engine fed with raw 210x160x3 emulator frames: the reference's front-end in the agents (host) against the
        # HIP front-end with device-resident frame queues and (agent, plane) rollouts
        half = max(4.0, args.e2e_seconds * 0.75)
        keys = ("predictions_per_sec", "training_steps_per_sec", "mean_predict_batch", "seconds", "agents", "predictors")
        raw = {}
        for mult, npred in ((1, 2), (2, 2)):      # Config.PREDICTORS = 2 for both (4 predictor threads + 2 trainers: see profiles/README.md)
            raw["agents_x%d" % mult] = {
                mode: {k: r[k] for k in keys}
                for mode, r in ((m, run_engine(None, half, mult * args.e2e_agents, B, A, frames="rgb-" + m, predictors=npred))
                                for m in ("host", "device"))}
        raw["note"] = ("synthetic emulator frames; 'host': ga3c_frame_preprocess (AVX2) in every agent process, states shipped, "
                       "~65 us of agent CPU per step; 'device': raw frames shipped, front-end + frame queues + plane history on "
                       "the GPU, ~27 us of agent CPU per step -- the host path is ahead while the box's CPU share lasts and "
                       "falls behind once the agents saturate it")
        out["e2e"]["raw_frames"] = raw
        if cpu_theta is not None:      # the same harness with the oracle's C port as the model: the CPU path beside it
            import ga3c_oracle_cport as oc
            # BASELINE.md section 4 item 2 = BASELINE.json configs[0]: 4 agents / 1 predictor / 1 trainer, the reference's default
            # TRAINING_MIN_BATCH_SIZE = 0 (one rollout per train step, Config.py:118), CPU model; and the HIP engine beside it
            oc.lib().ga3c_oc_set_threads(8)       # one predictor + one trainer thread call in concurrently: 2 x 8 = the 16-core share
            short = max(3.0, 0.5 * args.e2e_seconds)
            out["cpu_baseline"]["e2e_config0"] = dict(
                run_engine(CPortModel(oc, cpu_theta, A), short, 4, B, A, predictors=1, trainers=1, min_batch=0),
                note="BASELINE configs[0]: 4 agents / 1 predictor / 1 trainer, TRAINING_MIN_BATCH_SIZE = 0, model = "
                     "oracle/ga3c_oracle_c.c (8 OpenMP threads per calling thread)")
            out["e2e"]["config0"] = dict(
                run_engine(None, short, 4, B, A, predictors=1, trainers=1, min_batch=0),
                note="the same configs[0] shape on the HIP engine")
            oc.lib().ga3c_oc_set_threads(4)       # 2 predictor + 2 trainer threads call in concurrently: 4 x 4 = the 16-core share
            out["cpu_baseline"]["e2e"] = dict(
                run_engine(CPortModel(oc, cpu_theta, A), short, args.e2e_agents, B, A),
                note="extra: the 32-agent / 2 / 2 engine shape of the `e2e` leg with the C port as the model (host-buffer path, "
                     "4 OpenMP threads per calling thread)")

    # ---- N-rank engine: one Server per GPU (own agents, own transport), lock-step training over RCCL (DataParallel.EngineGroup)
    if world > 1 and args.e2e_seconds > 0 and dp_error is None:
        import threading

        def give_up():      # a rank that never joins a collective would hang the job: the line is printed without this leg
            if rank == 0:
                out["e2e"] = {"error": "the %d-rank engine leg did not finish within its time limit" % world}
                print(json.dumps(out), flush=True)
            os._exit(3)      # the line is kept, the status says that a collective leg hung
        guard = threading.Timer(args.e2e_seconds + 90.0, give_up)
        guard.daemon = True
        guard.start()
        Config.DEVICE = "gpu:%d" % local_rank
        Config.RANDOM_SEED += 100003 * rank
        group = DataParallel.EngineGroup(rank, world)
        res = run_engine(None, args.e2e_seconds, args.e2e_agents, B, A, engine_group=group)
        group.close()
        guard.cancel()
        tot = rv.reduce([res["predictions_per_sec"], res["training_steps_per_sec"], res["whole_run"]["training_steps_per_sec"]],
                        op="sum")
        if rank == 0:
            out["e2e"] = {"predictions_per_sec": float(tot[0]), "training_steps_per_sec": float(tot[1]) / world,
                          "agents": world * args.e2e_agents, "ranks": world, "rank0": res,
                          "note": "one engine per GPU (%d agents, 2 predictors, 2 trainers each); predictions are summed over "
                                  "the ranks, a train step is a GLOBAL step (every rank's %d-row shard + RCCL all-reduce)"
                                  % (args.e2e_agents, B)}
    rv.barrier()
    rv.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
