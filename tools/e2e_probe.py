"""Where does whole-engine time go?  Runs the Server (synthetic agents, zero-copy transport) and samples, after a
warm-up, the steady-state prediction / train-step rates, the mean prediction batch, and the CPU seconds burnt by
the server process and by the agent processes (psutil), so an agent-bound run can be told from a server-bound one.

    python tools/e2e_probe.py --agents 32 --predictors 2 --trainers 2 --seconds 10 [--no-train]

Prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=32)
    ap.add_argument("--predictors", type=int, default=2)
    ap.add_argument("--trainers", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--warm", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--train-min-batch", type=int, default=127)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--hogwild", action="store_true")
    ap.add_argument("--python-predictor", action="store_true", help="keep ThreadPredictor's loop in Python")
    ap.add_argument("--linger-us", type=int, default=0)
    ap.add_argument("--linger-batch", type=int, default=0)
    ap.add_argument("--dynamic", action="store_true", help="ThreadDynamicAdjustment random walk every 2 s (soak test)")
    ap.add_argument("--state-cache", type=int, default=-1, help="Config.STATE_CACHE (0 / 1; default: the package's)")
    ap.add_argument("--lanes", type=int, default=0, help="prediction lanes of the Network (0: one per predictor, the default)")
    ap.add_argument("--frames", choices=["planes", "planes-device", "rgb-host", "rgb-device"], default="planes",
                    help="frame source / where the reference's front-end runs (Config.FRAME_SOURCE, Config.FRONTEND)")
    args = ap.parse_args()

    import psutil
    import ga3c_amd  # noqa: F401
    from Config import Config
    from Server import Server

    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = args.agents, args.predictors, args.trainers
    Config.DYNAMIC_SETTINGS = bool(args.dynamic)
    Config.DYNAMIC_SETTINGS_STEP_WAIT, Config.DYNAMIC_SETTINGS_INITIAL_WAIT = 2, 2
    Config.PREDICTION_BATCH_SIZE = args.batch
    Config.PREDICTION_LINGER_US, Config.PREDICTION_LINGER_BATCH = args.linger_us, args.linger_batch
    Config.TRAINING_MIN_BATCH_SIZE = args.train_min_batch
    Config.TRAIN_MODELS = not args.no_train
    if args.state_cache >= 0:
        Config.STATE_CACHE = bool(args.state_cache)
    Config.HOGWILD = bool(args.hogwild)
    Config.NATIVE_PREDICTOR = not args.python_predictor
    if args.frames == "planes-device":
        Config.FRAME_SOURCE, Config.FRONTEND = "planes", "device"
    elif args.frames != "planes":
        Config.FRAME_SOURCE, Config.FRONTEND = "rgb", args.frames.split("-")[1]
    Config.SAVE_MODELS = False
    Config.LOAD_CHECKPOINT = False
    Config.RESULTS_FILENAME = "/tmp/e2e_probe_results.txt"
    Config.EPISODES = 10 ** 9

    if args.lanes:
        os.environ["GA3C_PREDICT_LANES"] = str(args.lanes)
    srv = Server(max_agents=args.agents)
    me = psutil.Process()

    def cgroup():
        try:
            d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
            return int(d.get("usage_usec", 0)), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
        except OSError:
            return 0, 0, 0
    snap = {}

    def cpu_of(procs):
        tot = 0.0
        for p in procs:
            try:
                c = p.cpu_times()
                tot += c.user + c.system
            except psutil.Error:
                pass
        return tot

    def take():
        kids = [p for p in me.children(recursive=True)]
        return {"t": time.perf_counter(), "pred": srv.predictions_served, "steps": srv.training_step,
                "batches": sum(p.batches for p in srv.predictors),
                "loop": {k: sum(p.seconds[k] for p in srv.predictors) for k in ("pop", "predict", "respond")}, "srv_cpu": cpu_of([me]), "agent_cpu": cpu_of(kids),
                "cg": cgroup(), "eng": srv.model.stats() if hasattr(srv.model, "stats") else {},
                "n_kids": len(kids), "n_agents": len(srv.agents), "alive": sum(1 for a in srv.agents if a.is_alive()), "n_pred": len(srv.predictors), "n_train": len(srv.trainers),
                "rss_mb": round(me.memory_info().rss / 2 ** 20), "spills": sum(t.spills for t in srv.trainers), "died": sum(1 for t in srv.predictors + srv.trainers if not t.is_alive()),
                "wake": srv.transport.wake_latency()}

    def sampler():
        time.sleep(args.warm)
        snap["a"] = take()
        time.sleep(max(0.5, args.seconds - args.warm - 0.5))
        snap["b"] = take()

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    srv.main(max_seconds=args.seconds)
    th.join(timeout=5)
    a, b = snap.get("a"), snap.get("b")
    if not a or not b:
        print(json.dumps({"error": "sampler did not finish"}))
        return
    dt = b["t"] - a["t"]
    pred, batches = b["pred"] - a["pred"], max(1, b["batches"] - a["batches"])
    eng = {k: b["eng"].get(k, 0) - a["eng"].get(k, 0) for k in b["eng"]}
    pc, tc = max(eng.get("predict_calls", 0), 1), max(eng.get("train_calls", 0), 1)
    engine = {"predict_us_per_call": {k[8:-3]: round(eng[k] / pc / 1e3, 1) for k in eng if k.startswith("predict_") and k.endswith("_ns")},
              "predict_weight_waits_per_call": round(eng.get("predict_weight_waits", 0) / pc, 3),
              "train_us_per_call": {k[6:-3]: round(eng[k] / tc / 1e3, 1) for k in eng if k.startswith("train_") and k.endswith("_ns")},
              "train_reader_waits_per_call": round(eng.get("train_reader_waits", 0) / tc, 3),
              "train_rows_per_call": round(eng.get("train_rows", 0) / tc, 1)}
    print(json.dumps({
        "placement": getattr(srv, "placement", None), "lost_train_batches": getattr(srv, "lost_train_batches", 0),
        "state_cache_depth": getattr(srv, "state_cache_depth", None), "state_cache": bool(getattr(srv, "state_cache", False)), "lanes": args.lanes or args.predictors, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "engine": engine,
        "cgroup": {"cpu_cores_used": round((b["cg"][0] - a["cg"][0]) / 1e6 / (b["t"] - a["t"]), 2), "throttled_periods": b["cg"][1] - a["cg"][1],
                   "throttled_s": round((b["cg"][2] - a["cg"][2]) / 1e6, 2)},
        "agents": args.agents, "predictors": args.predictors, "trainers": args.trainers, "train": not args.no_train,
        "hogwild": bool(args.hogwild), "frames": args.frames, "linger": [args.linger_us, args.linger_batch], "native_predictor": not args.python_predictor, "window_s": round(dt, 2), "host_cores": os.cpu_count(),
        "predictions_per_sec": round(pred / dt), "train_steps_per_sec": round((b["steps"] - a["steps"]) / dt, 1),
        "mean_predict_batch": round(pred / batches, 1), "predict_batches_per_sec": round(batches / dt),
        "predictor_us_per_batch": {k: round((b["loop"][k] - a["loop"][k]) / batches * 1e6, 1) for k in b["loop"]},
        "server_cpu_cores": round((b["srv_cpu"] - a["srv_cpu"]) / dt, 2),
        "agent_cpu_cores": round((b["agent_cpu"] - a["agent_cpu"]) / dt, 2),
        "agent_cpu_us_per_step": round((b["agent_cpu"] - a["agent_cpu"]) / max(1, pred) * 1e6, 1),
        "agent_wall_us_per_step": round(dt * args.agents / max(1, pred) * 1e6, 1),
        "workers_at_end": {"agents": b["n_agents"], "agents_alive": b["alive"], "predictors": b["n_pred"], "trainers": b["n_train"]},
        "server_rss_mb_start_end": [a["rss_mb"], b["rss_mb"]], "threads_died": b["died"],
        "answer_to_running_us": {k: {"answers": b["wake"][k][0] - a["wake"][k][0],
                                      "mean": round((b["wake"][k][1] * b["wake"][k][0] - a["wake"][k][1] * a["wake"][k][0]) / max(b["wake"][k][0] - a["wake"][k][0], 1), 1),
                                      "max_since_start": round(b["wake"][k][2], 1)} for k in ("ready", "slept")},
        "trainer_spills": sum(t.spills for t in srv.trainers) if srv.trainers else b["spills"]}))


if __name__ == "__main__":
    main()
