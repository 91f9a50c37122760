"""The engine's ceiling: the Server with its predictor / trainer threads and the HIP network, fed by NATIVE agent threads
(tests/native/native_agents.cpp: the agent side of the C ABI, an "emulator" that is one memcpy) instead of Python ProcessAgents.
What is left is the transport, the batching threads and the GPU, with states crossing PCIe out of the registered segment.

    python tools/engine_ceiling.py --agents 256 --predictors 2 --trainers 2 --seconds 12 [--no-train]

Prints one JSON line.  Development aid; not part of the product or the test suite.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def build_tool():
    exe = os.path.join(ROOT, "tools", "native_agents")
    src = os.path.join(ROOT, "tests", "native", "native_agents.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", exe, src,
                               "-L", os.path.join(ROOT, "ga3c_amd"), "-lga3c_host", "-Wl,-rpath," + os.path.join(ROOT, "ga3c_amd")])
    return exe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--predictors", type=int, default=2)
    ap.add_argument("--trainers", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--warm", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--train-min-batch", type=int, default=127)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--linger-us", type=int, default=0)
    ap.add_argument("--linger-batch", type=int, default=0)
    ap.add_argument("--hogwild", action="store_true")
    ap.add_argument("--frame-queue-on-device", action="store_true",
                    help="FRAME_SOURCE = 'planes', FRONTEND = 'device': agents ship their newest plane, states stay in HBM")
    args = ap.parse_args()
    exe = build_tool()

    import ga3c_amd  # noqa: F401
    from Config import Config
    from Server import Server

    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 0, args.predictors, args.trainers
    Config.DYNAMIC_SETTINGS = False
    Config.PREDICTION_BATCH_SIZE = args.batch
    Config.PREDICTION_LINGER_US, Config.PREDICTION_LINGER_BATCH = args.linger_us, args.linger_batch
    Config.TRAINING_MIN_BATCH_SIZE = args.train_min_batch
    Config.TRAIN_MODELS = not args.no_train
    Config.HOGWILD = bool(args.hogwild)
    if args.frame_queue_on_device:
        Config.FRAME_SOURCE, Config.FRONTEND = "planes", "device"
    Config.SAVE_MODELS = False
    Config.LOAD_CHECKPOINT = False
    Config.RESULTS_FILENAME = "/tmp/engine_ceiling_results.txt"
    Config.EPISODES = 10 ** 9

    srv = Server(max_agents=args.agents)
    snap = {}

    def cgroup():
        try:
            d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
            return int(d.get("usage_usec", 0)), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
        except OSError:
            return 0, 0, 0

    def thread_cpu(pid):
        """CPU seconds (user + system) of every thread of `pid`, summed by thread name (/proc/<pid>/task/*/stat)."""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        try:
            for tid in os.listdir("/proc/%d/task" % pid):
                try:
                    raw = open("/proc/%d/task/%s/stat" % (pid, tid)).read()
                except OSError:
                    continue
                name = raw[raw.index("(") + 1:raw.rindex(")")]
                f = raw[raw.rindex(")") + 2:].split()
                u, k = out.get(name, (0.0, 0.0))
                out[name] = (u + int(f[11]) / tick, k + int(f[12]) / tick)
        except OSError:
            pass
        return out

    procs = {}

    def take():
        return {"t": time.perf_counter(), "cpu": {k: thread_cpu(v) for k, v in dict(procs, server=os.getpid()).items()}, "pred": srv.predictions_served, "steps": srv.training_step, "cg": cgroup(),
                "eng": srv.model.stats() if hasattr(srv.model, "stats") else {},
                "batches": sum(p.batches for p in srv.predictors),
                "loop": {k: sum(p.seconds[k] for p in srv.predictors) for k in ("pop", "predict", "respond")},
                "died": sum(1 for t in srv.predictors + srv.trainers if not t.is_alive()), "wake": srv.transport.wake_latency(),
                "rss_mb": next((int(l.split()[1]) / 1024 for l in open("/proc/self/status") if l.startswith("VmRSS")), 0)}

    out = {}

    agent_cpus = (getattr(srv, "placement", None) or {}).get("agent_cpus")

    def driver():
        time.sleep(1.0)                                   # predictors and trainers are up
        proc = subprocess.Popen([exe, srv.transport.name, str(args.agents), str(args.seconds - 2.0), "0" if args.no_train else "1"] +
                                (["cache"] if getattr(srv, "state_cache", False) else []),
                                stdout=subprocess.PIPE, text=True,
                                preexec_fn=(lambda: os.sched_setaffinity(0, agent_cpus)) if agent_cpus else None)
        procs["agents"] = proc.pid
        time.sleep(args.warm)
        snap["a"] = take()
        time.sleep(max(0.5, args.seconds - 2.0 - args.warm - 1.0))
        snap["b"] = take()
        out["tool"] = proc.communicate(timeout=30)[0].strip().replace("\n", " | ")

    th = threading.Thread(target=driver, daemon=True)
    th.start()
    srv.main(max_seconds=args.seconds)
    th.join(timeout=40)
    a, b = snap.get("a"), snap.get("b")
    if not a or not b:
        print(json.dumps({"error": "sampler did not finish"}))
        return
    dt = b["t"] - a["t"]
    pred, batches = b["pred"] - a["pred"], max(1, b["batches"] - a["batches"])
    state_bytes = 84 * 84 if args.frame_queue_on_device else 84 * 84 * 4
    eng = {k: b["eng"].get(k, 0) - a["eng"].get(k, 0) for k in b["eng"]}
    pc, tc = max(eng.get("predict_calls", 0), 1), max(eng.get("train_calls", 0), 1)
    engine = {"predict_us_per_call": {k[8:-3]: round(eng[k] / pc / 1e3, 1) for k in eng if k.startswith("predict_") and k.endswith("_ns")},
              "train_us_per_call": {k[6:-3]: round(eng[k] / tc / 1e3, 1) for k in eng if k.startswith("train_") and k.endswith("_ns")},
              "train_reader_waits_per_call": round(eng.get("train_reader_waits", 0) / tc, 3)}
    cpu = {}
    for who in b["cpu"]:
        for name, (u, k) in b["cpu"][who].items():
            u0, k0 = a["cpu"].get(who, {}).get(name, (0.0, 0.0))
            if (u - u0) + (k - k0) > 0.02 * dt:
                cpu["%s:%s" % (who, name)] = [round((u - u0) / dt, 2), round((k - k0) / dt, 2)]
    print(json.dumps({
        "cpu_cores_by_thread_name_user_sys": cpu, "us_cpu_per_prediction": round(sum(x + y for x, y in cpu.values()) * dt / max(pred, 1) * 1e6, 2),
        "placement": getattr(srv, "placement", None), "lost_train_batches": getattr(srv, "lost_train_batches", 0),
        "state_cache_depth": getattr(srv, "state_cache_depth", None), "engine": engine, "cgroup": {"cpu_cores_used": round((b["cg"][0] - a["cg"][0]) / 1e6 / dt, 2), "throttled_periods": b["cg"][1] - a["cg"][1],
                                     "throttled_s": round((b["cg"][2] - a["cg"][2]) / 1e6, 2)},
        "agents": args.agents, "native_agents": True, "frame_queue_on_device": bool(args.frame_queue_on_device), "predictors": args.predictors, "trainers": args.trainers, "train": not args.no_train,
        "hogwild": bool(args.hogwild), "window_s": round(dt, 2), "host_cores": os.cpu_count(),
        "predictions_per_sec": round(pred / dt), "train_steps_per_sec": round((b["steps"] - a["steps"]) / dt, 1),
        "mean_predict_batch": round(pred / batches, 1), "predict_batches_per_sec": round(batches / dt),
        "predictor_us_per_batch": {k: round((b["loop"][k] - a["loop"][k]) / batches * 1e6, 1) for k in b["loop"]},
        "pcie_gb_per_s_states": round(pred / dt * state_bytes / 1e9, 2), "threads_died": b["died"],
        "answer_to_running_us": {k: {"answers": b["wake"][k][0] - a["wake"][k][0],
                                      "mean": round((b["wake"][k][1] * b["wake"][k][0] - a["wake"][k][1] * a["wake"][k][0]) / max(b["wake"][k][0] - a["wake"][k][0], 1), 1),
                                      "max_since_start": round(b["wake"][k][2], 1)} for k in ("ready", "slept")},
        "server_rss_mb_start_end": [round(a["rss_mb"]), round(b["rss_mb"])], "tool": out.get("tool")}))


if __name__ == "__main__":
    main()
