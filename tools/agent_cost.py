"""Development aid (CPU only): what ONE ProcessAgent costs per emulator step when its answers come back at once.
A child process plays the server -- pops prediction requests and answers them with a uniform policy, pops rollouts and gives
their slots back -- and the parent runs ProcessAgent.run() in-process for a few seconds, optionally under cProfile.
    python tools/agent_cost.py [--seconds 4] [--mode states|cache|device] [--profile]
Prints steps/s, us of agent CPU per step and (with --profile) the top of the profile."""
import argparse
import multiprocessing as MP
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ga3c_amd"))
import numpy as np  # noqa: E402


def server(name, A, stop):
    import Transport as tp
    t = tp.Transport.attach(name)
    ids = np.zeros(8, np.uint32)
    p = np.full((8, A), 1.0 / A, np.float32)
    v = np.zeros(8, np.float32)
    while not stop.value:
        n = t.pop_batch(ids, 1)
        if n > 0:
            t.respond(ids, n, p, v)
        while t.ready_count() > 0:
            s = t.pop_rollout(0)
            if s < 0:
                break
            t.release(s)
    t.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--mode", choices=["states", "cache", "device"], default="cache")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    from Config import Config
    import Transport as tp
    from ProcessAgent import ProcessAgent, config_snapshot
    Config.STATE_TRANSPORT = 'u8'
    Config.STATE_CACHE_ACTIVE = args.mode == "cache"
    if args.mode == "device":
        Config.FRONTEND = 'device'
    A = int(Config.NUM_ACTIONS)
    state_bytes = 84 * 84 * 4
    row_bytes = 16 if args.mode in ("cache", "device") else 0
    t = tp.Transport.create(tp.unique_name("cost"), 1, A, state_bytes, 16, Config.TIME_MAX + 1, row_bytes)
    stop = MP.Value('i', 0)
    srv = MP.Process(target=server, args=(t.name, A, stop))
    srv.start()
    agent = ProcessAgent(0, t.name, MP.Queue(), config_snapshot())
    threading.Timer(args.seconds, lambda: setattr(agent.exit_flag, "value", 1)).start()
    c0, w0 = time.process_time(), time.perf_counter()
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.runcall(agent.run)
    else:
        agent.run()
    cpu, wall = time.process_time() - c0, time.perf_counter() - w0
    steps = agent.planes_pushed if args.mode == "device" else agent.requests
    print("mode %s: %d steps in %.2f s wall = %.0f steps/s; %.1f us CPU per step (this process, all threads)"
          % (args.mode, steps, wall, steps / wall, cpu / max(steps, 1) * 1e6))
    if args.profile:
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    stop.value = 1
    srv.join()
    t.shutdown()
    t.unlink()
    t.close()


if __name__ == "__main__":
    main()
