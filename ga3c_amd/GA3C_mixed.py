"""Several games at once on one node (BASELINE configs[4]: "Boxing + Pong mixed ... dynamic NT/NP"): one GA3C engine --
its own Server, agents, transport, Network and action count -- per game, as separate processes.  The reference's Server
owns exactly one game (Config.ATARI_GAME, Server.py:70-78, one policy head of get_num_actions() outputs), so "mixed" is
a matter of launching, not of the model.

    python GA3C_mixed.py GAMES=Boxing:18:0,Pong:6:0 KEY=VALUE ...

GAMES is a comma-separated list of name:num_actions[:gpus] where gpus is a '+'-separated list of device ordinals (default
0).  One device = `python GA3C.py` on it; several = one rank per device under torch.distributed.run (RCCL data parallel,
DataParallel.py).  Engines may share a device.  Every other KEY=VALUE goes to every engine (GA3C.py's grammar).  Each
engine gets NETWORK_NAME=<name> (so checkpoints/<name>_%08d and logs/<name>/) and RESULTS_FILENAME=results_<name>.txt.
The exit status is the worst of the engines'.
"""
import os
import subprocess
import sys


def parse_games(spec):
    games = []
    for item in spec.split(","):
        parts = item.strip().split(":")
        if len(parts) < 2:
            raise ValueError("GAMES entry %r is not name:num_actions[:gpus]" % item)
        gpus = [int(g) for g in parts[2].split("+")] if len(parts) > 2 and parts[2] else [0]
        games.append((parts[0], int(parts[1]), gpus))
    if len({g[0] for g in games}) != len(games):
        raise ValueError("GAMES names must differ: they name the checkpoints and result files")
    return games


def commands(games, passthrough, base_port=29600):
    here = os.path.dirname(os.path.abspath(__file__))
    out = []
    for k, (name, actions, gpus) in enumerate(games):
        own = ["GAME=%s" % name, "NUM_ACTIONS=%d" % actions, "NETWORK_NAME=%s" % name, "RESULTS_FILENAME=results_%s.txt" % name]
        env = dict(os.environ)
        if len(gpus) == 1:
            cmd = [sys.executable, os.path.join(here, "GA3C.py"), "DEVICE=gpu:%d" % gpus[0]] + own + passthrough
        else:
            # LOCAL_RANK r drives the r-th device of this engine's list
            env["HIP_VISIBLE_DEVICES"] = ",".join(str(g) for g in gpus)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(gpus)),
                   "--master-addr", "127.0.0.1", "--master-port", str(base_port + k),
                   os.path.join(here, "GA3C.py")] + own + passthrough
        out.append((name, cmd, env))
    return out


def main(argv):
    spec = [a for a in argv if a.startswith("GAMES=")]
    if len(spec) != 1:
        sys.exit(__doc__)
    games = parse_games(spec[0].split("=", 1)[1])
    passthrough = [a for a in argv if not a.startswith("GAMES=")]
    procs = [(name, subprocess.Popen(cmd, env=env)) for name, cmd, env in commands(games, passthrough)]
    worst = 0
    for name, p in procs:
        rc = p.wait()
        print("engine %s ended with status %d" % (name, rc))
        worst = max(worst, abs(rc))
    return worst


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
