"""Entry point: `python GA3C.py KEY=VALUE ...` sets Config attributes, coerced to the type of the
current value, then runs the server (reference: ga3c/GA3C.py:38-59).  As there, a bool can only be
switched off with an empty value (`KEY=`), because bool("False") is True.
"""
import sys

import _native  # noqa: F401  (fail early if the libraries are not built)
from Config import Config


def apply_argv(argv):
    for arg in argv:
        key, value = arg.split('=', 1)
        setattr(Config, key, type(getattr(Config, key))(value))
    if Config.PLAY_MODE:
        Config.AGENTS = 1
        Config.PREDICTORS = 1
        Config.TRAINERS = 1
        Config.DYNAMIC_SETTINGS = False
        Config.LOAD_CHECKPOINT = True
        Config.TRAIN_MODELS = False
        Config.SAVE_MODELS = False


if __name__ == '__main__':
    apply_argv(sys.argv[1:])
    import DataParallel
    group = DataParallel.EngineGroup.from_env()          # None unless launched with WORLD_SIZE > 1
    if group is not None:
        _, local_rank, _ = DataParallel.env_rank_world()
        Config.DEVICE = 'gpu:%d' % local_rank
        Config.RANDOM_SEED += 100003 * group.rank        # different synthetic episodes on every rank
        Config.RESULTS_FILENAME = 'results_rank%d.txt' % group.rank
        Config.SAVE_MODELS = Config.SAVE_MODELS and group.rank == 0
    from Server import Server
    try:
        # raises (non-zero exit status) if a predictor / trainer thread died
        Server(engine_group=group).main(max_seconds=Config.MAX_SECONDS or None)
    except RuntimeError as e:
        if isinstance(e.__cause__, DataParallel.GroupStalled):
            # a rank kept the others inside an all-reduce: trainer threads and the train stream are stuck in it, so no
            # orderly teardown of the GPU side is possible -- say who was late and end the process with a failure status
            import os
            sys.stderr.write("GA3C: %s\n" % e.__cause__)
            sys.stderr.flush()
            os._exit(3)
        raise
    finally:
        if group is not None:
            group.close()
