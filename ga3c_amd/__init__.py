"""ga3c_amd -- MI355X-native GA3C actor-learner hot path.

Host side in Python (as the reference is), compute in libga3c_hip.so (hand-written gfx950
kernels behind include/ga3c_abi.h), transport in libga3c_host.so (include/ga3c_host.h).

The modules keep the reference's flat names (Config, Server, ProcessAgent, NetworkVP, ...) and
import each other the way the reference's do (`from Config import Config`), so this directory
works as the working directory of `_train.sh`.  `import ga3c_amd` puts it on sys.path so the same
flat imports work from anywhere; there is exactly one copy of each module.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
