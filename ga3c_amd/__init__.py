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

# HIP streams (prediction lanes, train lanes, frame queues, gradient exchange) are multiplexed onto GPU_MAX_HW_QUEUES hardware
# queues, 4 unless the environment says otherwise.  On resident batches a queue per lane pays (4 lanes: 7.5 -> 9.3 M
# predictions/s with 8 queues; bench.py sets it), but the whole engine with 4 predictor threads fell from 344 k to 203 k
# predictions/s on 8 queues (profiles/README.md): the package leaves the runtime's default alone.

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
