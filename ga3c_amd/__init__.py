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

# Every prediction lane, the train lane(s), the frame queues and the gradient exchange have HIP streams of their own, and
# the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default): two lanes that land on one queue
# run back to back.  Measured on an MI355X, 4 prediction lanes at batch 128: 7.5 M predictions/s on 4 queues, 9.0 M on 8
# (profiles/README.md).  Read by the HIP runtime when it initialises, so this must come before the first HIP call; a value
# set by the user wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
