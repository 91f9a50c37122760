"""-m gpu: bench.py prints ONE JSON line with the fields the driver's contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_line_contract(tmp_path):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2", "--cpu-seconds", "1",
                          "--e2e-seconds", "3", "--e2e-agents", "4"], capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "predictions_per_sec" and d["unit"] == "predictions/s" and d["n_gpus"] == 1
    assert d["steps"] == 20 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 1e5 and abs(d["ms_per_step"] - 128.0 / d["value"] * 1e3) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "predictions/s" and c["sample"]
    assert d["train"]["value"] > 100 and d["e2e"]["predictions_per_sec"] > 100
    # value is host wall-clock of the bracketed block; the GPU event span of the same block is an extra and can only be shorter
    assert 0 < d["gpu_span_ms_per_step"] <= d["ms_per_step"] and d["value_gpu_span"] >= d["value"]
    assert d["torch_imported"] is False                 # device sync and rank plumbing are the package's own
    dp1 = d["train"]["train_dp_1rank"]                  # the N-GPU code path timed with a one-rank communicator
    assert "error" not in dp1 and dp1["rccl_comm"]["ranks"] == 1
    assert dp1["rows_128"]["steps_per_sec"] > 100 and dp1["rows_132"]["steps_per_sec"] > 100


@pytest.mark.timeout(400)
def test_two_rank_launch_path(tmp_path):
    """The driver's multi-GPU launch line, rehearsed with two ranks.  On a one-GPU box both ranks are put on device 0:
    RCCL refuses duplicate GPUs, which exercises the fallback (prediction leg measured, train leg null); with two or
    more GPUs the real data-parallel path runs."""
    import socket
    import torch
    ngpu = torch.cuda.device_count()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
           "--cpu-seconds", "0", "--e2e-seconds", "0"]
    if ngpu < 2:
        cmd += ["--device-override", "0"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=380, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["value"] > 1e5
    if ngpu < 2:
        assert d["data_parallel_error"] and d["train"]["value"] is None
    else:
        assert d["data_parallel_error"] is None and d["train"]["value"] > 100
