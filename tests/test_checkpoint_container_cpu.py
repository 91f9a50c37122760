"""The checkpoint container of ga3c_net_save / ga3c_net_load (ga3c_amd/csrc/ga3c_checkpoint.hpp: an uncompressed .npz) against
numpy itself, without a GPU: what the C writer writes numpy.load reads, what numpy.savez writes the C reader reads (zip64
headers included), and compressed archives are refused by name."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ckpt") / "ckpt_tool")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "ckpt_tool.cpp")])
    return exe


def test_numpy_reads_what_the_c_writer_writes(tool, tmp_path):
    path = str(tmp_path / "c.npz")
    subprocess.check_call([tool, "write", path])
    with np.load(path, allow_pickle=False) as z:
        assert sorted(z.files) == ["b:0", "step", "w/x:0"]
        assert z["w/x:0"].dtype == np.float32 and z["w/x:0"].tolist() == [[0, 1, 2], [3, 4, 5]]
        assert z["b:0"].shape == (4,) and z["b:0"].tolist() == [0.0, 0.5, 1.0, 1.5]
        assert z["step"].dtype == np.int64 and z["step"].shape == () and int(z["step"]) == 1234567890123
    again = str(tmp_path / "c2.npz")
    subprocess.check_call([tool, "write", again])
    assert open(path, "rb").read() == open(again, "rb").read()          # no time stamps: same members, same bytes
    assert not os.path.exists(path + ".tmp")                             # written under a temporary name, then renamed


def test_the_c_reader_reads_what_numpy_writes(tool, tmp_path):
    path = str(tmp_path / "np.npz")
    rng = np.random.default_rng(3)
    w = rng.normal(size=(8, 8, 4, 16)).astype(np.float32)
    np.savez(path, **{"conv11/w:0": w, "conv11/w/RMSProp:0": np.ones((8, 8, 4, 16), np.float32), "logits_v/b:0": np.float32([0.25]),
                      "step": np.int64(77)})
    lines = subprocess.check_output([tool, "read", path], text=True).strip().splitlines()
    got = {ln.split("|")[0]: ln.split("|")[1:] for ln in lines}
    assert got["step"] == ["<i8", "", "77"]
    assert got["logits_v/b:0"] == ["<f4", "1", "0.25"]
    assert got["conv11/w:0"][:2] == ["<f4", "8,8,4,16"]
    assert [np.float32(v) for v in got["conv11/w:0"][2].split(",")] == w.ravel()[:8].tolist()
    assert got["conv11/w/RMSProp:0"][2] == ",".join(["1"] * 8)


def test_compressed_archives_and_garbage_are_refused(tool, tmp_path):
    path = str(tmp_path / "z.npz")
    np.savez_compressed(path, a=np.zeros(1000, np.float32))
    run = subprocess.run([tool, "read", path], capture_output=True, text=True)
    assert run.returncode == 1 and "compressed" in run.stderr
    junk = str(tmp_path / "junk.npz")
    open(junk, "wb").write(b"not a zip archive at all, just some bytes" * 3)
    run = subprocess.run([tool, "read", junk], capture_output=True, text=True)
    assert run.returncode == 1 and run.stderr.strip()
