"""Placement.plan on a made-up sysfs tree shaped like the MI355X host (2 sockets x 64 cores x 2 threads, 8 cores per L3,
4 GPUs per socket): which CPUs a server next to each GPU gets under a 16-CPU cgroup quota, and when nothing is changed."""
import os

import pytest


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text + "\n")


@pytest.fixture()
def fake_host(tmp_path):
    sysr, cg = str(tmp_path / "sys"), str(tmp_path / "cg")
    for cpu in range(256):
        core = cpu % 128                          # cpu and cpu + 128 are the two threads of one core
        l3 = core // 8
        members = list(range(l3 * 8, l3 * 8 + 8))
        _write("%s/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % (sysr, cpu),
               "%d-%d,%d-%d" % (members[0], members[-1], members[0] + 128, members[-1] + 128))
        _write("%s/devices/system/cpu/cpu%d/topology/thread_siblings_list" % (sysr, cpu), "%d,%d" % (core, core + 128))
    gpus = {0: ["0000:0a:00.0", "0000:23:00.0", "0000:3b:00.0", "0000:52:00.0"],
            1: ["0000:8d:00.0", "0000:a4:00.0", "0000:bd:00.0", "0000:d9:00.0"]}
    for node, devs in gpus.items():
        for bdf in devs:
            base = "%s/bus/pci/devices/%s" % (sysr, bdf)
            _write(base + "/vendor", "0x1002")
            _write(base + "/class", "0x038000")
            _write(base + "/numa_node", str(node))
            _write(base + "/local_cpulist", "0-63,128-191" if node == 0 else "64-127,192-255")
    _write("%s/bus/pci/devices/0000:00:01.0/vendor" % sysr, "0x1022")     # a bridge: not a GPU
    _write("%s/bus/pci/devices/0000:00:01.0/class" % sysr, "0x060400")
    _write(cg + "/cpu.max", "1600000 100000")
    return sysr, cg, gpus


def test_every_gpu_gets_a_window_that_starts_at_its_own_share_of_the_node(fake_host):
    import ga3c_amd  # noqa: F401
    import Placement
    sysr, cg, gpus = fake_host
    starts = set()
    for node, devs in gpus.items():
        for k, bdf in enumerate(devs):
            got = Placement.plan(bdf, allowed=range(256), sys_root=sysr, cgroup_root=cg)
            cpus = got["cpus"]
            assert len(cpus) == 4 * 16, got                  # WIDTH x the cgroup's quota
            lo = node * 64 + k * 16
            cores = [(lo - node * 64 + i) % 64 + node * 64 for i in range(32)]       # four L3 domains from its own share on,
            assert cpus == sorted(cores + [c + 128 for c in cores]), got             # wrapping inside the node; cores + siblings
            assert lo in cpus and lo not in starts           # every server's window starts at its own share of the node
            starts.add(lo)
    assert Placement.cpu_quota(cg) == 16.0


def test_an_explicit_width_of_the_quota_is_sixteen_physical_cores(fake_host):
    import Placement
    sysr, cg, gpus = fake_host
    for node, devs in gpus.items():
        for k, bdf in enumerate(devs):
            got = Placement.plan(bdf, want=16, allowed=range(256), sys_root=sysr, cgroup_root=cg)
            lo = node * 64 + k * 16
            assert got["cpus"] == list(range(lo, lo + 16))


def test_wanting_more_than_the_cores_takes_the_siblings_and_no_quota_takes_the_whole_share(fake_host):
    import Placement
    sysr, cg, gpus = fake_host
    got = Placement.plan("0000:23:00.0", want=24, allowed=range(256), sys_root=sysr, cgroup_root=cg)
    assert got["cpus"] == list(range(16, 32)) + list(range(144, 152))
    os.remove(cg + "/cpu.max")
    got = Placement.plan("0000:a4:00.0", allowed=range(256), sys_root=sysr, cgroup_root=cg)
    assert got["cpus"] == list(range(80, 96)) + list(range(208, 224))


def test_nothing_is_changed_when_there_is_nothing_to_choose(fake_host):
    import Placement
    sysr, cg, _ = fake_host
    assert Placement.plan("0000:23:00.0", allowed=range(8), sys_root=sysr, cgroup_root=cg)["cpus"] == []      # already confined
    got = Placement.plan(None, want=16, allowed=range(256), sys_root=sysr, cgroup_root=cg)                   # no GPU known
    assert got["cpus"] == list(range(16))
    got = Placement.plan("0000:ff:00.0", allowed=range(256), sys_root=sysr, cgroup_root=cg)                  # unknown device
    assert len(got["cpus"]) == 64
    assert Placement.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]


def test_place_applies_an_explicit_list_and_can_be_switched_off():
    import Placement
    before = os.sched_getaffinity(0)
    try:
        assert Placement.place("off") is None and os.sched_getaffinity(0) == before
        one = sorted(before)[:1]
        got = Placement.place(",".join(str(c) for c in one))
        assert got["cpus"] == one and os.sched_getaffinity(0) == set(one)
        assert Placement.place(",".join(str(c) for c in one)) is got       # a second Server of the process: placed already
    finally:
        Placement.apply(before)
    assert os.sched_getaffinity(0) == before
