"""Where the server's threads and its agent processes run: a compact set of cores next to the GPU.

The reference leaves placement to the operating system (its only device knob is Config.DEVICE, Config.py:62).  On a large
host that is what limits this engine: every prediction is one sleep / wake of an agent and of a batching thread, and with the
threads free to roam 256 logical CPUs each wake lands on a cold, idle core (a trip out of an idle state, every shared cache
line fetched across the socket).  Measured on the 256-CPU / 16-CPU-quota MI355X box with 256 native agents, 2 predictors, 2
trainers (profiles/README.md, round 4): 396 k predictions/s at 38 us of CPU per prediction and the cgroup throttled in every
period when unplaced; 934 k - 1.04 M predictions/s at 7.5 us per prediction and no throttling on 16 CPUs of the GPU's socket.

`plan()` picks the window: whole L3 domains of the GPU's NUMA node, physical cores before their hyper-thread siblings, WIDTH x
the cgroup's CPU quota in all (the quota bounds CPU time, the window bounds where the threads may be: agents that are heavy --
Python processes at ~21 us of CPU per step -- want a hardware thread each when they wake, 64 of them reach 404 k predictions/s
on 16 CPUs, 521 k on 64 and 465 k unplaced, while 256 light native agents do 1.0 M on 16, 32 or 64 CPUs alike), and --
several GPUs per node, one server each -- starting at a different share of the node's L3 domains for each GPU (ordered by PCI
address), so that eight servers on one host do not pile onto the same cores.  `apply()` moves every thread of this process
there; threads and processes started afterwards (batching threads, the fork server, agents, the stats process) inherit it.
A window of exactly the quota's width can never be throttled, and that was tried as the default and as a fallback once
throttling is seen (a governor in Server.main): dropped -- being throttled inside a compact window costs little (64 Python
agents on 32 CPUs: throttled in every period, 424 k predictions/s; on 16 CPUs, never throttled: 404 k), what cost a factor of
eight was roaming the whole host, and the burst of 64 agent processes starting up triggered the governor at once.
Config.CPU_AFFINITY: 'auto' (default), 'off', or an explicit list such as '0-15'.
"""
import glob
import math
import os


def parse_cpulist(text):
    cpus = []
    for part in str(text).strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return cpus


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cpu_quota(root="/sys/fs/cgroup"):
    """CPUs' worth of time the cgroup may use per period (cgroup v2 cpu.max, v1 cpu.cfs_*), or None when unlimited."""
    text = _read(os.path.join(root, "cpu.max"))
    if text:
        quota, _, period = text.partition(" ")
        if quota != "max" and period:
            return int(quota) / float(period)
        return None
    quota, period = _read(os.path.join(root, "cpu/cpu.cfs_quota_us")), _read(os.path.join(root, "cpu/cpu.cfs_period_us"))
    if quota and period and int(quota) > 0:
        return int(quota) / float(period)
    return None


def _l3_domains(cpus, sys_cpu):
    """The given CPUs grouped by the L3 cache they share, each group ordered physical cores first, siblings after."""
    seen, groups = set(), []
    allowed = set(cpus)
    for cpu in sorted(cpus):
        if cpu in seen:
            continue
        shared = _read("%s/cpu%d/cache/index3/shared_cpu_list" % (sys_cpu, cpu))
        members = [c for c in (parse_cpulist(shared) if shared else [cpu]) if c in allowed and c not in seen]
        seen.update(members)
        first, rest, done = [], [], set()
        for c in sorted(members):
            if c in done:
                continue
            sib = _read("%s/cpu%d/topology/thread_siblings_list" % (sys_cpu, c))
            threads = [t for t in (parse_cpulist(sib) if sib else [c]) if t in members]
            done.update(threads)
            first.append(min(threads) if threads else c)
            rest.extend(sorted(t for t in threads if t != min(threads)))
        groups.append((first, rest))
    return groups


def _gpus_on_node(node, sys_pci):
    """PCI addresses of the AMD display / accelerator functions on one NUMA node (one per physical GPU), sorted."""
    out = []
    for dev in sorted(glob.glob(os.path.join(sys_pci, "*"))):
        if _read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        cls = _read(os.path.join(dev, "class")) or ""
        if not (cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200")):
            continue
        if (_read(os.path.join(dev, "numa_node")) or "-1") == str(node):
            out.append(os.path.basename(dev))
    return out


WIDTH = 4          # CPUs in the window per CPU of quota


def plan(pci_bus_id=None, want=None, allowed=None, sys_root="/sys", cgroup_root="/sys/fs/cgroup"):
    """-> {"cpus": [...], "why": str}: the CPUs to run on (empty: leave the process where it is)."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    quota = cpu_quota(cgroup_root)
    n = int(want) if want else (WIDTH * int(math.ceil(quota)) if quota else 0)
    sys_cpu = os.path.join(sys_root, "devices/system/cpu")
    sys_pci = os.path.join(sys_root, "bus/pci/devices")
    node_cpus, rank, peers, node = allowed, 0, 1, None
    if pci_bus_id:
        dev = os.path.join(sys_pci, pci_bus_id.lower())
        local = _read(os.path.join(dev, "local_cpulist"))
        node = _read(os.path.join(dev, "numa_node"))
        if local:
            near = [c for c in parse_cpulist(local) if c in set(allowed)]
            if near:
                node_cpus = near
        if node is not None and node != "-1":
            gpus = _gpus_on_node(node, sys_pci)
            if pci_bus_id.lower() in gpus:
                rank, peers = gpus.index(pci_bus_id.lower()), len(gpus)
    domains = _l3_domains(node_cpus, sys_cpu)
    share = max(1, len(domains) // max(peers, 1))
    mine = domains[(rank * share) % len(domains):][:share] if domains else []
    if not mine:
        return {"cpus": [], "why": "no topology information"}
    if not n:                                   # no quota: the GPU's whole share of the node
        n = sum(len(f) + len(r) for f, r in mine)
    # a window larger than this GPU's share goes on into the following domains of the node; whole domains, so that the
    # cores come before the siblings in what is taken
    k = rank * share + share
    while sum(len(f) + len(r) for f, r in mine) < n and len(mine) < len(domains):
        mine.append(domains[k % len(domains)])
        k += 1
    order = [c for f, _ in mine for c in f] + [c for _, r in mine for c in r]
    cpus = sorted(order[:n])
    if len(cpus) >= len(allowed):
        return {"cpus": [], "why": "the process is already confined to %d CPUs" % len(allowed)}
    if len(cpus) < 2:
        return {"cpus": [], "why": "fewer than two CPUs to choose"}
    return {"cpus": cpus, "quota": quota,
            "why": "%d CPUs (quota %s) in %d L3 domain(s) of NUMA node %s, GPU %s is number %d of %d on it"
            % (len(cpus), ("%.1f" % quota) if quota else "none", len(mine), node, pci_bus_id, rank + 1, peers)}


def apply(cpus):
    """Every thread of this process onto `cpus` (threads and children started later inherit it)."""
    moved = 0
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
            moved += 1
        except OSError:
            pass                                # the thread ended meanwhile
    return moved


_applied = None      # what place() did to this process last (a second Server of the process finds itself placed already)


def place(setting="auto", device=0):
    """Config.CPU_AFFINITY -> the CPUs chosen (None when nothing was changed)."""
    global _applied
    if setting is None or setting is False or setting in ("", "off"):
        return None
    if _applied is not None and set(_applied.get("server_cpus", _applied["cpus"])) == os.sched_getaffinity(0) and \
            _applied.get("setting") == (setting, int(device)):
        return _applied
    if setting != "auto":
        cpus = parse_cpulist(setting)
        apply(cpus)
        _applied = {"cpus": cpus, "why": "Config.CPU_AFFINITY = %r" % (setting,), "setting": (setting, int(device))}
        return _applied
    import ctypes as C
    import _native as nat
    bus = None
    buf = C.create_string_buffer(64)
    if nat.hip_lib().ga3c_device_pci_bus_id(int(device), buf, 64) == 0:     # (no GPU: nothing to be next to)
        bus = buf.value.decode()
    got = plan(bus, want=int(os.environ.get("GA3C_PLACE_WIDTH", "0")) or None)
    if got["cpus"]:
        k = int(os.environ.get("GA3C_SERVER_CPUS", "0"))
        if 0 < k < len(got["cpus"]) - 1:
            got["server_cpus"], got["agent_cpus"] = got["cpus"][:k], got["cpus"][k:]
            apply(got["server_cpus"])
        else:
            apply(got["cpus"])
        got["setting"] = (setting, int(device))
        _applied = got
        return got
    return None
