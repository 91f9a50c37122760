"""Development aid: user / system CPU seconds of every thread of the processes whose command line contains a pattern."""
import os
import sys

pat = sys.argv[1]
tck = os.sysconf("SC_CLK_TCK")
for pid in filter(str.isdigit, os.listdir("/proc")):
    try:
        cmd = open("/proc/%s/cmdline" % pid).read().replace("\0", " ")
    except OSError:
        continue
    if pat not in cmd or "cpu_split" in cmd:
        continue
    ut = st = n = 0
    for tid in os.listdir("/proc/%s/task" % pid):
        try:
            f = open("/proc/%s/task/%s/stat" % (pid, tid)).read().rsplit(")", 1)[1].split()
        except OSError:
            continue
        ut += int(f[11]); st += int(f[12]); n += 1
    print("%s pid %s threads %d user %.1f s sys %.1f s | %s" % (pat, pid, n, ut / tck, st / tck, cmd[:60]))
