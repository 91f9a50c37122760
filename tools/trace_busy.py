#!/usr/bin/env python3
"""Development aid: how busy the GPU was in a window of a rocprofv3 --kernel-trace CSV.
usage: trace_busy.py kernel_trace.csv [start fraction 0..1] [window ms]
Prints the union of all kernels' intervals over the window (= time with at least one kernel running), the time with two or
more running, the share of each kernel name (sum of durations / window) and the longest idle gaps."""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.split('(')[0].replace('void ', '').replace('ga3c::', '')[:44]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')))
    rows.sort()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    win = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 200e6
    t0 = rows[0][0] + int((rows[-1][1] - rows[0][0]) * frac)
    t1 = t0 + int(win)
    sel = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in rows if e > t0 and s < t1]
    events = []
    for s, e, _, _ in sel:
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, last, busy1, busy2, gaps = 0, t0, 0, 0, []
    for t, d in events:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth == 0 and t > last:
            gaps.append(t - last)
        last = t
        depth += d
    w = t1 - t0
    print("window %.1f ms, %d dispatches (%.0f per ms); at least one kernel running %.1f %%, two or more %.1f %%"
          % (w / 1e6, len(sel), len(sel) / (w / 1e6), 100.0 * busy1 / w, 100.0 * busy2 / w))
    gaps.sort(reverse=True)
    print("idle gaps: %d, total %.1f %%, longest (us): %s" % (len(gaps), 100.0 * sum(gaps) / w, [round(g / 1e3, 1) for g in gaps[:8]]))
    share, count = defaultdict(int), defaultdict(int)
    for s, e, n, _ in sel:
        share[n] += e - s
        count[n] += 1
    for n, v in sorted(share.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-46s %6.1f %% of the window, %6d launches, %6.2f us each" % (n, 100.0 * v / w, count[n], v / count[n] / 1e3))
    byq = defaultdict(int)
    for s, e, _, q in sel:
        byq[q] += e - s
    print("  by queue:", {q: "%.1f %%" % (100.0 * v / w) for q, v in sorted(byq.items())})


if __name__ == '__main__':
    main()
