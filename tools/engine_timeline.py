#!/usr/bin/env python3
"""Development aid: what happens to a prediction step on the GPU while the engine trains?  Reads a rocprofv3 --kernel-trace
CSV of a running engine (tools/engine_ceiling.py: native agent threads, so nothing forks under the profiler) and prints,
per kind of step, the span from its first kernel's start to its last kernel's end against the sum of its kernels' own
durations, the gaps between its kernels, each kernel's duration against the same kernel when nothing else runs, and how
much of the time a train kernel holds the chip.
usage: engine_timeline.py kernel_trace.csv [window-rows]"""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.split('(')[0].replace('void ', '').replace('ga3c::', '')[:44]


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))] if v else float('nan')


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        r['n'] = short(r['Kernel_Name'])
    rows.sort(key=lambda r: r['s'])
    t_lo, t_hi = rows[len(rows) // 4]['s'], rows[3 * len(rows) // 4]['s']          # the middle half: steady state
    mid = [r for r in rows if t_lo <= r['s'] < t_hi]
    by_q = defaultdict(list)
    for r in mid:
        by_q[r['Queue_Id']].append(r)
    print("%d dispatches in the middle %.1f ms; queues: %s" % (len(mid), (t_hi - t_lo) / 1e6,
          {q: len(v) for q, v in by_q.items()}))
    # which queue is what: by the kernels on it
    kinds = {}
    for q, v in by_q.items():
        names = defaultdict(int)
        for r in v:
            names[r['n']] += 1
        top = max(names, key=names.get)
        kinds[q] = 'train' if any('bwd' in n or 'slab' in n for n in names) else ('gather' if 'gather' in top else 'predict')
    print("queue kinds:", kinds)
    # busy fractions
    for kind in ('train', 'predict', 'gather'):
        iv = sorted((r['s'], r['e']) for q, v in by_q.items() if kinds[q] == kind for r in v)
        busy, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        print("%-8s kernels hold some CU %.1f %% of the time" % (kind, 100.0 * busy / (t_hi - t_lo)))
    # prediction steps: conv stack (or conv1 + conv2) -> dense1 -> heads on one queue
    steps = []
    for q, v in by_q.items():
        if kinds[q] != 'predict':
            continue
        cur = []
        for r in v:
            if ('conv_stack_fwd' in r['n'] or 'conv1_fwd' in r['n']) and cur:
                cur = []
            cur.append(r)
            if 'heads_kernel' in r['n']:
                if len(cur) >= 3:
                    steps.append(cur)
                cur = []
    print("%d prediction steps" % len(steps))
    if steps:
        span = [(st[-1]['e'] - st[0]['s']) / 1e3 for st in steps]
        own = [sum(r['e'] - r['s'] for r in st) / 1e3 for st in steps]
        gaps = [sum(st[i + 1]['s'] - st[i]['e'] for i in range(len(st) - 1)) / 1e3 for st in steps]
        print("  span first start -> last end: median %.1f  p90 %.1f us;  kernels' own durations: median %.1f  p90 %.1f;  "
              "gaps between them: median %.1f  p90 %.1f" % (pct(span, .5), pct(span, .9), pct(own, .5), pct(own, .9),
                                                            pct(gaps, .5), pct(gaps, .9)))
    # durations per kernel name
    dur = defaultdict(list)
    for r in mid:
        dur[r['n']].append((r['e'] - r['s']) / 1e3)
    print("  per kernel (us): median / p90 / count")
    for n, v in sorted(dur.items(), key=lambda kv: -len(kv[1])):
        if len(v) >= 20:
            print("    %-46s %7.1f %7.1f %7d" % (n, pct(v, .5), pct(v, .9), len(v)))
    # how long does a prediction kernel wait for its predecessor's successor slot: start - previous kernel's end on the queue
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if w:
        first = len(mid) // 2
        t0 = mid[first]['s']
        print("window of %d dispatches:" % w)
        for r in mid[first:first + w]:
            print("  %9.2f -> %9.2f (%6.2f) %-7s %s" % ((r['s'] - t0) / 1e3, (r['e'] - t0) / 1e3, (r['e'] - r['s']) / 1e3,
                                                       kinds[r['Queue_Id']], r['n']))


if __name__ == '__main__':
    main()
