#!/usr/bin/env python3
"""Development aid: reads a rocprofv3 --kernel-trace CSV and prints, for a window of dispatches, start / end offsets per
stream, the overlap between consecutive kernels and the idle gaps.  usage: timeline.py kernel_trace.csv [first] [count]"""
import csv
import sys


def short(n):
    return n.split('(')[0].replace('void ', '').replace('ga3c::', '')[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    if len(sys.argv) > 2 and not sys.argv[2].lstrip('-').isdigit():
        # timeline.py trace.csv NAME N [count]: start at the N-th dispatch whose kernel name contains NAME
        hits = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
        first = hits[int(sys.argv[3])] if len(sys.argv) > 3 else hits[len(hits) // 2]
        count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    else:
        first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
        count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    t0 = int(rows[first]['Start_Timestamp'])
    busy_end = 0
    for r in rows[first:first + count]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        q = r.get('Queue_Id', '?')
        print("%9.2f -> %9.2f us  (%6.2f)  q=%s  %s  gap_since_any_end=%6.2f" % (s / 1e3, e / 1e3, (e - s) / 1e3, q, short(r['Kernel_Name']),
                                                                          (s - busy_end) / 1e3))
        busy_end = max(busy_end, e)


if __name__ == '__main__':
    main()
