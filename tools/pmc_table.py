#!/usr/bin/env python3
"""Builds the per-kernel table of profiles/rNN_*_pmc_*.md from rocprofv3 outputs.

usage: pmc_table.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_SQ/ (counter_collection.csv each)> <kernel_stats.csv of an
       un-profiled-counter run> <out.md> [traffic.json]
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz), duration from --kernel-trace --stats
of a run WITHOUT counters (counter collection stretches the kernels).
"""
import collections
import csv
import glob
import json
import sys


def short(n):
    return n.split('(')[0].replace('void ', '').replace('ga3c::', '')


def counters(pat):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def main():
    root, stats, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fe = counters(root + '/pmc_FETCH_SIZE/*/*counter_collection.csv')
    wr = counters(root + '/pmc_WRITE_SIZE/*/*counter_collection.csv')
    sq = counters(root + '/pmc_SQ/*/*counter_collection.csv')
    dur = {short(r['Name']): float(r['AverageNs']) / 1e3 for r in csv.DictReader(open(stats))}
    rows, traffic = [], {}
    for k in sorted(fe):
        if k.startswith('__'):
            continue
        mean = lambda v: sum(v) / len(v)
        f, w = mean(fe[k]['FETCH_SIZE']), mean(wr[k]['WRITE_SIZE'])
        m = mean(sq[k]['SQ_VALU_MFMA_BUSY_CYCLES']) if k in sq else 0.0
        hbm = (2 * f + w) * 1024
        us = dur.get(k)
        rows.append((k, f, w, hbm, m, us))
        traffic[k.replace('_kernel', '').split('<')[0] + ('_train' if '<true' in k else '') + '_B128'] = hbm
    with open(out, 'w') as o:
        o.write('| kernel | FETCH_SIZE (KB) | WRITE_SIZE (KB) | HBM bytes / launch | avg duration (us, no counters) | HBM GB/s | MFMA busy cycles | MFMA utilisation |\n')
        o.write('|---|---|---|---|---|---|---|---|\n')
        for k, f, w, hbm, m, us in rows:
            gbs = '%.0f' % (hbm / us / 1e3) if us else '-'
            util = '%.1f %%' % (100 * m / (1024 * us * 2400)) if us and m else '-'
            o.write('| %s | %.0f | %.0f | %.2f MB | %s | %s | %.0f | %s |\n' % (k, f, w, hbm / 1e6, '%.2f' % us if us else '-', gbs, m, util))
    if len(sys.argv) > 4:
        json.dump(traffic, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
