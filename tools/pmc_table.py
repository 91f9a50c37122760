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


def traffic_key(k):
    """conv_stack_fwd_kernel<true, false> -> conv_stack_fwd_train_B128; <false, true> -> conv_stack_fwd_u8_B128;
    conv1_dw_kernel<true> -> conv1_dw_u8_B128 (its only template argument is the input format)."""
    base = k.replace('_kernel', '').split('<')[0]
    args = [a.strip() for a in k.split('<', 1)[1].rstrip('>').split(',')] if '<' in k else []
    tag = ''
    if base in ('conv_stack_fwd',) and len(args) >= 2:
        tag = ('_train' if args[0] == 'true' else '') + ('_u8' if args[1] == 'true' else '')
    elif base in ('conv1_dw', 'conv1_fwd', 'conv_bwd') and args:
        tag = '_u8' if args[0] == 'true' else ''
    elif base in ('dense1_fwd_tile', 'dense1_fwd') and args:
        tag = '' if args[0] == '1' else '_mt' + args[0]
    elif base in ('dense1_bwd_tile', 'slab_reduce') and args:
        tag = '_upd' if args[0] == 'true' else ''
    elif base == 'heads' and args:
        tag = '_train' if args[0] == 'true' else ''
    elif args and base not in ('dense1_fwd', 'dense1_fwd_tile', 'rmsprop', 'frame_frontend', 'conv2_dx'):
        tag = '_' + '_'.join(args)
    return base + tag + '_B128'


# Bytes each launch has to move at batch 128, A = 6, given the kernel's own decomposition (inputs read once, outputs and
# partial slabs written once); what "traffic well above the algorithmic bytes" is measured against.
MB = 1e6
X_F32, X_U8, N1, N2, WD, PART = 128 * 112896 / MB, 128 * 28224 / MB, 128 * 28224 / MB, 128 * 15488 / MB, 3872 * 256 * 4 / MB, 16 * 128 * 1024 / MB
ALGORITHMIC_MB = {
    "conv_stack_fwd_kernel<false, false>": X_F32 + N2 + 0.05, "conv_stack_fwd_kernel<false, true>": X_U8 + N2 + 0.05,
    "conv_stack_fwd_kernel<true, false>": X_F32 + N2 + N1 + 0.05, "conv_stack_fwd_kernel<true, true>": X_U8 + N2 + N1 + 0.05,
    "conv1_fwd_kernel<false>": X_F32 + N1, "conv2_fwd_kernel": N1 + N2,
    "dense1_fwd_tile_kernel<1>": N2 + WD + PART, "dense1_fwd_kernel<1>": N2 + WD + PART,
    "heads_kernel<false, 8>": PART + 0.14, "heads_kernel<true, 8>": PART + 0.28,
    # (the step of dense1/w: ms read + written, weights written, packed copy written: 4 x 3.96 MB in the epilogue, where the
    # gradient is on chip and the weights are read for the dn2 product anyway; 6 x in conv_bwd, which reads both)
    # <0>: gradients only; <1>: + the optimizer step of dense1/w in the epilogue; <2>: the step of the small parameters only
    # (dense1/w is stepped by the next launch, conv_bwd<.., true>: gradient, ms and weights read, ms, weights and packed copy written)
    "dense1_bwd_tile_kernel<0>": N2 + 0.13 + WD + WD + N2, "dense1_bwd_tile_kernel<1>": N2 + 0.13 + WD + WD + N2 + 4 * WD,
    "dense1_bwd_tile_kernel<2>": N2 + 0.13 + WD + WD + N2,
    "dense1_bwd_kernel": N2 + 0.13 + WD + WD + N2,
    "dense1_dw_kernel": N2 + 0.13 + WD, "dense1_dx_kernel": 0.13 + WD + N2 + N2,
    "conv2_dw_kernel": N1 + N2 + 128 * 8224 * 4 / MB, "conv2_dx_kernel": N2 + N1 + N1,
    "conv1_dw_kernel<false>": X_F32 + N1 + 256 * 4112 * 4 / MB, "conv1_dw_kernel<true>": X_U8 + N1 + 256 * 4112 * 4 / MB,
    "conv_bwd_kernel<false, false>": X_F32 + N1 + N2 + 256 * (4112 + 8224) * 4 / MB, "conv_bwd_kernel<true, false>": X_U8 + N1 + N2 + 256 * (4112 + 8224) * 4 / MB,
    "conv_bwd_kernel<false, true>": X_F32 + N1 + N2 + 256 * (4112 + 8224) * 4 / MB + 6 * WD, "conv_bwd_kernel<true, true>": X_U8 + N1 + N2 + 256 * (4112 + 8224) * 4 / MB + 6 * WD,
    # slabs read once: train steps reduce the 256 slab pairs of conv_bwd (256 x (4112 + 8224) floats), the per-kernel timer
    # the 512 + 128 slabs of the split kernels -- 12.63 MB either way
    "slab_reduce_kernel<false>": 256 * (4112 + 8224) * 4 / MB,
    "slab_reduce_kernel<true>": 256 * (4112 + 8224) * 4 / MB + 5 * 12336 * 4 / MB,       # + RMSProp over the 12,336 conv parameters
    "rmsprop_kernel<false, false>": 5 * 4.0225 + WD + 0.03,
    "pack_wd_kernel": 2 * WD, "frame_frontend_kernel<3>": 256 * 157248 / MB,
}


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
        traffic[traffic_key(k)] = hbm
    with open(out, 'w') as o:
        o.write('| kernel | FETCH_SIZE (KB) | WRITE_SIZE (KB) | HBM bytes / launch | algorithmic bytes | ratio | avg duration (us, no counters) | HBM GB/s | MFMA busy cycles | MFMA utilisation |\n')
        o.write('|---|---|---|---|---|---|---|---|---|---|\n')
        for k, f, w, hbm, m, us in rows:
            gbs = '%.0f' % (hbm / us / 1e3) if us else '-'
            util = '%.1f %%' % (100 * m / (1024 * us * 2400)) if us and m else '-'
            alg = ALGORITHMIC_MB.get(k)
            o.write('| %s | %.0f | %.0f | %.2f MB | %s | %s | %s | %s | %.0f | %s |\n'
                    % (k, f, w, hbm / 1e6, '%.2f MB' % alg if alg else '-', '%.2f' % (hbm / 1e6 / alg) if alg else '-',
                       '%.2f' % us if us else '-', gbs, m, util))
    if len(sys.argv) > 4:
        json.dump(traffic, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
