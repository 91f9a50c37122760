#!/bin/bash
# Profiles the current build on the MI355X box: one rocprofv3 --kernel-trace --stats pass and three PMC passes
# (SQ_VALU_MFMA_BUSY_CYCLES...; FETCH_SIZE; WRITE_SIZE), program directly after `--`, single prediction lane so that
# kernels never overlap.  Writes profiles/<tag>_kernel_stats_B128_single_lane.csv, profiles/<tag>_pmc_B128.md and
# regenerates profiles/traffic.json with the commit the numbers belong to.
#   usage (through gpurun, from the repo root):  bash tools/profile_round.sh <tag> <commit>
set -e -o pipefail
TAG=$1; COMMIT=$2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 100 --warmup 10 --predictors 1 --no-lane-sweep --cpu-seconds 0 --e2e-seconds 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.err
echo "stats pass done"
SHORT="python3 $ROOT/bench.py --steps 20 --warmup 2 --predictors 1 --no-lane-sweep --cpu-seconds 0 --e2e-seconds 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_SQ -- $SHORT > /dev/null 2> $OUT/pmc_SQ.err
echo "SQ pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_FETCH_SIZE -- $SHORT > /dev/null 2> $OUT/pmc_FETCH.err
echo "FETCH pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_WRITE_SIZE -- $SHORT > /dev/null 2> $OUT/pmc_WRITE.err
echo "WRITE pass done"
cd $ROOT
STATS=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
cp $STATS profiles/${TAG}_kernel_stats_B128_single_lane.csv
python3 tools/pmc_table.py $OUT profiles/${TAG}_kernel_stats_B128_single_lane.csv $OUT/table.md $OUT/traffic_new.json
python3 - "$TAG" "$COMMIT" "$OUT" <<'PY'
import json, sys
tag, commit, out = sys.argv[1:4]
new = json.load(open(out + "/traffic_new.json"))
try:
    old = json.load(open("profiles/traffic.json"))
except OSError:
    old = {}
keep = {k: v for k, v in old.items() if k.startswith("frame_frontend")}      # collected by tools/frontend_bench.py passes
merged = dict(keep, **new)
import csv
for r in csv.DictReader(open("profiles/%s_kernel_stats_B128_single_lane.csv" % tag)):
    if "conv_stack_fwd_kernel<false, false>" in r["Name"]:
        merged["conv_stack_fwd_B128_rocprof_avg_us"] = float(r["AverageNs"]) / 1e3     # begin-to-end of each dispatch
merged["commit"] = commit
merged["source"] = "profiles/%s_pmc_B128.md" % tag
merged["frontend_source"] = old.get("frontend_source", "profiles/r01_l_frontend_pmc.md (round 1; the front-end kernel is unchanged since)")
json.dump(merged, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
json.dump(merged, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
PY
cp $OUT/table.md gpurun_out/${TAG}_pmc_table.md
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
cp profiles/${TAG}_kernel_stats_B128_single_lane.csv gpurun_out/
cp $OUT/stats_bench.json gpurun_out/${TAG}_bench_single_lane_under_rocprof.json
echo "profile $TAG done"
