#!/bin/bash
# HBM-side and L2 traffic of the 2xFP16 128x128 product at the decode's shapes: FETCH_SIZE / WRITE_SIZE / L2 hit-miss counters of
# tools/f16_probe.py (separate --pmc passes, no trace option).   usage: bash tools/f16_traffic.sh <tag>
set -e
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_f16_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/f16_probe.py --one"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
cd $ROOT
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
T=$(find $OUT/tcc -name "*counter_collection.csv" | head -1)
python3 profiles/summarize.py pmc $F $W gpurun_out/${TAG}_f16_pmc_traffic.json
python3 - $T <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    if "gemm" in k:
        print(k, {c: x / n[(k, c)] for c, x in v.items()})
PY
cat gpurun_out/${TAG}_f16_pmc_traffic.json
rm -rf $OUT/fetch $OUT/write $OUT/tcc
