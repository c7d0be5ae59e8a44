#!/bin/bash
# rocprofv3 kernel stats of constrained-decode calls (bench.py --mode decode-cbs --cbs-legs compiled): usage: bash tools/collect_cbs.sh <tag>
set -e
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_cbs_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --mode decode-cbs --cbs-legs compiled"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 profiles/summarize.py stats $S gpurun_out/${TAG}_cbs_kernel_stats.csv
tail -2 $OUT/stats.log
rm -rf $OUT/stats
head -30 gpurun_out/${TAG}_cbs_kernel_stats.csv
