#!/bin/bash
# launch-by-launch timeline of ONE 100-image beam-search call (between two feat_mask_kernel launches): bash tools/collect_decode_timeline.sh <tag>
set -e
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_dect_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode decode --images 300 --warmup 2 > $OUT/trace.log 2>&1
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $T feat_mask_kernel 45 5 > gpurun_out/${TAG}_decode_timeline.txt
rm -rf $OUT/trace
head -5 gpurun_out/${TAG}_decode_timeline.txt; tail -22 gpurun_out/${TAG}_decode_timeline.txt
