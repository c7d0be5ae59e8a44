#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool: bash tools/prof_cmd.sh <tag> <script.py> [args...]
set -e
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $SCRIPT "$@" > $OUT/stats.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 profiles/summarize.py stats $S gpurun_out/${TAG}_kernel_stats.csv
rm -rf $OUT/stats
head -16 gpurun_out/${TAG}_kernel_stats.csv
