#!/bin/bash
# rocprofv3 evidence for profiles/ (run on the GPU box through gpurun from the repo root; raw output under gpurun_out/prof_<tag>,
# summaries under gpurun_out/<tag>_*).  Every pass profiles the SAME command: the timed train steps only (bench.py --timed-only),
# so every launch in the tables belongs to the C2 train step.  Counters are collected in their own passes, without any trace option
# besides the implicit kernel dispatch records (gpurun refuses --pmc together with --sys-trace and friends).
#   usage: bash tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --timed-only --prewarm 0"
CMD2="python3 $ROOT/bench.py --steps 2 --warmup 1 --timed-only --prewarm 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD2 > $OUT/fetch.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD2 > $OUT/write.log 2>&1
echo "WRITE_SIZE pass done"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq -- $CMD2 > $OUT/sq.log 2>&1
echo "SQ pass done"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $CMD2 > $OUT/grbm.log 2>&1
echo "GRBM pass done"
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
Q=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
G=$(find $OUT/grbm -name "*counter_collection.csv" | head -1)
python3 profiles/summarize.py stats $S gpurun_out/${TAG}_kernel_stats.csv
python3 profiles/summarize.py pmc $F $W gpurun_out/${TAG}_pmc_traffic.json
python3 profiles/summarize.py sq $Q $G $S gpurun_out/${TAG}_sq_mfma.csv
python3 tools/timeline.py $T prep_tokens_kernel 0 > gpurun_out/${TAG}_timeline.txt
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/grbm
echo "summaries written: gpurun_out/${TAG}_*"
