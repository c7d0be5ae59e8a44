#!/bin/bash
# rocprofv3 kernel stats + matrix-pipe counters of the C4 decode leg (bench.py --mode decode): bash tools/collect_decode.sh <tag>
set -e
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_dec_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --mode decode --images 200 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $CMD > $OUT/grbm.log 2>&1 || echo "grbm pass failed"
cd $ROOT
S=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); Q=$(find $OUT/sq -name "*counter_collection.csv" | head -1); G=$(find $OUT/grbm -name "*counter_collection.csv" | head -1)
python3 profiles/summarize.py stats $S gpurun_out/${TAG}_decode_kernel_stats.csv
[ -n "$Q" ] && [ -n "$G" ] && python3 profiles/summarize.py sq $Q $G $S gpurun_out/${TAG}_decode_sq_mfma.csv || true
tail -c 600 $OUT/stats.log
rm -rf $OUT/stats $OUT/sq $OUT/grbm
head -24 gpurun_out/${TAG}_decode_kernel_stats.csv
