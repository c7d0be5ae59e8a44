#!/bin/bash
# Same-box alternating A/B of two builds of libssc_hip.so on the C2 train step (and optionally the C4 decode):
#   bash tools/ab.sh <variant dir under style-seqcvae_amd, e.g. _base> [pairs] [train|decode]
# A = the variant (SSC_DEBUG=1 SSC_LIB_PATH=...), B = the in-tree library.  Prints ms/step (train) or tokens/s (decode) per run.
VAR=${1:-_base}; PAIRS=${2:-3}; MODE=${3:-train}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIBA=$ROOT/style-seqcvae_amd/$VAR/libssc_hip.so
if [ "$MODE" = decode ]; then ARGS="--mode decode --images 500 --warmup 5"; KEY=value; else ARGS="--timed-only --steps 100 --warmup 20"; KEY=ms_per_step; fi
for i in $(seq 1 $PAIRS); do
  a=$(SSC_DEBUG=1 SSC_LIB_PATH=$LIBA python3 $ROOT/bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['$KEY'])")
  b=$(python3 $ROOT/bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['$KEY'])")
  echo "pair $i: A($VAR) $a   B(in-tree) $b"
done
