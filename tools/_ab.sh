set -e
for i in 1 2; do
echo "== base"; SSC_LIB_PATH=$PWD/style-seqcvae_amd/_base/libssc_hip.so timeout -k 10 200 python bench.py --steps 30 --warmup 5 --timed-only | tail -1
echo "== new"; timeout -k 10 200 python bench.py --steps 30 --warmup 5 --timed-only | tail -1
done
