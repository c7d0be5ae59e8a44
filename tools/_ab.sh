set -e
for i in 1 2; do
for w in 0 1; do echo "== WT $w"; SSC_STORE_WT=$w timeout -k 10 200 python bench.py --steps 30 --warmup 5 --timed-only | tail -1; done
done
SSC_STORE_WT=1 timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py -x -q | tail -1
