set -e
for m in 3 1 2; do echo "== X3B $m"; SSC_X3B=$m timeout -k 10 200 python bench.py --mode decode --images 1000 | tail -1 | cut -c1-200; done
for m in 3 2; do echo "== train X3B $m"; SSC_X3B=$m timeout -k 10 200 python bench.py --steps 30 --warmup 5 --timed-only | tail -1; done
