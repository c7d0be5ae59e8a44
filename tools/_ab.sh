set -e
timeout -k 10 500 python -m pytest tests/test_dp_gpu.py tests/test_train_gpu.py -x -q > gpurun_out/dp2.log 2>&1 || { tail -30 gpurun_out/dp2.log; exit 1; }
tail -2 gpurun_out/dp2.log
SSC_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-decode --prewarm 20 > gpurun_out/dp_bench.log 2>gpurun_out/dp_bench.err || { tail -20 gpurun_out/dp_bench.err; exit 1; }
python -c "
import json
for l in open('gpurun_out/dp_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('data_parallel'))
"
