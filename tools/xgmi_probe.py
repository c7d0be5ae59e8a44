#!/usr/bin/env python3
"""Two (or N) ranks on ONE GPU exercising the direct hipIpc all-reduce (ssc_runtime/xgmi.py): set-up, self-test, timing.
python tools/xgmi_probe.py [world] [MB]"""
import os
import sys
import time
import traceback

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "style-seqcvae_amd")]
import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, mb):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from ssc_runtime.xgmi import XgmiAllReduce
        n = mb * 1024 * 1024 // 4
        flat = torch.full((n,), float(rank + 1), device="cuda")
        print(f"[{rank}] setting up", flush=True)
        t0 = time.time()
        xg = XgmiAllReduce(flat, verify=False)
        print(f"[{rank}] mapped peers in {time.time() - t0:.2f}s", flush=True)
        dist.barrier()
        xg.allreduce(0, n)
        torch.cuda.synchronize()
        bad = (flat != 3.0).nonzero().flatten()
        print(f"[{rank}] result mismatches {bad.numel()} first {bad[:8].tolist()} last {bad[-8:].tolist()} vals {flat[bad[:8]].tolist() if bad.numel() else []}", flush=True)
        print(f"[{rank}] first allreduce done: err={int(xg.err.item())} head={flat[:4].tolist()} tail={flat[-4:].tolist()}", flush=True)
        xg.self_test()
        print(f"[{rank}] self-test ok", flush=True)
        for _ in range(3):
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            xg.allreduce(0, n)
            e1.record()
            torch.cuda.synchronize()
            print(f"[{rank}] allreduce {mb} MB: {e0.elapsed_time(e1):.3f} ms", flush=True)
        xg.check()
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        traceback.print_exc()
        sys.stdout.flush()
        os._exit(3)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, 29611, mb)) for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.time() + 120
    for p in procs:
        p.join(max(1, deadline - time.time()))
    for p in procs:
        if p.is_alive():
            print("rank still alive after the deadline: killing", p.pid, flush=True)
            p.kill()
    print("exit codes", [p.exitcode for p in procs])
