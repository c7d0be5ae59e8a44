"""GPU (two processes sharing cuda:0, gloo collectives - the box has one GPU): a data-parallel fused train step
(phased backward + overlapped all-reduce of the flat gradient ranges + clip/SGD with the 1/world scale) leaves every
rank with the parameters a single process gets on the whole batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import oracle

pytestmark = pytest.mark.gpu


def _join_then_get(procs, ret, timeout=180):
    """The result of the rank processes without a blocking get(): a rank that died never puts, and get() would hang the run.
    (The result is read BEFORE the producers exit: tensors travel through the queue as shared-memory handles.)"""
    import time
    deadline = time.time() + timeout
    while ret.empty() and time.time() < deadline and all(p.is_alive() or p.exitcode == 0 for p in procs):
        time.sleep(0.05)
    got = ret.get() if not ret.empty() else None
    for p in procs:
        p.join(max(1.0, deadline - time.time()))
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join(10)
    assert [p.exitcode for p in procs] == [0] * len(procs), [p.exitcode for p in procs]
    assert got is not None
    return got


CFG = dict(vocab_size=90, image_feature_size=32, embedding_size=20, hidden_size=24, attention_projection_size=16, z_space=8,
           max_caption_length=6, sentiment_vae=1, senti_prior_multip=0.5)


def _inputs():
    g = torch.Generator().manual_seed(21)
    B, R, L = 8, 4, 6
    feats = torch.randn(B, R, 32, generator=g)
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = 2 + b % 5
        caps[b, :n] = torch.randint(2, 90, (n,), generator=g)
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(L + 1, B, 8, generator=g)
    return feats, caps, senti, eps


def _step(eng, feats, caps, senti, eps, frozen):
    eng.train_step(feats.cuda(), caps.cuda(), senti.cuda(), eps.cuda().contiguous(), lr=0.02, kld_weight=750.0, momentum=0.9,
                   weight_decay=0.001, max_norm=0.7, decoder_frozen=frozen)


def _worker(rank, world, port, ret, algo="rccl"):
    import torch.distributed as dist
    from gpuutil import engine_from
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    eng.dp_algo = algo
    feats, caps, senti, eps = _inputs()
    per = feats.size(0) // world
    sl = slice(rank * per, (rank + 1) * per)
    for frozen in (True, False):
        _step(eng, feats[sl], caps[sl], senti[sl], eps[:, sl], frozen)
    torch.cuda.synchronize()
    if algo != "rccl":
        assert eng.dp_choice is not None and (algo == "auto" or eng.dp_choice["algo"] == "xgmi"), eng.dp_choice
        if eng._xgmi is not None:
            eng._xgmi.check()
    if rank == 0:
        ret.put({k: v.cpu().numpy() for k, v in eng.state_dict().items()})   # by value: the parent may read after this rank exits
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["rccl", "xgmi", "auto"])
def test_two_rank_train_step_equals_single_process(algo):
    """algo: the gradient exchange of the overlapped backward - torch.distributed all-reduce (gloo here), the direct hipIpc
    reduce-scatter + all-gather kernels (csrc/collective.hip; both ranks map each other's flat gradient buffer - IPC handles work
    device-local), or the start-up choice between the two."""
    from gpuutil import engine_from, maxdiff
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, algo)) for r in range(2)]
    for p in procs:
        p.start()
    got = {k: torch.from_numpy(v) for k, v in _join_then_get(procs, ret).items()}
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    feats, caps, senti, eps = _inputs()
    for frozen in (True, False):
        _step(eng, feats, caps, senti, eps, frozen)
    want = eng.state_dict()
    for k, v in want.items():
        assert maxdiff(got[k], v) < 2e-6, k


def _rccl_worker(port, ret):
    """One rank on the REAL RCCL backend ("nccl" on ROCm; the gloo tests above never touch it): process-group init bound to the
    device, the phased backward with its four asynchronous all-reduces of flat gradient ranges, work.wait() on the compute
    stream, the exposure timing, a barrier and the float64 MAX reduction bench.py uses for its clock."""
    import torch.distributed as dist
    from gpuutil import engine_from
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    eng.dp_force = True
    eng.dp_profile = True
    feats, caps, senti, eps = _inputs()
    for frozen in (True, False):
        _step(eng, feats, caps, senti, eps, frozen)
    exposure = eng.dp_exposure_ms()
    dist.barrier()
    t = torch.tensor([1.25], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    ret.put(({k: v.cpu().numpy() for k, v in eng.state_dict().items()}, float(t.item()), exposure))
    dist.destroy_process_group()


def test_overlapped_backward_on_the_rccl_backend_one_rank():
    from gpuutil import engine_from, maxdiff
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = ctx.Process(target=_rccl_worker, args=(port, ret))
    p.start()
    got, tmax, exposure = _join_then_get([p], ret)
    got = {k: torch.from_numpy(v) for k, v in got.items()}
    assert tmax == 1.25
    assert len(exposure) == 2 and all(x >= 0.0 for x in exposure)
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    feats, caps, senti, eps = _inputs()
    for frozen in (True, False):
        _step(eng, feats, caps, senti, eps, frozen)
    for k, v in eng.state_dict().items():
        assert maxdiff(got[k], v) < 2e-6, k


def _xgmi_worker(rank, world, port, ret):
    import torch.distributed as dist
    from ssc_runtime.xgmi import XgmiAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4 * 700_003                      # no multiple of the shard / grid sizes
    vals = [torch.randint(-500, 500, (n,), generator=torch.Generator().manual_seed(50 + r)).float() for r in range(world)]
    total = sum(vals)
    flat = vals[rank].cuda()
    xg = XgmiAllReduce(flat, verify=True)            # maps the peers, self-test against dist.all_reduce
    assert torch.equal(flat.cpu(), vals[rank])       # the self-test restores what it touched
    side = torch.cuda.Stream()
    want = vals[rank].clone()
    # whole buffer, a range with uneven shards, a 4-float range (one rank's shard is empty), a range on a side stream,
    # the same range twice (sums of sums), an empty range
    for lo, hi, st in [(0, n, None), (8, 8 + 4 * 1001, None), (n - 4, n, None), (4 * 1000, 4 * 300_000, side), (0, 64, None), (0, 64, None), (16, 16, None)]:
        dist.barrier()
        if st is not None:
            st.wait_stream(torch.cuda.current_stream())
        xg.allreduce(lo, hi, stream=st)
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        # expected: the sum over ranks of what each rank held in [lo, hi) before the call (tracked per rank in `want`)
        ret_vals = flat.cpu()
        dist.barrier()
        gathered = [torch.empty_like(want[lo:hi]) for _ in range(world)]
        dist.all_gather(gathered, want[lo:hi].contiguous())
        want[lo:hi] = sum(gathered)
        assert torch.equal(ret_vals, want), (lo, hi, (ret_vals - want).abs().max())
    xg.check()
    # a range that was reduced exactly once (by the whole-buffer call only) against the independently computed sum
    assert torch.equal(flat.cpu()[4 * 300_000:n - 4], total[4 * 300_000:n - 4])
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_xgmi_direct_allreduce_ranks_sharing_the_gpu(world):
    """ssc_xgmi_allreduce (reduce-scatter + all-gather kernels over hipIpc peer mappings, flag-ordered across the processes)
    against torch.distributed: exact sums of integer-valued floats over whole buffers, sub-ranges, one-unit ranges (some ranks'
    shards empty), a side stream; two ranks, and three (an odd world size: uneven shards, two peers per rank)."""
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_xgmi_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    assert _join_then_get(procs, ret) == "ok"
