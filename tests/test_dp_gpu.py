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


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from gpuutil import engine_from
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    feats, caps, senti, eps = _inputs()
    per = feats.size(0) // world
    sl = slice(rank * per, (rank + 1) * per)
    for frozen in (True, False):
        _step(eng, feats[sl], caps[sl], senti[sl], eps[:, sl], frozen)
    torch.cuda.synchronize()
    if rank == 0:
        ret.put({k: v.cpu() for k, v in eng.state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_equals_single_process():
    from gpuutil import engine_from, maxdiff
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
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
    ret.put(({k: v.cpu() for k, v in eng.state_dict().items()}, float(t.item()), exposure))
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
    got, tmax, exposure = ret.get()
    p.join(120)
    assert p.exitcode == 0
    assert tmax == 1.25
    assert len(exposure) == 2 and all(x >= 0.0 for x in exposure)
    cfg = oracle.OracleConfig(**CFG)
    eng = engine_from(cfg, oracle.init_params(cfg, seed=4))
    feats, caps, senti, eps = _inputs()
    for frozen in (True, False):
        _step(eng, feats, caps, senti, eps, frozen)
    for k, v in eng.state_dict().items():
        assert maxdiff(got[k], v) < 2e-6, k
