"""CPU (gloo, world_size 2): the data-parallel host logic - row sharding, the flat-buffer sum all-reduce and the
1/world scale - reproduces the full-batch gradients of the oracle; plus layout checks of the flat parameter store."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from ssc_runtime import dp
from ssc_runtime.engine import FlatStore, ModelDims


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CFG = dict(vocab_size=60, image_feature_size=24, embedding_size=12, hidden_size=16, attention_projection_size=8, z_space=6,
           max_caption_length=5, sentiment_vae=1, senti_prior_multip=0.5)


def _inputs():
    g = torch.Generator().manual_seed(7)
    B, R, L = 4, 3, 5
    feats = torch.randn(B, R, 24, generator=g)
    caps = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = 2 + b % 3
        caps[b, :n] = torch.randint(2, 60, (n,), generator=g)
    senti = torch.tensor([[1.0], [-1.0], [0.0], [1.0]])
    eps = torch.randn(L + 1, B, 6, generator=g)
    return feats, caps, senti, eps


def _grads(params, cfg, feats, caps, senti, eps):
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = oracle.train_forward(p, cfg, feats, caps, senti, eps)
    oracle.train_objective(out, cfg).backward()
    return {k: v.grad for k, v in p.items()}


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = oracle.OracleConfig(**CFG)
    params = oracle.init_params(cfg, seed=1)
    feats, caps, senti, eps = _inputs()
    lo, hi = dp.shard_rows(feats.size(0), rank, world)
    g = _grads(params, cfg, feats[lo:hi], caps[lo:hi], senti[lo:hi], eps[:, lo:hi].contiguous())
    dims = ModelDims(V=60, E=12, H=16, A=8, F=24, Z=6, S=1, kld_mode=1, pm_scale=0.5)
    store = FlatStore(dims.param_shapes(), "cpu")
    for k, v in g.items():
        store.views[k].copy_(v)
    w = dp.allreduce_flat(store.flat, n_buckets=3)
    store.flat.mul_(dp.gscale(w))
    if rank == 0:
        # numpy arrays travel through the queue by value; torch tensors would be handed over as shared-memory file descriptors,
        # which needs this process alive until the parent has received them (a race once the worker exits)
        ret.put({k: v.clone().numpy() for k, v in store.views.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradients_equal_full_batch():
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = {k: torch.from_numpy(v) for k, v in ret.get().items()}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = oracle.OracleConfig(**CFG)
    want = _grads(oracle.init_params(cfg, seed=1), cfg, *_inputs())
    for k, v in want.items():
        assert torch.allclose(got[k], v, atol=1e-6, rtol=1e-5), k


def test_shard_rows_and_buckets():
    assert dp.shard_rows(128, 3, 8) == (48, 64)
    with pytest.raises(ValueError):
        dp.shard_rows(10, 0, 4)
    b = dp.bucket_bounds(1000, 3)
    assert b[0][0] == 0 and b[-1][1] == 1000 and all(lo % 64 == 0 for lo, _ in b)
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))


def test_flat_store_layout_alignment_and_ranges():
    dims = ModelDims(V=100, E=37, H=50, A=27, F=70, Z=13, S=1, kld_mode=1)
    st = FlatStore(dims.param_shapes(), "cpu")
    for name, v in st.views.items():
        assert v.data_ptr() % 16 == 0, name
        if v.dim() == 2 and v.size(0) > 1:
            assert v.stride(0) % 4 == 0 and v.stride(1) == 1, name
    # fc_mean.weight / fc_log_var.weight are adjacent with the same ld: one (2Z, H) GEMM operand
    a, b = st.views["_updown_cell.fc_mean.weight"], st.views["_updown_cell.fc_log_var.weight"]
    assert b.data_ptr() == a.data_ptr() + 13 * a.stride(0) * 4
    # decoder LSTM is the trailing contiguous range (freeze schedule skips it as one slice)
    dec = [n for n in st.views if "_language_lstm_cell_decoder" in n]
    lo, hi = st.range_of(dec)
    assert hi == st.numel and lo > 0
