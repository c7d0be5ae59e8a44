#!/usr/bin/env python3
"""Headline benchmark: captions/sec of one Style-SeqCVAE (var_updown) train step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torch.distributed.run)

A "step" = one pass of the hot path over one minibatch per GPU: fused T-step forward + BPTT (HIP kernels behind
libssc_hip.so) + RCCL all-reduce of the flat gradient buffer + clip_grad_norm + SGD(momentum, wd)
(reference: var_updown/scripts/train.py:154-176).  Workload = BASELINE.json configs[1] ("C2": B=64/GPU, 36x2048
region features, 20-token captions, Z=128, V=10000; E/H/A from the reference Config defaults 1000/1200/768,
SENTIMENT_VAE=1).  Inputs are synthetic and resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode): before HIP loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)

C2 = dict(B=64, R=36, F=2048, L=20, Z=128, V=10000, E=1000, H=1200, A=768)
C5 = dict(B=128, R=100, F=2048, L=40, Z=128, V=30000, E=1000, H=1200, A=768)


def _dbg_env(name, default=None):
    """A/B switches of the tools (SSC_BENCH_*): honoured only in a process that opts in with SSC_DEBUG=1, like the library's own
    SSC_* switches - a stray variable must not change what the driver's run measures."""
    return os.environ.get(name, default) if os.environ.get("SSC_DEBUG", "") == "1" else default


def synth_batch(seed, B, R, F, L, V, Z, device):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, R, F, generator=g)
    lens = torch.randint(8, L + 1, (B,), generator=g)
    caps = torch.zeros(B, L, dtype=torch.long)
    ids = torch.randint(2, V, (B, L), generator=g)
    for b in range(B):
        caps[b, : lens[b]] = ids[b, : lens[b]]
    senti = torch.randint(-1, 2, (B, 1), generator=g).float()
    eps = torch.randn(L + 1, B, Z, generator=g)
    return tuple(t.to(device) for t in (feats, caps, senti, eps))


def fused_step_bytes(c):
    """algorithmic bytes of ONE fused attention+LSTM step (rows #7-#11, excl. vocab), fp32, S=1."""
    B, R, F, E, H, A, Z = c["B"], c["R"], c["F"], c["E"], c["H"], c["A"], c["Z"]
    H4, s = 4 * H, 1
    w_att = H4 * (E + F + 2 * H + H) + 2 * H4
    w_enc = H4 * (F + 2 * H + s + H) + 2 * H4
    w_dec = H4 * (F + 2 * H + s + Z + H) + 2 * H4
    weights = w_att + A * H + A + w_enc + 2 * (Z * H + Z) + w_dec
    acts = B * (R * F + R * A + F + E + 12 * H + 4 * Z + R)
    return 4 * (weights + acts)


def cpu_baseline(c, seconds_budget=40.0):
    """The CPU oracle (oracle/: pure-torch restatement pinned to the reference) timed on this box's host cores on the same
    workload: full train step (fwd + autograd bwd + clip + SGD) of the C2 minibatch.  Bounded sample: a short thread sweep
    (one timed step per count after one warm-up) picks the best torch thread count - the default, every core of the box, is
    oversubscribed for this step - then >= 3 timed steps at that count, true median."""
    import oracle

    cfg = oracle.OracleConfig(vocab_size=c["V"], image_feature_size=c["F"], embedding_size=c["E"], hidden_size=c["H"],
                              attention_projection_size=c["A"], z_space=c["Z"], max_caption_length=c["L"],
                              sentiment_vae=1, senti_prior_multip=0.5)
    params = {k: v.requires_grad_(True) for k, v in oracle.init_params(cfg, seed=2).items()}
    feats, caps, senti, eps = synth_batch(1234, c["B"], c["R"], c["F"], c["L"], c["V"], c["Z"], "cpu")
    opt = torch.optim.SGD(list(params.values()), lr=0.015, momentum=0.9, weight_decay=0.001)

    def step():
        t0 = time.time()
        opt.zero_grad()
        out = oracle.train_forward(params, cfg, feats, caps, senti, eps)
        oracle.train_objective(out, cfg).backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 12.5)
        opt.step()
        return time.time() - t0

    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    t_start = time.time()
    sweep = {}
    for n in sorted({min(x, ncpu) for x in (8, 16, 32, 64)} | {default_threads}):
        if time.time() - t_start > seconds_budget * 0.6 and sweep:
            break
        torch.set_num_threads(n)
        if not sweep:
            step()   # warm-up (allocator, first-touch)
        sweep[n] = step()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]]
    while len(times) < 3 or (len(times) < 10 and time.time() - t_start < seconds_budget):
        times.append(step())
    torch.set_num_threads(default_threads)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {"value": c["B"] / med, "unit": "captions/s", "cores": best, "kind": "port",
            "sample": f"{len(times)} timed train steps (fwd+bwd+clip+SGD) of the C2 minibatch (B={c['B']}) at {best} torch threads "
                      f"(best of the sweep {{threads: s/step}} = { {k: round(v, 2) for k, v in sweep.items()} }), median {med:.2f} s/step, "
                      f"torch {torch.__version__} CPU, os.cpu_count()={ncpu}"}


def attention_roofline(device):
    """HBM-roofline fraction of the fused attention step (logits + masked softmax + weighted sum; SURVEY §8(d) 'Attention
    kernel') at the C2 and C5 shapes: algorithmic bytes 4*(B*R*A + B*R*F + B*A + B*R + B*F) over the hipEvent-timed
    average of 50 back-to-back calls (two kernels per call)."""
    from ssc_runtime import lib as L
    lib = L.load()
    out = {}
    for name, (B, R, A, F) in {"C2": (64, 36, 768, 2048), "C5": (128, 100, 768, 2048)}.items():
        g = torch.Generator().manual_seed(1)
        q = torch.randn(B, A, generator=g).to(device)
        pv = torch.randn(B, R, A, generator=g).to(device)
        wa = torch.randn(A, generator=g).to(device)
        feats = torch.randn(B, R, F, generator=g).to(device)
        mask = torch.ones(B, R, device=device)
        logits = torch.empty(B, R, device=device)
        alpha = torch.empty(B, R, device=device)
        att = torch.empty(B, F, device=device)
        # rotate over several copies so that the 26-146 MB working set is not served from the 256 MB Infinity Cache
        n_copies = 12 if name == "C2" else 3
        pvs = [pv.clone() for _ in range(n_copies)]
        fts = [feats.clone() for _ in range(n_copies)]

        def call(i):
            lib.ssc_attn_fwd(L.ptr(q), A, L.ptr(pvs[i % n_copies]), L.ptr(wa), L.ptr(mask), L.ptr(fts[i % n_copies]), B, R, A, F, 1,
                             L.ptr(logits), L.ptr(alpha), L.ptr(att), F, L.stream_ptr())

        for i in range(5):
            call(i)
        torch.cuda.synchronize()
        # the 50 calls are captured into one hipGraph so that the measurement is device time, not ctypes launch overhead
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(50):
                call(i)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        nbytes = 4 * (B * R * A + B * R * F + B * A + B * R + B * F)
        out[name] = {"us": us, "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / us / 1e3,
                     "frac_of_8TBps": nbytes / us / 1e3 / HBM_PEAK_GBS}
    return out


def decode_roofline(dec, feats, senti, c):
    """16-bit-MFMA roofline of ONE beam-search call (100 images x 20 samples x beam 5 = 10000 rows, early stop off): a hipEvent pair
    around every GEMM launch (ssc_prof_enable); the large products (M >= 512 rows: attention-LSTM gates, decoder gates,
    vocabulary head of every step + the per-call tables) run `passes` matrix passes per fp32 product - 3 in the decode's default
    2xFP16 form (two fp16 pieces per operand, three partial products), 6 in the 3xBF16 form -, so
    achieved = passes * sum 2MNK / sum duration against the 2.5 PFLOP/s dense 16-bit (bf16 = fp16) peak."""
    from ssc_runtime import lib as L
    from ssc_runtime.inference import diverse_decode
    lib = L.load()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.ssc_prof_enable(1)
    e0.record()
    diverse_decode(dec, feats, senti, 20, 5, c["L"], 1, early_stop=False)
    e1.record()
    buf = torch.zeros(4096 * 8, dtype=torch.float32)
    n = lib.ssc_prof_collect(buf.data_ptr(), 4096)
    lib.ssc_prof_enable(0)
    call_ms = e0.elapsed_time(e1)
    rec = buf[: n * 8].view(n, 8).tolist()
    big = [r for r in rec if r[1] >= 512]
    if not big:
        return None
    ms = sum(r[5] for r in big)
    flops = sum(r[7] for r in big)
    passes = 3.0 if dec._cfg.gemm_mode == 3 else 6.0
    tf6 = passes * flops / (ms * 1e-3) / 1e12
    top = {}
    for r in big:
        e = top.setdefault((int(r[1]), int(r[2]), int(r[3])), [0, 0.0, 0.0])
        e[0] += 1
        e[1] += r[5]
        e[2] += r[7]
    shapes = [{"M": k[0], "N": k[1], "K": k[2], "launches": v[0], "avg_us": v[1] / v[0] * 1e3,
               "mfma_TFLOPs": passes * v[2] / (v[1] * 1e-3) / 1e12} for k, v in sorted(top.items(), key=lambda kv: -kv[1][1])[:4]]
    return {"bound": "mfma", "kernel": ("large products of one beam-search call, 2xFP16 form (gemm_x3w_kernel<NT,128x128,F16>: 3 fp16 MFMA passes "
            "per fp32 product)" if passes == 3.0 else "large 3xBF16 GEMMs of one beam-search call (gemm_x3w_kernel<128x128> / "
            "gemm_x3b_kernel<128x128>; 6 bf16 MFMA passes per fp32 product)"), "passes_per_fp32_product": passes,
            "achieved": tf6, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
            "frac": tf6 / MFMA_BF16_PEAK_TF, "traffic": None, "fp32_equivalent_TFLOPs": flops / (ms * 1e-3) / 1e12,
            "fp32_equivalent_vs_fp32_mfma_peak": flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
            "launches_per_call": len(big), "gemm_ms_per_call": ms, "call_ms_with_event_overhead": call_ms,
            "share_of_call_time": ms / call_ms, "top_shapes": shapes}


def measure_decode(model, c, rank, world, device, images, warmup):
    """BASELINE.json configs[3] (C4): beam 5 (per-node 2) x 20 latent samples per image, 36x2048 features, max 20
    steps, trivial one-state FSM, images sharded over ranks (no collective).  A "step" = one chunk of 100 images
    (10000 rows per beam-search call: how many images share a call is the harness's choice - the reference decodes one image at
    a time, inference.py:95; from 100 images on the rate is flat, 50 cost 4 %: 5000 rows leave the last 128-row tile row of
    every product 94 % empty and a thirteenth, a third full, round of workgroups).  Returns the result dict on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    from ssc_runtime.inference import count_tokens, diverse_decode

    was_training = model.training
    model.eval()
    dec = model._dec
    dec.weights_frozen = _dbg_env("SSC_BENCH_NO_REUSE") != "1"   # an inference run: the parameters do not change between the calls (what scripts/inference.py sets; the variable is the A/B switch)
    chunk = int(_dbg_env("SSC_BENCH_DECODE_CHUNK", "100"))   # images per beam-search call: 100 x 20 samples x 5 beams = 10000 rows (the variable is the A/B switch)
    chunk = max(1, min(chunk, images // world))
    per_rank = images // world
    n_chunks = max(1, per_rank // chunk)
    g = torch.Generator().manual_seed(4321 + rank)
    feats = [torch.randn(chunk, c["R"], c["F"], generator=g).to(device) for _ in range(min(n_chunks, 4))]
    senti = torch.ones(chunk, device=device)
    results = {True: [0.0, 0.0, 0.0], False: [0.0, 0.0, 0.0]}
    passes = []
    # warm-up: at least `warmup` calls AND at least 3 s of work - a GPU coming out of idle needs seconds to reach its sustained
    # state (BENCH_r03: the leg that ran first read 434 k tokens/s, the one behind it 498 k; alternating on one box they are 1 % apart)
    t_w = time.perf_counter()
    i = 0
    while i < warmup or time.perf_counter() - t_w < 3.0:
        diverse_decode(dec, feats[i % len(feats)], senti, 20, 5, c["L"], 1, early_stop=True)
        torch.cuda.synchronize()
        i += 1
    for i in range(n_chunks):   # one untimed pass of the timed loop itself (BENCH r4 rehearsal: the first timed pass still read 7 % low after 3 s of calls)
        count_tokens(diverse_decode(dec, feats[i % len(feats)], senti, 20, 5, c["L"], 1, early_stop=True)[0], 1)
    for early in (True, False, True, False):   # alternating passes of n_chunks calls each; a leg's figure is over both of its passes
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tokens = rows_steps = 0
        for i in range(n_chunks):
            pred, calls = diverse_decode(dec, feats[i % len(feats)], senti, 20, 5, c["L"], 1, early_stop=early)
            tokens += count_tokens(pred, 1)
            rows_steps += chunk * 20 * (1 + 5 * (calls - 1))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el, tokens, rows_steps], device=device, dtype=torch.float64)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            el, tokens, rows_steps = float(tmax[0]), float(t[1]), float(t[2])
        for k, v in enumerate((el, tokens, rows_steps)):
            results[early][k] += v
        passes.append({"early_stop": early, "tokens_per_s": tokens / el})
    droof = decode_roofline(dec, feats[0], senti, c) if rank == 0 else None
    # the same calls with every product in the 3xBF16 form (the decode's numerics until round 4): one pass, rank 0's own rate
    bf16x3 = None
    if rank == 0:
        mode0 = dec._cfg.gemm_mode
        dec._cfg.gemm_mode = 1
        dec._last_ctx = None
        try:
            diverse_decode(dec, feats[0], senti, 20, 5, c["L"], 1, early_stop=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tk = 0
            for i in range(n_chunks):
                pred, _ = diverse_decode(dec, feats[i % len(feats)], senti, 20, 5, c["L"], 1, early_stop=True)
                tk += count_tokens(pred, 1)
            torch.cuda.synchronize()
            bf16x3 = {"tokens_per_s_one_gpu": tk / (time.perf_counter() - t0)}
        finally:
            dec._cfg.gemm_mode = mode0
            dec._last_ctx = None
    dec.weights_frozen = False       # (the train leg that may follow changes the weights)
    dec._last_ctx = None
    if was_training:
        model.train()
    if rank != 0:
        return None
    el, tokens, rows_steps = results[True]
    el2, tokens2, rows_steps2 = results[False]
    n_timed = 2 * n_chunks   # (two passes per leg)
    return {"metric": "decode tokens/sec (beam 5 x 20 latent samples per image)", "value": tokens / el, "unit": "tokens/s",
            "n_gpus": world, "steps": n_timed, "warmup": warmup, "ms_per_step": el / n_timed * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_detail": "fp32 operands and results; the large per-step products (>= 512 rows) via a 2 x fp16 split of every fp32 operand "
                            "(power-of-two scaled per image context, 21-22 significant bits, three partial products on the fp16 matrix "
                            "cores, fp32 accumulate; ssc_model_cfg.gemm_mode 3), smaller products 3 x bf16 split (24 bits, six products), "
                            "pointwise exact fp32; reference fixtures hold at 1e-4.  bf16x3_mode = the same calls with gemm_mode 1",
            "bf16x3_mode": bf16x3,
            "config": {"workload": "C4 diverse decode: %d images (decoded twice: two alternating passes per leg), 36x2048 feats, beam 5 "
                                   "(per-node 2), N_Z=20, max 20 steps, trivial FSM, random-init weights" % (n_chunks * chunk * world),
                       "images_per_call": chunk, "rows_per_call": chunk * 100},
            "roofline": droof,
            "captions_per_s": n_timed * chunk * world * 20 / el, "row_steps_per_s": rows_steps / el,
            "early_stop_disabled": {"tokens_per_s": tokens2 / el2, "row_steps_per_s": rows_steps2 / el2,
                                    "captions_per_s": n_timed * chunk * world * 20 / el2},
            "passes": passes}


def measure_decode_cbs(model, c, device, n_calls=3, warmup=2, legs="all"):
    """SURVEY 8(f)-1 at C4's shapes: constrained beam search with k = 3 constraint classes per image (two word forms per word, every
    fifth class a two-word phrase -> sub-states), beam 5 (per-node 2), N_Z = 20, 36x2048 features, max 20 steps, machines from
    ssc_runtime.constraints.FiniteStateMachineBuilder (bit-identical to the reference builder, g9_fsm) - one machine per IMAGE,
    compiled on the device (ssc_fsm_compile), rows without a finite beam skipped.  Images per call are chosen for <= 20000 rows
    per step.  Beside the whole-call rate: the selection step (row kernel + merge) alone at the call's shape, compiled against
    dense, with hipEvents on the launching stream - `selection.achieved` = algorithmic bytes (the logits once: G*V*4) / time."""
    import ctypes as C
    import tempfile
    from ssc_runtime import lib as L
    from ssc_runtime.constraints import FiniteStateMachineBuilder
    from ssc_runtime.decode import CompiledFsm
    from ssc_runtime.inference import count_tokens, diverse_decode
    lib = L.load()
    was_training = model.training
    model.eval()
    dec = model._dec
    dec.weights_frozen = True
    V, n_z, beam, per_node, kmax = c["V"], 20, 5, 2, 3
    vocab = model._vocabulary
    g = torch.Generator().manual_seed(777)
    ncls = 300
    lines = []
    for j in range(ncls):   # class j: words w(100+4j) .. ; every fifth class is a two-word phrase
        a, b2, c2, d2 = (f"w{100 + 4 * j + o}" for o in range(4))
        lines.append(f"{a}\t{a},{b2}")
        lines.append(f"{c2}\t{c2},{d2}")
    with tempfile.TemporaryDirectory() as td:
        tsv = os.path.join(td, "wordforms.tsv")
        open(tsv, "w").write("\n".join(lines) + "\n")
        builder = FiniteStateMachineBuilder(vocab, tsv, None, max_given_constraints=kmax)

        def constraint(j):
            a, c2 = f"w{100 + 4 * j}", f"w{100 + 4 * j + 2}"
            return f"{a} {c2}" if j % 5 == 0 else a
        nimg_pool = 64
        picks = [torch.randperm(ncls, generator=g)[:kmax].tolist() for _ in range(nimg_pool)]
        built = [builder.build([constraint(j) for j in pk]) for pk in picks]
    S = max(b[1] for b in built)
    nimg = max(1, 20000 // (n_z * S * beam))
    built = built[:nimg] if nimg <= nimg_pool else built
    nimg = len(built)
    fsm = torch.zeros(nimg, S, S, V, dtype=torch.uint8)
    for i, (m, ns, _) in enumerate(built):
        fsm[i, :ns, :ns] = m[:ns, :ns]
    fsm = fsm.to(device)
    ncons = torch.full((nimg * n_z,), kmax, dtype=torch.long)
    feats = [torch.randn(nimg, c["R"], c["F"], generator=g).to(device) for _ in range(2)]
    senti = torch.ones(nimg, device=device)
    comp = CompiledFsm(fsm, fill=8)
    sparse_frac = float(comp.sparse_states().float().mean())

    def run(compiled, skip, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tokens = 0
        for i in range(n):
            pred, calls = diverse_decode(dec, feats[i % 2], senti, n_z, beam, c["L"], 1, fsm=fsm, num_constraints=ncons,
                                         min_constraints_to_satisfy=kmax, compiled=compiled, skip_dead=skip, early_stop=True)
            tokens += count_tokens(pred, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, tokens, pred

    run(comp, True, warmup)
    el, tokens, pred = run(comp, True, n_calls)
    if legs == "compiled":   # (what the rocprofv3 passes run: only the product path's launches in the tables)
        return {"value": tokens / el, "unit": "tokens/s", "ms_per_call": el / n_calls * 1e3, "rows_per_step": nimg * n_z * S * beam}
    # the same calls on the dense machine (one scan per row and target state, every row stepped): round 3's path
    def dense_call(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = 0
        for i in range(n):
            p2, _ = _dense_decode(dec, feats[i % 2], senti, n_z, beam, c["L"], fsm, ncons, kmax)
            tk += count_tokens(p2, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, tk
    dense_call(1)
    dense_n = max(1, n_calls // 2)
    dense_el, dense_tokens = dense_call(dense_n)
    # selection step alone at this shape
    B, G = nimg * n_z, nimg * n_z * S * beam
    logits = torch.randn(G, V, device=device)
    last_pred = torch.randint(2, V, (B, S * beam), device=device)
    last_lp = -torch.rand(B, S, beam, device=device) * 20
    mach = torch.arange(nimg, dtype=torch.int32, device=device).repeat_interleave(n_z)
    pred_o = torch.empty(B, S * beam, dtype=torch.int64, device=device)
    lp_o = torch.empty(B, S, beam, device=device)
    back = torch.empty(B, S * beam, dtype=torch.int64, device=device)
    sval = torch.empty(B * S * S * beam * per_node, device=device)
    sidx = torch.empty(B * S * S * beam * per_node, dtype=torch.int64, device=device)
    d = L.BeamDesc()
    d.scores, d.ld, d.raw_logits = L.ptr(logits), V, 1
    d.fsm, d.mach = L.ptr(fsm), L.ptr(mach)
    d.B, d.beam, d.per_node, d.end_index = B, beam, per_node, 1
    d.last_pred, d.last_lp, d.pred, d.lp_out, d.backptr = L.ptr(last_pred), L.ptr(last_lp), L.ptr(pred_o), L.ptr(lp_o), L.ptr(back)
    d.scratch_val, d.scratch_idx = L.ptr(sval), L.ptr(sidx)
    sel = {}
    for name, tables in (("compiled", comp), ("dense", None)):
        d.tables = L.ptr(tables.tables) if tables is not None else None
        d.dims = tables.dims if tables is not None else L.FsmDims(nimg, S, V, 0, 1)
        for _ in range(2):
            lib.ssc_beam_step_fsm(C.byref(d), L.stream_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            lib.ssc_beam_step_fsm(C.byref(d), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        alg = G * V * 4.0                                           # the logits, once
        moved = alg if tables is not None else S * G * V * 5.0       # dense: V floats + V mask bytes per (row, target state)
        sel[name] = {"us_per_step": us, "algorithmic_bytes": alg, "achieved_GBps": alg / us / 1e3, "frac_of_8TBps": alg / us / 1e3 / HBM_PEAK_GBS,
                     "bytes_scanned": moved}
        if name == "compiled":
            got = (pred_o.clone(), lp_o.clone(), back.clone())
        else:
            same = bool(torch.equal(got[0], pred_o) and torch.equal(got[1], lp_o) and torch.equal(got[2], back))
    dec.weights_frozen = False
    dec._last_ctx = None
    if was_training:
        model.train()
    return {"metric": "constrained decode tokens/sec (k=3 constraints, beam 5 x 20 latent samples per image)", "value": tokens / el,
            "unit": "tokens/s", "ms_per_call": el / n_calls * 1e3, "captions_per_s": n_calls * nimg * n_z / el,
            "config": {"workload": "C4 shapes with constrained beam search: %d images per call x 20 samples x %d states x beam 5 = %d rows "
                                   "per step, k = 3 constraint classes per image (one in five a two-word phrase), V = %d, max 20 steps, "
                                   "random-init weights" % (nimg, S, G, V),
                       "images_per_call": nimg, "states": S, "rows_per_step": G, "from_states_in_compiled_form": sparse_frac},
            "dense_machine_same_shape": {"tokens_per_s": dense_tokens / dense_el, "ms_per_call": dense_el / dense_n * 1e3},
            "selection": {"bound": "hbm", "kernel": "beam_row_fsm_kernel + beam_merge_kernel (one scan of the logits per row)",
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", **sel["compiled"],
                          "dense": sel["dense"], "dense_equals_compiled": same}}


def _dense_decode(dec, feats, senti, n_z, beam, L_, fsm, ncons, kmax):
    """Round 3's constrained decode, kept for the A/B: the dense machine, one masked scan per (row, target state), every row stepped."""
    from ssc_runtime.decode import cbs_search
    from ssc_runtime.decoding import select_best_beam_simple_batched
    dev = feats.device
    nimg = feats.size(0)
    B = nimg * n_z
    ctx = dec.prepare(feats)
    sent_b = senti.reshape(nimg, 1).expand(nimg, n_z).reshape(B)
    mach = torch.arange(nimg, dtype=torch.int32, device=dev).repeat_interleave(n_z)
    calls = {"k": 0}

    def step(tokens, state):
        G = tokens.numel()
        eps = torch.randn(G, dec.dims.Z, device=dev)
        calls["k"] += 1
        lp, st, _ = dec.step(ctx, tokens, state, sent_b.view(B, 1).expand(B, G // B).reshape(G), eps, raw_logits=True)
        return lp, {k: v for k, v in st.items() if k not in ("h_encoder", "c_encoder")}
    start = torch.full((B,), 1, dtype=torch.long, device=dev)
    beams, lps = cbs_search(start, None, step, fsm, 1, L_, beam, 2, early_stop=True, raw_logits=True,
                            ungathered_ok=lambda G, group: dec.ungathered_ok(ctx, G, group), mach=mach, compile_fsm=False)
    best, _ = select_best_beam_simple_batched(beams, lps, ncons, kmax)
    return best.view(nimg, n_z, -1), calls["k"]


def measure_c5_train(device, steps=5, warmup=2):
    """BASELINE.json configs[4] (C5, the stress shape) as a driver-timed train step on ONE GPU: B = 128, R = 100 regions, L = 40
    (T = 41), V = 30000, E / H / A = 1000 / 1200 / 768, Z = 128, SENTIMENT_VAE 1.  From 128 rows on the gate products are bound by
    the bf16 matrix pipe (3xBF16: six passes per fp32 product), not by HBM: `roofline` is that of the per-timestep gate products
    (M = 128 against 4H = 4800 columns, wave-specialised 128x128 kernels), from a hipEvent pair around every GEMM launch of one
    further forward + backward."""
    from ssc_runtime import lib as L
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner
    lib = L.load()
    c = dict(C5)
    torch.manual_seed(2)
    model = UpDownCaptioner(Vocabulary.synthetic(c["V"]), image_feature_size=c["F"], embedding_size=c["E"], hidden_size=c["H"],
                            attention_projection_size=c["A"], max_caption_length=c["L"], beam_size=5, z_space=c["Z"], prior_std=1.0,
                            simple_vae=False, latent_embedding="glove", sentiment_vae=1, senti_prior_multip=0.5, device=device).to(device)
    eng = model._engine()
    batches = [synth_batch(555 + i, c["B"], c["R"], c["F"], c["L"], c["V"], c["Z"], device) for i in range(2)]
    T = c["L"] + 1

    def step(i):
        feats, caps, senti, _ = batches[i % 2]
        eps = torch.randn(T, c["B"], c["Z"], device=device)
        eng.train_step(feats, caps, senti, eps, lr=0.015, kld_weight=750.0, momentum=0.9, weight_decay=0.001, max_norm=12.5,
                       decoder_frozen=False)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    B = c["B"]
    lib.ssc_prof_enable(1)
    feats, caps, senti, eps = batches[0]
    eng.forward(feats, caps, senti, eps)
    torch.cuda.synchronize()
    eng.backward(torch.full((B,), 1.0 / B, device=device), torch.full((B,), 1.0 / (B * 750.0), device=device))
    buf = torch.zeros(4096 * 8, dtype=torch.float32)
    n = lib.ssc_prof_collect(buf.data_ptr(), 4096)
    lib.ssc_prof_enable(0)
    rec = buf[: n * 8].view(n, 8).tolist()
    gate = [r for r in rec if int(r[1]) == B]      # the per-timestep minibatch products (forward x W^T and backward dG W)
    roof = None
    if gate:
        ms = sum(r[5] for r in gate)
        fl = sum(r[7] for r in gate)
        by = sum(r[6] for r in gate)
        tf6 = 6.0 * fl / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "per-timestep gate products at M = 128 (gemm_x3w_kernel<128x128>, grouped; 6 bf16 MFMA passes "
                "per fp32 product)", "achieved": tf6, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf6 / MFMA_BF16_PEAK_TF,
                "traffic": None, "launches": len(gate), "avg_launch_us": ms / len(gate) * 1e3,
                "algorithmic_GBps": by / (ms * 1e-3) / 1e9, "share_of_gemm_time": ms / sum(r[5] for r in rec)}
    del model, eng
    torch.cuda.empty_cache()
    return {"metric": "captions/sec (train step)", "value": B * steps / el, "unit": "captions/s", "steps": steps, "warmup": warmup,
            "ms_per_step": el / steps * 1e3, "dtype": "f32",
            "config": {"workload": "C5 train step on one GPU: fwd+bwd+clip+SGD, B=128, R=100, F=2048, L=40 (T=41), Z=128, V=30000, "
                                   "E=1000, H=1200, A=768, SENTIMENT_VAE=1"},
            "roofline": roof}


def bench_decode(args, model, eng, c, rank, world, device):
    import torch.distributed as dist
    if args.mode == "decode-cbs":
        print(json.dumps(measure_decode_cbs(model, c, device, legs=args.cbs_legs)), flush=True)
        return
    res = measure_decode(model, c, rank, world, device, args.images, args.warmup)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=C2["B"])
    ap.add_argument("--mode", default="train", choices=["train", "decode", "decode-cbs"],
                    help="train: headline captions/sec (default); decode: C4 diverse-decode tokens/sec")
    ap.add_argument("--images", type=int, default=1000, help="decode mode: synthetic images in total")
    ap.add_argument("--cbs-legs", default="all", choices=["all", "compiled"],
                    help="decode-cbs mode: all = compiled machines, the dense A/B and the selection step alone; compiled = the product path only")
    ap.add_argument("--dump-gemm", default="", help="write the per-shape GEMM timing table (roofline leg) to this file")
    ap.add_argument("--no-decode", action="store_true", help="skip the short decode leg of the default run")
    ap.add_argument("--dp-algo", default="rccl", choices=["auto", "rccl", "xgmi"],
                    help="gradient exchange at N > 1: rccl (default) = torch.distributed all-reduce; xgmi = direct reduce-scatter + "
                         "all-gather over hipIpc peer mappings (csrc/collective.hip); auto = verify the direct path against RCCL at start-up "
                         "(over the FULL gradient buffer), time both and keep the faster (falls back to rccl when peers cannot be mapped).  "
                         "The direct path has never run across two devices in any record of this repository (no multi-GPU node was "
                         "available to the build), so it is opt-in until one such run exists")
    ap.add_argument("--prewarm", type=int, default=150,
                    help="untimed extra train steps before the W warm-up steps when no decode leg ran first (a GPU coming out of "
                         "idle needs > 1 s to reach its sustained state); 0 for the profiler passes")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the timed train steps (no roofline / attention / decode / CPU legs): what the rocprofv3 passes run, "
                         "so that every profiled launch belongs to the C2 train step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    # SSC_BENCH_ONE_DEVICE=1 (rehearsal on a one-GPU box only): every rank uses cuda:0 and gloo carries the
    # collectives; the real multi-GPU run is one rank per GPU over RCCL ("nccl" backend on ROCm).
    rehearsal = _dbg_env("SSC_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rank == 0 and not rehearsal:
            # first look at what RCCL does with the 446 MB gradient all-reduce on this node (ring vs direct, protocol, channels;
            # a single ring moves it in ~5 ms, a direct reduce-scatter + all-gather over all 7 xGMI links in
            # ~0.7 ms): rank 0 logs RCCL's init and tuning decisions to stderr
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,GRAPH")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from ssc_runtime import lib as L
    from ssc_runtime.vocab import Vocabulary
    from var_updown.models import UpDownCaptioner

    c = dict(C2)
    c["B"] = args.batch
    torch.manual_seed(2)  # RANDOM_SEED of the shipped yaml; identical replicas on every rank (default torch inits)
    model = UpDownCaptioner(Vocabulary.synthetic(c["V"]), image_feature_size=c["F"], embedding_size=c["E"],
                            hidden_size=c["H"], attention_projection_size=c["A"], max_caption_length=c["L"], beam_size=5,
                            z_space=c["Z"], prior_std=1.0, simple_vae=False, latent_embedding="glove", sentiment_vae=1,
                            senti_prior_multip=0.5, device=device).to(device)
    eng = model._engine()  # flat parameter / gradient store + fused kernels; the module's parameters are views of it
    eng.dp_algo = args.dp_algo
    if args.mode in ("decode", "decode-cbs"):
        return bench_decode(args, model, eng, c, rank, world, device)
    # decode leg of the headline metric (BASELINE.json: "captions/sec (train step) + decode tokens/sec"): C4 itself
    # (1000 images per rank = 20 beam-search calls of 5000 rows), reported in the same JSON line.  It runs FIRST, on the
    # random-init weights C4 is defined on (BOUNDARY is then rarely emitted, so the searches run their full
    # length; after a few SGD steps on synthetic captions the model emits BOUNDARY at once).
    dres = None
    if not args.timed_only and not args.no_decode:
        dres = measure_decode(model, c, rank, world, device, 1000 * world, 5)   # C4: 1000 images per GPU (>= 5 calls and >= 2 s of warm-up: the leg runs first, on a GPU coming out of idle)
    cbs_res = None
    if rank == 0 and world == 1 and not args.timed_only and not args.no_decode:
        cbs_res = measure_decode_cbs(model, c, device)
    batches = [synth_batch(1234 + rank + 100 * i, c["B"], c["R"], c["F"], c["L"], c["V"], c["Z"], device) for i in range(4)]
    total_iters = 70000

    T_steps, Zdim = c["L"] + 1, c["Z"]

    def step(i):
        feats, caps, senti, _ = batches[i % len(batches)]
        # the reparameterisation noise is DRAWN INSIDE the timed step, on the device (the reference draws it inside the step too,
        # per time step on the host: updown_cell.py:206); the pre-generated eps of `batches` serves the parity / roofline legs
        eps = torch.randn(T_steps, c["B"], Zdim, device=device)
        lr = 0.015 * (1 - i / total_iters)
        eng.train_step(feats, caps, senti, eps, lr=lr, kld_weight=750.0, momentum=0.9, weight_decay=0.001,
                       max_norm=12.5, decoder_frozen=False)

    if args.timed_only or args.no_decode:   # no decode leg ran before: bring the GPU out of idle first (same count on every rank: the steps all-reduce)
        for i in range(args.prewarm):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
        eng.dp_profile = True     # exposed all-reduce time per step: compute stream idle between its last backward kernel and
        eng.dp_exposure_ms()      # the reduced gradients (the four gradient ranges are reduced behind their backward phases)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dp_exposure = None
    if world > 1:
        eng.dp_profile = False
        ex = eng.dp_exposure_ms()
        t = torch.tensor([sum(ex) / max(1, len(ex)), max(ex) if ex else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # the replicas must still be bit-identical (same reduced gradients, same update on every rank), and no bounded wait of the
        # direct exchange may have given up: a number measured on diverged replicas would be worthless
        chk = eng.params.flat.sum(dtype=torch.float64).reshape(1)
        lohi = torch.cat([chk, -chk])
        dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
        identical = bool((lohi[0] == -lohi[1]).item())
        timeouts = False
        if eng._xgmi is not None and (eng.dp_choice or {}).get("algo") == "xgmi":
            try:
                eng._xgmi.check()
            except Exception:   # noqa: BLE001
                timeouts = True
            flag = torch.tensor([1.0 if timeouts else 0.0], device=device, dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            timeouts = bool(flag.item() > 0)
        if rank == 0 and (not identical or timeouts):
            print(f"[bench] DATA-PARALLEL RUN INVALID: replicas identical = {identical}, direct-exchange timeouts = {timeouts}",
                  file=sys.stderr, flush=True)
        dp_exposure = {"replicas_identical": identical, "direct_exchange_timeouts": timeouts,
                       "allreduce_exposed_ms_per_step_mean": float(t[0]), "allreduce_exposed_ms_per_step_max": float(t[1]),
                       "bytes_per_step": int(eng.grads.flat.numel() * 4), "exchange": eng.dp_choice or {"algo": "rccl", "why": "requested"},
                       "scheme": "5 ranges (output head behind the BPTT loop, then embedding, decoder LSTM, encoder LSTM + fc heads, attention "
                       "LSTM + attention), each exchanged behind its backward phase (engine.backward_overlapped)"}
        if rank == 0:
            print("data-parallel exchange:", json.dumps(dp_exposure), file=sys.stderr, flush=True)
    loss_probe = None if args.timed_only else eng.forward(*batches[0])[0].mean().item()

    result = None
    if rank == 0 and args.timed_only:
        print(json.dumps({"metric": "captions/sec (train step)", "value": world * c["B"] * args.steps / elapsed, "unit": "captions/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "timed_only": True}), flush=True)
    elif rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * c["B"] * args.steps / elapsed
        # ---- roofline leg: in-situ hipEvent timing of every GEMM launch over 2 further steps -----------------------
        lib = L.load()
        nprof = 2
        B = c["B"]
        # (a) the T-step time loops alone (hipEvent pair around the loop inside ssc_train_fwd / ssc_train_bwd; no other
        #     instrumentation active): the fused attention+LSTM step of SURVEY 8(d) is loop time / T
        import ctypes as _C
        lib.ssc_prof_loop_enable(1)
        fwd_loop_ms, bwd_loop_ms = [], []
        for i in range(3):
            feats, caps, senti, eps = batches[i % len(batches)]
            eng.forward(feats, caps, senti, eps)
            eng.backward(torch.full((B,), 1.0 / B, device=device), torch.full((B,), 1.0 / (B * 750.0), device=device))
            f_ms, b_ms = _C.c_float(), _C.c_float()
            lib.ssc_prof_loop_ms(_C.byref(f_ms), _C.byref(b_ms))
            fwd_loop_ms.append(f_ms.value)
            bwd_loop_ms.append(b_ms.value)
        lib.ssc_prof_loop_enable(0)
        # (b) a hipEvent pair around every GEMM launch of two further steps
        lib.ssc_prof_enable(1)
        for i in range(nprof):
            feats, caps, senti, eps = batches[i % len(batches)]
            eng.forward(feats, caps, senti, eps)
            torch.cuda.synchronize()
            eng.backward(torch.full((B,), 1.0 / B, device=device), torch.full((B,), 1.0 / (B * 750.0), device=device))
        buf = torch.zeros(4096 * 8, dtype=torch.float32)
        n = lib.ssc_prof_collect(buf.data_ptr(), 4096)
        lib.ssc_prof_enable(0)
        rec = buf[: n * 8].view(n, 8)   # kind, M, N, K, splits, ms, algorithmic bytes, flops (exact also for grouped launches)
        names = {0: "gemm_kernel<NT> (forward: x W^T)", 1: "gemm_kernel<NN> (backward: dG W)", 3: "gemm_kernel<TN> (dW = dG^T X)"}
        agg = {}
        for kind, M, N, K, splits, msr, rbytes, rflops in rec.tolist():
            a = agg.setdefault(int(kind), dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
            a["ms"] += msr
            a["flops"] += rflops
            a["bytes"] += rbytes
            a["n"] += 1
        if args.dump_gemm:
            shapes = {}
            for kind, M, N, K, splits, msr, rbytes, rflops in rec.tolist():
                e = shapes.setdefault((int(kind), int(M), int(N), int(K), int(splits)), [0, 0.0, rbytes, rflops])
                e[0] += 1
                e[1] += msr
            with open(args.dump_gemm, "w") as f:
                f.write("kind,M,N,K,splits,calls_per_step,avg_us,total_ms_per_step,TFLOPs,algGBps\n")
                for (kind, M, N, K, sp), (cnt, tot, rbytes, rflops) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    us = tot / cnt * 1e3
                    f.write(f"{names[kind][12:14]},{M},{N},{K},{sp},{cnt / nprof:.1f},{us:.1f},{tot / nprof:.3f},"
                            f"{rflops / us / 1e6:.1f},{rbytes / us / 1e3:.0f}\n")
        # Two regimes (DESIGN.md "GEMM"): the per-timestep MINIBATCH products (M = B rows against a wide weight matrix,
        # forward x W^T and backward dG W) stream every weight once per launch -> HBM-bound; the LARGE products over the
        # (t, b) rows (hoisted gate terms, vocabulary head, weight gradients) are bound by the bf16 matrix pipe, on which
        # the 3xBF16 kernels spend six MFMA passes per fp32 product.
        fam = {"minibatch": dict(ms=0.0, bytes=0.0, flops=0.0, n=0), "large": dict(ms=0.0, bytes=0.0, flops=0.0, n=0)}
        for kind, M, N, K, splits, msr, rbytes, rflops in rec.tolist():
            f = fam["minibatch" if M <= c["B"] else "large"]
            f["ms"] += msr
            f["bytes"] += rbytes          # algorithmic: every operand and the result once (summed over a group's members)
            f["flops"] += rflops
            f["n"] += 1
        pm = {}
        try:  # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))["kernels"]
        except Exception:
            pm = {}

        def traffic_of(prefixes):
            sel = [v for k, v in pm.items() if any(k.startswith(f) for f in prefixes)]
            if not sel:
                return None
            return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / max(1, sum(v["launches"] for v in sel))

        mb, lg = fam["minibatch"], fam["large"]
        total_gemm_ms = mb["ms"] + lg["ms"]
        gbs = mb["bytes"] / (mb["ms"] * 1e-3) / 1e9
        roof_mb = {"bound": "hbm", "kernel": "minibatch GEMMs (M = B: gemm_x3w_kernel<64x256>, gemm_x3_kernel<64x64>, gemm_kernel<64x64>)",
                   "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                   "traffic": traffic_of(("gemm_x3w_kernel<NT,64x256", "gemm_x3w_kernel<NN,64x256", "gemm_x3_kernel", "gemm_kernel<NN,64x", "gemm_kernel<NT,64x")),
                   "algorithmic_bytes_per_launch": mb["bytes"] / max(1, mb["n"]), "launches_per_step": mb["n"] / nprof,
                   "avg_launch_us": mb["ms"] / max(1, mb["n"]) * 1e3, "share_of_gemm_time": mb["ms"] / total_gemm_ms}
        tf6 = 6.0 * lg["flops"] / (lg["ms"] * 1e-3) / 1e12
        roof_lg = {"bound": "mfma", "kernel": "large 3xBF16 GEMMs (128x128 tiles: wave-specialised gemm_x3w_kernel from 768 workgroups on "
                   "and for the grouped weight gradients, 4-wave gemm_x3b_kernel below; 6 bf16 MFMA passes per fp32 product; "
                   "TRUE flops: products with device-side row compaction are priced on the rows / k-rows they processed, the counts "
                   "read back behind each launch)",
                   "achieved": tf6, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf6 / MFMA_BF16_PEAK_TF,
                   "traffic": traffic_of(("gemm_x3b_kernel", "gemm_x3w_kernel<NT,128", "gemm_x3w_kernel<NN,128", "gemm_x3w_kernel<TN,128")),
                   "fp32_equivalent_TFLOPs": lg["flops"] / (lg["ms"] * 1e-3) / 1e12, "launches_per_step": lg["n"] / nprof,
                   "avg_launch_us": lg["ms"] / max(1, lg["n"]) * 1e3, "share_of_gemm_time": lg["ms"] / total_gemm_ms}
        # the single dominant kernel: the wave-specialised 64x256 minibatch kernel in its two layouts (NT: forward x W^T,
        # NN: backward dG W; N >= 1024) - algorithmic bytes per launch / average launch duration, both from the events
        dom = {}
        for kind, M, N, K, splits, msr, rbytes, rflops in rec.tolist():
            if M <= c["B"] and N >= 1024 and int(kind) in (0, 1):
                e = dom.setdefault(int(kind), dict(ms=0.0, bytes=0.0, n=0))
                e["ms"] += msr
                e["bytes"] += rbytes
                e["n"] += 1
        roof_dom = None
        if dom:
            kd = max(dom, key=lambda k: dom[k]["ms"])
            e = dom[kd]
            nm = "gemm_x3w_kernel<%s,64x256>" % ("NT" if kd == 0 else "NN")
            g2 = e["bytes"] / (e["ms"] * 1e-3) / 1e9
            roof_dom = {"bound": "hbm", "kernel": nm + (" (forward x W^T gate products)" if kd == 0 else " (backward dG W products)"),
                        "achieved": g2, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g2 / HBM_PEAK_GBS,
                        "traffic": traffic_of((nm,)), "algorithmic_bytes_per_launch": e["bytes"] / e["n"],
                        "launches_per_step": e["n"] / nprof, "avg_launch_us": e["ms"] / e["n"] * 1e3,
                        "share_of_gemm_time": e["ms"] / total_gemm_ms}
        roofline, roofline_other = (roof_mb, roof_lg) if mb["ms"] >= lg["ms"] else (roof_lg, roof_mb)
        if roof_dom is not None and roofline is roof_mb:
            roof_all_mb, roofline = roof_mb, roof_dom      # `roofline` = the dominant kernel; the family total stays beside it
        else:
            roof_all_mb = None
        # the recurrent (per-timestep) gate GEMMs of the fused attention+LSTM step: NT launches with M == B, N == 4H
        step_recs = [r for r in rec.tolist() if int(r[0]) == 0 and int(r[1]) == c["B"] and int(r[2]) == 4 * c["H"]]
        T = c["L"] + 1
        fused = fused_step_bytes(c)
        fwd_step_us = sorted(fwd_loop_ms)[len(fwd_loop_ms) // 2] / T * 1e3   # the time loop alone, median of 3 calls
        bwd_step_us = sorted(bwd_loop_ms)[len(bwd_loop_ms) // 2] / T * 1e3
        roofline_step = {"bound": "hbm", "scope": "fused attention+LSTM step (SURVEY 8(d): rows #7-#11, excl. the vocabulary head "
                         "and the hoisted per-sequence terms), forward: time of the T-step loop of ssc_train_fwd / T (hipEvent pair "
                         "around the loop, no other instrumentation)", "achieved": fused / (fwd_step_us * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fused / (fwd_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_step": fused, "us_per_step": fwd_step_us,
                         "gate_gemm_us_per_step_with_event_overhead": sum(r[5] for r in step_recs) / nprof / T * 1e3,
                         "bptt_us_per_step": bwd_step_us}
        result = {"metric": "captions/sec (train step)", "value": value, "unit": "captions/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                  "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": "C2 train step: fwd+bwd+allreduce+clip+SGD, B=%d/GPU, R=36, F=2048, L=20 (T=21), "
                                         "Z=128, V=10000, E=1000, H=1200, A=768, SENTIMENT_VAE=1" % c["B"],
                             "global_batch": world * c["B"], "parallelism": f"dp{world}", "loss_probe": loss_probe},
                  "roofline": roofline,
                  "roofline_large_gemm" if roofline_other is roof_lg else "roofline_minibatch_gemm": roofline_other,
                  "roofline_all_minibatch_gemms": roof_all_mb,
                  "roofline_step": roofline_step, "data_parallel": dp_exposure,
                  "gemm_time_ms_per_step": {names[k]: agg[k]["ms"] / nprof for k in agg}}
        result["attention_roofline"] = attention_roofline(device)
    if rank == 0 and dres is not None:
        result["decode_tokens_per_s"] = dres["value"]
        result["decode"] = {k: dres[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline", "captions_per_s",
                                                 "row_steps_per_s", "early_stop_disabled", "passes", "dtype_detail", "bf16x3_mode")}
    if rank == 0 and result is not None and not args.timed_only:
        result["dtype_detail"] = ("fp32 operands and results; products of 16-byte aligned operands via a 3 x bf16 split of every fp32 "
                                  "operand on the bf16 matrix cores (six partial products of order <= 2, fp32 accumulate: error ~ one "
                                  "fp32 rounding per product), everything else exact fp32; exact_fp32_mode = the same step with every "
                                  "product on v_mfma_f32_32x32x2_f32 (ssc_model_cfg.gemm_mode 2)")
    if rank == 0 and world == 1 and result is not None and not args.timed_only:
        # the same train step with every product on the exact-fp32 matrix instruction (per-engine numerics mode)
        eng._cfg.gemm_mode = 2
        try:
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_ex = max(3, min(args.steps, 10))
            for i in range(n_ex):
                step(i)
            torch.cuda.synchronize()
            el_ex = time.perf_counter() - t0
            result["exact_fp32_mode"] = {"value": c["B"] * n_ex / el_ex, "unit": "captions/s", "ms_per_step": el_ex / n_ex * 1e3,
                                         "steps": n_ex, "vs_default_mode": (c["B"] * n_ex / el_ex) / result["value"]}
        finally:
            eng._cfg.gemm_mode = 0
        result["c5_train"] = measure_c5_train(device)
    if rank == 0 and cbs_res is not None:
        result["decode_cbs"] = cbs_res
    if rank == 0 and not args.timed_only:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(c)
        print(json.dumps(result), flush=True)
    if world > 1:
        if eng._xgmi is not None:
            eng._xgmi.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
