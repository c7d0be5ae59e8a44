"""Eval-mode decode step + constrained beam search over the C ABI (ssc_decode_*, ssc_beam_*).

Reference: UpDownCaptioner._decode_step eval branch (var_updown/var_updown/models/updown_captioner.py:371-455),
ConstrainedBeamSearch.search (updown-baseline/updown/modules/cbs.py:59-277).
"""
import ctypes as C
from typing import Callable, Dict, Optional, Tuple

import torch

from . import lib as _lib
from .engine import ModelDims

STATE_KEYS = ("h1", "c1", "h_encoder", "c_encoder", "h_decoder", "c_decoder")


class ImageContext:
    """Per-image terms computed once per image set (mask, avg, pv, hoisted gate term): ssc_decode_prepare."""

    def __init__(self, feats: torch.Tensor, buf: torch.Tensor, obj: Optional[torch.Tensor] = None):
        self.feats = feats
        self.buf = buf
        self.obj = obj   # SENTIMENT_VAE = 2: per-region attribute means (nimg, R, Z) of these images (updown_cell.py:160-163)
        self.nimg, self.R, _ = feats.shape
        self.att_table_ready = False   # the per-image attended-feature table is formed by the first step that uses it


class DecodeEngine:
    DEFAULT_GEMM_MODE = 3

    def __init__(self, dims: ModelDims, params_struct_fn: Callable[[], "_lib.Params"], device):
        self.lib = _lib.load()
        self.dims = dims
        self.device = torch.device(device)
        self._cfg = dims.cfg()
        # numerics of the decode's products (ssc_model_cfg.gemm_mode).  Default 3: 3xBF16 everywhere except the large (>= 768
        # workgroups of 128x128) NT products - the per-step gate / vocabulary products of a 10000-row call, 80 % of its time -, which
        # take the 2xFP16 form: two fp16 pieces per fp32 operand (scaled by powers of two measured per image context), three
        # partial products instead of six on the matrix cores, 21-22 significant bits per operand instead of 24 (within 2e-6 of
        # sum|a||b| against float64, tests/test_gemm_gpu.py; the reference fixtures of the eval path hold at 1e-4 / identical
        # captions).  1 = 3xBF16 only, 2 = exact-fp32 MFMA; an engine-wide dims.gemm_mode other than 0 wins.
        if self._cfg.gemm_mode == 0:
            self._cfg.gemm_mode = self.DEFAULT_GEMM_MODE
        self._params = params_struct_fn
        self._ws = None
        self._ws_key = None
        # weights_frozen: the caller's promise that the parameters do not change between prepare() calls (an inference run over a
        # loaded checkpoint: scripts/inference.py, bench.py's decode leg).  What ssc_decode_prepare derives from the weights alone
        # (the per-token gate table) is then taken over from the previous image context instead of being formed per call.
        # Default off: the drop-in module's eval calls may be interleaved with training steps.
        self.weights_frozen = False
        self._last_ctx = None
        self._sws = None   # workspace of search()

    def prepare(self, feats: torch.Tensor, obj_means: Optional[torch.Tensor] = None) -> ImageContext:
        """obj_means (nimg, R, Z): the per-region attribute means of SENTIMENT_VAE = 2 (kld_mode 2), else None."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 3 and feats.size(2) == self.dims.F
        feats = feats.contiguous()
        nimg, R, _ = feats.shape
        if self.dims.kld_mode == 2:
            if obj_means is None or tuple(obj_means.shape) != (nimg, R, self.dims.Z):
                raise ValueError(f"SENTIMENT_VAE = 2 needs the per-region attribute means obj_atts ({nimg}, {R}, {self.dims.Z}), got "
                                 f"{None if obj_means is None else tuple(obj_means.shape)}")
            obj_means = obj_means.to(self.device, torch.float32).contiguous()
        else:
            obj_means = None
        nbytes = self.lib.ssc_decode_image_bytes(C.byref(self._cfg), nimg, R)
        buf = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
        p = self._params()
        prev = self._last_ctx if self.weights_frozen else None
        self.lib.ssc_decode_prepare_from(C.byref(self._cfg), C.byref(p), _lib.ptr(feats), nimg, R, _lib.ptr(buf), buf.numel() * 4,
                                         _lib.ptr(prev.buf) if prev is not None else None, prev.nimg if prev is not None else 0,
                                         prev.R if prev is not None else 0, _lib.stream_ptr())
        ctx = ImageContext(feats, buf, obj_means)
        # (kept alive by this reference: the copy above is stream-ordered before anything that could overwrite the old buffer)
        self._last_ctx = ctx if self.weights_frozen else None
        return ctx

    # The attended-feature term of the decoder gates from a per-image table (ssc_decode_step_desc.att_table): worth its one-off
    # product per image context once a step has a few hundred rows that share their image sixteen or more at a time (C4: 5000
    # rows, 100 per image: -13 % per call); below that the K = F segment is cheaper (one image x 100 rows: 3.8 vs 4.7 ms).
    ATT_TABLE_MIN_ROWS = 512
    ATT_TABLE_MIN_ROWS_PER_IMAGE = 16

    def _att_table_mode(self, ctx: "ImageContext", G: int, rpi: int) -> int:
        if ctx.R > 128 or G < self.ATT_TABLE_MIN_ROWS or rpi < self.ATT_TABLE_MIN_ROWS_PER_IMAGE:
            return 0
        on = C.c_int(1)   # ssc_debug_set("dec_att_table", 0): A/B switch of include/ssc_debug.h
        if self.lib._raw_ssc_debug_get(b"dec_att_table", C.byref(on)) != 0 or not on.value:
            return 0
        if not ctx.att_table_ready:
            ctx.att_table_ready = True
            return 2
        return 1

    def zero_states(self, G: int) -> Dict[str, torch.Tensor]:
        return {k: torch.zeros(G, self.dims.H, dtype=torch.float32, device=self.device) for k in STATE_KEYS}

    def step(self, ctx: ImageContext, tokens: torch.Tensor, states: Optional[Dict[str, torch.Tensor]],
             sentiment: Optional[torch.Tensor], eps: torch.Tensor, want_log_probs: bool = True,
             emb_table: Optional[torch.Tensor] = None, raw_logits: bool = False, prior_mean_out: Optional[torch.Tensor] = None,
             prior_mean: Optional[torch.Tensor] = None, prior_var: Optional[torch.Tensor] = None
             ) -> Tuple[Optional[torch.Tensor], Dict[str, torch.Tensor], torch.Tensor]:
        """One eval decode step for G rows (row g -> image g // (G / nimg)).  Returns (log_probs (G,V) or None,
        new states, alpha (G,R)).  h_encoder / c_encoder are carried through untouched (updown_cell.py:176-203).
        raw_logits: return the un-normalised vocabulary logits instead (for cbs_search(raw_logits=True), which takes the
        log-sum-exp inside its selection kernel: one pass over the (G,V) matrix less per step).
        prior_mean_out (G, Z), SENTIMENT_VAE = 2 only: receives the step's pooled prior mean (what the cell returns, updown_cell.py:231).
        prior_mean / prior_var (G, Z): the caller's own prior instead of the one the configuration implies."""
        d = self.dims
        G = tokens.numel()
        assert G % ctx.nimg == 0, (G, ctx.nimg)
        rpi = G // ctx.nimg
        if states is None:
            states = self.zero_states(G)
        # "_parent" (B, S*beam) int64: set by cbs_search after a beam re-ordering - row g descends from beam parent[g] of its group;
        # beams with the same parent hold identical states (ssc_decode_step_desc.parent: their shared products are formed once)
        parent = states.get("_parent")
        # "_ungathered": the states are the previous call's outputs in ITS row order (cbs_search left out the re-ordering by
        # back-pointer because ungathered_ok() said this call reads them through the parent lists)
        ungathered = bool(states.get("_ungathered", False))
        # "_skip": (running log-probs (B,S,beam), end index) from cbs_search(skip_dead=True): rows without a finite beam and rows
        # whose beam has ended need no step (ssc_decode_step_desc.row_lp)
        skip = states.get("_skip")
        st = {k: v.contiguous() for k, v in states.items() if not k.startswith("_")}
        tokens = tokens.to(torch.int64).contiguous()
        eps = eps.to(self.device, torch.float32).contiguous()
        assert tuple(eps.shape) == (G, d.Z), eps.shape
        sent = sentiment.reshape(G).to(torch.float32).contiguous() if sentiment is not None else None
        key = (G, ctx.R)
        if self._ws_key != key:
            nbytes = self.lib.ssc_decode_step_workspace_bytes(C.byref(self._cfg), G, ctx.R)
            self._ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
            self._ws_key = key
        new = {k: torch.empty_like(st[k]) for k in ("h1", "c1", "h_decoder", "c_decoder")}
        alpha = torch.empty(G, ctx.R, dtype=torch.float32, device=self.device)
        lp = torch.empty(G, d.V, dtype=torch.float32, device=self.device) if want_log_probs else None
        pm_in = prior_mean.to(self.device, torch.float32).contiguous() if prior_mean is not None else None
        pv_in = prior_var.to(self.device, torch.float32).contiguous() if prior_var is not None else None
        assert pm_in is None or tuple(pm_in.shape) == (G, d.Z)
        assert pv_in is None or tuple(pv_in.shape) == (G, d.Z)
        att_table = self._att_table_mode(ctx, G, rpi)
        has_parent = parent is not None and parent.numel() == G
        group = parent.shape[-1] if has_parent else 0
        if ungathered and not (has_parent and emb_table is None and
                               self.lib._raw_ssc_decode_ungathered_ok(C.byref(self._cfg), ctx.nimg, G, group, att_table)):
            raise ValueError("un-gathered states handed to a decode step that cannot read them through parent lists")
        desc = _lib.DecodeStepDesc(G, ctx.R, rpi, ctx.feats.data_ptr(), ctx.buf.data_ptr(), tokens.data_ptr(),
                                   sent.data_ptr() if sent is not None else None, eps.data_ptr(),
                                   st["h1"].data_ptr(), st["c1"].data_ptr(), st["h_decoder"].data_ptr(),
                                   st["c_decoder"].data_ptr(), new["h1"].data_ptr(), new["c1"].data_ptr(),
                                   new["h_decoder"].data_ptr(), new["c_decoder"].data_ptr(), alpha.data_ptr(),
                                   lp.data_ptr() if lp is not None else None, 1 if raw_logits else 0,
                                   1 if emb_table is not None else 0,
                                   parent.data_ptr() if has_parent else None, group, att_table, 1 if ungathered else 0,
                                   skip[0].data_ptr() if skip is not None else None, int(skip[1]) if skip is not None else 0,
                                   ctx.obj.data_ptr() if ctx.obj is not None else None,
                                   prior_mean_out.data_ptr() if (prior_mean_out is not None and ctx.obj is not None) else None,
                                   _lib.ptr(pm_in), _lib.ptr(pv_in))
        p = self._params()
        if emb_table is not None:  # rows of `emb_table` are the token embeddings themselves (UpDownCell.forward API)
            p.emb = emb_table.data_ptr()
            p.ld_emb = emb_table.stride(0)
        self.lib.ssc_decode_step(C.byref(self._cfg), C.byref(p), C.byref(desc), _lib.ptr(self._ws), self._ws.numel() * 4,
                                 _lib.stream_ptr())
        out_states = dict(st)
        out_states.update(new)
        return lp, out_states, alpha

    def ungathered_ok(self, ctx: ImageContext, G: int, group: int) -> bool:
        """May a step of G rows in groups of `group` beams over `ctx` take its previous states in the previous step's row order
        (states["_ungathered"], with the back-pointers in states["_parent"])?  For cbs_search(ungathered_ok=...)."""
        if G % ctx.nimg != 0:
            return False
        # (same decision as step(): the table must be in use; it has been formed by the time a search re-orders beams)
        att = 0 if (ctx.R > 128 or G < self.ATT_TABLE_MIN_ROWS or G // ctx.nimg < self.ATT_TABLE_MIN_ROWS_PER_IMAGE) else 1
        return bool(self.lib._raw_ssc_decode_ungathered_ok(C.byref(self._cfg), ctx.nimg, G, group, att))


    def search(self, ctx: ImageContext, sentiment: Optional[torch.Tensor], n_samples: int, beam: int, per_node: int, max_steps: int,
               end_index: int, eps0: torch.Tensor, eps: Optional[torch.Tensor], fsm: Optional[torch.Tensor] = None,
               compiled: Optional["CompiledFsm"] = None, mach: Optional[torch.Tensor] = None, skip_dead: bool = False,
               early_stop: bool = True):
        """The whole constrained beam search of one call in ONE library call (ssc_decode_search): ctx.nimg images x n_samples
        latent samples, batch entry b = (image, sample).  sentiment (B) or None; eps0 (B, Z), eps (max_steps - 1, B*S*beam, Z):
        the noise of every step, drawn by the caller.  fsm (M,S,S,V) / compiled / mach as in cbs_search.
        -> (predictions (B, S, beam, steps), log_probs (B, S, beam)); one wait at the end, for the number of steps."""
        d = self.dims
        B = ctx.nimg * n_samples
        S = 1 if fsm is None else fsm.size(1)
        G = B * S * beam
        dev = self.device
        if fsm is not None:
            assert fsm.is_cuda and fsm.dtype == torch.uint8 and fsm.is_contiguous()
            assert mach is not None or fsm.size(0) == B
        sd = _lib.SearchDesc()
        sd.nimg, sd.R, sd.n_samples = ctx.nimg, ctx.R, n_samples
        sd.S, sd.beam, sd.per_node, sd.max_steps, sd.end_index = S, beam, per_node, max_steps, end_index
        sd.feats, sd.imgbuf = ctx.feats.data_ptr(), ctx.buf.data_ptr()
        sent = sentiment.reshape(B).to(dev, torch.float32).contiguous() if sentiment is not None else None
        eps0 = eps0.to(dev, torch.float32).contiguous()
        assert tuple(eps0.shape) == (B, d.Z), eps0.shape
        if max_steps > 1:
            eps = eps.to(dev, torch.float32).contiguous()
            assert tuple(eps.shape) == (max_steps - 1, G, d.Z), (eps.shape, (max_steps - 1, G, d.Z))
        sd.sentiment, sd.eps0, sd.eps = _lib.ptr(sent), _lib.ptr(eps0), _lib.ptr(eps) if max_steps > 1 else None
        sd.obj_atts = _lib.ptr(ctx.obj)
        if mach is not None:   # one machine index per BATCH ENTRY (image, sample); the kernels index the machines with it unchecked
            assert fsm is not None and mach.numel() == B, (mach.shape, B)
            lo, hi = int(mach.min()), int(mach.max())
            assert 0 <= lo and hi < fsm.size(0), (lo, hi, fsm.size(0))
        mach = mach.to(dev, torch.int32).contiguous() if mach is not None else None
        sd.fsm, sd.mach = _lib.ptr(fsm), _lib.ptr(mach)
        if compiled is not None:
            assert compiled.dims.S == S and compiled.dims.P >= per_node
            sd.tables, sd.dims = _lib.ptr(compiled.tables), compiled.dims
        sd.skip_dead = 1 if (skip_dead and (compiled is not None or fsm is None)) else 0
        sd.early_stop = 1 if early_stop else 0
        pred = torch.empty(B, S * beam, max_steps, dtype=torch.int64, device=dev)
        lps = torch.empty(B, S, beam, dtype=torch.float32, device=dev)
        ctl = torch.empty(2 + 2 * max_steps, dtype=torch.int32, device=dev)
        sd.predictions, sd.log_probs, sd.ctl = _lib.ptr(pred), _lib.ptr(lps), _lib.ptr(ctl)
        flag = None
        if early_stop:
            flag, flag_dev = _host_flag()
            if flag_dev is not None:
                sd.host_flag, sd.host_flag_host = flag_dev, C.c_void_p(flag.data_ptr())
        nbytes = self.lib.ssc_decode_search_workspace_bytes(C.byref(self._cfg), C.byref(sd))
        if self._sws is None or self._sws.numel() < nbytes:
            self._sws = None   # (release before growing)
            self._sws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        p = self._params()
        self.lib.ssc_decode_search(C.byref(self._cfg), C.byref(p), C.byref(sd), _lib.ptr(self._sws), self._sws.numel(), _lib.stream_ptr())
        nsteps = int(ctl[0]) if early_stop else max_steps   # (the one wait of the call: its result is about to be read anyway)
        if flag is not None:
            _HOST_FLAGS.append(flag)
        return pred[:, :, :nsteps].contiguous().view(B, S, beam, nsteps), lps

    def _step_from_embedding(self, ctx, token_embedding, states, sentiment, eps, prior_mean_out=None, prior_mean=None, prior_var=None):
        G = token_embedding.size(0)
        table = token_embedding.to(self.device, torch.float32).contiguous()
        ids = torch.arange(G, dtype=torch.int64, device=self.device)
        return self.step(ctx, ids, states, sentiment, eps, want_log_probs=False, emb_table=table, prior_mean_out=prior_mean_out,
                         prior_mean=prior_mean, prior_var=prior_var)


DecodeEngine.step_from_embedding = DecodeEngine._step_from_embedding


class CompiledFsm:
    """Machines in the compiled form of ssc_fsm_compile (include/ssc.h): per (machine, from-state) a default target set plus a
    short list of exception tokens.  `fsm` (M,S,S,V) uint8 on the device stays referenced: a from-state that does not fit the
    form (more than `max_exceptions` tokens off the default) is flagged on the device and takes the dense scans.
    fill: how many of the smallest non-exception tokens are kept per from-state (>= per-node beam size of the search)."""

    def __init__(self, fsm: torch.Tensor, max_exceptions: int = 512, fill: int = 8):
        assert fsm.is_cuda and fsm.dtype == torch.uint8 and fsm.dim() == 4 and fsm.size(1) == fsm.size(2) <= 32, fsm.shape
        self.fsm = fsm.contiguous()
        M, S, _, V = self.fsm.shape
        lib = _lib.load()
        self.dims = _lib.FsmDims(M, S, V, int(max_exceptions), int(fill))
        nbytes = lib.ssc_fsm_tables_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise ValueError(f"ssc_fsm_tables_bytes: bad dims {fsm.shape}, E={max_exceptions}, P={fill}")
        self.tables = torch.empty(nbytes // 4, dtype=torch.int32, device=fsm.device)
        lib.ssc_fsm_compile(_lib.ptr(self.fsm), C.byref(self.dims), _lib.ptr(self.tables), nbytes, _lib.stream_ptr())

    def sparse_states(self) -> torch.Tensor:
        """(M, S) bool: which from-states took the compiled form (diagnostic; one device read)."""
        ms = self.dims.M * self.dims.S
        return self.tables[:ms].view(self.dims.M, self.dims.S) != 0


_HOST_FLAGS = []   # pinned words for the early-stop flag, recycled (allocating pinned memory is slow)


def _host_flag():
    """-> (pinned int32 tensor [2] holding zeros - stop flag, last completed step: ssc_beam_desc.host_flag -, its device-visible address)."""
    lib = _lib.load()
    t = _HOST_FLAGS.pop() if _HOST_FLAGS else torch.zeros(2, dtype=torch.int32).pin_memory()
    t.zero_()
    dp = C.c_void_p()
    if lib._raw_ssc_host_device_ptr(C.c_void_p(t.data_ptr()), C.byref(dp)) != 0:
        return t, None
    return t, dp


def cbs_search(start_predictions: torch.Tensor, start_state, step: Callable, fsm: Optional[torch.Tensor], end_index: int,
               max_steps: int, beam_size: int, per_node_beam_size: int, early_stop: bool = True,
               early_stop_every: int = 4, raw_logits: bool = False,
               ungathered_ok: Optional[Callable[[int, int], bool]] = None, compiled: Optional[CompiledFsm] = None,
               mach: Optional[torch.Tensor] = None, skip_dead: bool = False, compile_fsm: bool = True):
    """Constrained beam search with on-device bookkeeping (ssc_beam_first_fsm / ssc_beam_step_fsm / ssc_gather_rows /
    ssc_beam_backtrace_ctl).  `step(tokens (G,), state) -> (log_probs (G,V), state, ...)` as in cbs.py:127,170.
    Returns (predictions (B,S,beam,steps) int64, log_probs (B,S,beam)).
    fsm: (M,S,S,V) uint8 on the device, or None = the trivial one-state machine; batch entry b runs machine mach[b] (int32 (B);
    None: M = B, machine b).  compiled: the machines' compiled form (made here from `fsm` when S > 1 unless compile_fsm=False,
    which keeps the dense per-target scans): one scan per row instead of S, bit-identical selections.
    raw_logits: `step` returns un-normalised logits; the selection kernels normalise each row themselves (bit-identical
    selections and log-probs).
    early_stop: cbs.py:167 stops as soon as every beam has ended - a host sync per step in the reference.  Here the device itself
    notes the step after which all beams had ended (ssc_beam_desc.ctl) and turns every later step into a no-op, the host polls a
    pinned flag the device writes and merely stops QUEUEING steps once it sees it; the columns the reference would have produced
    are cut out at the end.  Same output for every machine, no host round trip.  early_stop_every <= 1: ask (and wait) after
    every step, as the reference does.
    skip_dead: ssc_beam_desc.skip_dead (rows without a finite beam are not scored from their logits; needs the compiled form).
    ungathered_ok(G, group) -> bool: the step function reads its previous states through the back-pointers itself
    (DecodeEngine.ungathered_ok): the states then stay in the previous step's row order, with state["_parent"] = back-pointers and
    state["_ungathered"] = True, and the re-ordering of cbs.py:236-250 (one gather per state tensor and step) is not done here."""
    lib = _lib.load()
    st = _lib.stream_ptr
    dev = start_predictions.device
    B = start_predictions.numel()
    if compiled is not None and fsm is None:
        fsm = compiled.fsm
    if fsm is None:   # the trivial one-state machine (every transition allowed): no mask is read on the device
        M, S, V = B, 1, None
    else:
        M, S, _, V = fsm.shape
        assert fsm.is_cuda and fsm.dtype == torch.uint8
        fsm = fsm.contiguous()
        assert mach is not None or M == B, (M, B)
        if compiled is None and compile_fsm and S > 1:
            compiled = CompiledFsm(fsm, fill=max(8, per_node_beam_size))
    if mach is not None:
        mach = mach.to(dev, torch.int32).contiguous()
        assert mach.numel() == B
        assert 0 <= int(mach.min()) and int(mach.max()) < M, "machine index out of range"   # (the kernels index the machines with it unchecked)
    assert not skip_dead or compiled is not None
    SB = S * beam_size
    preds = torch.empty(max_steps, B, SB, dtype=torch.int64, device=dev)
    backs = torch.empty(max(max_steps - 1, 1), B, SB, dtype=torch.int64, device=dev)
    last_lp = torch.empty(B, S, beam_size, dtype=torch.float32, device=dev)
    ctl = flag = flag_dev = None
    if early_stop:
        ctl = torch.zeros(2 + 2 * max_steps, dtype=torch.int32, device=dev)
        ctl[0] = max_steps
        flag, flag_dev = _host_flag()
    out = step(start_predictions, start_state)
    lp0, state = out[0], out[1]
    lp0 = lp0.contiguous()
    if V is None:
        V = lp0.shape[1]
    assert lp0.shape == (B, V), lp0.shape
    d = _lib.BeamDesc()
    d.raw_logits = 1 if raw_logits else 0
    d.fsm = _lib.ptr(fsm)
    d.tables = _lib.ptr(compiled.tables) if compiled is not None else None
    d.dims = compiled.dims if compiled is not None else _lib.FsmDims(M, S, V, 0, 1)
    d.mach = _lib.ptr(mach)
    d.B, d.beam, d.per_node, d.end_index = B, beam_size, per_node_beam_size, end_index
    d.skip_dead = 1 if skip_dead else 0
    d.ctl = _lib.ptr(ctl)
    d.max_steps = max_steps
    d.host_flag = flag_dev if ctl is not None else None
    d.scores, d.ld = _lib.ptr(lp0), lp0.stride(0)
    d.pred, d.lp_out = _lib.ptr(preds[0]), _lib.ptr(last_lp)
    lib.ssc_beam_first_fsm(C.byref(d), st())
    # enlarge states to (B*S*beam, *) batch-major (cbs.py:10-17,152-155)
    def enlarge(t):
        _, *rest = t.shape
        return t.view(B, 1, 1, *rest).expand(B, S, beam_size, *rest).reshape(-1, *rest).contiguous()

    state = {k: enlarge(v) for k, v in state.items() if not k.startswith("_")}
    state["_parent"] = torch.zeros(B, SB, dtype=torch.int64, device=dev)   # every beam of a group descends from the one start row
    sval = torch.empty(B * S * SB * per_node_beam_size, dtype=torch.float32, device=dev)
    sidx = torch.empty(B * S * SB * per_node_beam_size, dtype=torch.int64, device=dev)
    d.scratch_val, d.scratch_idx = _lib.ptr(sval), _lib.ptr(sidx)
    for t in range(1, max_steps):
        last = preds[t - 1].reshape(B * SB)
        if ctl is not None:
            if early_stop_every <= 1:
                if int(ctl[0]) <= t:   # (waits for the device, like the reference's `.all()`)
                    break
            elif int(flag[0]) != 0:    # a plain read of pinned host memory: nothing is queued, nothing waited for
                break
        state["_last_lp"] = last_lp
        if skip_dead:   # (for step functions that skip the rows whose logits will not be read: DecodeEngine.step)
            state["_skip"] = (last_lp, end_index)
        out = step(last, state)
        lp, state = out[0].contiguous(), out[1]
        new_lp = torch.empty_like(last_lp)
        d.scores, d.ld = _lib.ptr(lp), lp.stride(0)
        d.last_pred, d.last_lp = _lib.ptr(last), _lib.ptr(last_lp)
        d.pred, d.lp_out, d.backptr = _lib.ptr(preds[t]), _lib.ptr(new_lp), _lib.ptr(backs[t - 1])
        d.step_index = t
        lib.ssc_beam_step_fsm(C.byref(d), st())
        last_lp = new_lp
        new_state = {}
        # a step function that reads its previous states through the parent list (DecodeEngine.step at large G: `ungathered_ok`)
        # gets them as they are - the re-ordering of cbs.py:236-250 then happens inside its kernels' row lists
        leave = ungathered_ok is not None and SB > 1 and bool(ungathered_ok(B * SB, SB))
        for k, v in state.items():  # cbs.py:236-250
            if k.startswith("_"):
                continue
            if leave:
                new_state[k] = v
                continue
            v2 = v.reshape(B * SB, -1).contiguous()
            dst = torch.empty_like(v2)
            lib.ssc_gather_rows(_lib.ptr(v2), v2.stride(0), _lib.ptr(backs[t - 1]), B, SB, v2.size(1), _lib.ptr(dst), st())
            new_state[k] = dst.view_as(v)
        new_state["_parent"] = backs[t - 1]   # for the step function: which rows of a group now hold the same states
        if leave:
            new_state["_ungathered"] = True
        state = new_state
    if ctl is None:
        allp = torch.empty(B, SB, max_steps, dtype=torch.int64, device=dev)
        lib.ssc_beam_backtrace(_lib.ptr(preds), _lib.ptr(backs), max_steps, B, SB, _lib.ptr(allp), st())
        return allp.view(B, S, beam_size, max_steps), last_lp
    # a search the host stopped queueing at step t has written columns [0, t) and ctl[0] <= t; one that ran out has ctl[0] <= max_steps
    allp = torch.empty(B, SB, max_steps, dtype=torch.int64, device=dev)
    lib.ssc_beam_backtrace_ctl(_lib.ptr(preds), _lib.ptr(backs), _lib.ptr(ctl), max_steps, B, SB, end_index, _lib.ptr(allp), st())
    nsteps = int(ctl[0])   # (the one wait of the search: its result is about to be read anyway)
    _HOST_FLAGS.append(flag)
    return allp[:, :, :nsteps].contiguous().view(B, S, beam_size, nsteps), last_lp
