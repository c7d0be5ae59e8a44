"""Training / decoding engine over the C ABI: flat parameter + gradient buffers, the fused
T-step forward/BPTT, clip + SGD, and the data-parallel gradient exchange.

Reference for the behaviour reproduced here: UpDownCaptioner.forward
(var_updown/var_updown/models/updown_captioner.py:228-368) and the optimiser loop of
var_updown/scripts/train.py:154-176.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as _lib

P_CELL = "_updown_cell."
P_ATT = P_CELL + "_attention_lstm_cell."
P_ENC = P_CELL + "_language_lstm_cell_encoder."
P_DEC = P_CELL + "_language_lstm_cell_decoder."
P_BUTD = P_CELL + "_butd_attention."

# state_dict key -> ssc_params field
FIELD_OF = {
    "_embedding_layer.weight": "emb",
    P_ATT + "weight_ih": "att_w_ih", P_ATT + "weight_hh": "att_w_hh", P_ATT + "bias_ih": "att_b_ih", P_ATT + "bias_hh": "att_b_hh",
    P_BUTD + "_query_vector_projection_layer.weight": "wq",
    P_BUTD + "_image_features_projection_layer.weight": "wv",
    P_BUTD + "_attention_layer.weight": "wa",
    P_ENC + "weight_ih": "enc_w_ih", P_ENC + "weight_hh": "enc_w_hh", P_ENC + "bias_ih": "enc_b_ih", P_ENC + "bias_hh": "enc_b_hh",
    P_CELL + "fc_mean.weight": "fc_mean_w", P_CELL + "fc_mean.bias": "fc_mean_b",
    P_CELL + "fc_log_var.weight": "fc_lv_w", P_CELL + "fc_log_var.bias": "fc_lv_b",
    "_output_layer.weight": "out_w", "_output_layer.bias": "out_b",
    "_output_projection.0.weight": "proj_w", "_output_projection.0.bias": "proj_b",
    P_DEC + "weight_ih": "dec_w_ih", P_DEC + "weight_hh": "dec_w_hh", P_DEC + "bias_ih": "dec_b_ih", P_DEC + "bias_hh": "dec_b_hh",
}
HAS_LD = dict(_lib.PARAM_FIELDS)


class _EventWork:
    """wait() makes the compute stream wait for a collective that was issued on the communication stream (same call as the
    torch.distributed work handle's)."""

    def __init__(self, event, stream):
        self.event, self.stream = event, stream

    def wait(self):
        self.stream.wait_event(self.event)


@dataclass
class ModelDims:
    """Hot-path hyper-parameters (mirror of ssc_model_cfg)."""
    V: int
    E: int
    H: int
    A: int
    F: int
    Z: int
    S: int = 0            # conditioning columns on the language LSTMs (updown_cell.py:47-81): 0, 1 (sentiment) or, with kld_mode 2
                          # (SENTIMENT_VAE = 2: attention-pooled attribute means, Z wide), Z (LATENT_EMBEDDING "glove": all of them) or
                          # 1 ("senti_word_net": the first one, updown_cell.py:169-172)
    tied: bool = False    # E in {300,600}: frozen tied embedding (updown_captioner.py:75,112-119)
    kld_mode: int = 0     # 0: SENTIMENT_VAE == 0 formula, 1 otherwise (updown_captioner.py:298-303), 2: formula 1 with the prior
                          # mean of each step = the attention-pooled obj_atts (SENTIMENT_VAE = 2, updown_cell.py:160-163)
    pm_scale: float = 0.0  # prior_mean = pm_scale * sentiment
    prior_var: float = 1.0
    pad: int = 0
    boundary: int = 1
    gemm_mode: int = 0    # numerics of this engine's products: 0 = process default (ssc_set_gemm_mode), 1 = 3xBF16, 2 = exact-fp32 MFMA

    def cfg(self) -> _lib.ModelCfg:
        return _lib.ModelCfg(self.V, self.E, self.H, self.A, self.F, self.Z, self.S, int(self.tied), self.kld_mode,
                             float(self.pm_scale), float(self.prior_var), self.pad, self.boundary, int(self.gemm_mode))

    def param_shapes(self) -> "Dict[str, Tuple[int, ...]]":
        V, E, H, A, F, Z, S = self.V, self.E, self.H, self.A, self.F, self.Z, self.S
        sh = {"_embedding_layer.weight": (V, E)}
        sh[P_ATT + "weight_ih"] = (4 * H, E + F + 2 * H)
        sh[P_ATT + "weight_hh"] = (4 * H, H)
        sh[P_ATT + "bias_ih"] = (4 * H,)
        sh[P_ATT + "bias_hh"] = (4 * H,)
        sh[P_BUTD + "_query_vector_projection_layer.weight"] = (A, H)
        sh[P_BUTD + "_image_features_projection_layer.weight"] = (A, F)
        sh[P_BUTD + "_attention_layer.weight"] = (1, A)
        sh[P_ENC + "weight_ih"] = (4 * H, S + F + 2 * H)
        sh[P_ENC + "weight_hh"] = (4 * H, H)
        sh[P_ENC + "bias_ih"] = (4 * H,)
        sh[P_ENC + "bias_hh"] = (4 * H,)
        sh[P_CELL + "fc_mean.weight"] = (Z, H)
        sh[P_CELL + "fc_log_var.weight"] = (Z, H)      # adjacent to fc_mean.weight: one (2Z,H) GEMM operand
        sh[P_CELL + "fc_mean.bias"] = (Z,)
        sh[P_CELL + "fc_log_var.bias"] = (Z,)
        if self.tied:
            sh["_output_projection.0.weight"] = (E, H)
            sh["_output_projection.0.bias"] = (E,)
        else:
            sh["_output_layer.weight"] = (V, H)
            sh["_output_layer.bias"] = (V,)
        # decoder LSTM last: one contiguous range that the freeze schedule (train.py:156-161) can skip
        sh[P_DEC + "weight_ih"] = (4 * H, S + F + 2 * H + Z)
        sh[P_DEC + "weight_hh"] = (4 * H, H)
        sh[P_DEC + "bias_ih"] = (4 * H,)
        sh[P_DEC + "bias_hh"] = (4 * H,)
        return sh


def _r4(x):
    return (x + 3) // 4 * 4


class FlatStore:
    """One flat fp32 device buffer holding every tensor of `shapes` (16-B aligned offsets, rows of 2-D
    weights padded to a multiple of 4 floats so that 16 B/lane loads stay legal), exposed as views."""

    def __init__(self, shapes: "Dict[str, Tuple[int, ...]]", device, zero=True):
        self.shapes = dict(shapes)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, shp in shapes.items():
            n = shp[0] * _r4(shp[1]) if len(shp) == 2 and shp[0] > 1 else _r4(shp[-1])
            self.offsets[name] = (off, n)
            off += n  # n is a multiple of 4 floats: every tensor starts 16-B aligned, neighbours stay adjacent
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device) if zero else torch.empty(
            off, dtype=torch.float32, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        for name, shp in shapes.items():
            o, n = self.offsets[name]
            if len(shp) == 2 and shp[0] > 1:
                ld = _r4(shp[1])
                self.views[name] = self.flat[o:o + n].view(shp[0], ld)[:, :shp[1]]
            else:
                self.views[name] = self.flat[o:o + shp[-1]].view(*shp)

    def range_of(self, names: Sequence[str]) -> Tuple[int, int]:
        """[lo, hi) flat range covering `names` (must be laid out consecutively)."""
        lo = min(self.offsets[n][0] for n in names)
        hi = max(self.offsets[n][0] + self.offsets[n][1] for n in names)
        return lo, hi

    def c_struct(self, only: Optional[Sequence[str]] = None) -> _lib.Params:
        s = _lib.Params()
        for name, view in self.views.items():
            if only is not None and name not in only:
                continue
            f = FIELD_OF[name]
            setattr(s, f, view.data_ptr())
            if HAS_LD[f]:
                setattr(s, "ld_" + f, view.stride(0))
        return s


def params_struct(tensors: "Dict[str, torch.Tensor]") -> _lib.Params:
    """ssc_params over arbitrary (row-major, unit inner stride) tensors."""
    s = _lib.Params()
    for name, t in tensors.items():
        if name not in FIELD_OF:
            continue
        f = FIELD_OF[name]
        if t.dim() == 2 and t.stride(1) != 1 and t.size(0) > 1:
            raise ValueError(f"{name}: inner stride must be 1")
        setattr(s, f, t.data_ptr())
        if HAS_LD[f]:
            setattr(s, "ld_" + f, t.stride(0) if t.size(0) > 1 else t.size(1))
    return s


class TrainEngine:
    """Owns parameters, gradients and the activation workspace of one model replica on one GPU."""

    def __init__(self, dims: ModelDims, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("TrainEngine needs a ROCm GPU (no CPU fallback)")
        self.dims = dims
        self.device = torch.device(device if device is not None else "cuda")
        shapes = dims.param_shapes()
        self.params = FlatStore(shapes, self.device)
        self.grads = FlatStore(shapes, self.device)
        self.momentum = None
        self._ws = None
        self._ws_key = None
        self._cfg = dims.cfg()
        self._keep = None
        self._scratch = torch.zeros(1024 + 16, dtype=torch.float32, device=self.device)
        names = list(shapes)
        self.decoder_names = [n for n in names if n.startswith(P_DEC)]
        self.frozen_names = ["_embedding_layer.weight"] if dims.tied else []
        self.steps_done = 0
        self.fwd_version = 0
        self.dp_buckets = 1
        self.dp_overlap = True
        self.dp_force = False         # take the phased / overlapped path in a one-rank group too (tests: the real RCCL backend on one GPU)
        self.dp_autograd = False      # the autograd path (UpDownCaptioner.forward + loss.backward()) all-reduces the flat
                                      # gradient buffer inside backward (scripts/train.py without --fused-optimizer)
        self.dp_profile = False       # record the exposed all-reduce time of every overlapped backward (bench.py --gpus N)
        self.dp_exposure_events = []  # [(bwd kernels done, gradients reduced)] torch.cuda.Event pairs
        # gradient exchange of the overlapped backward: "rccl" = torch.distributed all-reduce (RCCL rings / trees); "xgmi" = the
        # direct reduce-scatter + all-gather over hipIpc peer mappings (ssc_runtime/xgmi.py, csrc/collective.hip: all xGMI links
        # at once); "auto" = map the peers, verify the direct path against torch.distributed, time both on the real
        # gradient buffer and keep the faster - falling back to "rccl" whenever the direct path cannot be set up on every rank
        self.dp_algo = "rccl"
        self.dp_choice = None         # what "auto" / "xgmi" resolved to: {"algo": ..., "rccl_ms": ..., "xgmi_ms": ..., "why": ...}
        self._xgmi = None
        self._comm_stream = None

    def adopt(self, named: "Dict[str, torch.nn.Parameter]"):
        """Re-home nn.Parameters into the flat store (copy once, then `param.data` IS the view).  Cheap when they
        already live there; re-adopts after .to()/.cuda() or any external re-binding of param.data."""
        for name, p in named.items():
            view = self.params.views[name]
            if p.data_ptr() != view.data_ptr() or p.stride() != view.stride() or p.device != view.device:
                with torch.no_grad():
                    view.copy_(p.data.to(self.device, torch.float32))
                p.data = view

    def param_version(self):
        return (self.params.flat._version, self.steps_done)

    # ---- parameters --------------------------------------------------------------------------------
    def load_state_dict(self, sd: "Dict[str, torch.Tensor]"):
        for name, view in self.params.views.items():
            src = sd[name]
            view.copy_(src.to(self.device, torch.float32).view_as(view))

    def state_dict(self) -> "Dict[str, torch.Tensor]":
        out = {k: v.detach().clone().contiguous() for k, v in self.params.views.items()}
        if self.dims.tied:
            out["_output_layer.weight"] = out["_embedding_layer.weight"]
        return out

    def grad_dict(self) -> "Dict[str, torch.Tensor]":
        return {k: v.detach().clone().contiguous() for k, v in self.grads.views.items()}

    # ---- fused forward / backward ----------------------------------------------------------------------
    def _workspace(self, B, R, L):
        key = (B, R, L)
        if self._ws_key != key:
            nbytes = self.lib.ssc_train_workspace_bytes(C.byref(self._cfg), B, R, L)
            self._ws = torch.empty(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
            self._ws_key = key
        return self._ws

    def _batch(self, feats, caps, sentiment, eps, obj_atts=None):
        B, R, F = feats.shape
        L = caps.shape[1]
        assert F == self.dims.F and feats.is_contiguous() and caps.is_contiguous() and eps.is_contiguous()
        assert caps.dtype == torch.int64 and feats.dtype == torch.float32 and eps.dtype == torch.float32
        assert tuple(eps.shape) == (L + 1, B, self.dims.Z), eps.shape
        sent = None
        if sentiment is not None:
            sent = sentiment.reshape(B).to(torch.float32).contiguous()
        if self.dims.kld_mode == 2:
            assert obj_atts is not None and tuple(obj_atts.shape) == (B, R, self.dims.Z), "SENTIMENT_VAE = 2 needs obj_atts (B,R,Z)"
            obj_atts = obj_atts.to(torch.float32).contiguous()
        else:
            obj_atts = None
        bt = _lib.Batch(B, R, L, feats.data_ptr(), caps.data_ptr(), sent.data_ptr() if sent is not None else None,
                        eps.data_ptr(), obj_atts.data_ptr() if obj_atts is not None else None)
        return bt, (sent, obj_atts)

    def forward(self, feats, caps, sentiment, eps, obj_atts=None):
        """-> (loss (B,), kld (B,)); keeps activations for backward().  obj_atts (B,R,S): per-region attribute means, kld_mode 2 only."""
        bt, sent = self._batch(feats, caps, sentiment, eps, obj_atts)
        ws = self._workspace(bt.B, bt.R, bt.L)
        loss = torch.empty(bt.B, dtype=torch.float32, device=self.device)
        kld = torch.empty(bt.B, dtype=torch.float32, device=self.device)
        p = self.params.c_struct()
        self.lib.ssc_train_fwd(C.byref(self._cfg), C.byref(p), C.byref(bt), _lib.ptr(ws), ws.numel() * 4, _lib.ptr(loss),
                               _lib.ptr(kld), _lib.stream_ptr())
        self._keep = (bt, feats, caps, sent, eps)
        self.fwd_version += 1
        return loss, kld

    # gradient ranges that are final after each backward phase (flat layout order: emb | att LSTM | attention |
    # enc LSTM | fc | output head | dec LSTM)
    def phase_ranges(self):
        names = list(self.grads.views)
        head = [n for n in names if n.startswith("_output_")]
        dec = self.decoder_names
        emb = [n for n in names if n.startswith("_embedding")]
        first = [n for n in names if n.startswith(P_ATT) or n.startswith(P_BUTD)]
        mid = [n for n in names if n.startswith(P_ENC) or n.startswith(P_CELL + "fc_")]
        # (mask, range final after it): the head's gradients travel under the whole BPTT loop, which finishes no range itself;
        # the three weight-gradient phases are independent of each other - the two lighter ones run first so that the first
        # large all-reduce starts ~0.2 ms earlier and the heaviest phase computes under two reductions
        # the embedding gradient (one product + a scatter, 40 MB at C2) goes first: its range travels under all other phases, and
        # the last range - what stays exposed - is the attention LSTM's alone (five ranges: VERDICT r2 next-4a)
        return [(16, self.grads.range_of(head)), (32, None), (64, self.grads.range_of(emb)), (8, self.grads.range_of(dec)),
                (4, self.grads.range_of(mid)), (128, self.grads.range_of(first))]

    def backward_overlapped(self, gl, gk, skip: Sequence[str] = (), group=None):
        """Backward in four phases; the sum all-reduce of each finished gradient range is issued asynchronously
        (RCCL runs it on its own stream) while the next phase's GEMMs run.  Returns the world size."""
        import torch.distributed as dist

        bt = self._keep[0]
        ws = self._workspace(bt.B, bt.R, bt.L)
        skipset = set(skip) | set(self.frozen_names)
        only = [n for n in self.grads.views if n not in skipset]
        p = self.params.c_struct()
        g = self.grads.c_struct(only=only)
        gl = gl.to(torch.float32).contiguous()
        gk = gk.to(torch.float32).contiguous()
        world = dist.get_world_size(group)
        works = []
        xg = self._resolve_dp_algo(group) if world > 1 else None
        cur = torch.cuda.current_stream(self.device)
        # only what the update will read travels: a frozen decoder LSTM / tied embedding keeps its (stale) gradient range at home
        t_lo, t_hi = self.trainable_range(bool(set(self.decoder_names) & skipset))
        for mask, rng in self.phase_ranges():
            self.lib.ssc_train_bwd_phases(C.byref(self._cfg), C.byref(p), C.byref(bt), _lib.ptr(ws), ws.numel() * 4,
                                          _lib.ptr(gl), _lib.ptr(gk), C.byref(g), mask, _lib.stream_ptr())
            if rng is not None:
                lo, hi = max(rng[0], t_lo), min(rng[1], t_hi)
                if hi > lo and xg is not None:
                    # direct path: the collective's kernels run on the communication stream behind this phase; the compute
                    # stream goes on with the next phase and waits for the `done` event before it reads the range
                    ready = torch.cuda.Event()
                    ready.record(cur)
                    self._comm_stream.wait_event(ready)
                    xg.allreduce(lo, hi, stream=self._comm_stream)
                    done = torch.cuda.Event()
                    done.record(self._comm_stream)
                    works.append((_EventWork(done, cur), (lo, hi)))
                elif hi > lo:
                    works.append((dist.all_reduce(self.grads.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True), (lo, hi)))
        if xg is not None:
            xg.poll(self._comm_stream)   # a bounded wait that gave up leaves the gradients un-summed: reported one step later at most
        if self.dp_profile:   # exposure = time the compute stream sits between its last backward kernel and the reduced gradients
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # the squared norm of every range is summed as soon as that range has arrived, under the reductions still in flight:
        # behind the last one only its own range's norm and the update are left
        n_parts = 0
        for w, (lo, hi) in works:
            w.wait()
            self.lib.ssc_sq_norm(_lib.ptr(self.grads.flat[lo:hi]), hi - lo, _lib.ptr(self._scratch),
                                 _lib.ptr(self._scratch[1026 + n_parts:1027 + n_parts]), _lib.stream_ptr())
            n_parts += 1
        self._sq_parts = self._scratch[1026:1026 + n_parts].sum(dim=0, keepdim=True) if n_parts else None
        if self.dp_profile:
            e1.record()
            self.dp_exposure_events.append((e0, e1))
        return world

    def _resolve_dp_algo(self, group):
        """The XgmiAllReduce to use for this group, or None for torch.distributed (see dp_algo).  Collective on first use."""
        if self.dp_algo == "rccl":
            return None
        if self.dp_choice is not None:
            return self._xgmi if self.dp_choice["algo"] == "xgmi" else None
        import sys

        import torch.distributed as dist

        from . import xgmi
        rank = dist.get_rank(group)
        log = (lambda m: print("[ssc dp]", m, file=sys.stderr, flush=True)) if rank == 0 else None
        self._comm_stream = torch.cuda.Stream(device=self.device)
        obj = xgmi.try_create(self.grads.flat, group=group, log=log)
        choice = {"algo": "rccl", "rccl_ms": None, "xgmi_ms": None, "why": "direct path unavailable"}
        if obj is not None and self.dp_algo == "xgmi":
            choice.update(algo="xgmi", why="requested")
        elif obj is not None:   # auto: time both on the whole gradient buffer (contents are scratch between steps)
            def timed(fn):
                torch.cuda.synchronize(self.device)
                dist.barrier(group=group)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fn()   # warm-up
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize(self.device)
                t = torch.tensor([e0.elapsed_time(e1) / 3], dtype=torch.float64,
                                 device=self.device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                return float(t.item())
            # the start-up self-test over the WHOLE gradient buffer (the construction's own covers 64 K floats): ordering / visibility
            # problems of a 446 MB exchange do not show on a buffer that fits the caches
            try:
                obj.self_test(self.grads.flat.numel())
                full_ok = True
            except Exception as e:   # noqa: BLE001 - XgmiError is raised on every rank or on none
                full_ok = False
                if log:
                    log(f"direct path failed its full-buffer self-test ({e}): using torch.distributed")
            self.grads.flat.zero_()
            r_ms = timed(lambda: dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM, group=group))
            x_ms = timed(lambda: obj.allreduce(0, self.grads.flat.numel()))
            try:
                obj.check()
                healthy = True
            except Exception:   # noqa: BLE001 - a bounded wait gave up on this rank
                healthy = False
            healthy = obj._agree(healthy and full_ok)   # collective: every rank takes the same path
            choice.update(rccl_ms=r_ms, xgmi_ms=x_ms)
            if not healthy:
                choice.update(why="a rank timed out in the direct path")
            elif x_ms < r_ms:
                choice.update(algo="xgmi", why="faster than torch.distributed on this node")
            else:
                choice.update(why="torch.distributed is faster on this node")
        self._xgmi = obj
        self.dp_choice = choice
        if log:
            log(f"gradient exchange: {choice}")
        return obj if choice["algo"] == "xgmi" else None

    def dp_exposure_ms(self):
        """Exposed all-reduce time per overlapped backward since the last call (synchronises)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self.dp_exposure_events]
        self.dp_exposure_events = []
        return out

    def backward_phased(self, gl, gk, masks: Sequence[int], skip: Sequence[str] = ()):
        """The backward as a given sequence of ssc_train_bwd_phases calls, no collective (tests: every legal order of the phases
        leaves the gradients of the one-call backward)."""
        bt = self._keep[0]
        ws = self._workspace(bt.B, bt.R, bt.L)
        skipset = set(skip) | set(self.frozen_names)
        p = self.params.c_struct()
        g = self.grads.c_struct(only=[n for n in self.grads.views if n not in skipset])
        gl = gl.to(torch.float32).contiguous()
        gk = gk.to(torch.float32).contiguous()
        for mask in masks:
            self.lib.ssc_train_bwd_phases(C.byref(self._cfg), C.byref(p), C.byref(bt), _lib.ptr(ws), ws.numel() * 4,
                                          _lib.ptr(gl), _lib.ptr(gk), C.byref(g), mask, _lib.stream_ptr())

    def backward(self, gl, gk, skip: Sequence[str] = ()):
        """Writes d(sum_b gl_b loss_b + gk_b kld_b)/dparam into self.grads for every parameter not in `skip`
        (nor frozen)."""
        bt = self._keep[0]
        ws = self._workspace(bt.B, bt.R, bt.L)
        skipset = set(skip) | set(self.frozen_names)
        only = [n for n in self.grads.views if n not in skipset]
        p = self.params.c_struct()
        g = self.grads.c_struct(only=only)
        gl = gl.to(torch.float32).contiguous()
        gk = gk.to(torch.float32).contiguous()
        self.lib.ssc_train_bwd(C.byref(self._cfg), C.byref(p), C.byref(bt), _lib.ptr(ws), ws.numel() * 4, _lib.ptr(gl),
                               _lib.ptr(gk), C.byref(g), _lib.stream_ptr())

    def workspace_view(self, which, T1=None):
        """Saved activations of the last forward (test hook; see ssc_train_workspace_view)."""
        bt = self._keep[0]
        ws = self._workspace(bt.B, bt.R, bt.L)
        ld = C.c_int(0)
        p = self.lib.ssc_train_workspace_view(C.byref(self._cfg), bt.B, bt.R, bt.L, _lib.ptr(ws), which, C.byref(ld))
        off = (p - ws.data_ptr()) // 4
        T, B = bt.L + 1, bt.B
        d = self.dims
        if which <= 5:
            return ws[off:off + (T + 1) * B * ld.value].view(T + 1, B, ld.value)[:, :, :d.H]
        if which == 6:
            return ws[off:off + T * B * bt.R].view(T, B, bt.R)
        if which in (7, 8):
            return ws[off:off + T * B * ld.value].view(T, B, ld.value)[:, :, :d.Z]
        if which == 9:
            return ws[off:off + T * B * ld.value].view(T, B, ld.value)[:, :, :d.V]
        if which == 11:
            return ws[off:off + T * B * ld.value].view(T, B, ld.value)[:, :, :d.F]
        raise ValueError(which)

    # ---- optimiser (train.py:126-134,173-176) ---------------------------------------------------------
    def trainable_range(self, decoder_frozen: bool) -> Tuple[int, int]:
        """The flat layout is [frozen tied embedding | always-trainable | decoder LSTM], so the trainable part is
        one contiguous range whatever the freeze schedule (train.py:156-161) says."""
        lo, hi = 0, self.params.numel
        if self.frozen_names:
            lo = self.params.range_of(self.frozen_names)[1]
        if decoder_frozen:
            hi = self.params.range_of(self.decoder_names)[0]
        return lo, hi

    def clip_sgd_step(self, lr, momentum=0.9, weight_decay=0.001, max_norm=12.5, decoder_frozen=False, gscale=1.0, sq_norm=None):
        """clip_grad_norm_(max_norm) + SGD(momentum, weight_decay) on the flat buffers, skipping frozen ranges
        (torch>=2 semantics: parameters without a gradient are not touched).  Momentum buffers start at zero,
        which reproduces torch's lazily created buffer (first step: buf = d_p).  Returns the squared grad norm."""
        if self.momentum is None:
            self.momentum = torch.zeros_like(self.params.flat)
        lo, hi = self.trainable_range(decoder_frozen)
        st = _lib.stream_ptr()
        if sq_norm is not None:   # already summed range by range (backward_overlapped)
            sq = sq_norm
        else:
            sq = self._scratch[1024:1025]
            self.lib.ssc_sq_norm(_lib.ptr(self.grads.flat[lo:hi]), hi - lo, _lib.ptr(self._scratch), _lib.ptr(sq), st)
        self.lib.ssc_sgd_step(_lib.ptr(self.params.flat[lo:hi]), _lib.ptr(self.grads.flat[lo:hi]),
                              _lib.ptr(self.momentum[lo:hi]), hi - lo, _lib.ptr(sq), float(gscale), float(max_norm),
                              float(lr), float(momentum), float(weight_decay), 0, st)
        self.steps_done += 1
        return sq

    def optimizer_state_dict(self, named_parameters, lr, momentum, weight_decay, iteration) -> dict:
        """The fused optimiser's state in torch.optim.SGD's state_dict layout (one momentum_buffer per parameter, indexed
        in `named_parameters` order = model.parameters() order), so that a checkpoint written on either path of
        scripts/train.py resumes on the other and `optimizer.load_state_dict` of the reference's train.py:142-151 accepts
        it.  `iteration` rides along inside this entry: the reference loads every OTHER top-level key into the model."""
        names = [n for n, _ in named_parameters]
        state = {}
        if self.momentum is not None:
            for i, n in enumerate(names):
                o, cnt = self.params.offsets[n]
                shp = self.params.shapes[n]
                if len(shp) == 2 and shp[0] > 1:
                    buf = self.momentum[o:o + cnt].view(shp[0], _r4(shp[1]))[:, :shp[1]]
                else:
                    buf = self.momentum[o:o + shp[-1]].view(*shp)
                state[i] = {"momentum_buffer": buf.detach().clone().contiguous()}
        group = {"lr": float(lr), "momentum": float(momentum), "dampening": 0, "weight_decay": float(weight_decay),
                 "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "iteration": int(iteration)}

    def load_optimizer_state_dict(self, named_parameters, sd: dict):
        """Inverse of optimizer_state_dict; also accepts a torch.optim.SGD state_dict (parameters without a buffer - never
        updated so far - start from zero, torch's lazily created buffer)."""
        names = [n for n, _ in named_parameters]
        if self.momentum is None:
            self.momentum = torch.zeros_like(self.params.flat)
        self.momentum.zero_()
        for i, st in sd.get("state", {}).items():
            buf = st.get("momentum_buffer")
            if buf is None:
                continue
            n = names[int(i)]
            o, cnt = self.params.offsets[n]
            shp = self.params.shapes[n]
            if len(shp) == 2 and shp[0] > 1:
                dst = self.momentum[o:o + cnt].view(shp[0], _r4(shp[1]))[:, :shp[1]]
            else:
                dst = self.momentum[o:o + shp[-1]].view(*shp)
            dst.copy_(buf.to(self.device, torch.float32).view_as(dst))

    # ---- data parallel -----------------------------------------------------------------------------------
    def allreduce_grads(self, group=None):
        """One RCCL all-reduce (sum) over the flat gradient buffer; the 1/world_size is folded into the SGD kernel."""
        from . import dp

        return dp.allreduce_flat(self.grads.flat, group=group, n_buckets=self.dp_buckets)

    def train_step(self, feats, caps, sentiment, eps, lr, kld_weight=750.0, momentum=0.9, weight_decay=0.001,
                   max_norm=12.5, decoder_frozen=False, group=None, obj_atts=None):
        """fwd + bwd + (all-reduce) + clip + SGD: one iteration of train.py:154-176.  Returns (loss, kld) per row."""
        loss, kld = self.forward(feats, caps, sentiment, eps, obj_atts)
        B = loss.numel()
        key = (B, float(kld_weight))
        if getattr(self, "_upstream_key", None) != key:   # d(mean loss + mean kld / KLD_WEIGHT) / d(loss_b, kld_b): constant per (B, weight)
            self._upstream = (torch.full((B,), 1.0 / B, dtype=torch.float32, device=self.device),
                              torch.full((B,), 1.0 / (B * kld_weight), dtype=torch.float32, device=self.device))
            self._upstream_key = key
        gl, gk = self._upstream
        import torch.distributed as dist

        skip = self.decoder_names if decoder_frozen else ()
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.dp_force) and self.dp_overlap:
            world = self.backward_overlapped(gl, gk, skip=skip, group=group)
            sq_parts = self._sq_parts
        else:
            self.backward(gl, gk, skip=skip)
            world = self.allreduce_grads(group)
            sq_parts = None
        self.clip_sgd_step(lr, momentum, weight_decay, max_norm, decoder_frozen, gscale=1.0 / world, sq_norm=sq_parts)
        return loss, kld
