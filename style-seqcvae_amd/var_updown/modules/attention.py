"""Mirror of updown-baseline/updown/modules/attention.py:27-97: three bias-free Linear layers with the reference's attribute
names (same state_dict keys, same default-init RNG consumption) and a stand-alone `forward` on the HIP path.

Inside UpDownCell / UpDownCaptioner the attention step is part of the fused sequence kernels (csrc/sequence.hip,
csrc/decode.hip) and this `forward` is not called; it is the module-level API of the reference (attention weights for a
query vector), computed by the same kernels through the C ABI: the two projections by ssc_gemm, logits + masked softmax by
ssc_attn_fwd.  No autograd (the differentiable path is UpDownCaptioner.forward)."""
import ctypes as C
from typing import Optional

import torch
from torch import nn


class BottomUpTopDownAttention(nn.Module):
    def __init__(self, query_size: int, image_feature_size: int, projection_size: int):
        super().__init__()
        self._query_vector_projection_layer = nn.Linear(query_size, projection_size, bias=False)
        self._image_features_projection_layer = nn.Linear(image_feature_size, projection_size, bias=False)
        self._attention_layer = nn.Linear(projection_size, 1, bias=False)

    @torch.no_grad()
    def forward(self, query_vector: torch.Tensor, image_features: torch.Tensor,
                image_features_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query_vector (B, query_size), image_features (B, R, F), optional mask (B, R) -> attention weights (B, R)
        (attention.py:36-97; masked rows get weight 0, allennlp masked_softmax semantics; without a mask the reference takes
        a plain softmax, which the all-ones mask reproduces to below fp32 resolution)."""
        from ssc_runtime import lib as L
        lib = L.load()
        wq = self._query_vector_projection_layer.weight
        wv = self._image_features_projection_layer.weight
        wa = self._attention_layer.weight
        if not wq.is_cuda:
            raise RuntimeError("BottomUpTopDownAttention.forward runs on the HIP path only (no CPU fallback): move the module "
                               "and its inputs to a ROCm device")
        dev = wq.device
        q_in = query_vector.to(dev, torch.float32).contiguous()
        feats = image_features.to(dev, torch.float32).contiguous()
        B, R, F = feats.shape
        A = wq.size(0)
        mask = (torch.ones(B, R, device=dev) if image_features_mask is None
                else image_features_mask.to(dev, torch.float32).contiguous())

        def gemm(x, w, M, N, K, out):
            d = L.GemmDesc()
            d.nseg = 1
            d.seg[0].A, d.seg[0].lda, d.seg[0].B, d.seg[0].ldb, d.seg[0].K = x.data_ptr(), K, w.data_ptr(), w.stride(0), K
            d.M, d.N, d.a_kc, d.b_kc = M, N, 1, 1
            d.C, d.ldc = out.data_ptr(), N
            ws = torch.empty(8 * M * N + 64, device=dev)
            d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
            lib.ssc_gemm(C.byref(d), L.stream_ptr())

        q = torch.empty(B, A, device=dev)
        pv = torch.empty(B * R, A, device=dev)
        gemm(q_in, wq.detach().contiguous(), B, A, q_in.size(1), q)
        gemm(feats.view(B * R, F), wv.detach().contiguous(), B * R, A, F, pv)
        logits = torch.empty(B, R, device=dev)
        alpha = torch.empty(B, R, device=dev)
        att = torch.empty(B, F, device=dev)
        wa_flat = wa.detach().reshape(-1).contiguous()
        lib.ssc_attn_fwd(L.ptr(q), A, L.ptr(pv), L.ptr(wa_flat), L.ptr(mask), L.ptr(feats), B, R, A, F, 1, L.ptr(logits),
                         L.ptr(alpha), L.ptr(att), F, L.stream_ptr())
        return alpha
