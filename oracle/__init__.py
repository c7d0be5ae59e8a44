"""CPU oracle for the Style-SeqCVAE (var_updown) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The product
path (``style-seqcvae_amd/``) never imports this package and fails loudly when
the HIP extension is missing.

Parity status: pinned against the reference's own ``UpDownCaptioner`` /
``UpDownCell`` / ``BottomUpTopDownAttention`` imported from ``/root/reference``
(see ``tests/golden/make_golden.py``), EXCEPT at the allennlp==0.8.4 boundary
(``masked_softmax``, ``masked_mean``, ``add_sentence_boundary_token_ids``,
``sequence_cross_entropy_with_logits``): allennlp is not vendored in the
reference and not installable here, so those four utilities are restated from
their published 0.8.4 semantics -> "parity unpinned" at that boundary only.
The CBS search driver (``cbs.py``) cannot execute under torch >= 1.2, so
``oracle.cbs_search`` is pinned by step-level goldens and invariants only.
"""
from .seqcvae_oracle import *  # noqa: F401,F403
