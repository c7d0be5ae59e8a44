"""ConstrainedBeamSearch with the reference's constructor / search signature
(updown-baseline/updown/modules/cbs.py:20-66); the bookkeeping runs in HIP kernels (ssc_beam_*)."""
from typing import Optional

from ssc_runtime.decode import cbs_search


class ConstrainedBeamSearch(object):
    def __init__(self, end_index: int, max_steps: int = 20, beam_size: int = 5, per_node_beam_size: Optional[int] = None):
        self._end_index = end_index
        self.max_steps = max_steps
        self.beam_size = beam_size
        self.per_node_beam_size = per_node_beam_size or self.beam_size

    def search(self, start_predictions, start_state, step, fsm):
        """-> (predictions (B, S, beam, steps), log_probabilities (B, S, beam)); `step` returns the 5-tuple of
        var_updown's eval _decode_step (cbs.py:127,170)."""
        return cbs_search(start_predictions, start_state, step, fsm, self._end_index, self.max_steps, self.beam_size,
                          self.per_node_beam_size, early_stop=True)
