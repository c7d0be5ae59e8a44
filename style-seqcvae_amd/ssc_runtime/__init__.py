"""Host-side runtime over the libssc_hip.so C ABI (include/ssc.h).

PyTorch is used for device memory, streams and torch.distributed only; every
compute step of the hot path is a HIP kernel behind the C ABI.  Importing this
package never falls back to a CPU implementation: `lib.load()` raises if the
extension is missing.
"""
from .lib import load, SscError  # noqa: F401
