"""Direct gradient all-reduce over peer-mapped buffers (hipIpc) - the hand-written fallback SURVEY 8(e) plans for the case that
RCCL keeps the 446 MB gradient all-reduce on rings: every rank maps every other rank's flat gradient buffer and a flag block into
its address space, then `ssc_xgmi_allreduce` (csrc/collective.hip) runs reduce-scatter + all-gather as plain HIP kernels that read
the peers' memory directly, all xGMI links of a GPU at once.  Replaces the reduce-add of nn.DataParallel
(var_updown/scripts/train.py:123-124); `torch.distributed` (RCCL) stays the reference it is verified against at set-up.

Set-up is collective: every rank of `group` constructs XgmiAllReduce on its own flat buffer.  The owner exports the hipMalloc
allocation that contains its buffer (`ssc_xgmi_ipc_export`: hipIpcGetMemHandle of the allocation base + the buffer's offset), the
64-byte handles travel through `dist.all_gather_object`, and every peer opens them UNDER ITS OWN DEVICE (`ssc_xgmi_ipc_open`:
the mapping is made for the device whose kernels read it - torch.multiprocessing's CUDA-IPC rebuild would map it for the owner's
device index instead).
`HSA_ENABLE_IPC_MODE_LEGACY=0` must be in the environment before the first HIP call (dmabuf IPC: this pool's driver has no other).
"""
import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import lib as L


class XgmiError(RuntimeError):
    pass


class XgmiAllReduce:
    def __init__(self, flat: torch.Tensor, group=None, verify: bool = True, timeout_polls: int = 0):
        """timeout_polls: bound of every cross-process wait in s_sleep(32) polls (0 = the library's default, 2**22: a few seconds).
        A rank that is late by more than that - a checkpoint write, a validation pass on one rank only - makes its peers' waits
        give up: the error word is set, the data kernels of that and every later collective leave the buffers untouched, and
        `poll()` / `check()` raise.  Keep slow one-rank work out of the step loop, or raise the bound."""
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.lib = L.load()
        self.flat = flat
        self.device = flat.device
        self.timeout_polls = int(timeout_polls)
        self._err_pending = None
        # preconditions and the export can fail on ONE rank only (an expandable-segment / VMM allocation has no IPC handle): the
        # failure is carried INTO the collective below and voted on, so that no rank is left waiting in all_gather_object
        pre = None
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.data_ptr() % 16 == 0):
            pre = "flat buffer must be a contiguous, 16-byte aligned fp32 CUDA tensor"
        elif self.world > L.SSC_XGMI_MAX_RANKS:
            pre = f"world size {self.world} > {L.SSC_XGMI_MAX_RANKS}"
        # 3 stages x MAX_RANKS sequence words, then (from word 32) a 32-word pattern block the peers read back through the copy
        # engine before any kernel touches a fresh mapping (_probe_peers)
        self.flags = torch.zeros(64, dtype=torch.int32, device=self.device)
        self.flags[32:] = torch.arange(32, dtype=torch.int32, device=self.device) * 7919 + 1000003 * (self.rank + 1)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.seq = 0
        torch.cuda.synchronize(self.device)
        self._opened = {}   # (rank, handle bytes) -> mapped allocation base (an allocation may hold both buffers)
        mine = None
        if pre is None:
            try:
                mine = (self.device.index, self._export(flat), self._export(self.flags), flat.numel())
            except Exception as e:   # noqa: BLE001 - reported through the vote
                pre = f"export failed: {type(e).__name__}: {e}"
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)   # (always entered, with None on a rank that could not export)
        if not self._agree(pre is None and all(h is not None for h in handles)):
            raise XgmiError(pre or "a peer rank could not export its buffers")
        # Every step that can fail on ONE rank only (mapping a peer, the self-test's comparison) is followed by a collective vote,
        # so that all ranks raise - or go on - together and never wait for each other in different collectives.
        why = self._map_peers(handles)
        if not self._agree(why is None):
            raise XgmiError(why or "a peer rank could not map the buffers")
        why = self._probe_peers()
        if not self._agree(why is None):
            raise XgmiError(why or "a peer rank could not read a mapped buffer")
        if verify and not self._agree(self._self_test_local()):
            raise XgmiError("xgmi all-reduce self-test: result differs from torch.distributed.all_reduce (on this or a peer rank)")

    def _agree(self, ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    def _export(self, t: torch.Tensor):
        handle = C.create_string_buffer(64)
        off = C.c_size_t(0)
        with torch.cuda.device(self.device):
            self.lib.ssc_xgmi_ipc_export(L.ptr(t), handle, C.byref(off))
        return bytes(handle.raw), int(off.value)

    def _open(self, j: int, exported) -> int:
        handle, off = exported
        key = (j, handle)
        if key not in self._opened:
            base = C.c_void_p()
            with torch.cuda.device(self.device):   # the mapping is made for THIS device
                self.lib.ssc_xgmi_ipc_open(C.create_string_buffer(handle, 64), C.byref(base))
            self._opened[key] = int(base.value)
        return self._opened[key] + off

    def _probe_peers(self) -> Optional[str]:
        """First touch of every fresh mapping by the runtime's copy engine (an error code on failure, never a GPU fault): each
        peer's pattern block must read back as that peer wrote it."""
        try:
            import numpy as np
            for j in range(self.world):
                if j == self.rank:
                    continue
                got = np.zeros(32, dtype=np.int32)
                with torch.cuda.device(self.device):
                    self.lib.ssc_xgmi_peek(C.c_void_p(int(self.comm.flags[j]) + 32 * 4), got.ctypes.data_as(C.c_void_p), 32 * 4)
                want = np.arange(32, dtype=np.int64) * 7919 + 1000003 * (j + 1)
                if not np.array_equal(got.astype(np.int64), want):
                    return f"rank {j}'s mapped flag block does not read back as written"
            return None
        except Exception as e:   # noqa: BLE001 - reported through the vote
            return f"{type(e).__name__}: {e}"

    def _map_peers(self, handles) -> Optional[str]:
        try:
            comm = L.XgmiComm()
            comm.world, comm.rank = self.world, self.rank
            for j, (dev_j, h_flat, h_flags, numel) in enumerate(handles):
                if j == self.rank:
                    comm.buf[j], comm.flags[j] = self.flat.data_ptr(), self.flags.data_ptr()
                    continue
                if numel != self.flat.numel():
                    return "ranks disagree on the buffer size"
                if dev_j != self.device.index:   # a peer GPU: kernels on this device must be allowed to touch its memory
                    with torch.cuda.device(self.device):
                        self.lib.ssc_xgmi_enable_peer(dev_j)
                comm.buf[j], comm.flags[j] = self._open(j, h_flat), self._open(j, h_flags)
            self.comm = comm
            return None
        except Exception as e:   # noqa: BLE001 - reported through the vote
            return f"{type(e).__name__}: {e}"

    def allreduce(self, lo: int = 0, hi: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None):
        """Enqueue the in-place sum over all ranks of flat[lo:hi] on `stream` (default: the current stream).  Every rank must
        issue the same sequence of calls.  lo / hi: multiples of 4 floats."""
        hi = self.flat.numel() if hi is None else hi
        if hi == lo:
            return
        self.seq += 1
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.lib.ssc_xgmi_allreduce(C.byref(self.comm), lo, hi, self.seq, self.timeout_polls, L.ptr(self.err), C.c_void_p(st.cuda_stream))

    def poll(self, stream: Optional[torch.cuda.Stream] = None):
        """Non-blocking look at the error word: raises XgmiError if a wait of a collective issued BEFORE the previous poll() gave
        up (the word travels to pinned host memory behind `stream` and is read once that copy has completed).  The training step
        calls it once per step, so a timed-out exchange - whose data kernels have left the gradients un-summed - is reported one
        step later at most instead of silently diverging the replicas."""
        if self._err_pending is not None and self._err_pending[0].query():
            code = int(self._err_pending[1][0])
            self._err_pending = None
            if code:
                raise XgmiError(f"xgmi all-reduce: rank {self.rank} timed out waiting for its peers at stage {code - 1}; the gradients "
                                "of that step were NOT summed")
        if self._err_pending is None:
            st = stream if stream is not None else torch.cuda.current_stream(self.device)
            host = torch.empty(1, dtype=torch.int32).pin_memory()
            with torch.cuda.stream(st):
                host.copy_(self.err, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(st)
            self._err_pending = (ev, host)

    def close(self):
        """Collective: unmap the peers' buffers (before the owning processes exit)."""
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        for base in self._opened.values():
            try:
                self.lib.ssc_xgmi_ipc_close(C.c_void_p(base))
            except Exception:   # noqa: BLE001 - best effort at shutdown
                pass
        self._opened = {}
        self.comm = None
        dist.barrier(group=self.group)

    def check(self):
        """Synchronises; raises if any bounded wait of the collectives issued so far gave up (a peer never arrived)."""
        code = int(self.err.item())
        if code:
            raise XgmiError(f"xgmi all-reduce: rank {self.rank} timed out waiting for its peers at stage {code - 1}")

    def self_test(self, n: int = 1 << 16):
        """Collective.  Raises on every rank if the direct path's result differs from torch.distributed's on any rank."""
        if not self._agree(self._self_test_local(n)):
            raise XgmiError("xgmi all-reduce self-test: result differs from torch.distributed.all_reduce (on this or a peer rank)")

    def _self_test_local(self, n: int = 1 << 16) -> bool:
        """A small all-reduce of known per-rank values through the direct path, compared with torch.distributed's result of the
        same input - exact equality is required (integer-valued floats: no rounding in either).  Uses the head of the buffer and
        restores it.  The collectives inside are issued unconditionally; the verdict is this rank's own."""
        n = min(n, self.flat.numel()) & ~3
        if n == 0:
            return True
        keep = self.flat[:n].clone()
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        vals = torch.randint(-1000, 1000, (n,), generator=g).float().to(self.device)
        want = vals.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM, group=self.group)
        self.flat[:n].copy_(vals)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        ok = True
        try:
            self.allreduce(0, n)
            self.check()
        except Exception:   # noqa: BLE001
            ok = False
        ok = ok and torch.equal(self.flat[:n], want)
        dist.barrier(group=self.group)   # nobody restores its buffer while a peer may still read it
        self.flat[:n].copy_(keep)
        torch.cuda.synchronize(self.device)
        return ok


def try_create(flat: torch.Tensor, group=None, log=None) -> Optional[XgmiAllReduce]:
    """XgmiAllReduce, or None (with the reason logged) when the peers cannot be mapped or the self-test fails on ANY rank: the
    caller then stays on torch.distributed.  The decision is made collectively, so that all ranks take the same path."""
    try:
        return XgmiAllReduce(flat, group=group, verify=True)   # raises on every rank or on none (votes after each fallible step)
    except Exception as e:   # noqa: BLE001 - any failure means "use RCCL"
        if log:
            log(f"xgmi all-reduce unavailable ({type(e).__name__}: {e}): using torch.distributed")
        return None
