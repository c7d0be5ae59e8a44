"""Build libssc_hip.so (gfx950) in-tree with hipcc.  `python style-seqcvae_amd/build.py [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libssc_hip.so")
SOURCES = ["gemm.hip", "pointwise.hip", "attention.hip", "sequence.hip", "decode.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + INCLUDE, "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "ssc.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def check_staged_loads(verbose=True):
    """Gate: the GEMM kernels' hand-scheduled register prefetch must survive code generation (tools/check_staged_loads.py:
    no instruction may touch a staged load's destination registers before the hand-counted wait).  Same flags, ISA only."""
    import tempfile
    tool = os.path.join(os.path.dirname(HERE), "tools", "check_staged_loads.py")
    if not os.path.exists(tool):
        return
    with tempfile.TemporaryDirectory() as td:
        isa = os.path.join(td, "gemm.s")
        cmd = [_hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(CSRC, "gemm.hip"), "-o", isa]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        rc = subprocess.call([sys.executable, tool, isa])
        if rc != 0:
            raise RuntimeError("gemm.hip: a staged-load destination register is touched before its wait in the generated ISA "
                               "(see the findings above); the library is NOT built")


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), s))
        objs.append(obj)
    for p, s in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    check_staged_loads(verbose)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
