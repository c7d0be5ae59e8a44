"""Build libssc_hip.so (gfx950) in-tree with hipcc.  `python style-seqcvae_amd/build.py [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libssc_hip.so")
SOURCES = ["gemm.hip", "pointwise.hip", "attention.hip", "sequence.hip", "decode.hip", "fsm.hip", "search.hip", "collective.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + INCLUDE, "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


STAMP = os.path.join(HERE, ".build_stamp")


def source_hash():
    """sha256 over the compiler flags and every source / header the library is built from (file names + contents): the
    library is current iff the stamp written next to it by the build that produced it carries this hash AND the library's own
    sha256 (mtimes are not evidence: a checkout or a copy to another machine resets them)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).replace(HERE, "").replace(os.path.dirname(HERE), "").encode())
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _file_hash(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        src, lib = open(STAMP).read().split()
    except ValueError:
        return True
    return src != source_hash() or lib != _file_hash(LIB)


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
    with open(STAMP, "w") as f:
        f.write(source_hash() + " " + _file_hash(LIB) + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
