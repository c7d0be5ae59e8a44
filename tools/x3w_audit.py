#!/usr/bin/env python3
"""Address audit of the wave-specialised GEMM kernels (diagnostics; not part of the product build).

  python tools/x3w_audit.py build                  # here: second library with -DSSC_X3W_AUDIT under style-seqcvae_amd/_audit
  python tools/x3w_audit.py run <command ...>      # on the GPU: run the command with that library loaded

In the audit library every hand-issued operand load of gemm_x3w_kernel (64x256 and 128x128, NT / NN / TN, grouped or not,
with or without device-side row lists) reports the byte range it touches relative to the operand base of its K segment
(csrc/gemm.hip, SSC_AUDIT_TOUCH); each launch site synchronises and compares the ranges with the spans its descriptors imply
(rows x leading dimension, row lists read back from the device).  A violation is printed with the product's shape; the library
prints "[x3w audit] N launches, R ranges checked, V VIOLATIONS" at process exit.  The loads themselves are unchanged, so the
audited run computes the same results (the test-suite passes under it), only slower: every audited launch synchronises.
Purpose: ADVICE r2 / DESIGN 9 - the address arithmetic of the M = 5000 four-segment decode products (and of every other
shape the tests and the benchmark exercise) is CHECKED on the device at the exact shapes, instead of argued from the source."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "style-seqcvae_amd")
OUT = os.path.join(PKG, "_audit")
sys.path.insert(0, PKG)


def build():
    import build as B
    os.makedirs(OUT, exist_ok=True)
    flags = B.FLAGS + ["-DSSC_X3W_AUDIT"]
    objs = []
    procs = []
    for s in B.SOURCES:
        obj = os.path.join(OUT, s.replace(".hip", ".o"))
        procs.append(subprocess.Popen([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj]))
        objs.append(obj)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    lib = os.path.join(OUT, "libssc_hip.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    print(lib)


def run(cmd):
    env = dict(os.environ, SSC_DEBUG="1", SSC_LIB_PATH=os.path.join(OUT, "libssc_hip.so"))
    raise SystemExit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        raise SystemExit(__doc__)
