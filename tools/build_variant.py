#!/usr/bin/env python3
"""Build the library of the CURRENT source tree into style-seqcvae_amd/<dir>/libssc_hip.so (extra compiler flags optional) - the
second build of a same-box A/B (`SSC_DEBUG=1 SSC_LIB_PATH=... python bench.py`).  python tools/build_variant.py <dir> [flags...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "style-seqcvae_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402

out = os.path.join(PKG, sys.argv[1])
os.makedirs(out, exist_ok=True)
flags = B.FLAGS + sys.argv[2:]
objs, procs = [], []
for s in B.SOURCES:
    obj = os.path.join(out, s.replace(".hip", ".o"))
    procs.append(subprocess.Popen([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj]))
    objs.append(obj)
for p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed")
lib = os.path.join(out, "libssc_hip.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
for o in objs:
    os.remove(o)
print(lib)
