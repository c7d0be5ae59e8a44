#!/usr/bin/env python3
"""Lists the kernels of every HIP source that use scratch memory (register spills) in the gfx950 ISA of the product flags.
A spill in a latency-bound kernel is a vector-memory round trip nobody asked for (and in gemm.hip it would break the hand-counted
waits: build.py gates that file itself).  Exit code 1 if any kernel spills.   python tools/scratch_check.py"""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ssc_build", os.path.join(ROOT, "style-seqcvae_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for src in B.SOURCES:
        out = os.path.join(td, src + ".s")
        subprocess.check_call([B._hipcc()] + [f for f in B.FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
        names = re.findall(r"\.amdhsa_kernel (\S+)", text)
        spills = []
        for name in names:
            blk = text[text.index(".amdhsa_kernel " + name):]
            blk = blk[:blk.index(".end_amdhsa_kernel")]
            n = int(re.search(r"private_segment_fixed_size (\d+)", blk).group(1))
            if n:
                spills.append((subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110], n))
        print(f"{src}: {len(names)} kernels, {len(spills)} with scratch")
        for n, b in spills:
            print(f"   {b:5d} B  {n}")
        bad += len(spills)
sys.exit(1 if bad else 0)
