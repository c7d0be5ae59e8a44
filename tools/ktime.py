#!/usr/bin/env python3
"""Run tools/gemm_probe.py under rocprofv3 --kernel-trace and print true device durations per GEMM kernel dispatch group
(the probe's own timings are host-bound below ~45 us because of the Python-side descriptor building)."""
import collections, csv, glob, os, subprocess, sys, tempfile
tag = sys.argv[1]
args = sys.argv[2:]
out = f"gpurun_out/kt_{tag}"
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", "python3", "tools/gemm_probe.py", "6"] + args,
               env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
f = glob.glob(out + "/*/*_kernel_trace.csv")[0]
groups = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" not in n:
        continue
    key = (n.split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in groups.items():
    v = sorted(v)
    print(f"{tag:18s} {k[0]:42s} grid {k[1]:>7s}x{k[2]}x{k[3]:>3s}  n={len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
