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
rows = sorted((r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]),
              key=lambda r: int(r["Start_Timestamp"]))
# the probe launches each shape 3 (warm-up) + 6 times back to back: one group per shape, in command-line order
for i in range(0, len(rows), 9):
    grp = rows[i:i + 9]
    v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    r = grp[0]
    shape = args[i // 9] if i // 9 < len(args) else "?"
    print(f"{tag:8s} {shape:28s} {r['Kernel_Name'].split('(')[0][-34:]:34s} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}x{r.get('Grid_Size_Y')}x{r.get('Grid_Size_Z')}"
          f"  n={len(v)}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
