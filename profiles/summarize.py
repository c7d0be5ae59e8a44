#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small files committed under profiles/.

  python profiles/summarize.py pmc  <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
  python profiles/summarize.py stats <kernel_stats.csv> <out.csv>
  python profiles/summarize.py sq    <sq_counter_collection.csv> <grbm_counter_collection.csv> <kernel_stats.csv> <out.csv>

PMC correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
half the bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled for the GEMM / streaming
kernels (all of ours read 16 B/lane); WRITE_SIZE is exact for 16-B-per-lane stores and is used as is.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"gemm_kernel<(\w+), (\w+), (\d), (\d), (\d), (\w+)>", name)
    if m:
        a, b, wm, wn, pf, vec = m.groups()
        kind = {("true", "true"): "NT", ("true", "false"): "NN", ("false", "false"): "TN", ("false", "true"): "TT"}[(a, b)]
        return f"gemm_kernel<{kind},{64 * int(wm)}x{64 * int(wn)},{'vec' if vec == 'true' else 'scalar'}>"
    m = re.search(r"gemm_x3b_kernel<(\w+), (\w+), (\w+)>", name)
    if m:
        a, b, kg = m.groups()
        kind = {("true", "true"): "NT", ("true", "false"): "NN", ("false", "false"): "TN"}[(a, b)]
        return f"gemm_x3b_kernel<{kind},128x128{',rowgather' if kg == 'true' else ''}>"
    m = re.search(r"gemm_x3w_kernel<(\w+), (\w+), (\w+), (\d+), (\d+), (\d)(?:, \d)?(?:, (\w+))?>", name)   # (+ producer waves since r02_b, + 2xFP16 flag since r04)
    if m:
        a, b, kg, tm, tn, pf, f16 = m.groups()
        kind = {("true", "true"): "NT", ("true", "false"): "NN", ("false", "false"): "TN"}[(a, b)]
        return f"gemm_x3w_kernel<{kind},{tm}x{tn}{',rowgather' if kg == 'true' else ''}{',f16x2' if f16 == 'true' else ''}>"
    m = re.search(r"gemm_x3_kernel<(\d), (\d), (\d)>", name)
    if m:
        return f"gemm_x3_kernel<NT,64x{64 * int(m.group(2))},lds{m.group(3)}>"
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+)(<[\w, ]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:50]


def pmc(fetch_csv, write_csv, out):
    acc = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
    for path, key in ((fetch_csv, "fetch_kib"), (write_csv, "write_kib")):
        n = collections.Counter()
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            acc[k][key] += float(r["Counter_Value"])
            n[k] += 1
        for k, c in n.items():
            acc[k]["launches"] = max(acc[k]["launches"], c)
    res = {}
    for k, v in acc.items():
        L = max(v["launches"], 1)
        res[k] = {"launches": L, "hbm_read_bytes_per_launch": 2.0 * v["fetch_kib"] * 1024 / L,
                  "hbm_write_bytes_per_launch": v["write_kib"] * 1024 / L,
                  "hbm_bytes_per_launch": (2.0 * v["fetch_kib"] + v["write_kib"]) * 1024 / L}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1 "
                       "--no-cpu-baseline`; read side x2 (gfx950 FETCH_SIZE half-count of 16 B/lane streams)",
               "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))},
              open(out, "w"), indent=1)


def stats(stats_csv, out):
    rows = list(csv.DictReader(open(stats_csv)))
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for r in rows:
            f.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},"
                    f"{float(r['AverageNs']) / 1e3:.2f},{r['Percentage']}\n")


def sq(sq_csv, grbm_csv, stats_csv, out):
    """Matrix-pipe utilisation per kernel from the SQ pass (summed over all dispatches of the kernel):
      mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES per SIMD): share of the time a SIMD with resident waves has its matrix
                        pipe busy.  SQ_BUSY_CYCLES is reported per SE-level SQ and summed; the ratio below uses
                        SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CU-busy) with CU-busy taken as SQ_BUSY_CYCLES / SEs-per-CU-normalisation
                        folded into `busy_ratio_raw` - read it as a RELATIVE number between kernels of one pass;
      bf16_mops / f32_mops = SQ_INSTS_VALU_MFMA_MOPS_* (units of 512 flops per the counter definition) -> achieved matrix
                        TFLOP/s = mops * 512 / kernel time, against the 2.5 PFLOP/s dense bf16 peak (mfma_frac_of_bf16_peak).
    Kernel time = total duration of the kernel in the kernel-stats pass of the same command scaled to this pass's dispatch count."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for path in (sq_csv, grbm_csv):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        dur[short(r["Name"])] = (float(r["TotalDurationNs"]), int(r["Calls"]))
    with open(out, "w") as f:
        f.write("kernel,dispatches,avg_us,bf16_or_f16_mfma_TFLOPs,frac_of_2500TF_16bit_peak,f32_mfma_TFLOPs,mfma_busy_cycles_per_wave_cycle,"
                "valu_insts_per_dispatch,effective_clock_GHz\n")
        rows = []
        for k, c in acc.items():
            if k not in dur:
                continue
            disp = max(n[k].values())
            t_ns = dur[k][0] / dur[k][1] * disp          # time of `disp` dispatches at the stats pass's average duration
            # (16-bit matrix work: bf16 and, for the 2xFP16 form of round 4, fp16 - the same pipe and the same 2.5 PFLOP/s dense peak)
            bf = (c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)) * 512.0 / t_ns / 1e3
            f32 = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / t_ns / 1e3
            wave = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0    # quad-cycles -> cycles (MI355X_MICROARCH.md cycle-constants table)
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / wave if wave else 0.0
            clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t_ns if t_ns else 0.0
            rows.append((t_ns, f"{k},{disp},{dur[k][0] / dur[k][1] / 1e3:.2f},{bf:.1f},{bf / 2500.0:.3f},{f32:.1f},{busy:.3f},"
                               f"{c.get('SQ_INSTS_VALU', 0.0) / disp:.0f},{clk:.2f}\n"))
        for _, line in sorted(rows, reverse=True):
            f.write(line)


if __name__ == "__main__":
    if sys.argv[1] == "pmc":
        pmc(*sys.argv[2:5])
    elif sys.argv[1] == "sq":
        sq(*sys.argv[2:6])
    else:
        stats(*sys.argv[2:4])
