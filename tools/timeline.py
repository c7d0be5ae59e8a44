#!/usr/bin/env python3
"""Timeline of one train step from a rocprofv3 kernel trace (kernel_trace.csv): for the LAST complete step found, every
launch with its start offset, duration and the gap to its predecessor; then totals per kernel of busy time and of the gaps
that precede it.  usage: python tools/timeline.py <..._kernel_trace.csv> [first-kernel-substring] [max rows] [skip]
(skip: take the step that many occurrences before the last complete one)"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/profiles")
from summarize import short  # noqa: E402


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "prep_tokens_kernel"
    maxrows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    if len(starts) < 2:
        print("need two occurrences of", first)
        return
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    if len(starts) < 2 + skip:
        print("need", 2 + skip, "occurrences of", first)
        return
    lo, hi = starts[-2 - skip], starts[-1 - skip]
    step = rows[lo:hi]
    t0 = step[0][0]
    print(f"step: {len(step)} launches, {(step[-1][1] - t0) / 1e3:.1f} us from first start to last end")
    busy = collections.Counter(); gaps = collections.Counter(); cnt = collections.Counter()
    prev_end = None
    for i, (s, e, n) in enumerate(step):
        k = short(n)
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        if i < maxrows:
            print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {k}")
        busy[k] += (e - s) / 1e3; gaps[k] += max(gap, 0.0); cnt[k] += 1
        prev_end = max(prev_end, e) if prev_end is not None else e
    print("\nkernel, calls, busy_us, gap_before_us")
    for k, v in busy.most_common():
        print(f"{k:48s} {cnt[k]:5d} {v:9.1f} {gaps[k]:9.1f}")
    print(f"{'TOTAL':48s} {sum(cnt.values()):5d} {sum(busy.values()):9.1f} {sum(gaps.values()):9.1f}")


if __name__ == "__main__":
    main()
