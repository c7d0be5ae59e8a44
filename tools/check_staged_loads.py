#!/usr/bin/env python3
"""Static hazard check of the hand-scheduled register prefetch in the GEMM kernels (gemm.hip).

The staged operand loads are inline-asm `global_load_dwordx4` whose destination registers must not be read, copied or
overwritten by any other instruction until a hand-counted `s_waitcnt vmcnt(N)` has retired them: hipcc believes the
value is present as soon as the asm statement has executed and is free to insert a copy (live-range split, phi
resolution) or - past the last use - to reuse the register, and the returning data then lands in the wrong place
(silently wrong products, or a garbage pointer).  Which of these it does depends on register allocation, i.e. on
unrelated edits.  This script re-derives the property from the generated ISA:

  per basic block, the registers with a staged load in flight (+ issue order); `s_waitcnt vmcnt(N)` retires all but
  the N youngest staged loads; any other instruction touching an in-flight register is reported.  Compiler-issued
  vector memory loads are ignored for the count (they only make a wait stricter).  A copy of an in-flight register
  reads the register's OLD contents every time (the copy issues within cycles, the load takes hundreds), so this
  failure is deterministic; the numerics tests catch it on every path they exercise, this check on all paths.

usage: check_staged_loads.py <isa.s> [kernel-name-regex]     exit status 1 if a hazard is found
"""
import re
import sys


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def parse_kernels(path, pat):
    kernels, cur, name = {}, None, None
    for line in open(path):
        t = line.strip()
        m = re.match(r"^(_Z\S+):\s*(;.*)?$", t)
        if m:
            name = m.group(1)
            cur = [] if re.search(pat, name) else None
            if cur is not None:
                kernels[name] = cur
            continue
        if cur is None or not t or t.startswith(";"):
            continue
        if t.startswith(".") and not t.startswith(".LBB"):
            if t.startswith(".end_amdhsa_kernel") or t.startswith(".section"):
                cur = None
            continue
        cur.append(t)
    return kernels


def check(path, pat):
    """Basic-block-local and therefore precise: a copy hipcc inserts for a live-range split or a phi lands in the block
    that issued the load, before the next hand-counted wait."""
    bad = []
    for kname, lines in parse_kernels(path, pat).items():
        inflight = []   # staged loads in flight in this basic block, oldest first (each = its 4 destination registers)
        bname = "entry"
        for t in lines:
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                bname, inflight = m.group(1), []
                continue
            if t.startswith("global_load_dwordx4"):
                ops = [x.strip() for x in t[len("global_load_dwordx4"):].split(",")]
                dst, addr = regs(ops[0]), regs(ops[1])
                live = set().union(*inflight) if inflight else set()
                if (addr | dst) & live:
                    bad.append((kname, bname, t, sorted((addr | dst) & live)))
                inflight.append(frozenset(dst))
                continue
            if t.startswith("s_waitcnt") and "vmcnt" in t:
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                inflight = inflight[len(inflight) - n:] if n and n < len(inflight) else ([] if n == 0 else inflight)
                continue
            if t.startswith("s_"):
                continue
            used = set()
            for tk in re.findall(r"[va]\[\d+:\d+\]|\bv\d+\b", t):
                used |= regs(tk)
            live = set().union(*inflight) if inflight else set()
            if used & live:
                bad.append((kname, bname, t, sorted(used & live)))
    return bad


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else r"gemm_x3w_kernel|gemm_x3b_kernel|gemm_x3_kernel|gemm_kernel"
    bad = check(path, pat)
    for kname, bname, t, rr in bad[:40]:
        print(f"HAZARD {kname[-70:]} {bname}: {t[:90]}   <-- in-flight registers {rr[:8]}")
    print(f"staged-load hazard check: {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
