#!/usr/bin/env python3
"""Static hazard check of the hand-scheduled register prefetch in the GEMM kernels (gemm.hip).

The staged operand loads are inline-asm `global_load_dwordx4` whose destination registers must not be read, copied or
overwritten by any other instruction until a hand-counted `s_waitcnt vmcnt(N)` has retired them: hipcc believes the
value is present as soon as the asm statement has executed and is free to insert a copy (live-range split, phi
resolution) or - past the last use - to reuse the register, and the returning data then lands in the wrong place
(silently wrong products, or a garbage pointer).  Which of these it does depends on register allocation, i.e. on
unrelated edits.  This script re-derives the property from the generated ISA.

Round 2: the analysis follows the control-flow graph.  The kernel is cut into basic blocks at its `.LBB` labels and
branch instructions; the set of staged loads in flight (ordered, oldest first) is carried along every fall-through
edge, every taken branch and every loop back-edge, path by path: a (block, in-flight state) pair is explored once,
so a prefetch that lives across the loop back-edge (PF = 2) is checked in the block it returns to as well.
`s_waitcnt vmcnt(N)` retires all but the N youngest staged loads; compiler-issued vector-memory operations are not
counted (they can only make a wait stricter, so ignoring them errs on the side of reporting).  Any instruction other
than the retiring wait that reads or writes a register with a load in flight is reported, as is a program end
(`s_endpgm`) reached with staged loads outstanding... that last one is legal on the hardware (a wave may end with
loads in flight) and only listed with --strict.

usage: check_staged_loads.py [--strict] <isa.s> [kernel-name-regex]     exit status 1 if a hazard is found
"""
import re
import sys

MAX_INFLIGHT = 48   # longer lists are truncated from the old end (a loop that issues without ever waiting)


def regs(tok):
    """VGPR numbers named by an operand (accumulator registers a[...] are a separate name space: numbered from 1000)"""
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        off = 1000 if m.group(1) == "a" else 0
        return set(range(off + int(m.group(2)), off + int(m.group(3)) + 1))
    m = re.match(r"([va])(\d+)$", tok)
    if m:
        return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))}
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
        cur.append(t.split(";")[0].strip())
    return kernels


PRUNE = False    # --prune: drop paths on which a counted wait vmcnt(N > 0) meets a number of staged loads other than 2N
DBG_OFF = None   # --dbg-off 0xNNN: kernarg offset of a diagnostics word that is 0 in production; a branch taken only when one of
                 # its bits is set (s_load_dword sN, .., off ; s_bitcmp1_b32 sN, k ; [s_cselect/s_and] ; s_cbranch_scc1|vccnz) is not followed


def dbg_branch_never_taken(insts):
    if DBG_OFF is None:
        return False
    reg = None
    seen_bit = False
    for t in insts:
        m = re.match(r"s_load_dword (s\d+), s\[\d+:\d+\], (0x[0-9a-f]+)", t)
        if m and int(m.group(2), 16) == DBG_OFF:
            reg, seen_bit = m.group(1), False
            continue
        if reg and re.match(r"s_bitcmp1_b32 %s, \d+" % reg, t):
            seen_bit = True
            continue
        if seen_bit and re.match(r"s_(cselect_b64|and_b64|waitcnt|andn2_b64)", t):
            continue
        if seen_bit and (t.startswith("s_cbranch_scc1") or t.startswith("s_cbranch_vccnz")):
            return True
        if seen_bit and not t.startswith("s_cbranch"):
            seen_bit = False
    return False


def build_cfg(lines):
    """-> blocks: list of (name, [instructions], [successor indices])"""
    blocks, cur_name, cur = [], "entry", []
    for t in lines:
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            blocks.append([cur_name, cur, None])
            cur_name, cur = m.group(1), []
            continue
        cur.append(t)
        if t.startswith("s_branch") or t.startswith("s_cbranch") or t.startswith("s_endpgm") or t.startswith("s_setpc"):
            blocks.append([cur_name, cur, None])
            cur_name, cur = cur_name + "+", []
    blocks.append([cur_name, cur, None])
    index = {}
    for i, b in enumerate(blocks):
        index.setdefault(b[0], i)
    for i, b in enumerate(blocks):
        succ = []
        last = b[1][-1] if b[1] else ""
        tgt = re.search(r"(\.LBB\w+)", last)
        if last.startswith("s_branch"):
            succ = [index[tgt.group(1)]]
        elif last.startswith("s_cbranch"):
            # s_cbranch_execz only skips a region that would run with no active lane: following the fall-through covers it
            succ = [] if (dbg_branch_never_taken(b[1]) or last.startswith("s_cbranch_execz")) else [index[tgt.group(1)]]
            if i + 1 < len(blocks):
                succ.append(i + 1)
            if len(succ) == 2 and (last.startswith("s_cbranch_vccnz") or last.startswith("s_cbranch_vccz")):
                b.append((last.split()[0], succ[0], succ[1]))   # decided by a known vcc in check()
        elif last.startswith("s_endpgm") or last.startswith("s_setpc"):
            succ = []
        elif i + 1 < len(blocks):
            succ = [i + 1]
        b[2] = succ
    return blocks


def sregs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def run_block(kname, bname, insts, state, bad, strict, consts=None):
    """state: staged loads in flight on entry.  consts (optional dict, updated in place): SGPR pairs known to hold 0 / -1
    (hipcc lowers if/else to a flag: `s_mov_b64 s[a:b], -1 ... s_mov_b64 s[a:b], 0 ... s_andn2_b64 vcc, exec, s[a:b];
    s_cbranch_vccnz`), so that the two arms of one source-level branch are not combined into a path that cannot run.
    consts["vcc"] = True (non-zero) / False (zero) when the block's final vcc is known."""
    inflight = list(state)
    if consts is None:
        consts = {}
    consts.pop("vcc", None)
    for t in insts:
        # ---- scalar constant tracking -------------------------------------------------------------------------------
        if t.startswith("s_") or t.startswith("v_cmp") or "vcc" in t:
            ops = [x.strip() for x in t.split(None, 1)[1].split(",")] if " " in t else []
            m = re.match(r"s_mov_b64 (s\[\d+:\d+\]), (-1|0)$", t)
            m2 = re.match(r"s_(and|andn2)_b64 vcc, exec, (s\[\d+:\d+\])$", t)
            if m:
                for r in sregs(m.group(1)):
                    consts.pop(r, None)
                consts[m.group(1)] = int(m.group(2))
            elif m2:
                v = consts.get(m2.group(2))
                consts.pop("vcc", None)
                if v is not None:
                    consts["vcc"] = (v != 0) if m2.group(1) == "and" else (v == 0)
            elif not t.startswith("s_cbranch") and not t.startswith("s_waitcnt") and not t.startswith("s_barrier") and not t.startswith("s_nop"):
                if ops:
                    dead = sregs(ops[0])
                    for k in [k for k in consts if k != "vcc" and sregs(k) & dead]:
                        consts.pop(k)
                if "vcc" in t or t.startswith("v_cmp"):
                    consts.pop("vcc", None)
        if t.startswith("global_load_dwordx4"):
            ops = [x.strip() for x in t[len("global_load_dwordx4"):].split(",")]
            dst, addr = regs(ops[0]), regs(ops[1])
            live = set().union(*inflight) if inflight else set()
            if (addr | dst) & live:
                bad.add((kname, bname, t, tuple(sorted((addr | dst) & live))))
            inflight.append(frozenset(dst))
            if len(inflight) > MAX_INFLIGHT:
                inflight = inflight[-MAX_INFLIGHT:]
            continue
        if t.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                if PRUNE and n > 0 and inflight and len(inflight) != 2 * n:
                    return None   # a hand-counted wait of a PF = 2 pipeline always sees 2N loads: this path cannot run
                if n == 0:
                    inflight = []
                elif n < len(inflight):
                    inflight = inflight[len(inflight) - n:]
            continue
        if t.startswith("s_endpgm"):
            if strict and inflight:
                bad.add((kname, bname, t, tuple(sorted(set().union(*inflight)))))
            continue
        if t.startswith("s_"):
            continue
        used = set()
        for tk in re.findall(r"[va]\[\d+:\d+\]|\b[va]\d+\b", t):
            used |= regs(tk)
        if inflight and used & set().union(*inflight):
            bad.add((kname, bname, t, tuple(sorted(used & set().union(*inflight)))))
    return tuple(inflight)


def check(path, pat, strict=False):
    bad = set()
    stats = {}
    for kname, lines in parse_kernels(path, pat).items():
        blocks = build_cfg(lines)
        seen = set()
        work = [(0, (), ())]
        carried = 0   # edges along which staged loads were still in flight
        while work:
            bi, state, cst = work.pop()
            if (bi, state, cst) in seen:
                continue
            seen.add((bi, state, cst))
            name, insts, succ = blocks[bi][:3]
            consts = dict(cst)
            found = set()
            out = run_block(kname, name, insts, state, found, strict, consts)
            if out is None:
                continue
            bad |= found
            vcc = consts.pop("vcc", None)
            if len(blocks[bi]) > 3 and vcc is not None:
                kind, taken, fall = blocks[bi][3]
                succ = [taken] if (vcc == (kind == "s_cbranch_vccnz")) else [fall]
            cout = tuple(sorted(consts.items()))
            for s in succ:
                if out:
                    carried += 1
                work.append((s, out, cout))
        stats[kname] = (len(blocks), len(seen), carried)
    return sorted(bad), stats


def scratch_users(path, pat):
    """Kernels (matching pat) with a private segment: a stack object or a spill.  In a kernel with hand-counted vmcnt waits a
    scratch reload is one more vector-memory operation: hipcc waits for it with vmcnt(0), which also drains the staged loads
    (seen: two stores of gathered row numbers sunk into one store through a pointer phi kept a 2-element array in scratch and
    made the grouped weight-gradient launches 20 % slower)."""
    out, name = [], None
    for line in open(path):
        t = line.strip()
        m = re.match(r"\.amdhsa_kernel (\S+)", t)
        if m:
            name = m.group(1)
        m = re.match(r"\.amdhsa_private_segment_fixed_size (\d+)", t)
        if m and name and int(m.group(1)) > 0 and re.search(pat, name):
            out.append((name, int(m.group(1))))
    return out


def main():
    global DBG_OFF, PRUNE
    argv = sys.argv[1:]
    PRUNE = "--prune" in argv
    if "--dbg-off" in argv:
        i = argv.index("--dbg-off")
        DBG_OFF = int(argv[i + 1], 16)
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    strict = "--strict" in sys.argv
    verbose = "--stats" in sys.argv
    path = args[0]
    pat = args[1] if len(args) > 1 else r"gemm_x3w_kernel|gemm_x3b_kernel|gemm_x3_kernel|gemm_kernel"
    bad, stats = check(path, pat, strict)
    for kname, bname, t, rr in bad[:60]:
        print(f"HAZARD {kname[-70:]} {bname}: {t[:90]}   <-- in-flight registers {list(rr[:8])}")
    if verbose:
        for k, (nb, ns, nc) in stats.items():
            print(f"  {k[-80:]}: {nb} blocks, {ns} (block, state) pairs, {nc} edges crossed with loads in flight")
    scr = scratch_users(path, pat)
    for kname, n in scr:
        print(f"SCRATCH {kname[-70:]}: {n} bytes/lane of private memory (stack object or spill) in a hand-scheduled kernel")
    print(f"staged-load hazard check (CFG-aware): {len(stats)} kernel(s), {len(bad)} finding(s), {len(scr)} kernel(s) with scratch")
    return 1 if (bad or scr) else 0


if __name__ == "__main__":
    sys.exit(main())
