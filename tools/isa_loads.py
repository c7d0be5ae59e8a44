"""Per kernel of a HIP source: how its global loads are waited for in the generated gfx950 ISA.  hipcc turns `p ? *p : 0` (a load
used on one side of a select) into a branch with `s_waitcnt vmcnt(0)` at the join: a lane's loads then go out one at a time.
Prints, per kernel, the number of global loads, the number of full waits that are followed by further global loads (serialisation
points) and the longest run of loads issued back to back.
  python tools/isa_loads.py style-seqcvae_amd/csrc/pointwise.hip [kernel-name-substring]  [--dump]
"""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa_of(src):
    spec = importlib.util.spec_from_file_location("ssc_build", os.path.join(ROOT, "style-seqcvae_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    cmd = [m._hipcc()] + [f for f in m.FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dump = "--dump" in sys.argv
    text = isa_of(args[0])
    pat = args[1] if len(args) > 1 else ""
    for name in re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M):
        if pat not in name:
            continue
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")]
        ops = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        mem = [l for l in ops if re.match(r"(global_load|buffer_load|s_waitcnt|global_store|s_barrier|v_mfma|ds_)", l)]
        loads = [i for i, l in enumerate(mem) if l.startswith(("global_load", "buffer_load"))]
        serial = 0
        for i, l in enumerate(mem):
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l and any(j > i for j in loads) and any(j < i for j in loads):
                serial += 1
        run = best = 0
        for l in mem:
            if l.startswith(("global_load", "buffer_load")):
                run += 1
                best = max(best, run)
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                run = 0
        vg = re.search(r"\.amdhsa_kernel " + re.escape(name) + r".*?next_free_vgpr (\d+)", text, re.S).group(1)
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
        print(f"{short:100s} loads {len(loads):3d}  full waits before more loads {serial:3d}  longest run {best:3d}  vgpr {vg}")
        if dump:
            print("\n".join("    " + l[:100] for l in mem))


if __name__ == "__main__":
    main()
