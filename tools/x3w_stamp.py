#!/usr/bin/env python3
"""Phase timing inside the wave-specialised GEMM kernel (diagnostics; not part of the product build).

  python tools/x3w_stamp.py build                      # here or on the box: second library with -DSSC_X3W_STAMP (+ ISA gate)
  python tools/x3w_stamp.py run [shape] [wg]           # on the GPU: one product, phase durations of workgroup `wg`

The stamped kernel records the shader clock (s_memtime) of producer wave 4 and consumer wave 0 of ONE workgroup at the
phase boundaries of every k-step (csrc/gemm.hip, SSC_STAMP) plus shader clock and 100 MHz wall clock at kernel entry / exit,
which gives the effective shader frequency of that launch.  shape = kind:M:N:K1+K2+.. as in tools/gemm_probe.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "style-seqcvae_amd")
OUT = os.path.join(PKG, "_stamp")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    import build as B
    os.makedirs(OUT, exist_ok=True)
    flags = B.FLAGS + ["-DSSC_X3W_STAMP"]
    objs = []
    for s in B.SOURCES:
        obj = os.path.join(OUT, s.replace(".hip", ".o"))
        subprocess.check_call([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj])
        objs.append(obj)
    isa = os.path.join(OUT, "gemm.s")
    subprocess.check_call([B._hipcc()] + [f for f in flags if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(B.CSRC, "gemm.hip"), "-o", isa],
                          stderr=subprocess.DEVNULL)
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_staged_loads.py"), isa])
    os.remove(isa)
    if rc and not os.environ.get("SSC_STAMP_ALLOW_SCRATCH"):   # (the stamps cost registers: the 128-VGPR two-workgroup 2xFP16 kernel spills two dwords in this build - stamp its SSC_F16_NPW=8 form)
        raise SystemExit("stamped build fails the staged-load gate")
    lib = os.path.join(OUT, "libssc_hip.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    print(lib)


def run(shape, wg):
    os.environ["SSC_DEBUG"] = "1"
    os.environ["SSC_LIB_PATH"] = os.path.join(OUT, "libssc_hip.so")
    import ctypes as C
    import torch
    from gpuutil import gemm
    from ssc_runtime import lib as L
    lib = L.load()
    setup = lib._cdll.ssc_debug_stamp_setup   # exists in the stamped library only
    setup.argtypes = [C.c_void_p, C.c_int]
    setup.restype = C.c_int
    parts = shape.split(":")
    kind, M, N = parts[0], int(parts[1]), int(parts[2])
    Ks = [int(k) for k in parts[3].split("+")]
    a_kc, b_kc = {"NT": (1, 1), "NN": (1, 0), "TN": (0, 0)}[kind]
    As = [torch.randn((M, K) if a_kc else (K, M), device="cuda") for K in Ks]
    Bs = [torch.randn((N, K) if b_kc else (K, N), device="cuda") for K in Ks]
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(40 * 64 * 4800, 10 * M * N) + 4096, device="cuda")
    segs = [(a, a.stride(0), b, b.stride(0), K) for a, b, K in zip(As, Bs, Ks)]
    buf = torch.zeros(2048, dtype=torch.int64, device="cuda")
    compact = None
    if os.environ.get("SSC_STAMP_GATHER") and kind == "TN":   # weight-gradient form: both operands gathered by k-row lists
        keep = torch.rand(Ks[0]) < 0.71
        rows = torch.nonzero(keep).flatten().to(torch.int32)
        cnt = torch.tensor([rows.numel(), 0, 0, 0], dtype=torch.int32)
        lst = torch.cat([rows, torch.zeros(Ks[0] - rows.numel(), dtype=torch.int32)]).cuda()
        compact = {"k_count": cnt.cuda(), "ka_rows": lst, "kb_rows": lst}
        run.keepalive = compact
    planes = None
    if os.environ.get("SSC_STAMP_PLANES") and kind == "NT":   # 2xFP16 form (SSC_GEMM_F16=1) on pre-split operands
        from gpuutil import split_f16
        planes = [(split_f16(a), split_f16(b)) for a, b in zip(As, Bs)]
    for _ in range(5):
        gemm(segs, M, N, a_kc, b_kc, out, ws=ws, compact=compact, planes=planes)
    torch.cuda.synchronize()
    assert setup(buf.data_ptr(), wg) == 0
    gemm(segs, M, N, a_kc, b_kc, out, ws=ws, compact=compact, planes=planes)
    torch.cuda.synchronize()
    setup(None, -1)
    st = buf.cpu().tolist()
    if wg == -2:   # entry / exit wall clock (100 MHz) of every workgroup of the launch
        ent = [(st[2 * i], st[2 * i + 1]) for i in range(1024) if st[2 * i] > 0]
        t0 = min(e for e, _ in ent)
        import statistics
        print(f"{shape}: {len(ent)} workgroups; first entry -> last exit {(max(x for _, x in ent) - t0) * 0.01:.2f} us")
        print(f"  entry (us after the first): median {statistics.median((e - t0) * 0.01 for e, _ in ent):.2f}  max {max((e - t0) * 0.01 for e, _ in ent):.2f}")
        print(f"  in-kernel time: min {min((x - e) * 0.01 for e, x in ent):.2f}  median {statistics.median((x - e) * 0.01 for e, x in ent):.2f}  max {max((x - e) * 0.01 for e, x in ent):.2f}")
        print(f"  exit (us after the first entry): min {min((x - t0) * 0.01 for _, x in ent):.2f}  median {statistics.median((x - t0) * 0.01 for _, x in ent):.2f}")
        return
    prod, cons = st[:128], st[128:256]
    cyc, wall = prod[124] - prod[126], prod[125] - prod[127]
    if wall <= 0 or cyc <= 0:
        raise SystemExit(f"no stamps from workgroup {wg} (not a 64x256 / 128x128 wave-specialised launch, or wg out of range)")
    ghz = cyc / (wall * 10.0)   # wall clock: 100 MHz
    print(f"{shape} workgroup {wg}: kernel entry -> exit {cyc} shader cycles, {wall * 0.01:.2f} us wall -> {ghz:.2f} GHz shader clock")
    t0 = prod[126]
    us = lambda c: c / (ghz * 1e3)
    print(f"prologue (us since kernel entry): setup done {us(prod[120] - t0):.2f} | first tiles requested {us(prod[121] - t0):.2f} | "
          f"first tile landed {us(prod[122] - t0):.2f}")
    print("producer wave 4  (us since kernel entry: data landed | planes stored | next loads issued | barrier passed)")
    r = 0
    while 4 * r + 3 < 120 and t0 < prod[4 * r] < prod[124]:
        p = prod[4 * r: 4 * r + 4]
        prev = prod[4 * r - 1] if r else t0
        print(f"  k-step {r:2d}: {us(p[0] - t0):7.2f} {us(p[1] - t0):7.2f} {us(p[2] - t0):7.2f} {us(p[3] - t0):7.2f}   "
              f"wait {us(p[0] - prev):5.2f}  split+store {us(p[1] - p[0]):5.2f}  issue {us(p[2] - p[1]):5.2f}  barrier {us(p[3] - p[2]):5.2f}")
        r += 1
    print(f"  exit {us(prod[124] - t0):7.2f}")
    print("consumer wave 0  (start | fragments + MFMAs issued | barrier passed)")
    r = 0
    c0 = cons[126]
    while 4 * r + 2 < 124 and c0 < cons[4 * r] < cons[124]:
        p = cons[4 * r: 4 * r + 3]
        print(f"  k-step {r:2d}: {us(p[0] - c0):7.2f} {us(p[1] - c0):7.2f} {us(p[2] - c0):7.2f}   compute {us(p[1] - p[0]):5.2f}  barrier {us(p[2] - p[1]):5.2f}")
        r += 1
    print(f"  exit {us(cons[124] - c0):7.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "NT:64:4800:2048+1200+1200+1200", int(sys.argv[3]) if len(sys.argv) > 3 else 100)
