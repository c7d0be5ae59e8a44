"""Times ssc_lstm_fwd_img (decoder cell + per-image attended-feature table contraction) at the decode shape against
ssc_lstm_fwd on the same rows without the table, and checks it against a torch fp64 evaluation.
  python tools/img_cell_probe.py [images rows_per_image R H]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "style-seqcvae_amd"))
import torch  # noqa: E402

from ssc_runtime import lib as L  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    nimg, rpi, R, H = (a + [50, 100, 36, 1200][len(a):])[:4]
    lib = L.load()
    dev = torch.device("cuda:0")
    G, H4 = nimg * rpi, 4 * H
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    slabs, P = rnd(G, H4) * 0.3, rnd(nimg, R, H4) * 0.3
    nuniq = G // 5
    slabs2 = rnd(nuniq, H4) * 0.3
    slot = torch.randint(0, nuniq, (G,), generator=g).int().to(dev)
    alpha = torch.softmax(rnd(G, R), dim=1).contiguous()
    cprev, b_ih, b_hh = rnd(G, H), rnd(H4) * 0.1, rnd(H4) * 0.1
    c_out, h_out = torch.empty(G, H, device=dev), torch.empty(G, H, device=dev)

    f = L.LstmFwdDesc()
    f.B, f.H = G, H
    f.slabs, f.nslab, f.slab_stride = slabs.data_ptr(), 1, G * H4
    f.slabs2, f.nslab2, f.slab2_stride, f.slab2_rows = slabs2.data_ptr(), 1, G * H4, slot.data_ptr()
    f.b_ih, f.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
    f.c_prev, f.ld_cprev = cprev.data_ptr(), H
    f.c_out, f.ld_cout, f.h_out, f.ld_hout = c_out.data_ptr(), H, h_out.data_ptr(), H
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run_img():
        lib.ssc_lstm_fwd_img(C.byref(f), L.ptr(alpha), R, L.ptr(P), R, rpi, st)

    def run_plain():
        lib.ssc_lstm_fwd(C.byref(f), st)

    def time(fn, n=50):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    run_img()
    torch.cuda.synchronize()
    pre = (slabs.double() + slabs2.double()[slot.long()] + b_ih.double() + b_hh.double()
           + torch.bmm(alpha.double().view(nimg, rpi, R), P.double()).view(G, H4))
    i, fg, gg, o = pre.view(G, 4, H).unbind(1)
    c = torch.sigmoid(fg) * cprev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    print("max |c - ref|", float((c_out.double() - c).abs().max()), " max |h - ref|", float((h_out.double() - h).abs().max()))
    bytes_ = (slabs.numel() + G * H4 // 5 + 3 * G * H + P.numel() + alpha.numel()) * 4
    t_img, t_plain = time(run_img), time(run_plain)
    print(f"G={G} R={R} H={H}: img {t_img:.1f} us ({bytes_ / t_img / 1e6:.2f} TB/s algorithmic)  plain cell {t_plain:.1f} us")


if __name__ == "__main__":
    main()
