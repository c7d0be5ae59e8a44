"""Row-list compaction of the 128x128 GEMM kernels: time of a 5000 x 4800 x 2400 product dense, with device-side row lists at several live-row counts, and dense at the reduced row count (the live-row tile order of gemm.hip makes the three agree).  python tools/compact_probe.py"""
import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+"/style-seqcvae_amd", ROOT+"/tests"]
import torch
from gpuutil import gemm
M,N,K=5000,4800,2400
A=torch.randn(M,K,device="cuda"); B=torch.randn(N,K,device="cuda"); C=torch.empty(M,N,device="cuda")
ws=torch.empty(40*1024*1024,device="cuda")
def run(compact):
    for _ in range(3): gemm([(A,K,B,K,K)],M,N,1,1,C,ws=ws,compact=compact)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gemm([(A,K,B,K,K)],M,N,1,1,C,ws=ws,compact=compact)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/10*1e3
print("dense 5000 rows: %.0f us" % run(None))
for cnt in (5000, 3300, 1000):
    rows=torch.arange(M,dtype=torch.int32,device="cuda")
    c={"m_count": torch.tensor([cnt,0,0,0],dtype=torch.int32,device="cuda"), "a_rows": rows, "c_rows": rows}
    print("row lists, count %d: %.0f us" % (cnt, run(c)))
A2=A[:3300].contiguous(); C2=torch.empty(3300,N,device="cuda")
def run2():
    for _ in range(3): gemm([(A2,K,B,K,K)],3300,N,1,1,C2,ws=ws)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gemm([(A2,K,B,K,K)],3300,N,1,1,C2,ws=ws)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/10*1e3
print("dense 3300 rows: %.0f us" % run2())
