import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "style-seqcvae_amd", "libssc_hip.so"))
a = (C.c_int * 4)()
print("rc", lib.ssc_debug_gemm_occupancy(a), "x3 lds2:", a[0], "x3 lds1:", a[1], "f32 64x64 pf4:", a[2], "f32 NN 64x128:", a[3])
print("rc", lib.ssc_debug_gemm_occupancy_x3b(a), "x3b NT:", a[0], "NN:", a[1], "TN:", a[2], "TN gather:", a[3])
