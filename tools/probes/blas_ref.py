"""Practical ceiling check: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the GEMM shapes of the
res4 layers at batch 8 (M = 12512) and batch 1 (M = 1564), float16 and float32.  Plain GEMMs: no gather, no epilogue."""
import time

import torch

dev = torch.device("cuda", 0)


def bench(m, n, k, dtype, reps=20):
    """20 back-to-back GEMMs replayed as one hipGraph (no host launch cost between them), best of 5 replays."""
    a = torch.randn(m, k, device=dev, dtype=dtype)
    b = torch.randn(k, n, device=dev, dtype=dtype)
    c = torch.empty(m, n, device=dev, dtype=dtype)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            torch.matmul(a, b, out=c)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                torch.matmul(a, b, out=c)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best * 1e6, 2.0 * m * n * k / best / 1e12


torch.backends.cuda.matmul.allow_tf32 = False
for dtype in (torch.float16, torch.float32):
    for m in (12512, 1564):
        for (n, k) in ((256, 2304), (256, 1024), (1024, 256), (512, 4608), (2048, 512)):
            us, tf = bench(m, n, k, dtype)
            print("%-8s M=%5d N=%4d K=%4d  %7.1f us  %7.1f TFLOP/s" % (str(dtype).split(".")[1], m, n, k, us, tf))
