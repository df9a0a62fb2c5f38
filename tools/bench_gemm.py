"""Wall time per call of the token-encoder Linears (UNETR, 216 tokens) through mi355seg.functional.linear beside torch's
(hipBLASLt) F.linear: usage: python tools/bench_gemm.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mi355seg
from mi355seg import functional as F
dev = "cuda"
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(216, 768, device=dev); w = torch.randn(768, 768, device=dev); b = torch.randn(768, device=dev)
w2 = torch.randn(3072, 768, device=dev); b2 = torch.randn(3072, device=dev); x2 = torch.randn(216, 3072, device=dev)
print("linear 216x768->768  ours %.1f us   torch %.1f us" % (bench(lambda: F.linear(x, w, b)), bench(lambda: torch.nn.functional.linear(x, w, b))))
print("linear 216x768->3072 ours %.1f us   torch %.1f us" % (bench(lambda: F.linear(x, w2, b2)), bench(lambda: torch.nn.functional.linear(x, w2, b2))))
w3 = torch.randn(768, 3072, device=dev)
print("linear 216x3072->768 ours %.1f us   torch %.1f us" % (bench(lambda: F.linear(x2, w3, b)), bench(lambda: torch.nn.functional.linear(x2, w3, b))))
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True)
def fb():
    y = F.linear(xg, wg, b); y.backward(x)
    xg.grad = None; wg.grad = None
print("linear fwd+bwd 768->768 ours %.1f us" % bench(fb))
