#!/usr/bin/env python3
"""Time one Conv3d layer (fwd / dgrad / wgrad) through the C-ABI with HIP events.
usage: bench_layer.py N D H W Cin Cout [k] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
N, D, H, W, Cin, Cout = [int(v) for v in sys.argv[1:7]]
k = int(sys.argv[7]) if len(sys.argv) > 7 else 3
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 10
stride = int(sys.argv[9]) if len(sys.argv) > 9 else 1
pad = (k // 2) if len(sys.argv) <= 10 else int(sys.argv[10])
x = torch.randn(N, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, k, k, k, device="cuda") * 0.05
b = torch.randn(Cout, device="cuda")
Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
y = torch.empty(N, Do, Ho, Wo, Cout, device="cuda")
dx = torch.empty_like(x)
dw = torch.empty_like(w)
ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, k, stride, pad), x.device)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * N * Do * Ho * Wo * k ** 3 * Cin * Cout
def run(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")
run("fwd", lambda: L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, stride, pad, None, None, ws.data_ptr(), ws.numel(), st))
run("dgrad", lambda: L.call("mi355seg_conv3d_dgrad_f32", y.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ws.data_ptr(), ws.numel(), st))
run("wgrad", lambda: L.call("mi355seg_conv3d_wgrad_f32", y.data_ptr(), Cout, x.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ws.data_ptr(), ws.numel(), st))
