#!/usr/bin/env python3
"""Time the experimental bf16-operand conv against the fp32 MFMA conv.  usage: bench_bf16.py N D H W Cin Cout [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
N, D, H, W, Cin, Cout = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
x = torch.randn(N, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
y = torch.empty(N, D, H, W, Cout, device="cuda")
ws = F.workspace(max(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, 3, 1, 1), L.query("mi355seg_conv3d_bf16mma_ws_bytes", Cin, Cout)), x.device)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * N * D * H * W * 27 * Cin * Cout
def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:10s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s (fp32-equivalent)")
run("fp32 mfma", lambda: L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, D, H, W, Cin, Cout, 3, 1, 1, None, None, ws.data_ptr(), ws.numel(), st))
ws6 = torch.empty(L.query("mi355seg_conv3d_bf16x6_ws_bytes", Cin, Cout), dtype=torch.uint8, device="cuda")
run("bf16x6", lambda: L.call("mi355seg_conv3d_bf16x6_f32", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, D, H, W, Cin, Cout, 0, ws6.data_ptr(), ws6.numel(), st))
run("bf16 mfma", lambda: L.call("mi355seg_conv3d_bf16mma_f32", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, D, H, W, Cin, Cout, 0, ws.data_ptr(), ws.numel(), st))
