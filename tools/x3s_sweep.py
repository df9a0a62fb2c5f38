#!/usr/bin/env python3
"""Compare the bf16x6 convolution (conv_x3s.hip / igemm MATH_X3) with the exact-fp32 MFMA path on the same inputs, forward
(with bias and BatchNorm statistics) and input gradient, over a list of shapes.  Prints the worst relative deviation per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
SHAPES = [(1, 32, 32, 32, 32, 32), (1, 16, 16, 16, 32, 64), (1, 16, 16, 16, 64, 64), (1, 16, 16, 16, 64, 32), (2, 16, 16, 16, 128, 128),
          (1, 8, 12, 16, 32, 32), (1, 9, 7, 20, 32, 64), (1, 16, 16, 48, 64, 32), (2, 5, 6, 17, 32, 32), (1, 64, 64, 64, 32, 32)]
st = lambda: torch.cuda.current_stream().cuda_stream
bad = 0
for (N, D, H, W, Cin, Cout) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(N, D, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda")
    dy = torch.randn(N, D, H, W, Cout, device="cuda")
    ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, 3, 1, 1), x.device)
    res = {}
    for math in ("fp32", "bf16x6"):
        mi355seg.set_conv_math(math)
        y = torch.zeros(N, D, H, W, Cout, device="cuda")
        dx = torch.zeros(N, D, H, W, Cin, device="cuda")
        ssum = torch.zeros(Cout, dtype=torch.float64, device="cuda"); ssq = torch.zeros_like(ssum)
        L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, 3, 1, 1,
               ssum.data_ptr(), ssq.data_ptr(), ws.data_ptr(), ws.numel(), st())
        L.call("mi355seg_conv3d_dgrad_f32", dy.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, 3, 1, 1, ws.data_ptr(), ws.numel(), st())
        torch.cuda.synchronize()
        res[math] = (y, dx, ssum, ssq)
    out = []
    for i, nm in enumerate(("y", "dx", "sum", "sq")):
        a, c = res["fp32"][i].double(), res["bf16x6"][i].double()
        out.append(float((a - c).abs().max() / a.abs().max().clamp_min(1e-30)))
    flag = "" if max(out) < 2e-5 else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"N={N} {D}x{H}x{W} {Cin}->{Cout}: rel dev y {out[0]:.2e} dx {out[1]:.2e} sum {out[2]:.2e} sq {out[3]:.2e}{flag}")
sys.exit(1 if bad else 0)
