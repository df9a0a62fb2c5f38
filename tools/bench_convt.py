#!/usr/bin/env python3
"""Time one ConvTranspose3d(k2, s2) layer (fwd / dgrad / wgrad) through the C-ABI with HIP events; the fine tensor may be a
channel slice of a wider buffer (the concat-free decoder writes the up-convolution into its half of the concat buffer).
usage: bench_convt.py N D H W Cin Cout [reps] [ld_fine] [--dtype f32|bf16] [--conv-math fp32|bf16x6] [--only fwd|dgrad|wgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
argv = sys.argv[1:]
dtype, math, only = "f32", None, None
if "--dtype" in argv:
    i = argv.index("--dtype"); dtype = argv[i + 1]; del argv[i:i + 2]
if "--conv-math" in argv:
    i = argv.index("--conv-math"); math = argv[i + 1]; del argv[i:i + 2]
if "--only" in argv:
    i = argv.index("--only"); only = argv[i + 1]; del argv[i:i + 2]
if math:
    mi355seg.set_conv_math(math)
N, D, H, W, Cin, Cout = [int(v) for v in argv[:6]]
reps = int(argv[6]) if len(argv) > 6 else 20
ldf = int(argv[7]) if len(argv) > 7 else Cout
td = torch.bfloat16 if dtype == "bf16" else torch.float32
eb = 2 if dtype == "bf16" else 4
x = torch.randn(N, D, H, W, Cin, device="cuda").to(td)
w = torch.randn(Cin, Cout, 2, 2, 2, device="cuda") * 0.05
b = torch.randn(Cout, device="cuda")
y = torch.randn(N, 2 * D, 2 * H, 2 * W, ldf, device="cuda").to(td)
dx = torch.empty_like(x)
dw, db = torch.empty_like(w), torch.empty_like(b)
ws = F.workspace(L.query("mi355seg_convt3d_k2s2_ws_bytes", N, D, H, W, Cin, Cout), x.device)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * N * D * H * W * 8 * Cin * Cout
nbytes = eb * N * D * H * W * (Cin + 8 * Cout)
sfx = "_" + dtype
def run(name, fn):
    if only and only != name: return
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:6s} {ms:8.4f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  {nbytes / ms / 1e9:6.2f} TB/s   [{dtype}] N={N} {D}x{H}x{W} {Cin}->{Cout} ld_fine={ldf} (pack kernel included)")
run("fwd", lambda: L.call("mi355seg_convt3d_k2s2_fwd" + sfx, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), ldf, N, D, H, W, Cin, Cout, ws.data_ptr(), ws.numel(), st))
run("dgrad", lambda: L.call("mi355seg_convt3d_k2s2_dgrad" + sfx, y.data_ptr(), ldf, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, ws.data_ptr(), ws.numel(), st))
run("wgrad", lambda: L.call("mi355seg_convt3d_k2s2_wgrad" + sfx, y.data_ptr(), ldf, x.data_ptr(), Cin, dw.data_ptr(), db.data_ptr(), N, D, H, W, Cin, Cout, ws.data_ptr(), ws.numel(), st))
