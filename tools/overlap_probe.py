#!/usr/bin/env python3
"""Does a layer's weight gradient run BESIDE its input gradient?  Both read dy and are independent; conv_wgrad_lowp (one 8-wave workgroup per
CU, 89.6 KB of LDS, 248 registers per SIMD lane pair) and conv_x3s (69.6 KB, 256 registers) fit one CU together.
usage: overlap_probe.py N D H W Cin Cout [reps]   -- serial on one stream vs the weight gradient on a second stream (launched first)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
N, D, H, W, Cin, Cout = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
k, stride, pad = 3, 1, 1
dev = torch.device("cuda")
x = torch.randn(N, D, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, k, k, k, device=dev) * 0.05
dy = torch.randn(N, D, H, W, Cout, device=dev)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
need = L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, k, stride, pad)
ws1 = torch.empty(need, dtype=torch.uint8, device=dev)
ws2 = torch.empty(need, dtype=torch.uint8, device=dev)
am = torch.zeros(3, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
st = main.cuda_stream
L.call("mi355seg_amax_f32", x.data_ptr(), Cin, N * D * H * W, Cin, am.data_ptr(), st)
L.call("mi355seg_amax_f32", w.data_ptr(), w.numel(), 1, w.numel(), am.data_ptr() + 4, st)
L.call("mi355seg_amax_f32", dy.data_ptr(), Cout, N * D * H * W, Cout, am.data_ptr() + 8, st)
ax, aw, ay = am.data_ptr(), am.data_ptr() + 4, am.data_ptr() + 8
flops = 2.0 * N * D * H * W * 27 * Cin * Cout


def dgrad(s, ws):
    L.call("mi355seg_conv3d_dgrad_ax_f32", dy.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ay, aw, ws.data_ptr(), ws.numel(), s.cuda_stream)


def wgrad(s, ws):
    L.call("mi355seg_conv3d_wgrad_ax_f32", dy.data_ptr(), Cout, x.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ay, ax, ws.data_ptr(), ws.numel(), s.cuda_stream)


def serial():
    dgrad(main, ws1); wgrad(main, ws1)


def overlapped(wfirst):
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    if wfirst:
        wgrad(side, ws2); dgrad(main, ws1)
    else:
        dgrad(main, ws1); wgrad(side, ws2)
    main.wait_stream(side)


def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps): fn()
    e1.record(main); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:28s} {ms:8.3f} ms   {2 * flops / ms / 1e9:7.1f} TFLOP/s (both)   N={N} {D}x{H}x{W} {Cin}->{Cout}")
    return ms


a = run("dgrad alone", lambda: dgrad(main, ws1))
b = run("wgrad alone", lambda: wgrad(main, ws1))
c = run("serial (one stream)", serial)
d = run("two streams, wgrad first", lambda: overlapped(True))
e = run("two streams, dgrad first", lambda: overlapped(False))
dw1 = dw.clone(); dx1 = dx.clone()
serial(); torch.cuda.synchronize()
print(f"   alone sum {a + b:.3f}  serial {c:.3f}  overlapped {d:.3f} / {e:.3f}  -> x{c / min(d, e):.3f};  results equal: {torch.equal(dw1, dw) and torch.equal(dx1, dx)}")
