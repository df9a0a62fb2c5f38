#!/usr/bin/env python3
"""Time one Conv3d layer (fwd / dgrad / wgrad) through the C-ABI with HIP events.
usage: bench_layer.py N D H W Cin Cout [k] [reps] [stride] [pad] [--dtype f32|bf16] [--conv-math fp32|bf16x6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
argv = sys.argv[1:]
dtype, math = "f32", None
if "--dtype" in argv:
    i = argv.index("--dtype"); dtype = argv[i + 1]; del argv[i:i + 2]
presplit_mode = "--presplit" in argv
if presplit_mode:
    argv.remove("--presplit")
if "--conv-math" in argv:
    i = argv.index("--conv-math"); math = argv[i + 1]; del argv[i:i + 2]
if math:
    mi355seg.set_conv_math(math)
if os.environ.get("MI355SEG_B16_TILES"):
    mi355seg.set_b16_tiles(int(os.environ["MI355SEG_B16_TILES"]))
N, D, H, W, Cin, Cout = [int(v) for v in argv[:6]]
k = int(argv[6]) if len(argv) > 6 else 3
reps = int(argv[7]) if len(argv) > 7 else 10
stride = int(argv[8]) if len(argv) > 8 else 1
pad = (k // 2) if len(argv) <= 9 else int(argv[9])
td = torch.bfloat16 if dtype == "bf16" else torch.float32
x = torch.randn(N, D, H, W, Cin, device="cuda").to(td)
w = torch.randn(Cout, Cin, k, k, k, device="cuda") * 0.05
b = torch.randn(Cout, device="cuda")
Do, Ho, Wo = [(e + 2 * pad - k) // stride + 1 for e in (D, H, W)]
y = torch.randn(N, Do, Ho, Wo, Cout, device="cuda").to(td)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes_bf16" if dtype == "bf16" else "mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, k, stride, pad), x.device)
st = torch.cuda.current_stream().cuda_stream
flops = 2.0 * N * Do * Ho * Wo * k ** 3 * Cin * Cout
sfx = "_" + dtype
def run(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s   [{dtype}, conv math {mi355seg.get_conv_math() if dtype == 'f32' else 'bf16'}] N={N} {D}x{H}x{W} {Cin}->{Cout} k{k} s{stride}")
if dtype == "f32" and L.query("mi355seg_conv_math_takes_amax") and "--measure-amax" not in sys.argv:
    # f16x3: the operand maxima are handed over (as the training path does), so the lines time the convolution kernels alone
    am = torch.zeros(3, device="cuda")
    L.call("mi355seg_amax_f32", x.data_ptr(), Cin, N * D * H * W, Cin, am.data_ptr(), st)
    L.call("mi355seg_amax_f32", w.data_ptr(), w.numel(), 1, w.numel(), am.data_ptr() + 4, st)
    L.call("mi355seg_amax_f32", y.data_ptr(), Cout, N * Do * Ho * Wo, Cout, am.data_ptr() + 8, st)
    ax, aw, ay = am.data_ptr(), am.data_ptr() + 4, am.data_ptr() + 8
    if presplit_mode:
        # TUNE build + MI355SEG_DBG=256 (staging = copy): hand the kernels tensors that ARE split -- per channel quad four fp16 h | four fp16 l
        # of v 2^s in the quad's 16 bytes -- so the timing runs on real operand bits and the results can be compared with the plain run's
        def presplit(t, amax):
            e = (int(amax.view(torch.int32).item()) >> 23) & 0xff
            sc = min(141 - e, 126) if e else 0
            v = t * (2.0 ** sc)
            h = v.half(); l = (v - h.float()).half()
            q = torch.stack((h.reshape(-1, t.shape[-1] // 4, 4), l.reshape(-1, t.shape[-1] // 4, 4)), dim=2).contiguous()
            return q.view(torch.float32).reshape(t.shape)
        # reference: the exact-fp32 MFMA kernels (another code path, untouched by the probe switch)
        mi355seg.set_conv_math("fp32")
        ref = {}
        L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, stride, pad, None, None, ws.data_ptr(), ws.numel(), st)
        ref["fwd"] = y.clone()
        dyt = torch.randn_like(y)
        L.call("mi355seg_amax_f32", dyt.data_ptr(), Cout, N * Do * Ho * Wo, Cout, am.data_ptr() + 8, st)
        L.call("mi355seg_conv3d_dgrad_f32", dyt.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ws.data_ptr(), ws.numel(), st)
        ref["dgrad"] = dx.clone()
        L.call("mi355seg_conv3d_wgrad_f32", dyt.data_ptr(), Cout, x.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ws.data_ptr(), ws.numel(), st)
        ref["wgrad"] = dw.clone()
        mi355seg.set_conv_math(math or "f16x3")
        xs, ds = presplit(x, am[0]), presplit(dyt, am[2])
        torch.cuda.synchronize()
        assert os.environ.get("MI355SEG_DBG") == "256", "--presplit needs the TUNE build and MI355SEG_DBG=256"
        run("fwd", lambda: L.call("mi355seg_conv3d_fwd_ax_f32", xs.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, stride, pad, None, None, ax, aw, ws.data_ptr(), ws.numel(), st))
        print("   fwd   max |presplit - plain| =", float((y - ref["fwd"]).abs().max()), " of", float(ref["fwd"].abs().max()))
        run("dgrad", lambda: L.call("mi355seg_conv3d_dgrad_ax_f32", ds.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ay, aw, ws.data_ptr(), ws.numel(), st))
        print("   dgrad max |presplit - plain| =", float((dx - ref["dgrad"]).abs().max()), " of", float(ref["dgrad"].abs().max()))
        run("wgrad", lambda: L.call("mi355seg_conv3d_wgrad_ax_f32", ds.data_ptr(), Cout, xs.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ay, ax, ws.data_ptr(), ws.numel(), st))
        print("   wgrad max |presplit - plain| =", float((dw - ref["wgrad"]).abs().max()), " of", float(ref["wgrad"].abs().max()))
        sys.exit(0)
    run("fwd", lambda: L.call("mi355seg_conv3d_fwd_ax_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, stride, pad, None, None, ax, aw, ws.data_ptr(), ws.numel(), st))
    L.call("mi355seg_amax_f32", y.data_ptr(), Cout, N * Do * Ho * Wo, Cout, am.data_ptr() + 8, st)
    run("dgrad", lambda: L.call("mi355seg_conv3d_dgrad_ax_f32", y.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ay, aw, ws.data_ptr(), ws.numel(), st))
    if k == 3 and stride == 1:
        # the input gradient with the BatchNorm-backward column sums of the layer in front in its epilogue (conv2 of a double-conv block)
        bnx = torch.randn(N, D, H, W, Cin, device="cuda")
        mean, rstd, gam, bet = torch.zeros(Cin, device="cuda"), torch.ones(Cin, device="cuda"), torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        s12 = torch.zeros(4, Cin, device="cuda")
        wsn = F.workspace(max(ws.numel(), L.query("mi355seg_norm_ws_bytes", N * D * H * W, 1, Cin)), x.device) if hasattr(L, "query") else ws
        try:
            run("dgbn", lambda: L.call("mi355seg_conv3d_dgrad_bnsums_ax_f32", y.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad,
                                       bnx.data_ptr(), Cin, mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1, 0.0,
                                       s12[0].data_ptr(), s12[1].data_ptr(), s12[2].data_ptr(), s12[3].data_ptr(), ay, aw, wsn.data_ptr(), wsn.numel(), st))
        except Exception as e:
            print("dgbn: ", repr(e)[:200])
    run("wgrad", lambda: L.call("mi355seg_conv3d_wgrad_ax_f32", y.data_ptr(), Cout, x.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ay, ax, ws.data_ptr(), ws.numel(), st))
    sys.exit(0)
run("fwd", lambda: L.call("mi355seg_conv3d_fwd" + sfx, x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, stride, pad, None, None, ws.data_ptr(), ws.numel(), st))
run("dgrad", lambda: L.call("mi355seg_conv3d_dgrad" + sfx, y.data_ptr(), Cout, w.data_ptr(), dx.data_ptr(), Cin, N, D, H, W, Cin, Cout, k, stride, pad, ws.data_ptr(), ws.numel(), st))
run("wgrad", lambda: L.call("mi355seg_conv3d_wgrad" + sfx, y.data_ptr(), Cout, x.data_ptr(), Cin, dw.data_ptr(), None, N, D, H, W, Cin, Cout, k, stride, pad, 0, ws.data_ptr(), ws.numel(), st))
