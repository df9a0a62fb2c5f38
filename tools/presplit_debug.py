#!/usr/bin/env python3
"""Which quad layout does the DBG=256 staging probe of conv_x3s implement?  (TUNE build, MI355SEG_DBG=256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mi355seg
F = mi355seg.functional
L = mi355seg.lib()
N, D, H, W, Cin, Cout, k = 1, 16, 16, 32, 32, 32, 3
torch.manual_seed(0)
x = torch.randn(N, D, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, k, k, k, device="cuda") * 0.05
b = torch.zeros(Cout, device="cuda")
y = torch.empty(N, D, H, W, Cout, device="cuda")
ws = F.workspace(L.query("mi355seg_conv3d_ws_bytes", N, D, H, W, Cin, Cout, k, 1, 1), x.device)
st = torch.cuda.current_stream().cuda_stream
am = torch.zeros(2, device="cuda")
L.call("mi355seg_amax_f32", x.data_ptr(), Cin, N * D * H * W, Cin, am.data_ptr(), st)
L.call("mi355seg_amax_f32", w.data_ptr(), w.numel(), 1, w.numel(), am.data_ptr() + 4, st)
mi355seg.set_conv_math("fp32")
L.call("mi355seg_conv3d_fwd_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, 1, 1, None, None, ws.data_ptr(), ws.numel(), st)
ref = y.clone()
mi355seg.set_conv_math("f16x3")
e = (int(am[0].view(torch.int32).item()) >> 23) & 0xff
sc = min(141 - e, 126) if e else 0
v = x * (2.0 ** sc)
h = v.half(); l = (v - h.float()).half()
hq, lq = h.reshape(-1, Cin // 4, 4), l.reshape(-1, Cin // 4, 4)
layouts = {
    "A [h0 h1 h2 h3 | l0 l1 l2 l3]": torch.stack((hq, lq), dim=2),
    "B [h0 l0 h1 l1 | h2 l2 h3 l3]": torch.stack((hq, lq), dim=3),
    "C [h0 h1 l0 l1 | h2 h3 l2 l3]": torch.stack((hq.reshape(-1, Cin // 4, 2, 2), lq.reshape(-1, Cin // 4, 2, 2)), dim=3),
}
print("dbg env", os.environ.get("MI355SEG_DBG"), "scale exp", sc, "ref max", float(ref.abs().max()))
# D: values on the fp16 grid (l = 0): only the h mapping matters
xd = (h.float() / (2.0 ** sc))
mi355seg.set_conv_math("fp32")
L.call("mi355seg_conv3d_fwd_f32", xd.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, 1, 1, None, None, ws.data_ptr(), ws.numel(), st)
refd = y.clone()
mi355seg.set_conv_math("f16x3")
zq = torch.zeros_like(hq)
for name, q in {"D [h | 0]": torch.stack((hq, zq), dim=2), "E [0 | h]": torch.stack((zq, hq), dim=2), "F [h | h]": torch.stack((hq, hq), dim=2)}.items():
    xs = q.contiguous().view(torch.float32).reshape(x.shape)
    L.call("mi355seg_conv3d_fwd_ax_f32", xs.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, 1, 1, None, None, am.data_ptr(), am.data_ptr() + 4, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    print(f"{name}: max |probe - ref(h only)| = {float((y - refd).abs().max()):.3e}   max |probe - 2 ref| = {float((y - 2 * refd).abs().max()):.3e}  max|probe| {float(y.abs().max()):.3e}")
for name, q in layouts.items():
    xs = q.contiguous().view(torch.float32).reshape(x.shape)
    L.call("mi355seg_conv3d_fwd_ax_f32", xs.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, 1, 1, None, None, am.data_ptr(), am.data_ptr() + 4, ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    print(f"{name}: max |probe - fp32 ref| = {float((y - ref).abs().max()):.3e}")
L.call("mi355seg_conv3d_fwd_ax_f32", x.data_ptr(), Cin, w.data_ptr(), b.data_ptr(), y.data_ptr(), Cout, N, D, H, W, Cin, Cout, k, 1, 1, None, None, am.data_ptr(), am.data_ptr() + 4, ws.data_ptr(), ws.numel(), st)
torch.cuda.synchronize()
print(f"plain fp32 tensor through the same call: {float((y - ref).abs().max()):.3e}")
