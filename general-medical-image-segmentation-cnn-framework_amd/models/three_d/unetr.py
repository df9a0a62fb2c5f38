"""UNETR on the MI355X kernels -- drop-in for the reference's models/three_d/unetr.py.

Interface contract kept from the reference: the constructor ``UNETR(img_shape, input_dim, output_dim, embed_dim,
patch_size, num_heads, dropout)`` (unetr.py:195), the same 338 ``state_dict`` keys (SURVEY.md appendix D -- which
fixes the class nesting ``decoderN.i.block.j.block``) and the forward semantics of unetr.py:277-294.

How it runs here: the ViT encoder (12 x [LayerNorm, 12-head attention, FFN 768->2048->768 ReLU], taps after layers
3/6/9/12; ``encoder_norm`` is constructed but never applied, as in the reference) uses the strided batched MFMA
GEMM + LayerNorm + softmax kernels.  A token tensor [N, P, E] IS the channel-last volume [N, p0, p1, p2, E] that
the reference builds with ``transpose(-1, -2).view(...)`` (unetr.py:280-283), so the hand-over to the conv decoder
is a free view.  The decoder reuses the U-Net conv / ConvT / fused conv+BN+ReLU kernels; its stages are declared
in ``_DECODER_PLAN`` below instead of being spelled out one by one.
"""
import torch
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, Dropout, LayerNorm, Linear, ReLU

_FFN_WIDTH = 2048
_DEPTH = 12
_TAPS = (3, 6, 9, 12)


# ------------------------------------------------------------------------------------------- decoder bricks
class _Holder(nn.Module):
    """A module with a single child called ``block`` (the nesting the reference's state_dict keys encode)."""

    def __init__(self, child):
        super().__init__()
        self.block = child

    def forward(self, x):
        return self.block(x)


class SingleDeconv3DBlock(_Holder):
    """ConvT k2 s2 (unetr.py:9-16)."""

    def __init__(self, cin, cout):
        super().__init__(ConvTranspose3d(cin, cout, kernel_size=2, stride=2))


class SingleConv3DBlock(_Holder):
    """'same' conv of odd kernel size (unetr.py:19-27)."""

    def __init__(self, cin, cout, ksize):
        super().__init__(Conv3d(cin, cout, kernel_size=ksize, stride=1, padding=(ksize - 1) // 2))


class _FusedTail(nn.Sequential):
    """[..., SingleConv3DBlock, BatchNorm3d, ReLU]: the last three run as ONE conv + statistics + BN + ReLU pass."""

    def forward(self, x):
        *head, conv, bn, _relu = self.children()
        for m in head:
            x = m(x)
        return F.conv_bn_act(x, conv.block, bn, F.ACT_RELU)


class Conv3DBlock(_Holder):
    """conv -> BN -> ReLU (unetr.py:30-41)."""

    def __init__(self, cin, cout, ksize=3):
        super().__init__(_FusedTail(SingleConv3DBlock(cin, cout, ksize), BatchNorm3d(cout), ReLU(True)))


class Deconv3DBlock(_Holder):
    """ConvT k2 s2 -> conv -> BN -> ReLU (unetr.py:44-56)."""

    def __init__(self, cin, cout, ksize=3):
        super().__init__(_FusedTail(SingleDeconv3DBlock(cin, cout), SingleConv3DBlock(cout, cout, ksize),
                                    BatchNorm3d(cout), ReLU(True)))


_BRICKS = {"conv": Conv3DBlock, "deconv": Deconv3DBlock, "up": SingleDeconv3DBlock,
           "head": lambda cin, cout: SingleConv3DBlock(cin, cout, 1)}


def _chain(spec, subst):
    """spec: ((brick, cin, cout), ...) with the placeholders 'E' / 'in' / 'out' resolved through ``subst``."""
    mods = [_BRICKS[kind](subst.get(a, a), subst.get(b, b)) for kind, a, b in spec]
    return mods[0] if len(mods) == 1 else nn.Sequential(*mods)


# attribute -> stages, in the reference's registration order (unetr.py:214-275)
_DECODER_PLAN = (
    ("decoder0", (("conv", "in", 32), ("conv", 32, 64))),
    ("decoder3", (("deconv", "E", 512), ("deconv", 512, 256), ("deconv", 256, 128))),
    ("decoder6", (("deconv", "E", 512), ("deconv", 512, 256))),
    ("decoder9", (("deconv", "E", 512),)),
    ("decoder12_upsampler", (("up", "E", 512),)),
    ("decoder9_upsampler", (("conv", 1024, 512), ("conv", 512, 512), ("conv", 512, 512), ("up", 512, 256))),
    ("decoder6_upsampler", (("conv", 512, 256), ("conv", 256, 256), ("up", 256, 128))),
    ("decoder3_upsampler", (("conv", 256, 128), ("conv", 128, 128), ("up", 128, 64))),
    ("decoder0_header", (("conv", 128, 64), ("conv", 64, 64), ("head", 64, "out"))),
)


# ------------------------------------------------------------------------------------------- ViT encoder
def _token_sum(a, b):
    """a + b on token tensors [N, P, E] (the residual adds of unetr.py:160,166) through the fused add kernel."""
    n, p, e = a.shape
    as5d = lambda t: t.reshape(n, 1, 1, p, e)
    return F.activation(as5d(a), F.ACT_NONE, residual=as5d(b)).view(n, p, e)


class SelfAttention(nn.Module):
    """Multi-head attention with separate q/k/v/out projections (unetr.py:59-110)."""

    def __init__(self, num_heads, embed_dim, dropout):
        super().__init__()
        self.num_attention_heads = num_heads
        self.attention_head_size = embed_dim // num_heads
        self.all_head_size = self.attention_head_size * num_heads
        for name in ("query", "key", "value"):
            setattr(self, name, Linear(embed_dim, self.all_head_size))
        self.out = Linear(embed_dim, embed_dim)
        self.attn_dropout, self.proj_dropout = Dropout(dropout), Dropout(dropout)
        self.softmax = nn.Softmax(dim=-1)            # kept for interface parity; the fused kernel does the work
        self.vis = False
        self._seat_qkv()

    def _seat_qkv(self):
        """Seat query / key / value .weight and .bias as consecutive slices of one buffer each: the fused [3E, E] projection then exists
        in place (no torch.cat per step) while the module keeps the reference's six parameters and state_dict keys (unetr.py:66-68).
        load_state_dict copies into the slices; Module.to() / .cuda() re-seats them (``_apply`` below)."""
        ws = [self.query.weight, self.key.weight, self.value.weight]
        bs = [self.query.bias, self.key.bias, self.value.bias]
        with torch.no_grad():
            fw, fb = torch.cat([w.detach() for w in ws], dim=0), torch.cat([b.detach() for b in bs], dim=0)
            rows = ws[0].shape[0]
            for i, (w, b) in enumerate(zip(ws, bs)):
                w.data, b.data = fw[i * rows:(i + 1) * rows], fb[i * rows:(i + 1) * rows]

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self._seat_qkv()
        return self

    def forward(self, hidden_states, residual=None):
        """``residual``: the block's residual stream -- the result is then residual + proj_dropout(out(attention)) (unetr.py:98-100,160),
        the dropout product and the sum in the out-projection's GEMM epilogue."""
        # the three projections as ONE GEMM on the concatenated weights (the parameters stay separate: state_dict keys query / key /
        # value of unetr.py:66-68; the concat's backward hands each its rows of the fused weight gradient)
        params = (self.query.weight, self.key.weight, self.value.weight, self.query.bias, self.key.bias, self.value.bias)
        if F.qkv_params_are_fused(*params):          # the parameters ARE the rows of the fused weight (``_seat_qkv``)
            qkv = F.linear_qkv(hidden_states, *params)
        else:
            w = torch.cat(params[:3], dim=0)
            b = torch.cat(params[3:], dim=0)
            qkv = F.linear(hidden_states, w, b)
        mask = None
        if self.training and self.attn_dropout.p > 0.0:
            n, p = qkv.shape[0], qkv.shape[1]
            mask = self.attn_dropout.draw((n, self.num_attention_heads, p, p), qkv.device)
        mixed = F.attention_qkv(qkv, self.num_attention_heads, mask)
        if residual is None:
            return self.proj_dropout(self.out(mixed)), None
        return self.out(mixed, mask=self.proj_dropout.mask_for(tuple(mixed.shape[:-1]) + (self.out.out_features,), mixed.device), residual=residual), None


class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(relu(w_1 x))) (unetr.py:127-138); the 786 default is the reference's (never used) typo."""

    def __init__(self, d_model=786, d_ff=_FFN_WIDTH, dropout=0.1):
        super().__init__()
        self.w_1, self.w_2 = Linear(d_model, d_ff), Linear(d_ff, d_model)
        self.dropout = Dropout(dropout)

    def forward(self, x, residual=None):
        # bias + ReLU + the dropout product in w_1's GEMM epilogue; bias + the block's residual sum (unetr.py:166) in w_2's
        hidden = self.w_1.forward_relu(x, mask=self.dropout.mask_for(tuple(x.shape[:-1]) + (self.w_1.out_features,), x.device))
        return self.w_2(hidden, residual=residual)


def _patch_count(cube, patch):
    return int((cube[0] * cube[1] * cube[2]) / (patch * patch * patch))


class Embeddings(nn.Module):
    """Patch embedding (conv kernel = stride = patch) + learned positions + dropout (unetr.py:141-156)."""

    def __init__(self, input_dim, embed_dim, cube_size, patch_size, dropout):
        super().__init__()
        self.n_patches, self.patch_size, self.embed_dim = _patch_count(cube_size, patch_size), patch_size, embed_dim
        self.patch_embeddings = Conv3d(input_dim, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, embed_dim))
        self.dropout = Dropout(dropout)

    def forward(self, x):
        """x: channel-last [N, D, H, W, Cin] -> tokens [N, P, E]."""
        tokens = self.patch_embeddings(x)
        batch = tokens.shape[0]
        tokens = tokens.reshape(batch, self.n_patches, self.embed_dim)
        return self.dropout(_token_sum(tokens, self.position_embeddings.expand(batch, -1, -1)))


class TransformerBlock(nn.Module):
    """Pre-norm block: x + attn(LN x), then x + ffn(LN x) (unetr.py:159-192)."""

    def __init__(self, embed_dim, num_heads, dropout, cube_size, patch_size):
        super().__init__()
        self.attention_norm = LayerNorm(embed_dim, eps=1e-6)
        self.mlp_norm = LayerNorm(embed_dim, eps=1e-6)
        self.mlp_dim = _patch_count(cube_size, patch_size)
        self.mlp = PositionwiseFeedForward(embed_dim, _FFN_WIDTH)     # FFN dropout stays at its 0.1 default, as upstream
        self.attn = SelfAttention(num_heads, embed_dim, dropout)

    def forward(self, x):
        # x + f(LN(x)) twice (unetr.py:159-166): the norm node hands x on unchanged so that its backward kernel sums both gradients of x
        n, x = self.attention_norm.forward_fork(x)
        x, weights = self.attn(n, residual=x)
        n, x = self.mlp_norm.forward_fork(x)
        return self.mlp(n, residual=x), weights


class Transformer(nn.Module):
    def __init__(self, input_dim, embed_dim, cube_size, patch_size, num_heads, num_layers, dropout, extract_layers):
        super().__init__()
        self.embeddings = Embeddings(input_dim, embed_dim, cube_size, patch_size, dropout)
        self.layer = nn.ModuleList(TransformerBlock(embed_dim, num_heads, dropout, cube_size, patch_size)
                                   for _ in range(num_layers))
        self.encoder_norm = LayerNorm(embed_dim, eps=1e-6)            # constructed but never applied (unetr.py:176)
        self.extract_layers = extract_layers

    def forward(self, x):
        h, taps = self.embeddings(x), []
        for index, block in enumerate(self.layer, start=1):
            h, _ = block(h)
            if index in self.extract_layers:
                taps.append(h)
        return taps


# ------------------------------------------------------------------------------------------- the network
class UNETR(nn.Module):
    def __init__(self, img_shape=(128, 128, 128), input_dim=4, output_dim=3, embed_dim=768, patch_size=16, num_heads=12, dropout=0.1):
        super().__init__()
        self.input_dim, self.output_dim, self.embed_dim = input_dim, output_dim, embed_dim
        self.img_shape, self.patch_size, self.num_heads, self.dropout = img_shape, patch_size, num_heads, dropout
        self.num_layers, self.ext_layers = _DEPTH, list(_TAPS)
        self.patch_dim = [int(extent / patch_size) for extent in img_shape]
        self.transformer = Transformer(input_dim, embed_dim, img_shape, patch_size, num_heads, _DEPTH, dropout, self.ext_layers)
        subst = {"E": embed_dim, "in": input_dim, "out": output_dim}
        for attr, spec in _DECODER_PLAN:
            setattr(self, attr, _chain(spec, subst))

    def forward(self, x):
        # the ViT encoder (2 % of the work: 0.09 of 4.7 TFLOP at 96^3) keeps fp32 token tensors under autocast too -- LayerNorm and
        # softmax are fp32 ops under torch autocast anyway; its four taps and the input volume enter the convolutional decoder
        # in the autocast storage type (bf16 under mi355seg.autocast, a no-op view otherwise)
        z0f = F.to_channels_last(x, dtype=torch.float32)
        batch = z0f.shape[0]
        z0 = F.cast(z0f)
        z3, z6, z9, z12 = (F.cast(t.reshape(batch, *self.patch_dim, self.embed_dim)) for t in self.transformer(z0f))
        up = self.decoder12_upsampler(z12)
        for lateral, tap, merge in ((self.decoder9, z9, self.decoder9_upsampler), (self.decoder6, z6, self.decoder6_upsampler),
                                    (self.decoder3, z3, self.decoder3_upsampler), (self.decoder0, z0, self.decoder0_header)):
            up = merge(F.cat_channels(lateral(tap), up))
        return F.to_channels_first(up)
