"""UNETR on the MI355X kernels -- drop-in for the reference's models/three_d/unetr.py.

Same constructor (``UNETR(img_shape, input_dim, output_dim, embed_dim, patch_size, num_heads, dropout)``,
unetr.py:195), the same 338 ``state_dict`` keys and forward semantics
(unetr.py:277-294).  The ViT encoder (12 x [LayerNorm, 12-head attention, FFN 768->2048->768 ReLU], taps
after layers 3/6/9/12; ``encoder_norm`` is constructed but never applied, as in the reference) runs on the
strided batched MFMA GEMM + LayerNorm + softmax kernels; token tensors [N, P, E] ARE the channel-last
volumes [N, p0, p1, p2, E] the reference builds with ``transpose(-1, -2).view(...)`` (unetr.py:280-283), so
the hand-over to the conv decoder is a free view.  The decoder reuses the U-Net conv / ConvT / BN+ReLU kernels.
"""
import copy

import torch
import torch.nn as nn

from ... import functional as F
from ...layers import BatchNorm3d, Conv3d, ConvTranspose3d, Dropout, LayerNorm, Linear, ReLU


class SingleDeconv3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.block = ConvTranspose3d(in_planes, out_planes, kernel_size=2, stride=2, padding=0, output_padding=0)

    def forward(self, x):
        return self.block(x)


class SingleConv3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size):
        super().__init__()
        self.block = Conv3d(in_planes, out_planes, kernel_size=kernel_size, stride=1, padding=((kernel_size - 1) // 2))

    def forward(self, x):
        return self.block(x)


class _ConvBnRelu(nn.Sequential):
    def forward(self, x):
        mods = list(self.children())
        for m in mods[:-3]:
            x = m(x)
        return F.conv_bn_act(x, mods[-3].block, mods[-2], F.ACT_RELU)     # conv + statistics + BN + ReLU fused


class Conv3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size=3):
        super().__init__()
        self.block = _ConvBnRelu(SingleConv3DBlock(in_planes, out_planes, kernel_size), BatchNorm3d(out_planes), ReLU(True))

    def forward(self, x):
        return self.block(x)


class Deconv3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size=3):
        super().__init__()
        self.block = _ConvBnRelu(SingleDeconv3DBlock(in_planes, out_planes), SingleConv3DBlock(out_planes, out_planes, kernel_size),
                                 BatchNorm3d(out_planes), ReLU(True))

    def forward(self, x):
        return self.block(x)


def _add(a, b):
    """a + b on token tensors [N, P, E] (residual adds, unetr.py:160,166)."""
    N, P, E = a.shape
    return F.activation(a.reshape(N, 1, 1, P, E), F.ACT_NONE, residual=b.reshape(N, 1, 1, P, E)).view(N, P, E)


class SelfAttention(nn.Module):
    def __init__(self, num_heads, embed_dim, dropout):
        super().__init__()
        self.num_attention_heads = num_heads
        self.attention_head_size = int(embed_dim / num_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = Linear(embed_dim, self.all_head_size)
        self.key = Linear(embed_dim, self.all_head_size)
        self.value = Linear(embed_dim, self.all_head_size)
        self.out = Linear(embed_dim, embed_dim)
        self.attn_dropout = Dropout(dropout)
        self.proj_dropout = Dropout(dropout)
        self.softmax = nn.Softmax(dim=-1)
        self.vis = False

    def forward(self, hidden_states):
        q, k, v = self.query(hidden_states), self.key(hidden_states), self.value(hidden_states)
        keep = None
        if self.training and self.attn_dropout.p > 0.0:
            B, P, _ = q.shape
            keep = self.attn_dropout.draw((B, self.num_attention_heads, P, P), q.device)
        context = F.attention(q, k, v, self.num_attention_heads, keep)
        return self.proj_dropout(self.out(context)), None


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model=786, d_ff=2048, dropout=0.1):
        super().__init__()
        self.w_1 = Linear(d_model, d_ff)
        self.w_2 = Linear(d_ff, d_model)
        self.dropout = Dropout(dropout)

    def forward(self, x):
        return self.w_2(self.dropout(self.w_1.forward_relu(x)))


class Embeddings(nn.Module):
    def __init__(self, input_dim, embed_dim, cube_size, patch_size, dropout):
        super().__init__()
        self.n_patches = int((cube_size[0] * cube_size[1] * cube_size[2]) / (patch_size * patch_size * patch_size))
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.patch_embeddings = Conv3d(in_channels=input_dim, out_channels=embed_dim, kernel_size=patch_size, stride=patch_size)
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, embed_dim))
        self.dropout = Dropout(dropout)

    def forward(self, x):
        """x: channel-last [N, D, H, W, Cin] -> tokens [N, P, E]."""
        t = self.patch_embeddings(x)
        N = t.shape[0]
        t = t.reshape(N, self.n_patches, self.embed_dim)
        return self.dropout(_add(t, self.position_embeddings.expand(N, -1, -1)))


class TransformerBlock(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout, cube_size, patch_size):
        super().__init__()
        self.attention_norm = LayerNorm(embed_dim, eps=1e-6)
        self.mlp_norm = LayerNorm(embed_dim, eps=1e-6)
        self.mlp_dim = int((cube_size[0] * cube_size[1] * cube_size[2]) / (patch_size * patch_size * patch_size))
        self.mlp = PositionwiseFeedForward(embed_dim, 2048)      # the reference leaves the FFN dropout at its 0.1 default
        self.attn = SelfAttention(num_heads, embed_dim, dropout)

    def forward(self, x):
        a, weights = self.attn(self.attention_norm(x))
        x = _add(a, x)
        return _add(self.mlp(self.mlp_norm(x)), x), weights


class Transformer(nn.Module):
    def __init__(self, input_dim, embed_dim, cube_size, patch_size, num_heads, num_layers, dropout, extract_layers):
        super().__init__()
        self.embeddings = Embeddings(input_dim, embed_dim, cube_size, patch_size, dropout)
        self.layer = nn.ModuleList()
        self.encoder_norm = LayerNorm(embed_dim, eps=1e-6)       # constructed but never applied (unetr.py:176)
        self.extract_layers = extract_layers
        for _ in range(num_layers):
            self.layer.append(copy.deepcopy(TransformerBlock(embed_dim, num_heads, dropout, cube_size, patch_size)))

    def forward(self, x):
        taps = []
        h = self.embeddings(x)
        for depth, blk in enumerate(self.layer):
            h, _ = blk(h)
            if depth + 1 in self.extract_layers:
                taps.append(h)
        return taps


class UNETR(nn.Module):
    def __init__(self, img_shape=(128, 128, 128), input_dim=4, output_dim=3, embed_dim=768, patch_size=16, num_heads=12, dropout=0.1):
        super().__init__()
        self.input_dim, self.output_dim, self.embed_dim = input_dim, output_dim, embed_dim
        self.img_shape, self.patch_size, self.num_heads, self.dropout = img_shape, patch_size, num_heads, dropout
        self.num_layers = 12
        self.ext_layers = [3, 6, 9, 12]
        self.patch_dim = [int(x / patch_size) for x in img_shape]
        self.transformer = Transformer(input_dim, embed_dim, img_shape, patch_size, num_heads, self.num_layers, dropout, self.ext_layers)
        self.decoder0 = nn.Sequential(Conv3DBlock(input_dim, 32, 3), Conv3DBlock(32, 64, 3))
        self.decoder3 = nn.Sequential(Deconv3DBlock(embed_dim, 512), Deconv3DBlock(512, 256), Deconv3DBlock(256, 128))
        self.decoder6 = nn.Sequential(Deconv3DBlock(embed_dim, 512), Deconv3DBlock(512, 256))
        self.decoder9 = Deconv3DBlock(embed_dim, 512)
        self.decoder12_upsampler = SingleDeconv3DBlock(embed_dim, 512)
        self.decoder9_upsampler = nn.Sequential(Conv3DBlock(1024, 512), Conv3DBlock(512, 512), Conv3DBlock(512, 512),
                                                SingleDeconv3DBlock(512, 256))
        self.decoder6_upsampler = nn.Sequential(Conv3DBlock(512, 256), Conv3DBlock(256, 256), SingleDeconv3DBlock(256, 128))
        self.decoder3_upsampler = nn.Sequential(Conv3DBlock(256, 128), Conv3DBlock(128, 128), SingleDeconv3DBlock(128, 64))
        self.decoder0_header = nn.Sequential(Conv3DBlock(128, 64), Conv3DBlock(64, 64), SingleConv3DBlock(64, output_dim, 1))

    def forward(self, x):
        z0 = F.to_channels_last(x)
        N = z0.shape[0]
        z3, z6, z9, z12 = [t.reshape(N, *self.patch_dim, self.embed_dim) for t in self.transformer(z0)]
        z12 = self.decoder12_upsampler(z12)
        z9 = self.decoder9_upsampler(torch.cat([self.decoder9(z9), z12], dim=-1))
        z6 = self.decoder6_upsampler(torch.cat([self.decoder6(z6), z9], dim=-1))
        z3 = self.decoder3_upsampler(torch.cat([self.decoder3(z3), z6], dim=-1))
        out = self.decoder0_header(torch.cat([self.decoder0(z0), z3], dim=-1))
        return F.to_channels_first(out)
