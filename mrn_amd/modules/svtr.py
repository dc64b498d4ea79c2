"""SVTR backbone on the HIP path (reference modules/svtr.py: PatchEmbed :211-254, Attention :90-152, Mlp :46-67,
Block :154-204, SubSample :265-312, SVTR :315-531; wrapper modules/feature_extraction.py:724-732).

Same constructor defaults, attribute names and state_dict keys (including the reference's unused `linear`,
`last_conv`, `norm` parameters).  Tokens are the NHWC pixels of the feature map, so `flatten(2).transpose(1, 2)` and
the reshapes before every SubSample are views.  Frozen experts (MRN's router phase) take the fused inference kernels
(or run in lock-step groups, modules/expert_group.py); an expert being trained (loop A) runs the same stages as autograd
Functions of mrn_amd.functional (LayerNormFn, TrainLinearFn, SvtrAttentionFn, ResidualScaleFn, GeluFn, ConvBlockFn).
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..functional import (AddPosFn, GeluFn, LayerNormFn, ResidualScaleFn, SvtrAttentionFn, TrainLinearFn, frozen_linear,
                          needs_grad)
from ._nn import conv_block, from_nhwc, to_nhwc


class Identity(nn.Module):
    def forward(self, input):
        return input


class DropPath(nn.Module):
    """per-sample stochastic depth; `forced_masks` (a list of [B] 0/1 tensors, one popped per call) pins the Bernoulli
    draws for parity tests"""

    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep
        self.forced_masks = None

    def scale(self, B, device):
        """[B] multiplier of the residual branch, or None when the path is always kept"""
        if self.drop_prob == 0. or not self.training:
            return None
        keep = 1 - self.drop_prob
        if self.forced_masks:
            m = self.forced_masks.pop(0).to(device).float()
        else:
            m = torch.empty(B, device=device).bernoulli_(keep)
        return m / keep if (keep > 0.0 and self.scale_by_keep) else m


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


def local_attention_mask(H, W, hk, wk):
    """[H*W, H*W] additive mask: 0 inside the (hk x wk) window centred on the query token, -inf outside (svtr.py:117-128)"""
    ys, xs = np.divmod(np.arange(H * W), W)
    dy = np.abs(ys[:, None] - ys[None, :])
    dx = np.abs(xs[:, None] - xs[None, :])
    inside = (dy <= hk // 2) & (dx <= wk // 2)
    return torch.from_numpy(np.where(inside, 0.0, -np.inf).astype(np.float32))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, mixer='Global', HW=(8, 25), local_k=[7, 11], qkv_bias=False, qk_scale=None,
                 attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.HW = HW
        self.mixer = mixer
        self.mask = None                                      # plain attribute, not a buffer (reference quirk 11)
        if HW is not None:
            self.N, self.C = HW[0] * HW[1], dim
            if mixer == 'Local':
                self.mask = local_attention_mask(HW[0], HW[1], local_k[0], local_k[1])

    def _mask_on(self, device):
        if self.mask is not None and self.mask.device != device:
            self.mask = self.mask.to(device)
        return self.mask

    def forward_train(self, x):
        """autograd path of an expert being trained: x [B,N,C] -> proj(softmax(q k^T * scale + mask) v)"""
        qkv = TrainLinearFn.apply(x, self.qkv.weight, self.qkv.bias, True)      # (leaves the range of qkv for the attention pass)
        ctx = SvtrAttentionFn.apply(qkv, self._mask_on(x.device), self.num_heads, self.scale)
        return TrainLinearFn.apply(ctx, self.proj.weight, self.proj.bias)

    def forward_tokens(self, x, residual=None):
        """x [B,N,C] -> proj(softmax(q k^T * scale + mask) v) (+ residual fused into the proj GEMM)"""
        B, N, C = x.shape
        h, d = self.num_heads, C // self.num_heads
        qkv = frozen_linear(x, self.qkv.weight, self.qkv.bias)                    # [B,N,3C] = (3, h, d) innermost
        mask = None
        if self.mask is not None:
            if self.mask.device != x.device:
                self.mask = self.mask.to(x.device)
            mask = self.mask
        if d == 32 and not torch.is_grad_enabled() and ops.SVTR_FUSED_ATTENTION:
            # frozen experts: one fused launch (online softmax, scores never reach HBM) instead of 2h batched GEMMs + softmax
            ctx = ops.svtr_attention(qkv, h, self.scale, mask, x3=ops.SVTR_ATTENTION_X3)
            return frozen_linear(ctx, self.proj.weight, self.proj.bias, residual=residual)
        attn = torch.empty(B, h, N, N, device=x.device, dtype=torch.float32)
        ctx = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        for hd in range(h):
            q = qkv[:, :, hd * d:(hd + 1) * d]                                    # views into qkv
            k = qkv[:, :, C + hd * d:C + (hd + 1) * d]
            v = qkv[:, :, 2 * C + hd * d:2 * C + (hd + 1) * d]
            a = attn[:, hd]
            # S[b] = scale * q k^T
            ops.gemm_raw(q, k, a, N, N, d, B, (N * 3 * C, 3 * C, 1), (N * 3 * C, 3 * C, 1), (h * N * N, N, 1), alpha=self.scale)
        ops.softmax_rows_(attn, mask)
        for hd in range(h):
            v = qkv[:, :, 2 * C + hd * d:2 * C + (hd + 1) * d]
            o = ctx[:, :, hd * d:(hd + 1) * d]
            # O[b][n][dd] = sum_m P[b][n][m] v[b][m][dd]  ->  "W operand"[dd][m] = v[m][dd]
            ops.gemm_raw(attn[:, hd], v, o, N, d, N, B, (h * N * N, N, 1), (N * 3 * C, 1, 3 * C), (N * C, C, 1))
        return frozen_linear(ctx, self.proj.weight, self.proj.bias, residual=residual)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mixer='Global', local_mixer=[7, 11], HW=[8, 25], mlp_ratio=4., qkv_bias=False,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer='nn.LayerNorm', epsilon=1e-6):
        super().__init__()
        self.norm1 = norm_layer(dim)
        if mixer not in ('Global', 'Local'):
            raise NotImplementedError("only the attention mixers of the shipped configuration run on the HIP path")
        self.mixer = Attention(dim, num_heads=num_heads, mixer=mixer, HW=HW, local_k=local_mixer, qkv_bias=qkv_bias,
                               qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else Identity()
        self.norm2 = norm_layer(dim)
        self.mlp_ratio = mlp_ratio
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward_train(self, x):
        """Block.forward (svtr.py:200-204) under autograd, same DropPath draw order as the inference path"""
        B, N, C = x.shape
        sc = self.drop_path.scale(B, x.device) if isinstance(self.drop_path, DropPath) else None
        y = LayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, True)
        x = ResidualScaleFn.apply(x, self.mixer.forward_train(y), sc, N)
        sc = self.drop_path.scale(B, x.device) if isinstance(self.drop_path, DropPath) else None
        y = LayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, True)
        hdn = GeluFn.apply(TrainLinearFn.apply(y, self.mlp.fc1.weight, self.mlp.fc1.bias, True))
        return ResidualScaleFn.apply(x, TrainLinearFn.apply(hdn, self.mlp.fc2.weight, self.mlp.fc2.bias, False, True), sc, N)

    def forward(self, x):
        if needs_grad(self, x):
            return self.forward_train(x)
        B, N, C = x.shape
        sc = self.drop_path.scale(B, x.device) if isinstance(self.drop_path, DropPath) else None
        y, _, _ = ops.layernorm_fwd(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if sc is None:
            x = self.mixer.forward_tokens(y, residual=x)
        else:
            x = ops.residual_scale_rows(x, self.mixer.forward_tokens(y), sc, N)
        sc = self.drop_path.scale(B, x.device) if isinstance(self.drop_path, DropPath) else None
        y, _, _ = ops.layernorm_fwd(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        hdn = frozen_linear(y, self.mlp.fc1.weight, self.mlp.fc1.bias, act=ops.ACT_GELU)
        if sc is None:
            return frozen_linear(hdn, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x)
        return ops.residual_scale_rows(x, frozen_linear(hdn, self.mlp.fc2.weight, self.mlp.fc2.bias), sc, N)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=[32, 100], in_channels=3, embed_dim=768, sub_num=2):
        super().__init__()
        self.img_size = img_size
        self.num_patches = (img_size[1] // (2 ** sub_num)) * (img_size[0] // (2 ** sub_num))
        self.embed_dim = embed_dim
        self.norm = None
        if sub_num != 2:
            raise NotImplementedError("sub_num=2 is the shipped configuration")
        self.proj = nn.Sequential(
            nn.Conv2d(in_channels, embed_dim // 2, 3, 2, 1), nn.BatchNorm2d(embed_dim // 2), nn.GELU(),
            nn.Conv2d(embed_dim // 2, embed_dim, 3, 2, 1), nn.BatchNorm2d(embed_dim), nn.GELU())

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if needs_grad(self, x):                               # expert training: conv + BatchNorm under autograd, then GELU
            y = GeluFn.apply(conv_block(to_nhwc(x), self.proj[0], self.proj[1], relu=False, precision="f32"))
            y = GeluFn.apply(conv_block(y, self.proj[3], self.proj[4], relu=False, precision="f32"))
        else:
            y = conv_block(to_nhwc(x), self.proj[0], self.proj[1], relu=False, act="gelu", precision="f32")
            y = conv_block(y, self.proj[3], self.proj[4], relu=False, act="gelu", precision="f32")
        return y.reshape(B, -1, y.shape[-1])                   # NHWC pixels are the tokens


class SubSample(nn.Module):
    def __init__(self, in_channels, out_channels, types='Pool', stride=[2, 1], sub_norm='nn.LayerNorm', act=None):
        super().__init__()
        self.types = types
        if types != 'Conv':
            raise NotImplementedError("patch_merging='Conv' is the shipped configuration")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)
        self.norm = eval(sub_norm)(out_channels)
        self.act = act() if act is not None else None
        if self.act is not None:
            raise NotImplementedError("SubSample activation is unused by SVTR")

    def forward_tokens(self, x, HW):
        """tokens [B, H*W, C] -> conv stride (2,1) -> LayerNorm -> tokens [B, (H/2)*W, C']"""
        B, N, C = x.shape
        y = conv_block(x.reshape(B, HW[0], HW[1], C), self.conv, None, relu=False)
        y = y.reshape(B, -1, y.shape[-1])
        if needs_grad(self, y):
            return LayerNormFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps)
        out, _, _ = ops.layernorm_fwd(y, self.norm.weight, self.norm.bias, self.norm.eps)
        return out


class SVTR(nn.Module):
    def __init__(self, img_size=[32, 256], in_channels=3, embed_dim=[64, 128, 256], depth=[3, 6, 3], num_heads=[2, 4, 8],
                 mixer=['Local'] * 6 + ['Global'] * 6, local_mixer=[[7, 11], [7, 11], [7, 11]], patch_merging='Conv',
                 mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0., last_drop=0.1, attn_drop_rate=0.,
                 drop_path_rate=0.1, norm_layer='nn.LayerNorm', sub_norm='nn.LayerNorm', epsilon=1e-6, out_channels=192,
                 out_char_num=25, block_unit='Block', act='nn.GELU', last_stage=True, sub_num=2, **kwargs):
        super().__init__()
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("dropout inside SVTR is 0 in the shipped configuration")
        self.img_size = img_size
        self.num_features = self.embed_dim = embed_dim
        self.out_channels = out_channels
        norm = partial(eval(norm_layer), eps=epsilon)
        self.patch_embed = PatchEmbed(img_size=img_size, in_channels=in_channels, embed_dim=embed_dim[0], sub_num=sub_num)
        self.HW = [img_size[0] // (2 ** sub_num), img_size[1] // (2 ** sub_num)]
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim[0]))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = np.linspace(0, drop_path_rate, sum(depth))
        bounds = np.cumsum([0] + list(depth))

        def stage(i, HW):
            return nn.ModuleList([
                Block(dim=embed_dim[i], num_heads=num_heads[i], mixer=mixer[bounds[i] + j], HW=HW, local_mixer=local_mixer[i],
                      mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, act_layer=eval(act),
                      attn_drop=attn_drop_rate, drop_path=dpr[bounds[i] + j], norm_layer=norm, epsilon=epsilon)
                for j in range(depth[i])])

        self.patch_merging = patch_merging
        if patch_merging is None:
            raise NotImplementedError("patch_merging=None is not a shipped configuration")
        self.blocks1 = stage(0, self.HW)
        self.sub_sample1 = SubSample(embed_dim[0], embed_dim[1], sub_norm=sub_norm, stride=[2, 1], types=patch_merging)
        self.blocks2 = stage(1, [self.HW[0] // 2, self.HW[1]])
        self.sub_sample2 = SubSample(embed_dim[1], embed_dim[2], sub_norm=sub_norm, stride=[2, 1], types=patch_merging)
        self.blocks3 = stage(2, [self.HW[0] // 4, self.HW[1]])
        self.sub_sample3 = SubSample(embed_dim[2], out_channels, sub_norm=sub_norm, stride=[2, 1], types=patch_merging)
        self.last_stage = last_stage
        if last_stage:                                    # parameters the reference creates but never uses (:465-479)
            self.avg_pool = nn.AdaptiveAvgPool2d([1, out_char_num])
            self.linear = nn.Linear(384, 512)
            self.last_conv = nn.Conv2d(embed_dim[2], self.out_channels, 1, 1, 0, bias=False)
            self.hardswish = nn.Hardswish()
            self.dropout = nn.Dropout(p=last_drop)
        self.norm = norm(embed_dim[-1])
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 1.0)                # reference quirk: LayerNorm bias initialised to 1 (:494-496)
        elif isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_in')

    def forward_features(self, x):
        B = x.shape[0]
        t = self.patch_embed(x)                                               # [B, 8*64, 64]
        if needs_grad(self, t):
            t = AddPosFn.apply(t, self.pos_embed)
        else:
            NC = t.shape[1] * t.shape[2]
            t = ops.ew_rows(ops.EW_ADD, t.view(B, NC), self.pos_embed.view(1, NC).expand(B, NC)).view(t.shape)   # row stride 0: no copy
        H, W = self.HW
        for blk in self.blocks1:
            t = blk(t)
        t = self.sub_sample1.forward_tokens(t, (H, W))
        for blk in self.blocks2:
            t = blk(t)
        t = self.sub_sample2.forward_tokens(t, (H // 2, W))
        for blk in self.blocks3:
            t = blk(t)
        t = self.sub_sample3.forward_tokens(t, (H // 4, W))                   # [B, (H/8)*W, out]
        return from_nhwc(t.reshape(B, H // 8, W, self.out_channels))         # logical [B, out, H/8, W]

    def forward(self, x):
        return self.forward_features(x)


class SVTR_FeatureExtractor(nn.Module):
    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        self.ConvNet = SVTR(in_channels=input_channel, out_channels=output_channel)

    def forward(self, input):
        return self.ConvNet(input)
