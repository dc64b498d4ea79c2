"""DM-Router (reference modules/dm_router.py:4-67) as a parameter container whose forward/backward run on the HIP
path (mrn_amd.functional.DMRouterFn).  Same constructor, attribute names and state_dict keys."""
import torch.nn as nn

from ..functional import DMRouterFn


class SpatialDomainGating(nn.Module):
    def __init__(self, d_ffn, seq_len):
        super().__init__()
        self.norm = nn.LayerNorm(d_ffn // 2)
        self.proj = nn.Linear(seq_len, seq_len)


class ChannelDomainGating(nn.Module):
    def __init__(self, d_ffn, seq_len):
        super().__init__()
        self.norm = nn.LayerNorm(d_ffn)
        self.proj = nn.Linear(seq_len, seq_len)


class DM_Router(nn.Module):
    def __init__(self, channel, d_ffn, patch, domain):
        super().__init__()
        self.patch = patch
        self.channel = channel
        self.domain = domain
        self.norm = nn.LayerNorm(channel)
        self.proj_1 = nn.Linear(channel, d_ffn)
        self.activation = nn.GELU()
        self.spatial_gating = SpatialDomainGating(d_ffn, patch * domain)
        self.channel_gating = ChannelDomainGating(patch, domain * channel)
        self.proj_2 = nn.Linear(d_ffn // 2, channel)
        self.proj_3 = nn.Linear(channel, channel)
        assert d_ffn == 2 * channel, "the HIP path assumes d_ffn = 2 * channel (reference: model.py:445)"

    def forward_l2(self, x):
        """x [B, P, I, C] (router-internal layout) -> same layout"""
        sg, cg = self.spatial_gating, self.channel_gating
        return DMRouterFn.apply(x, self.norm.weight, self.norm.bias, self.proj_1.weight, self.proj_1.bias,
                                sg.norm.weight, sg.norm.bias, sg.proj.weight, sg.proj.bias,
                                self.proj_2.weight, self.proj_2.bias, cg.norm.weight, cg.norm.bias,
                                cg.proj.weight, cg.proj.bias, self.proj_3.weight, self.proj_3.bias)

    def forward(self, x):
        """reference layout: x [B, I(domain), P(patch), C] -> same"""
        return self.forward_l2(x.permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3)
