"""Transformation stage operator: TPS spatial transformer on the HIP path.

Same API / state_dict keys as the reference's modules/transformation.py (TPS_SpatialTransformerNetwork :9-50,
LocalizationNetwork :53-112, GridGenerator :115-216).  The localization convs reuse the implicit-GEMM conv
kernels; grid generation and bilinear sampling are one fused kernel (csrc/tps.hip).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..functional import AvgPoolFn, LinearFn, LinearReluFn, TPSSampleFn, needs_grad
from ._nn import conv_block, from_nhwc, to_nhwc


def fiducial_layout(num_fid, y_top, y_bottom):
    """[F,2] control points: x evenly spaced on both rows, y given per row (scalars or arrays)."""
    half = num_fid // 2
    xs = np.linspace(-1.0, 1.0, half)
    top = np.stack([xs, np.broadcast_to(y_top, (half,))], axis=1)
    bottom = np.stack([xs, np.broadcast_to(y_bottom, (half,))], axis=1)
    return np.concatenate([top, bottom], axis=0)


class LocalizationNetwork(nn.Module):
    """Predicts the F fiducial points C' from the input image (reference :53-112)."""

    def __init__(self, F, I_channel_num):
        super().__init__()
        self.F = F
        self.I_channel_num = I_channel_num
        chans = [I_channel_num, 64, 128, 256, 512]
        layers = []
        for i in range(4):
            layers += [nn.Conv2d(chans[i], chans[i + 1], 3, 1, 1, bias=False), nn.BatchNorm2d(chans[i + 1]), nn.ReLU(True)]
            layers.append(nn.MaxPool2d(2, 2) if i < 3 else nn.AdaptiveAvgPool2d(1))
        self.conv = nn.Sequential(*layers)
        self.localization_fc1 = nn.Sequential(nn.Linear(512, 256), nn.ReLU(True))
        self.localization_fc2 = nn.Linear(256, F * 2)
        # RARE initialisation: zero weight, bias = fiducials on the image border bowed towards the centre line
        half = F // 2
        init = fiducial_layout(F, np.linspace(0.0, -1.0, half), np.linspace(1.0, 0.0, half))
        self.localization_fc2.weight.data.fill_(0)
        self.localization_fc2.bias.data = torch.from_numpy(init).float().view(-1)

    def forward_nhwc(self, x):
        c = self.conv
        pool = ((2, 2), (2, 2), (0, 0))
        # exact fp32 throughout: the fiducials feed an ill-conditioned grid + sampler
        x = conv_block(x, c[0], c[1], pool=pool, precision=ops.LOCNET_CONV_PRECISION)
        x = conv_block(x, c[4], c[5], pool=pool, precision=ops.LOCNET_CONV_PRECISION)
        x = conv_block(x, c[8], c[9], pool=pool, precision=ops.LOCNET_CONV_PRECISION)
        x = conv_block(x, c[12], c[13], precision=ops.LOCNET_CONV_PRECISION)
        fc1, fc2 = self.localization_fc1[0], self.localization_fc2
        if needs_grad(self, x):
            x = AvgPoolFn.apply(x)
            x = LinearReluFn.apply(x, fc1.weight, fc1.bias)
            x = LinearFn.apply(x, fc2.weight, fc2.bias)
        else:
            x = ops.avgpool_nhwc(x)                                                 # [B,512]
            x = ops.linear(x, fc1.weight, fc1.bias, act=ops.ACT_RELU)
            x = ops.linear(x, fc2.weight, fc2.bias)
        return x.view(x.shape[0], self.F, 2)

    def forward(self, batch_I):
        return self.forward_nhwc(to_nhwc(batch_I))


def _reference_persists_tps_buffers():
    """The reference registers inv_delta_C / P_hat as (persistent) buffers only on hosts with more than one GPU and keeps
    them as plain tensors otherwise (:127-146), so its checkpoints carry two extra keys per TPS stage when written on a
    multi-GPU host.  Same rule here, so checkpoints saved on a host have the key set the reference would write there."""
    return torch.cuda.device_count() > 1


class GridGenerator(nn.Module):
    """Constant TPS matrices (reference :115-216): buffers that follow .to(device); in the state_dict exactly when the
    reference would put them there (see _reference_persists_tps_buffers).  Loading accepts both checkpoint flavours: the two
    keys are constants of (F, I_r_size), so a checkpoint without them keeps the computed values and one with them is
    accepted under strict=True even when this host would not save them."""

    def __init__(self, F, I_r_size):
        super().__init__()
        self.eps = 1e-6
        self.I_r_height, self.I_r_width = I_r_size
        self.F = F
        C = fiducial_layout(F, -1.0, 1.0)
        inv_delta_C, P_hat = self._build(F, C, self.I_r_width, self.I_r_height)
        persistent = _reference_persists_tps_buffers()
        self.register_buffer("inv_delta_C", torch.tensor(inv_delta_C).float(), persistent=persistent)
        self.register_buffer("P_hat", torch.tensor(P_hat).float(), persistent=persistent)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in ("inv_delta_C", "P_hat"):
            key = prefix + name
            persistent = name not in self._non_persistent_buffers_set
            if key in state_dict and not persistent:
                state_dict.pop(key)                       # multi-GPU-host checkpoint on a single-GPU host
            elif key not in state_dict and persistent:
                state_dict[key] = getattr(self, name)     # single-GPU-host checkpoint on a multi-GPU host
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def _build(self, F, C, W, H):
        # pairwise TPS kernel U(r) = r^2 log r between fiducials (diagonal: r := 1 -> 0)
        diff = C[:, None, :] - C[None, :, :]
        r = np.sqrt((diff ** 2).sum(-1))
        np.fill_diagonal(r, 1.0)
        K = r ** 2 * np.log(r)
        delta = np.zeros((F + 3, F + 3), dtype=np.float64)
        delta[:F, 0] = 1.0
        delta[:F, 1:3] = C
        delta[:F, 3:] = K
        delta[F:F + 2, 3:] = C.T
        delta[F + 2, 3:] = 1.0
        inv_delta = np.linalg.inv(delta)
        # pixel-centre sampling grid of the rectified image and its TPS lift [1, x, y, U(|p - c_j|)]
        gx = (np.arange(-W, W, 2) + 1.0) / W
        gy = (np.arange(-H, H, 2) + 1.0) / H
        P = np.stack(np.meshgrid(gx, gy), axis=2).reshape(-1, 2)
        d = np.linalg.norm(P[:, None, :] - C[None, :, :], ord=2, axis=2)
        rbf = np.square(d) * np.log(d + self.eps)
        P_hat = np.concatenate([np.ones((P.shape[0], 1)), P, rbf], axis=1)
        return inv_delta, P_hat

    def build_P_prime(self, batch_C_prime):
        """[B,F,2] -> sampling grid [B, H*W, 2] (only needed by callers that want the grid itself)."""
        raise NotImplementedError("the grid is produced inside mrn_tps_grid_sample_f32; use TPS_SpatialTransformerNetwork")


class TPS_SpatialTransformerNetwork(nn.Module):
    def __init__(self, F, I_size, I_r_size, I_channel_num=1):
        super().__init__()
        self.F = F
        self.I_size = I_size
        self.I_r_size = I_r_size
        self.I_channel_num = I_channel_num
        self.LocalizationNetwork = LocalizationNetwork(F, I_channel_num)
        self.GridGenerator = GridGenerator(F, I_r_size)

    def forward(self, batch_I, return_aux=False):
        x = to_nhwc(batch_I)
        cprime = self.LocalizationNetwork.forward_nhwc(x)
        g = self.GridGenerator
        if needs_grad(self, batch_I) and not return_aux:
            return from_nhwc(TPSSampleFn.apply(x, cprime, g.inv_delta_C, g.P_hat, self.I_r_size))
        if return_aux:
            out, grid = ops.tps_grid_sample(x, cprime, g.inv_delta_C, g.P_hat, self.I_r_size, want_grid=True)
            return from_nhwc(out), cprime, grid
        return from_nhwc(ops.tps_grid_sample(x, cprime, g.inv_delta_C, g.P_hat, self.I_r_size))
