"""Glue shared by the stage modules: torch.nn layers are used as PARAMETER CONTAINERS only (so state_dict keys,
shapes and default initialisation are the reference's); their math runs through mrn_amd.ops (HIP kernels).

Activations travel between stages as ordinary torch tensors with the reference's logical NCHW shape but
channels_last (NHWC) memory, which is what the kernels consume.
"""
import weakref

import torch
import torch.nn as nn

from .. import ops


def require_no_grad(module, what):
    """Stages whose backward is not on the HIP path yet fail loudly instead of silently dropping gradients."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            f"{what}: backward through this stage is not implemented in the HIP path yet; freeze the expert "
            f"(requires_grad=False) or run under torch.no_grad()")


def to_nhwc(x):
    """logical [B,C,H,W] (any memory format) -> physical [B,H,W,C] contiguous tensor"""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    return ops.nchw_to_nhwc(x)


def from_nhwc(y):
    """physical [B,H,W,C] -> logical [B,C,H,W] view (channels_last memory)"""
    return y.permute(0, 3, 1, 2)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


_PACKED = weakref.WeakKeyDictionary()      # module -> (key, packed weight, event, stream): kept off the module (deepcopy of a model)


def packed_weight(conv):
    """[O,I,kh,kw] parameter -> cached [O,kh,kw,I] device tensor (re-packed when the parameter changes)."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device)
    cache = _PACKED.get(conv)
    if cache is None or cache[0] != key:
        ev = torch.cuda.Event()                      # (the build may run on the side stream: ops.prepack_trained)
        packed = ops.pack_conv_weight(w.detach())
        ev.record(torch.cuda.current_stream())
        cache = (key, packed, ev, torch.cuda.current_stream())
        _PACKED[conv] = cache
    elif cache[3] != torch.cuda.current_stream():
        torch.cuda.current_stream().wait_event(cache[2])
    return cache[1]


def bn_scale_shift(bn, stats, count):
    """(scale, shift) of a BatchNorm2d for this batch: batch statistics (+ running update) in training mode,
    running statistics otherwise -- torch.nn.BatchNorm2d.forward semantics."""
    if bn.training:
        mom = 0.1 if bn.momentum is None else bn.momentum
        scale, shift, _, _ = ops.bn_finalize(stats, bn.num_features, count, bn.weight, bn.bias, bn.running_mean,
                                             bn.running_var, mom, bn.eps)
        ops.count_batch(bn)
        return scale, shift
    return ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


def conv_block(x, conv, bn=None, relu=True, residual=None, pool=None, precision=None, act=None):
    """x NHWC -> conv (+bias) [-> BatchNorm] [-> +residual] [-> ReLU] [-> MaxPool], NHWC.
    pool = (kernel, stride, padding) fuses BN-apply + ReLU into the pooling pass.
    When a gradient is required the same kernels run inside autograd Functions (mrn_amd.functional)."""
    from ..functional import ConvBlockFn, MaxPoolFn, needs_grad
    trainable = needs_grad(conv, x, residual) or (bn is not None and needs_grad(bn))
    if trainable and act == "gelu":
        raise NotImplementedError("backward through conv + BatchNorm + GELU (SVTR PatchEmbed) is not implemented")
    if trainable:
        y = ConvBlockFn.apply(x, conv.weight, conv.bias, bn.weight if bn is not None else None,
                              bn.bias if bn is not None else None, residual,
                              (conv, bn, relu, precision or ops.TRAIN_CONV_PRECISION))
        if pool is not None:
            y = MaxPoolFn.apply(y, *pool)
        return y
    w = packed_weight(conv)
    stride, padding = _pair(conv.stride), _pair(conv.padding)
    if bn is None:
        y, _ = ops.conv2d_nhwc(x, w, conv.bias, stride, padding, act=ops.ACT_RELU if relu else ops.ACT_NONE,
                               precision=precision)
        if pool is not None:
            y = ops.maxpool_nhwc(y, *pool)
        return y
    y, stats = ops.conv2d_nhwc(x, w, conv.bias, stride, padding, act=ops.ACT_NONE, want_stats=bn.training,
                               precision=precision)
    count = y.shape[0] * y.shape[1] * y.shape[2]
    scale, shift = bn_scale_shift(bn, stats, count)
    if pool is not None and residual is None:
        return ops.maxpool_nhwc(y, pool[0], pool[1], pool[2], scale=scale, shift=shift, relu=relu)
    y = ops.scale_shift_act(y, scale, shift, relu=(2 if act == "gelu" else relu), residual=residual)
    if pool is not None:
        y = ops.maxpool_nhwc(y, *pool)
    return y
