"""Lock-step execution of the Transformation + FeatureExtraction stages of G frozen experts.

In MRN's router phase every sample visits every expert (reference modules/model.py:399-401, il_modules/mrn.py:323-337)
and the experts share one architecture, so their conv stacks are the same sequence of GEMM shapes with different
weights.  Running them one after the other leaves the chip with a ragged tail of tiles at the end of each of the
~200 conv launches (an expert's deepest layers are 520 tiles of 256x256 on 256 CUs); running them as ONE grouped
launch per layer (G x tiles) fills the chip, cuts the launch count by G and lets the convolutions use the 256-wide
tiles of csrc/conv_x3.hip.  Activations live as [G,B,H,W,C] stacks; between convolutions one elementwise pass applies
BatchNorm (+ residual + ReLU [+ MaxPool]) and writes the fp32 tensor and / or the HL32 split-fp16 operand of the next
convolution (csrc/group_ops.hip).

Numerics are those of the per-expert path with CONV_PRECISION "fp16x3" (22-bit split products, fp32 accumulation, fp32
BatchNorm statistics); layers the grouped kernel cannot take (Cin % 32 != 0) run per expert on the exact-fp32 kernel.
The module parameters stay where they are (state_dict layout untouched): weights are re-packed into per-layer HL32
stacks that are cached until a parameter changes.
"""
import os

import torch

from .. import ops
from ._nn import _pair, packed_weight, require_no_grad, to_nhwc


EVAL_BN_FOLD = os.environ.get("MRN_EVAL_BN_FOLD", "1") != "0"      # eval-mode BatchNorm folded into the conv epilogue (A/B switch)
# eval-mode layers that qualify for the Winograd form run conv (F(4,3)) + a producer pass with the running-statistics affine instead of
# the one-launch direct conv with the BatchNorm folded into its epilogue (A/B switch)
SVTR_GROUPED_EMBED = os.environ.get("MRN_SVTR_EMBED", "grouped") == "grouped"      # SVTR PatchEmbed of the frozen experts in lock-step
TPS_WINO = os.environ.get("MRN_TPS_WINO", "1") == "1"          # TPS localisation network: convs 3 and 4 on the Winograd form
EVAL_WINO = os.environ.get("MRN_EVAL_WINO", "1") != "0"
RESIDUAL_FROM_F32 = bool(int(os.environ.get("MRN_RESIDUAL_F32", "0")))      # True: keep an fp32 copy of every identity-shortcut source (one extra 4 B/element write)


class Act:
    """A [G,B,H,W,C] activation stack: fp32 tensor and / or HL32 bytes.  shared=True: one [B,H,W,C] input for all groups."""

    def __init__(self, shape, f32=None, hl=None, shared=False, wino=None, wino_R=0):
        self.shape = tuple(shape)          # (G, B, H, W, C)
        self.f32 = f32
        self.hl = hl
        self.shared = shared
        self.wino = wino                   # Winograd-domain operand bytes [G][B][H][ceil(W/R)][R+2][C/32][128] (ops.bn_apply_wino_grouped)
        self.wino_R = wino_R


def _bn_modules(extractor):
    got = getattr(extractor, "_mrn_bn_list", None)
    if got is None:
        got = [m for m in extractor.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        extractor._mrn_bn_list = got
    return got


def supported(experts):
    """Grouped execution covers TPS/None + ResNet/VGG and None + SVTR experts of identical configuration, all frozen, same
    BN / DropPath mode."""
    if len(experts) < 2:
        return False
    e0 = experts[0]
    for e in experts:
        if e.stages != e0.stages or e.stages["Feat"] not in ("ResNet", "VGG", "SVTR") or e.stages["Trans"] not in ("TPS", "None"):
            return False
    if e0.stages["Feat"] == "SVTR":
        if e0.stages["Trans"] != "None" or not ops.SVTR_FUSED_ATTENTION or len({e.training for e in experts}) != 1:
            return False
    for mods in zip(*[_bn_modules(e) for e in experts]):
        if len({m.training for m in mods}) != 1:
            return False
    return ops.CONV_PRECISION in ("auto", "fp16x3")


class _GroupedLinear:
    """cache of per-layer stacked operands + grouped Linear on the x3 GEMM (mrn_conv2d_x3_hl32 with a 1x1 kernel)"""

    def _cached(self, name, params, build):
        key = tuple((p.data_ptr(), p._version) for p in params)
        got = self._cache.get(name)
        if got is None or got[0] != key:
            with torch.no_grad():
                got = (key, build())
            self._cache[name] = got
        return got[1]

    def _linear(self, name, x_hl, rows, K, weights, biases, out=None, out_row_stride=0, out_group_stride=0, groups=None,
                act=ops.ACT_NONE, hl_only=False):
        """grouped y[g] = x[g] @ W[g]^T + b[g]; weights: list of [N,K] tensors (any stride), biases: list or None.
        hl_only: return the result as the HL32 operand of the next grouped Linear instead of fp32"""
        G = len(weights)
        N = weights[0].shape[0]
        params = list(weights) + ([b for b in biases] if biases is not None else [])

        def build():
            w_hl, scale = ops.pack_weights_hl32([w.detach().contiguous().view(N, 1, 1, K) for w in weights])
            bias = torch.stack([b.detach() for b in biases]).contiguous() if biases is not None else None
            return w_hl, scale, bias
        w_hl, scale, bias = self._cached(name, params, build)
        y, _ = ops.conv2d_x3(x_hl, G, False, rows, 1, 1, K, w_hl, scale, N, (1, 1), bias=bias, act=act, out=out,
                             out_row_stride=out_row_stride, out_group_stride=out_group_stride, hl_only=hl_only,
                             products=ops.X3_PRODUCTS)
        return y


class BackboneGroup(_GroupedLinear):
    def __init__(self, experts):
        self.experts = list(experts)       # Model_Extractor modules
        self.G = len(self.experts)
        self._cache = {}
        self._wcache = {}
        self._bncache = {}
        self._nbt = []

    # ---- caches ------------------------------------------------------------------------------------------------
    def _weights_hl(self, convs):
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs)
        got = self._wcache.get(id(convs[0]))
        if got is None or got[0] != key:
            got = (key, ops.pack_weights_hl32([packed_weight(c).ohwi for c in convs]))
            self._wcache[id(convs[0])] = got
        return got[1]

    def _weights_wino(self, convs, R):
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs) + (R, ops.wino_dense(), ops.REDUCED_BF16)
        got = self._wcache.get(("wino", id(convs[0])))
        if got is None or got[0] != key:
            got = (key, ops.pack_weights_wino([packed_weight(c).ohwi for c in convs], R))
            self._wcache[("wino", id(convs[0]))] = got
        return got[1]

    @staticmethod
    def wino_for(conv, bns):
        """R of the Winograd form this conv would run in when its input arrives as a Winograd-domain operand, else 0: 3x3 / stride 1
        / pad 1, wide enough, and NOT followed by an eval-mode BatchNorm (that fold keeps its one-launch direct form)"""
        if bns is not None and not bns[0].training and not EVAL_WINO:
            return 0
        ok = ops.wino_eligible(_pair(conv.kernel_size), _pair(conv.stride), _pair(conv.padding), conv.in_channels, conv.out_channels)
        return ops.WINO_R if ok else 0

    def _bn_table(self, bns):
        key = tuple(t.data_ptr() for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var))
        got = self._bncache.get(id(bns[0]))
        if got is None or got[0] != key:
            rows = [[b.weight.data_ptr() for b in bns], [b.bias.data_ptr() for b in bns],
                    [b.running_mean.data_ptr() for b in bns], [b.running_var.data_ptr() for b in bns]]
            got = (key, torch.tensor(rows, dtype=torch.int64, device=bns[0].weight.device))
            self._bncache[id(bns[0])] = got
        return got[1]

    def _bias_stack(self, convs):
        if convs[0].bias is None:
            return None
        return torch.stack([c.bias.detach() for c in convs]).contiguous()

    # ---- one conv (+BN) (+residual) (+ReLU) (+pool) layer for all groups ----------------------------------------------
    def layer(self, x, convs, bns=None, relu=True, residual=None, pool=None, want_f32=False, want_hl=True, want_wino=0, gelu=False):
        """want_wino = R: the result also as the Winograd-domain operand of a following 3x3 conv (only from the BatchNorm-apply
        pass, i.e. with bns and no pool); a layer whose INPUT carries x.wino and qualifies (wino_for) runs as F(R,3).
        gelu: BatchNorm -> GELU instead of ReLU (SVTR PatchEmbed; plain BatchNorm-apply pass only)"""
        if gelu:
            assert bns is not None and bns[0].training and pool is None and not want_wino and residual is None
            relu = True
        G = self.G
        _, B, H, W, Cin = x.shape
        c0 = convs[0]
        Cout = c0.out_channels
        ksize, stride, padding = _pair(c0.kernel_size), _pair(c0.stride), _pair(c0.padding)
        Ho, Wo = ops.conv_out_hw(H, W, ksize, stride, padding)
        dev = c0.weight.device
        training = bns is not None and bns[0].training
        fuse_act = bns is None                       # no BatchNorm: bias + ReLU go into the conv epilogue
        act = ops.ACT_RELU if (fuse_act and relu) else ops.ACT_NONE
        stats = None
        res = residual.f32 if residual is not None else None
        res_hl = residual.hl if residual is not None and res is None else None
        use_wino = x.wino is not None and x.wino_R == self.wino_for(c0, bns)
        # (an eval-mode layer with a 2x2 / 2 pool behind it that the patch-resident kernel takes runs there too: the pool in the conv
        #  epilogue and the running-statistics affine on the pooled map beat the one-launch fold + a pooling pass over the full map)
        patch_pool = (pool == ((2, 2), (2, 2), (0, 0)) and Ho % 2 == 0 and Wo % 2 == 0
                      and ops.patch_conv_supported(ksize, stride, padding, Cin, Cout))
        if (EVAL_BN_FOLD and bns is not None and not training and Cin % 32 == 0 and Cout >= 64 and Cout % 32 == 0
                and (want_hl or res_hl is None) and not use_wino and not want_wino and not patch_pool):
            # frozen experts in EVAL mode (DERNet's old extractors, LwF's previous network, validation): the BatchNorm is a fixed
            # per-channel affine, so conv -> BN -> (+ identity) -> ReLU -> operand split is ONE launch: the affine, the shortcut
            # and the activation run in the conv epilogue, which writes the HL32 operand of the next layer directly
            if x.hl is None:
                assert x.f32 is not None
                x.hl = ops.split_hl32(x.f32)
            w_hl, w_scale = self._weights_hl(convs)
            # the affine is rebuilt from the CURRENT running statistics on every call (one launch through the pointer table): the
            # statistics are updated by kernels through raw pointers while the experts run in train mode between two validations,
            # so nothing keyed on tensor versions may cache it
            scale, shift = ops.bn_eval_affine_grouped(self._bn_table(bns), G, Cout, bns[0].eps)
            fused = dict(bias=self._bias_stack(convs), act=ops.ACT_RELU if relu else ops.ACT_NONE, ch_scale=scale, ch_shift=shift,
                         residual=res, residual_hl=res_hl, products=ops.X3_PRODUCTS)
            if pool is not None:
                assert res is None and res_hl is None
                y, _ = ops.conv2d_x3(x.hl, G, x.shared, B, H, W, Cin, w_hl, w_scale, Cout, ksize, stride, padding, **fused)
                f32, hl, (Hp, Wp) = ops.maxpool_grouped(y, pool[0], pool[1], pool[2], None, None, relu=False, want_f32=want_f32,
                                                        want_hl=want_hl)
                return Act((G, B, Hp, Wp, Cout), f32, hl)
            got, _ = ops.conv2d_x3(x.hl, G, x.shared, B, H, W, Cin, w_hl, w_scale, Cout, ksize, stride, padding,
                                   hl_only=want_hl and not want_f32, also_hl=want_hl and want_f32, **fused)
            if want_hl and want_f32:
                return Act((G, B, Ho, Wo, Cout), got[0], got[1])
            return Act((G, B, Ho, Wo, Cout), None, got) if want_hl else Act((G, B, Ho, Wo, Cout), got, None)
        # the narrow early 3x3 layers (32 -> 64, 64 -> 128): patch-resident weight-stationary kernel (csrc/conv_patch.hip); a 2x2 / 2
        # max-pool behind it is taken in ITS epilogue -- per window and channel the maximum where the BatchNorm weight is >= 0, the
        # minimum where it is negative (BatchNorm is monotone per channel) -- so the pass below runs on a quarter of the pixels
        use_patch = not use_wino and (x.hl is not None or x.f32 is not None) and ops.patch_conv_supported(ksize, stride, padding, Cin, Cout)
        use_c4 = (not use_patch and not use_wino and Cin == 4 and ksize == (3, 3) and stride == (1, 1) and padding == (1, 1)
                  and Cout in (32, 64) and x.f32 is not None)       # first conv of the stacks (csrc/conv_first.hip): pools the same way
        pooled = ((use_patch or (use_c4 and ops.PATCH_CONV) or (use_wino and ops.wino_pool_supported(H, W, x.wino_R, Cout)))
                  and pool == ((2, 2), (2, 2), (0, 0)) and Ho % 2 == 0 and Wo % 2 == 0)
        y = torch.empty(G, B, Ho // 2 if pooled else Ho, Wo // 2 if pooled else Wo, Cout, device=dev, dtype=torch.float32)
        if use_patch:
            if x.hl is None:
                x.hl = ops.split_hl32(x.f32)
            w_hl, w_scale = self._weights_hl(convs)
            _, stats = ops.conv3x3_patch_x3(x.hl, G, x.shared, B, H, W, Cin, w_hl, w_scale, Cout, bias=self._bias_stack(convs), act=act,
                                            want_stats=training, pool=pooled, gamma_ptrs=self._bn_table(bns)[0] if bns is not None else None,
                                            out=y)
            if pooled:
                pool = None                      # (done: what follows is the plain BatchNorm-apply pass on the pooled map)
        elif use_wino:
            u_hl, u_scale = self._weights_wino(convs, x.wino_R)
            _, stats = ops.conv2d_x3_wino(x.wino, G, x.shared, B, H, W, Cin, u_hl, u_scale, Cout, x.wino_R,
                                          bias=self._bias_stack(convs), act=act, want_stats=training, out=y, pool=pooled,
                                          gamma_ptrs=self._bn_table(bns)[0] if (pooled and bns is not None) else None)
            if pooled:
                pool = None
        elif Cin % 32 == 0 and Cout >= 64:
            if x.hl is None:
                assert x.f32 is not None
                x.hl = ops.split_hl32(x.f32)
            w_hl, w_scale = self._weights_hl(convs)
            _, stats = ops.conv2d_x3(x.hl, G, x.shared, B, H, W, Cin, w_hl, w_scale, Cout, ksize, stride, padding,
                                     bias=self._bias_stack(convs), act=act, want_stats=training, out=y, products=ops.X3_PRODUCTS)
        elif use_c4:
            # first conv of the stacks: one launch for all experts on the dedicated Cin = 4 kernel (csrc/conv_first.hip)
            w = self._cached("c4w%d" % id(c0), [c.weight for c in convs],
                             lambda: torch.stack([packed_weight(c).ohwi for c in convs]).contiguous())
            _, stats = ops.conv3x3_c4_grouped(x.f32, w, self._bias_stack(convs), act=act, want_stats=training, out=y, pool=pooled,
                                              gamma_ptrs=self._bn_table(bns)[0] if bns is not None else None)
            if pooled:
                pool = None
        else:   # other small-Cin layers: exact-fp32 kernel per expert, written into the stack
            assert x.f32 is not None
            n = ops.call("mrn_conv2d_stats_floats", B, Ho, Wo, Cout) if training else 0
            stats = torch.empty(G, n, device=dev, dtype=torch.float32) if training else None
            for g, c in enumerate(convs):
                xg = x.f32 if x.shared else x.f32[g]
                ops.conv2d_nhwc(xg, packed_weight(c), c.bias, stride, padding, act=act, want_stats=training, precision="f32",
                                out=y[g], stats_out=stats[g] if training else None)
        scale = shift = None
        if bns is not None:
            if training:
                mom = 0.1 if bns[0].momentum is None else bns[0].momentum
                scale, shift = ops.bn_finalize_grouped(stats, G, Cout, B * Ho * Wo, self._bn_table(bns), mom, bns[0].eps)
                self._nbt += [b.num_batches_tracked for b in bns if b.num_batches_tracked is not None]
            else:
                scale, shift = ops.bn_eval_affine_grouped(self._bn_table(bns), G, Cout, bns[0].eps)
        # identity shortcut: the fp32 tensor when it exists, else the HL32 image (hi + lo) the block input already has
        post_relu = relu and not fuse_act
        if pool is not None:
            assert res is None and res_hl is None
            if want_wino:
                f32, hl, v, (Hp, Wp) = ops.maxpool_wino_grouped(y, pool[0], pool[1], pool[2], want_wino, scale, shift, relu=post_relu,
                                                                want_f32=want_f32, want_hl=want_hl)
                return Act((G, B, Hp, Wp, Cout), f32, hl, wino=v, wino_R=want_wino)
            f32, hl, (Hp, Wp) = ops.maxpool_grouped(y, pool[0], pool[1], pool[2], scale, shift, relu=post_relu,
                                                    want_f32=want_f32, want_hl=want_hl)
            return Act((G, B, Hp, Wp, Cout), f32, hl)
        Ho, Wo = y.shape[2], y.shape[3]              # (a pooled patch convolution left the pooled map)
        if scale is None and res is None and res_hl is None and not post_relu and not want_hl and not want_wino:
            return Act((G, B, Ho, Wo, Cout), y, None)
        if want_wino:
            f32, hl, v = ops.bn_apply_wino_grouped(y, scale, shift, want_wino, relu=post_relu, residual=res, residual_hl=res_hl,
                                                   want_f32=want_f32, want_hl=want_hl)
            return Act((G, B, Ho, Wo, Cout), f32, hl, wino=v, wino_R=want_wino)
        f32, hl = ops.bn_apply_grouped(y, scale, shift, relu=2 if (gelu and post_relu) else post_relu, residual=res, want_f32=want_f32,
                                       want_hl=want_hl, residual_hl=res_hl)
        return Act((G, B, Ho, Wo, Cout), f32, hl)

    # ---- network programs ---------------------------------------------------------------------------------------
    def _basic_block(self, x, blocks, next_needs_f32, out_wino=0, out_hl=True):
        """out_wino / out_hl: what the consumer of the block's result reads (a Winograd-form conv / the plain HL32 operand)"""
        b0 = blocks[0]
        w2 = self.wino_for(b0.conv2, [b.bn2 for b in blocks])
        out = self.layer(x, [b.conv1 for b in blocks], [b.bn1 for b in blocks], want_wino=w2, want_hl=not w2)
        if b0.downsample is not None:
            res = self.layer(x, [b.downsample[0] for b in blocks], [b.downsample[1] for b in blocks], relu=False,
                             want_f32=True, want_hl=False)
        else:
            res = x
            assert res.f32 is not None or res.hl is not None
        return self.layer(out, [b.conv2 for b in blocks], [b.bn2 for b in blocks], relu=True, residual=res,
                          want_f32=next_needs_f32, want_hl=out_hl, want_wino=out_wino)

    def _batched_linear_f32(self, name, x, layers, act=ops.ACT_NONE):
        """x [G,R,K] fp32 -> [G,R,N]: y[g] = act(x[g] @ W_g^T + b_g) on the exact-fp32 GEMM, batch = expert"""
        G, R, K = x.shape
        N = layers[0].out_features
        w, b = self._cached(name, [l.weight for l in layers] + [l.bias for l in layers],
                            lambda: (torch.stack([l.weight.detach() for l in layers]).contiguous(),
                                     torch.stack([l.bias.detach() for l in layers]).contiguous()))
        y = torch.empty(G, R, N, device=x.device, dtype=torch.float32)
        ops.gemm_raw(x, w, y, R, N, K, G, (R * K, K, 1), (N * K, K, 1), (R * N, N, 1), bias=b, bias_batch_stride=N, act=act)
        return y

    def _resnet(self, x, last_hl=False):
        nets = [e.FeatureExtraction.ConvNet for e in self.experts]
        n0 = nets[0]
        p22, p2_21 = ((2, 2), (2, 2), (0, 0)), ((2, 2), (2, 1), (0, 1))

        def stage(x, name, after, after_bn):
            """`after`: the conv that follows the stage (its Winograd eligibility decides what the last block emits)"""
            blocks = [list(getattr(n, name)) for n in nets]
            nb = len(blocks[0])
            for i in range(nb):
                # identity shortcuts read the block input back from its HL32 image: no fp32 copy is ever written
                if i + 1 < nb:      # the next block reads the plain operand (shortcut / 1x1 downsample) and, if it qualifies, the Winograd one
                    nxt = [b[i + 1] for b in blocks]
                    ow, oh = self.wino_for(nxt[0].conv1, [b.bn1 for b in nxt]), True
                else:
                    ow = self.wino_for(getattr(n0, after), [getattr(n, after_bn) for n in nets])
                    oh = not ow
                x = self._basic_block(x, [b[i] for b in blocks], RESIDUAL_FROM_F32 and i + 1 < nb and blocks[0][i + 1].downsample is None,
                                      out_wino=ow, out_hl=oh)
            return x

        def first_wino(name):
            blk = [getattr(n, name)[0] for n in nets]
            return self.wino_for(blk[0].conv1, [b.bn1 for b in blk])

        def first_block_identity(name):
            return RESIDUAL_FROM_F32 and getattr(n0, name)[0].downsample is None

        x = self.layer(x, [n.conv0_1 for n in nets], [n.bn0_1 for n in nets])
        x = self.layer(x, [n.conv0_2 for n in nets], [n.bn0_2 for n in nets], pool=p22, want_f32=first_block_identity("layer1"))
        x = stage(x, "layer1", "conv1", "bn1")
        x = self.layer(x, [n.conv1 for n in nets], [n.bn1 for n in nets], pool=p22, want_f32=first_block_identity("layer2"),
                       want_wino=first_wino("layer2"))
        x = stage(x, "layer2", "conv2", "bn2")
        x = self.layer(x, [n.conv2 for n in nets], [n.bn2 for n in nets], pool=p2_21, want_f32=first_block_identity("layer3"),
                       want_wino=first_wino("layer3"))
        x = stage(x, "layer3", "conv3", "bn3")
        x = self.layer(x, [n.conv3 for n in nets], [n.bn3 for n in nets], want_f32=first_block_identity("layer4"),
                       want_wino=first_wino("layer4"))
        x = stage(x, "layer4", "conv4_1", "bn4_1")
        x = self.layer(x, [n.conv4_1 for n in nets], [n.bn4_1 for n in nets])
        return self.layer(x, [n.conv4_2 for n in nets], [n.bn4_2 for n in nets], want_f32=not last_hl, want_hl=last_hl)

    def _vgg(self, x, last_hl=False):
        nets = [e.FeatureExtraction.ConvNet for e in self.experts]
        p22, p21 = ((2, 2), (2, 2), (0, 0)), ((2, 1), (2, 1), (0, 0))

        def L(x, i, bn=None, pool=None, last=False, nxt=None):
            """nxt = (conv index, bn index or None) of the layer that consumes the result: decides the operand form it is written in"""
            ww = 0
            if nxt is not None:
                ww = self.wino_for(nets[0][nxt[0]], None if nxt[1] is None else [n[nxt[1]] for n in nets])
            return self.layer(x, [n[i] for n in nets], None if bn is None else [n[bn] for n in nets], pool=pool,
                              want_f32=last and not last_hl, want_hl=(not last or last_hl) and not ww, want_wino=ww)
        x = L(x, 0, pool=p22, nxt=(3, None))
        x = L(x, 3, pool=p22, nxt=(6, None))
        x = L(x, 6, nxt=(8, None))
        x = L(x, 8, pool=p21, nxt=(11, 12))
        x = L(x, 11, bn=12, nxt=(14, 15))
        x = L(x, 14, bn=15, pool=p21, nxt=(18, None))
        return L(x, 18, last=True)

    # ---- SVTR: tokens [G*B, N, C] fp32 residual stream, every Linear one grouped x3 GEMM ------------------------------------
    def _drop_scales(self, blks, B, dev):
        """[G*B] DropPath multipliers of one residual branch (modules/svtr.py:17-22), or None when every path is kept"""
        dp = blks[0].drop_path
        if not hasattr(dp, "drop_prob") or dp.drop_prob == 0. or not dp.training:
            return None
        if any(b.drop_path.forced_masks for b in blks):       # parity tests pin the draws per expert
            return torch.cat([b.drop_path.scale(B, dev) for b in blks])
        bank = getattr(self, "_drop_bank", None)
        if bank is not None and bank[1] < bank[0].shape[0] and bank[2][bank[1]] == dp.drop_prob:
            bank[1] += 1                      # (the next row of the forward's one-launch draw, see _svtr)
            return bank[0][bank[1] - 1]
        keep = 1 - dp.drop_prob
        m = torch.empty(self.G * B, device=dev).bernoulli_(keep)
        return m / keep if (keep > 0.0 and dp.scale_by_keep) else m

    def _draw_drop_bank(self, nets, B, dev):
        """every DropPath multiplier of one grouped SVTR forward in TWO launches (one Bernoulli draw over [draws, G*B] with a per-row
        keep probability, one scaling) instead of two per residual branch -- 44 short launches per sub-group and forward.  Rows in
        Block.forward's draw order (mixer branch, then MLP branch, block by block); _drop_scales hands them out.  Pinned draws
        (forced_masks: the parity tests) and eval mode bypass the bank."""
        self._drop_bank = None
        probs = []
        for si in range(3):
            stages = [list(getattr(n, "blocks%d" % (si + 1))) for n in nets]
            for i, blk in enumerate(stages[0]):
                dp = blk.drop_path
                if hasattr(dp, "drop_prob") and dp.drop_prob != 0. and dp.training:
                    if any(st[i].drop_path.forced_masks for st in stages) or not dp.scale_by_keep:
                        return
                    probs += [dp.drop_prob, dp.drop_prob]
        if not probs or max(probs) >= 1.0:
            return
        keep = self._cached("dropkeep%d" % len(probs), [], lambda: torch.tensor([1.0 - p_ for p_ in probs], device=dev, dtype=torch.float32).view(-1, 1))
        m = torch.bernoulli(keep.expand(len(probs), self.G * B))
        self._drop_bank = [m.div_(keep), 0, probs]

    def _ln_params(self, name, norms):
        return self._cached(name, [n.weight for n in norms] + [n.bias for n in norms],
                            lambda: (torch.stack([n.weight.detach() for n in norms]).contiguous(),
                                     torch.stack([n.bias.detach() for n in norms]).contiguous()))

    def _svtr_block(self, name, x, pending, blks, B, N):
        """one Block (svtr.py:154-204) of every expert.  x [G*B,N,C] residual stream; pending = (branch, drop) of the previous
        block's MLP, folded into this block's first LayerNorm pass.  Returns (x, pending)."""
        G = self.G
        C = x.shape[-1]
        rows = B * N
        b0 = blks[0]
        mixer = b0.mixer
        if mixer.mask is not None and mixer.mask.device != x.device:
            mixer.mask = mixer.mask.to(x.device)
        drop1 = self._drop_scales(blks, B, x.device)          # same draw order as Block.forward: mixer branch, then MLP branch
        g1, b1 = self._ln_params(name + ".ln1", [b.norm1 for b in blks])
        br, dr = pending if pending is not None else (None, None)
        if mixer.num_heads * 32 == C and ops.svtr_mixer_supported(N, C, B, mixer.mask):
            # LayerNorm1 -> qkv -> attention -> proj -> + residual -> LayerNorm2 in one kernel (csrc/svtr_mixer.hip)
            qkvs, projs = [b.mixer.qkv for b in blks], [b.mixer.proj for b in blks]

            def build_mixer():
                wq, sq = ops.pack_weights_hl32([m.weight.detach().contiguous().view(3 * C, 1, 1, C) for m in qkvs])
                perm = ops.mlp_hidden_permutation(C, x.device)
                wp, sp = ops.pack_weights_hl32([m.weight.detach().index_select(1, perm).contiguous().view(C, 1, 1, C) for m in projs])
                bq = torch.stack([m.bias.detach() for m in qkvs]).contiguous() if qkvs[0].bias is not None else None
                return wq, sq, bq, wp, sp, torch.stack([m.bias.detach() for m in projs]).contiguous()
            wq, sq, bq, wp, sp, bp = self._cached(name + ".mixer", [t_ for m in qkvs + projs for t_ in (m.weight, m.bias) if t_ is not None],
                                                  build_mixer)
            drop2 = self._drop_scales(blks, B, x.device)
            g2, b2 = self._ln_params(name + ".ln2", [b.norm2 for b in blks])
            Ch = blks[0].mlp.fc1.out_features
            x, y_hl = ops.svtr_mixer_fused(x, br, dr, g1, b1, b0.norm1.eps, wq, sq, bq, mixer.mask, mixer.scale, wp, sp, bp, drop1,
                                           g2, b2, b0.norm2.eps, B, hw=mixer.HW if mixer.mask is not None else None)
        else:
            if mixer.num_heads * 32 == C and ops.svtr_attention_block_supported(N, C, B, mixer.mask):
                # stage 3 (C = 256): LayerNorm1 -> qkv -> attention in one kernel, proj and the rest unfused (csrc/svtr_mixer.hip ATTN form)
                qkvs = [b.mixer.qkv for b in blks]

                def build_qkv():
                    wq, sq = ops.pack_weights_hl32([m.weight.detach().contiguous().view(3 * C, 1, 1, C) for m in qkvs])
                    return wq, sq, torch.stack([m.bias.detach() for m in qkvs]).contiguous() if qkvs[0].bias is not None else None
                wq, sq, bq = self._cached(name + ".attnblk", [t_ for m in qkvs for t_ in (m.weight, m.bias) if t_ is not None], build_qkv)
                x, ctx_hl = ops.svtr_attention_block_fused(x, br, dr, g1, b1, b0.norm1.eps, wq, sq, bq, mixer.mask, mixer.scale, B)
                Ch = blks[0].mlp.fc1.out_features
                if ops.SVTR_FUSED_TAIL and ops.SVTR_FUSED_MLP and C == 256 and Ch == 4 * C and ops.X3_PRODUCTS == 3:
                    # ... and the rest of the block as ONE more launch: proj -> + residual -> LayerNorm2 -> fc1 -> GELU -> fc2 (csrc/svtr_mlp.hip TAIL)
                    projs, fc1s, fc2s = [b.mixer.proj for b in blks], [b.mlp.fc1 for b in blks], [b.mlp.fc2 for b in blks]

                    def build_tail():
                        wp, sp = ops.pack_weights_hl32([m.weight.detach().contiguous().view(C, 1, 1, C) for m in projs])
                        pin = ops.mlp_hidden_permutation(C, x.device)
                        w1, s1 = ops.pack_weights_hl32([m.weight.detach().index_select(1, pin).contiguous().view(Ch, 1, 1, C) for m in fc1s])
                        ph = ops.mlp_hidden_permutation(Ch, x.device)
                        w2, s2 = ops.pack_weights_hl32([m.weight.detach().index_select(1, ph).contiguous().view(C, 1, 1, Ch) for m in fc2s])
                        return (wp, sp, torch.stack([m.bias.detach() for m in projs]).contiguous(), w1, s1,
                                torch.stack([m.bias.detach() for m in fc1s]).contiguous(), w2, s2,
                                torch.stack([m.bias.detach() for m in fc2s]).contiguous())
                    wp, sp, bp, w1, s1, b1_, w2, s2, b2_ = self._cached(name + ".tail", [t_ for m in projs + fc1s + fc2s for t_ in (m.weight, m.bias)],
                                                                        build_tail)
                    drop2 = self._drop_scales(blks, B, x.device)
                    g2, b2 = self._ln_params(name + ".ln2", [b.norm2 for b in blks])
                    br = ops.svtr_tail_fused(ctx_hl, x, G * rows, rows, G, C, wp, sp, bp, drop1, N, g2, b2, b0.norm2.eps, w1, s1, b1_, w2, s2, b2_)
                    return x, (br.view(G * B, N, C), drop2)
            else:
                t, _, y_hl = ops.add_layernorm_grouped(x, br, dr, N, g1, b1, rows, b0.norm1.eps, want_sum=br is not None)
                x = t if t is not None else x
                qkv = self._linear(name + ".qkv", y_hl, rows, C, [b.mixer.qkv.weight for b in blks],
                                   [b.mixer.qkv.bias for b in blks] if mixer.qkv.bias is not None else None)
                ctx_hl = ops.svtr_attention(qkv.view(G * B, N, 3 * C), mixer.num_heads, mixer.scale, mixer.mask, want_f32=False, want_hl=True,
                                            x3=ops.SVTR_ATTENTION_X3)
            br = self._linear(name + ".proj", ctx_hl, rows, C, [b.mixer.proj.weight for b in blks], [b.mixer.proj.bias for b in blks])
            drop2 = self._drop_scales(blks, B, x.device)
            g2, b2 = self._ln_params(name + ".ln2", [b.norm2 for b in blks])
            x, _, y_hl = ops.add_layernorm_grouped(x, br.view(G * B, N, C), drop1, N, g2, b2, rows, b0.norm2.eps, want_sum=True)
        Ch = blks[0].mlp.fc1.out_features
        if ops.SVTR_FUSED_MLP and (C in (64, 128) or (C == 256 and ops.SVTR_FUSED_MLP256)) and Ch == 4 * C and ops.X3_PRODUCTS == 3:
            # fc1 -> GELU -> fc2 in one kernel: the 4C-wide hidden tensor stays in registers (csrc/svtr_mlp.hip)
            fc1s, fc2s = [b.mlp.fc1 for b in blks], [b.mlp.fc2 for b in blks]

            def build():
                w1, s1 = ops.pack_weights_hl32([m.weight.detach().contiguous().view(Ch, 1, 1, C) for m in fc1s])
                perm = ops.mlp_hidden_permutation(Ch, x.device)
                w2, s2 = ops.pack_weights_hl32([m.weight.detach().index_select(1, perm).contiguous().view(C, 1, 1, Ch) for m in fc2s])
                return (w1, s1, torch.stack([m.bias.detach() for m in fc1s]).contiguous(), w2, s2,
                        torch.stack([m.bias.detach() for m in fc2s]).contiguous())
            w1, s1, b1_, w2, s2, b2_ = self._cached(name + ".mlp", [t_ for m in fc1s + fc2s for t_ in (m.weight, m.bias)], build)
            br = ops.svtr_mlp_fused(y_hl, G * rows, rows, G, C, w1, s1, b1_, w2, s2, b2_)
            return x, (br.view(G * B, N, C), drop2)
        # fc1 + GELU lands straight in the HL32 layout fc2 reads: the 4C-wide hidden tensor crosses HBM once each way
        hdn_hl = self._linear(name + ".fc1", y_hl, rows, C, [b.mlp.fc1.weight for b in blks], [b.mlp.fc1.bias for b in blks],
                              act=ops.ACT_GELU, hl_only=True)
        br = self._linear(name + ".fc2", hdn_hl, rows, Ch, [b.mlp.fc2.weight for b in blks], [b.mlp.fc2.bias for b in blks])
        return x, (br.view(G * B, N, C), drop2)

    def _svtr(self, image, last_hl=False):
        """image [B,H,W,C] fp32 (shared) -> features Act [G,B,1,W,out] (SVTR.forward_features, modules/svtr.py:500-528)"""
        G = self.G
        nets = [e.FeatureExtraction.ConvNet for e in self.experts]
        n0 = nets[0]
        B = image.shape[0]
        logical = image.permute(0, 3, 1, 2)                   # [B,C,H,W] view of the NHWC image: what PatchEmbed takes
        H, W = n0.HW
        N, C = H * W, n0.embed_dim[0]
        x = torch.empty(G * B, N, C, device=image.device, dtype=torch.float32)
        pes = [n.patch_embed for n in nets]
        if SVTR_GROUPED_EMBED and all(p_.proj[1].training for p_ in pes) and C % 32 == 0 and C >= 64:
            # PatchEmbed in lock-step: conv 1 (Cin = 4, stride 2: exact fp32 per expert into one stack) and conv 2 (Cin = 32 -> the grouped
            # split-fp16 x3 kernel) each followed by ONE grouped BatchNorm finalise + apply-with-GELU pass
            a = self.layer(Act((G, B, image.shape[1], image.shape[2], image.shape[3]), image, None, shared=True),
                           [p_.proj[0] for p_ in pes], [p_.proj[1] for p_ in pes], gelu=True)
            tok = self.layer(a, [p_.proj[3] for p_ in pes], [p_.proj[4] for p_ in pes], gelu=True, want_f32=True, want_hl=False).f32
            for g, n in enumerate(nets):
                ops.ew_rows(ops.EW_ADD, tok[g].view(B, N * C), n.pos_embed.view(1, N * C).expand(B, N * C), out=x[g * B:(g + 1) * B].view(B, N * C))
        else:
            for g, n in enumerate(nets):
                # PatchEmbed (Cin = 4 and 32: exact-fp32 convs + BatchNorm + GELU, as on the per-expert path) + position embedding
                tok = n.patch_embed(logical)
                ops.ew_rows(ops.EW_ADD, tok.view(B, N * C), n.pos_embed.view(1, N * C).expand(B, N * C), out=x[g * B:(g + 1) * B].view(B, N * C))
        self._draw_drop_bank(nets, B, image.device)
        for si in range(3):
            blocks = [list(getattr(n, "blocks%d" % (si + 1))) for n in nets]
            pending = None
            for i in range(len(blocks[0])):
                x, pending = self._svtr_block("svtr%d.%d" % (si, i), x, pending, [b[i] for b in blocks], B, N)
            # SubSample (:298-305): fold the last residual add, 3x3 stride-(2,1) conv, LayerNorm
            pb, pd = pending if pending is not None else (None, None)      # (None: the last block ran as one fused launch)
            _, _, hl = ops.add_layernorm_grouped(x, pb, pd, N, want_sum=False, want_f32=False, want_hl=True)
            subs = [getattr(n, "sub_sample%d" % (si + 1)) for n in nets]
            y = self.layer(Act((G, B, H, W, C), None, hl), [s.conv for s in subs], None, relu=False, want_f32=True, want_hl=False)
            _, _, H, W, C = y.shape
            N = H * W
            gm, bt = self._ln_params("svtr_sub%d" % si, [s.norm for s in subs])
            last = si == 2
            _, x, hl = ops.add_layernorm_grouped(y.f32.view(G * B, N, C), None, None, N, gm, bt, B * N, subs[0].norm.eps,
                                                 want_f32=not (last and last_hl), want_hl=last and last_hl)
        return Act((G, B, H, W, C), x.view(G, B, H, W, C) if x is not None else None, hl if last_hl else None)

    def _tps(self, image):
        """image [B,H,W,C] fp32 (shared) -> rectified images [G,B,H,W,C]"""
        G = self.G
        B, H, W, C = image.shape
        tps = [e.Transformation for e in self.experts]
        loc = [t.LocalizationNetwork for t in tps]
        pool = ((2, 2), (2, 2), (0, 0))
        x = Act((G, B, H, W, C), image, None, shared=True)
        # (the pooling pass in front of a 3x3 conv with Cin >= 128 writes the Winograd-domain operand: conv 3 and conv 4 run as F(4,3))
        w2, w3, w4 = (self.wino_for(loc[0].conv[i], [l.conv[i + 1] for l in loc]) if TPS_WINO else 0 for i in (4, 8, 12))
        x = self.layer(x, [l.conv[0] for l in loc], [l.conv[1] for l in loc], pool=pool, want_wino=w2, want_hl=not w2)
        x = self.layer(x, [l.conv[4] for l in loc], [l.conv[5] for l in loc], pool=pool, want_wino=w3, want_hl=not w3)
        x = self.layer(x, [l.conv[8] for l in loc], [l.conv[9] for l in loc], pool=pool, want_wino=w4, want_hl=not w4)
        x = self.layer(x, [l.conv[12] for l in loc], [l.conv[13] for l in loc], want_f32=True, want_hl=False)
        out = torch.empty(G, B, tps[0].I_r_size[0], tps[0].I_r_size[1], C, device=image.device, dtype=torch.float32)
        _, _, Hl, Wl, Cl = x.shape
        v = ops.avgpool_nhwc(x.f32.view(G * B, Hl, Wl, Cl))                          # [G*B, 512], one launch
        fc1s, fc2s = [l.localization_fc1[0] for l in loc], [l.localization_fc2 for l in loc]
        # the two fully connected layers stay on the exact-fp32 MFMA (the fiducials feed the ill-conditioned TPS grid,
        # DESIGN.md section 2), as ONE batched launch over the experts instead of G tiny ones
        v = self._batched_linear_f32("loc_fc1", v.view(G, B, Cl), fc1s, act=ops.ACT_RELU)
        cp = self._batched_linear_f32("loc_fc2", v, fc2s).view(G, B, loc[0].F, 2)
        for g, t in enumerate(tps):
            gg = t.GridGenerator
            ops.tps_grid_sample(image, cp[g], gg.inv_delta_C, gg.P_hat, t.I_r_size, out=out[g])
        return Act(tuple(out.shape), out, None)

    def visual_all(self, image, as_act=False):
        """image: logical [B,C,H,W] -> backbone features [G,B,T,C'] (the reference's permute + AdaptiveAvgPool + squeeze
        is the identity on the height-1 NHWC map).  as_act: return the [G,B,1,T,C'] Act holding only the HL32 operand
        (what HeadsGroup's first grouped Linear consumes)."""
        img = to_nhwc(image)
        B, H, W, C = img.shape
        self._nbt = []
        if self.experts[0].stages["Trans"] == "TPS":
            x = self._tps(img)
        else:
            x = Act((self.G, B, H, W, C), img, None, shared=True)
        feat = self.experts[0].stages["Feat"]
        if feat == "SVTR":
            for e in self.experts:
                require_no_grad(e.FeatureExtraction.ConvNet, "SVTR")
            x = self._svtr(img, as_act)
        else:
            x = self._resnet(x, as_act) if feat == "ResNet" else self._vgg(x, as_act)
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        G, B, Ho, Wo, Cf = x.shape
        if Ho != 1:
            raise NotImplementedError("HIP path expects a height-1 feature map (32x256 inputs); got H=%d" % Ho)
        return x if as_act else x.f32.view(G, B, Wo, Cf)


class SequenceGroup(_GroupedLinear):
    """SequenceModeling of G frozen extractors in lock-step (BiLSTM x 2, or the SVTR-style single Linear): every Linear is a
    grouped split-fp16 x3 GEMM (csrc/conv_x3.hip with a 1x1 kernel), every recurrence ONE launch for all extractors
    (mrn_lstm_layer_fwd_grouped_f32).  Used on its own by DERNet (frozen old extractors, reference modules/model.py:223-254)
    and as the first half of HeadsGroup."""

    def __init__(self, extractors):
        self.extractors = list(extractors)      # Model_Extractor modules
        self.G = len(self.extractors)
        self._cache = {}

    @staticmethod
    def sequence_supported(extractors):
        e0 = extractors[0]
        if len(extractors) < 2 or e0.stages["Seq"] not in ("BiLSTM", "None"):
            return False
        if any(e.stages != e0.stages for e in extractors):
            return False
        return ops.CONV_PRECISION in ("auto", "fp16x3")

    def _bilstm(self, idx, x_hl, rows_shape, K, **dst):
        """BidirectionalLSTM number idx of every extractor: x [G, B*T, K] (HL32) -> [G,B,T,256] fp32 (or the strided `out`)"""
        G = self.G
        B, T = rows_shape
        mods = [e.SequenceModeling[idx] for e in self.extractors]
        H = mods[0].hidden_size
        packed = [m._packed() for m in mods]                     # (w_ih [2*4H,in], w_hh frag-major [2,...], b_ih, b_hh)
        xproj = self._linear("ih%d" % idx, x_hl, B * T, K, [p[0] for p in packed], [p[2] for p in packed])
        w_hh = self._cached("hh%d" % idx, [p[1] for p in packed], lambda: torch.stack([p[1] for p in packed]).contiguous())
        b_hh = self._cached("bhh%d" % idx, [p[3] for p in packed], lambda: torch.stack([p[3] for p in packed]).contiguous())
        if ops.RECURRENT_X3:
            # recurrent product on the f16 MFMA: W_hh pre-split into a fragment-major fp16 stream (cached), h split in LDS
            def build():
                packs = [[ops.pack_fragment_major_h(w) for w in (m.rnn.weight_hh_l0, m.rnn.weight_hh_l0_reverse)] for m in mods]
                return (torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous(),
                        torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous())
            w_h, w_inv = self._cached("hh16_%d" % idx, [w for m in mods for w in (m.rnn.weight_hh_l0, m.rnn.weight_hh_l0_reverse)], build)
            rec = ops.lstm_layer_x3_grouped(xproj.view(G, B, T, 2 * 4 * H), w_h, w_inv, b_hh, H, 2)
        else:
            rec = ops.lstm_layer_grouped(xproj.view(G, B, T, 2 * 4 * H), w_hh, b_hh, H, 2)
        y = self._linear("lin%d" % idx, ops.split_hl32(rec), B * T, 2 * H, [m.linear.weight for m in mods],
                         [m.linear.bias for m in mods], **dst)
        return y if dst else y.view(G, B, T, -1)

    def sequence(self, visual, out=None, out_row_stride=0, out_group_stride=0):
        """visual: Act with the backbone features [G,B,1,T,C'] -> contextual features [G,B,T,hidden]; with `out` and the two
        strides (floats) extractor g's rows land at out + g * out_group_stride + row * out_row_stride instead"""
        _, B, _, T, Cf = visual.shape
        x_hl = visual.hl if visual.hl is not None else ops.split_hl32(visual.f32)
        dst = dict(out=out, out_row_stride=out_row_stride, out_group_stride=out_group_stride) if out is not None else {}
        if self.extractors[0].stages["Seq"] == "BiLSTM":
            y1 = self._bilstm(0, x_hl, (B, T), Cf)
            return self._bilstm(1, ops.split_hl32(y1), (B, T), y1.shape[-1], **dst)      # [G,B,T,hidden]
        lins = [e.SequenceModeling[0] for e in self.extractors]                      # Seq "None": one Linear (model.py sequence())
        y = self._linear("seq", x_hl, B * T, Cf, [l.weight for l in lins], [l.bias for l in lins], **dst)
        return y if out is not None else y.view(self.G, B, T, -1)


class HeadsGroup(SequenceGroup):
    """SequenceModeling + Prediction of G frozen experts in lock-step (BiLSTM x 2 + CTC Linear / teacher-forced attention
    decoder): every Linear is a grouped split-fp16 x3 GEMM (csrc/conv_x3.hip with a 1x1 kernel), every recurrence ONE
    launch for all experts (mrn_lstm_layer_fwd_grouped_f32 / mrn_attn_decoder_fwd_grouped_f32).  Replaces six concurrent
    streams of 16-32-workgroup launches whose overlap was left to the hardware queues."""

    def __init__(self, experts):
        self.experts = list(experts)        # Model modules
        super().__init__([e.model for e in self.experts])

    @staticmethod
    def supported(experts, is_train):
        e0 = experts[0]
        if not SequenceGroup.sequence_supported([e.model for e in experts]) or any(e.stages != e0.stages for e in experts):
            return False
        if e0.stages["Pred"] == "Attn" and not is_train:       # greedy decoding feeds argmax back step by step: per expert
            return False
        return True

    def run(self, visual, text, feats_out, logits_out):
        """visual: Act with the backbone features [G,B,1,T,C'] (HL32 and / or fp32); feats_out [B,T,G,hidden] (router
        layout, expert g -> slice [:, :, g, :]); logits_out: list of G [B,T_pred,C_g] views with padded rows."""
        G = self.G
        _, B, _, T, Cf = visual.shape
        feat = self.sequence(visual)                                              # [G,B,T,hidden]
        feats_out.copy_(feat.permute(1, 2, 0, 3))
        hidden = feat.shape[-1]
        feat_hl = ops.split_hl32(feat)
        heads = [e.Prediction for e in self.experts]
        if self.experts[0].stages["Pred"] == "CTC":
            src_hl, rows, K = feat_hl, B * T, hidden
            gens = heads
        else:
            S = self.experts[0].opt.batch_max_length + 1
            cells = [h.attention_cell for h in heads]
            D = cells[0].input_size
            Hproj = self._linear("i2h", feat_hl, B * T, hidden, [c.i2h.weight for c in cells], None).view(G, B, T, -1)
            E = heads[0].num_char_embeddings
            emb = torch.empty(G, B, S, E, device=feat.device, dtype=torch.float32)
            for g, h in enumerate(heads):
                ops.embed_gather(text[:, :S], h.char_embeddings.weight, h.num_class, out=emb[g])
            eproj = self._linear("emb", ops.split_hl32(emb), B * S, E, [c.rnn.weight_ih[:, D:] for c in cells],
                                 [c.rnn.bias_ih for c in cells]).view(G, B, S, -1)
            x3 = heads[0].x3_ok()
            packed = [h._packed_x3() if x3 else h._packed() for h in heads]    # fragment-major (w_h2h, w_ih[:, :D], w_hh[, w_inv])
            hid = ops.attn_decoder_grouped(feat, Hproj, eproj, [p[0] for p in packed], [c.h2h.bias for c in cells],
                                           [c.score.weight for c in cells], [p[1] for p in packed], [p[2] for p in packed],
                                           [c.rnn.bias_hh for c in cells], cells[0].hidden_size,
                                           w_inv=[p[3] for p in packed] if x3 else None)
            src_hl, rows, K = ops.split_hl32(hid), B * S, hid.shape[-1]
            gens = [h.generator for h in heads]
        per = rows * K * 4                                         # HL32 bytes of one expert's generator input
        for g, gen in enumerate(gens):
            out = logits_out[g]
            assert out.stride(2) == 1 and out.stride(0) == out.shape[1] * out.stride(1)
            self._linear("gen%d" % g, src_hl[g * per:(g + 1) * per], rows, K, [gen.weight], [gen.bias], out=out,
                         out_row_stride=out.stride(1))
