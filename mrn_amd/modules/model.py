"""Model containers with the reference's four-stage operator API (modules/model.py): Model_Extractor (:17-101),
Model (:105-199), MRNNet (:314-496).  Same constructor arguments (`opt`), attribute names, forward signatures,
return conventions and state_dict keys; the math runs on the HIP path.

MI355X-first differences that do not change results:
  * experts write their contextual features straight into one [B, P, I, C] buffer and their logits into
    16-byte-aligned padded rows, so torch.stack / pad-with-ones / permute / contiguous / sum of
    MRNNet.cross_forward (:399-423) collapse into the DM-Router kernels plus one fan-in pass;
  * the input image is converted to NHWC once and shared by all experts.
"""
import contextlib
import copy
import os

import torch
import torch.nn as nn

from .. import ops
from ..functional import FaninFn, GateTailFn, LinearFn, needs_grad
from ._nn import to_nhwc
from .dm_router import DM_Router
from .feature_extraction import ResNet_FeatureExtractor, VGG_FeatureExtractor
from .prediction import Attention
from .sequence_modeling import BidirectionalLSTM
from .transformation import TPS_SpatialTransformerNetwork



# Loop B: the heads of the frozen experts (BiLSTM x 2 + attention decoder / CTC Linear: latency-bound launches of 64-192 workgroups,
# 8 ms of a TRBA x 6 step) run on a stream of their own behind their backbones, so that the NEXT batch's backbones -- issued one step
# ahead by experts_prefetch -- start as soon as this batch's backbones are done instead of queueing behind its recurrences.
# MRN_HEADS_STREAM=0/1 is the A/B switch.
HEADS_STREAM = os.environ.get("MRN_HEADS_STREAM", "1") == "1"

class Model_Extractor(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.stages = {"Trans": opt.Transformation, "Feat": opt.FeatureExtraction, "Seq": opt.SequenceModeling,
                       "Pred": opt.Prediction}
        if opt.Transformation == "TPS":
            self.Transformation = TPS_SpatialTransformerNetwork(
                F=opt.num_fiducial, I_size=(opt.imgH, opt.imgW), I_r_size=(opt.imgH, opt.imgW),
                I_channel_num=opt.input_channel)
        else:
            print("No Transformation module specified")
        if opt.FeatureExtraction == "VGG":
            self.FeatureExtraction = VGG_FeatureExtractor(opt.input_channel, opt.output_channel)
        elif opt.FeatureExtraction == "ResNet":
            self.FeatureExtraction = ResNet_FeatureExtractor(opt.input_channel, opt.output_channel)
        elif opt.FeatureExtraction == "RCNN":
            from .feature_extraction import RCNN_FeatureExtractor
            self.FeatureExtraction = RCNN_FeatureExtractor(opt.input_channel, opt.output_channel)
        elif opt.FeatureExtraction == "SVTR":
            from .svtr import SVTR_FeatureExtractor
            self.FeatureExtraction = SVTR_FeatureExtractor(opt.input_channel, opt.output_channel)
        else:
            raise Exception("No FeatureExtraction module specified")
        self.FeatureExtraction_output = opt.output_channel
        self.AdaptiveAvgPool = nn.AdaptiveAvgPool2d((None, 1))
        if opt.SequenceModeling == "BiLSTM":
            self.SequenceModeling = nn.Sequential(
                BidirectionalLSTM(self.FeatureExtraction_output, opt.hidden_size, opt.hidden_size),
                BidirectionalLSTM(opt.hidden_size, opt.hidden_size, opt.hidden_size))
        else:
            self.SequenceModeling = nn.Sequential(nn.Linear(self.FeatureExtraction_output, opt.hidden_size))
            print("No SequenceModeling module specified")
        self.SequenceModeling_output = opt.hidden_size

    def visual(self, image):
        """Transformation + FeatureExtraction (+ the reference's permute / AdaptiveAvgPool / squeeze): [B,C,H,W] -> [B,T,C']"""
        with ops.batch_counters():                           # (the train-mode BatchNorm layers' num_batches_tracked: one launch)
            if not self.stages["Trans"] == "None":
                image = self.Transformation(image)
            fmap = self.FeatureExtraction(image)             # logical [B,C,H,W], NHWC memory
        B, C, H, W = fmap.shape
        if H != 1:
            raise NotImplementedError("HIP path expects a height-1 feature map (32x256 inputs); got H=%d" % H)
        # permute(0,3,1,2) + AdaptiveAvgPool((None,1)) + squeeze(3) of the reference is the identity on [B,1,W,C]
        return to_nhwc(fmap).view(B, W, C)

    def sequence(self, visual, out=None):
        """SequenceModeling stage: [B,T,C'] -> [B,T,hidden]; `out` = optional (strided) destination."""
        if self.stages["Seq"] == "BiLSTM":
            x = self.SequenceModeling[0](visual)
            return self.SequenceModeling[1](x, out=out)
        lin = self.SequenceModeling[0]
        if needs_grad(lin, visual):
            return LinearFn.apply(visual, lin.weight, lin.bias)
        return ops.linear(visual, lin.weight, lin.bias, out=out)

    def forward(self, image, out=None):
        """image [B,C,H,W] -> contextual feature [B,T,hidden]"""
        return self.sequence(self.visual(image), out=out)


class Model(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.model = Model_Extractor(opt)
        self.SequenceModeling_output = self.model.SequenceModeling_output
        self.stages = {"Pred": opt.Prediction}
        self.fc = None
        self.Prediction = None

    def reset_class(self, opt, device):
        if opt.Prediction == "CTC":
            self.Prediction = nn.Linear(self.SequenceModeling_output, opt.num_class)
        elif opt.Prediction == "Attn":
            self.Prediction = Attention(self.SequenceModeling_output, opt.hidden_size, opt.num_class,
                                        nn.Linear(opt.hidden_size, opt.num_class))
        else:
            raise Exception("Prediction is neither CTC or Attn")
        self.Prediction.to(device)

    def forward(self, image, text=None, is_train=True, feature_out=None, predict_out=None):
        return self.heads(self.model.visual(image), text, is_train, feature_out, predict_out)

    def heads(self, visual, text=None, is_train=True, feature_out=None, predict_out=None):
        """SequenceModeling + Prediction on precomputed backbone features"""
        feat = self.model.sequence(visual, out=feature_out)
        if self.stages["Pred"] == "CTC":
            if needs_grad(self.Prediction, feat):
                pred = LinearFn.apply(feat, self.Prediction.weight, self.Prediction.bias)
            else:
                pred = ops.linear(feat, self.Prediction.weight, self.Prediction.bias, out=predict_out)
        else:
            pred = self.Prediction(feat, text, is_train, batch_max_length=self.opt.batch_max_length, out=predict_out)
        return {"predict": pred, "feature": feat}

    def update_fc(self, hidden_size, nb_classes, device=None):
        fc = nn.Linear(hidden_size, nb_classes)
        if self.fc is not None:
            nb_output = self.fc.out_features
            fc = fc.to(self.fc.weight.device)
            fc.weight.data[:nb_output] = self.fc.weight.data
            fc.bias.data[:nb_output] = self.fc.bias.data
        self.fc = fc

    def new_fc(self, hidden_size, nb_classes):
        self.fc = nn.Linear(hidden_size, nb_classes)

    def weight_align(self, increment):
        """rescale the newest `increment` rows of fc so their mean L2 norm matches the old rows' (reference :166-174)"""
        gamma = ops.weight_align_(self.fc.weight.data, increment)
        torch.autograd.graph.increment_version(self.fc.weight)      # written through a raw pointer: invalidate repacked copies
        print("alignweights,gamma=", float(gamma))
        return gamma

    def build_prediction(self, opt, num_class):
        if opt.Prediction == "CTC":
            self.Prediction = self.fc
        elif opt.Prediction == "Attn":
            self.Prediction = Attention(self.SequenceModeling_output, opt.hidden_size, num_class, self.fc)
        else:
            raise Exception("Prediction is neither CTC or Attn")

    def copy(self):
        return copy.deepcopy(self)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        return self


class DERNet(Model):
    """Dynamically expanding network (reference modules/model.py:203-312): one Model_Extractor per task, features
    concatenated on the channel axis, main head over all of them, auxiliary head over the newest 256."""

    def __init__(self, opt):
        super().__init__(opt)
        self.model = nn.ModuleList()
        self.out_dim = None
        self.fc = None
        self.aux_fc = None
        self.task_sizes = []
        self.expert_grouping = True         # frozen extractors run their conv backbones in lock-step
        self.frozen_stream = True           # ... on their own HIP stream, side by side with the trained extractor's forward
        self._group = None
        self._side = None

    @property
    def feature_dim(self):
        return 0 if self.out_dim is None else self.out_dim * len(self.model)

    def _frozen_lockstep(self, image):
        """(frozen indices, BackboneGroup, SequenceGroup) when the frozen extractors can run conv backbones AND BiLSTMs in lock-step
        and occupy one contiguous channel range of the feature buffer; else None"""
        from . import expert_group
        trainable = [needs_grad(ext, image) for ext in self.model]
        frozen = [i for i, t in enumerate(trainable) if not t]
        if not (self.expert_grouping and len(frozen) >= 2):
            return None
        exts = [self.model[i] for i in frozen]
        key = tuple(id(e) for e in exts)
        if self._group is None or self._group[0] != key:
            self._group = (key, expert_group.BackboneGroup(exts), expert_group.SequenceGroup(exts))
        if not expert_group.supported(exts):
            return None
        if not (expert_group.SequenceGroup.sequence_supported(exts) and frozen == list(range(frozen[0], frozen[0] + len(frozen)))):
            return (frozen, self._group[1], None)
        return (frozen, self._group[1], self._group[2])

    def _run_frozen(self, image, frozen, bg, sg):
        """the frozen group's slices of a fresh [B,T,out_dim*N] feature buffer (current stream)"""
        with torch.no_grad():
            stack = bg.visual_all(image, as_act=True)
            _, B, _, T, _ = stack.shape
            buf = torch.empty(B, T, self.feature_dim, device=image.device, dtype=torch.float32)
            sg.sequence(stack, out=buf[:, :, frozen[0] * self.out_dim:], out_row_stride=self.feature_dim, out_group_stride=self.out_dim)
        return buf

    def frozen_prefetch(self, image):
        """Issue the FROZEN extractors' forward of a (future) batch on the side stream; pass the handle to
        forward(..., frozen=handle).  They are frozen and in eval mode (reference il_modules/der.py:101-104,137-141), so their
        features for batch n+1 do not depend on the update of batch n: only the launch order changes.  None when the frozen
        extractors cannot run in lock-step (the caller then simply does not pass a handle)."""
        image = to_nhwc(image).permute(0, 3, 1, 2)
        plan = self._frozen_lockstep(image)
        if plan is None or plan[2] is None or not (self.frozen_stream and image.is_cuda):
            return None
        if self._side is None:
            self._side = ops.aux_stream(0, image.device)
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf = self._run_frozen(image, *plan)
            done = torch.cuda.Event()
            done.record(side)
        image.record_stream(side)
        return {"buf": buf, "done": done, "batch": image.shape[0], "frozen": plan[0]}

    def _features(self, image, frozen_handle=None):
        """[B,T,out_dim*N]: frozen extractors write straight into their channel slice (no torch.cat pass); two or more
        frozen extractors of one architecture run conv backbones and BiLSTMs in lock-step (modules/expert_group.py), on a side
        stream next to the trained extractor's forward -- or earlier still, through frozen_prefetch()"""
        image = to_nhwc(image).permute(0, 3, 1, 2)
        trainable = [needs_grad(ext, image) for ext in self.model]
        visuals, seq_done, buf, join = {}, set(), None, None
        if frozen_handle is not None:
            assert frozen_handle["batch"] == image.shape[0]
            buf, join, seq_done = frozen_handle["buf"], frozen_handle["done"], set(frozen_handle["frozen"])
        else:
            plan = self._frozen_lockstep(image)
            if plan is not None and plan[2] is not None:
                # the frozen group is independent of the trained extractor until the heads: its (MFMA-bound) convolutions share
                # the chip with the trained extractor's forward, whose BatchNorm / pooling passes are HBM-bound
                if self.frozen_stream and any(trainable) and image.is_cuda:
                    handle = self.frozen_prefetch(image)
                    buf, join = handle["buf"], handle["done"]
                else:
                    buf = self._run_frozen(image, *plan)
                seq_done = set(plan[0])
            elif plan is not None:
                with torch.no_grad():
                    stack = plan[1].visual_all(image)
                visuals = {i: stack[k] for k, i in enumerate(plan[0])}
        trained = {i: ext(image) for i, ext in enumerate(self.model) if trainable[i]}      # (main stream; the side stream runs)
        if join is not None:
            main = torch.cuda.current_stream()
            main.wait_event(join)
            buf.record_stream(main)
        outs = []
        for i, ext in enumerate(self.model):
            if trainable[i]:
                outs.append(trained[i])
                continue
            with torch.no_grad():
                if i not in seq_done:
                    vis = visuals[i] if i in visuals else ext.visual(image)
                    if buf is None:
                        buf = torch.empty(vis.shape[0], vis.shape[1], self.feature_dim, device=vis.device, dtype=torch.float32)
                    ext.sequence(vis, out=buf[:, :, i * self.out_dim:(i + 1) * self.out_dim])
                outs.append(buf[:, :, i * self.out_dim:(i + 1) * self.out_dim])
        return buf if not any(trainable) else torch.cat(outs, -1)

    def _head(self, head, feat, text, is_train):
        if self.stages["Pred"] == "CTC":
            if needs_grad(head, feat):
                return LinearFn.apply(feat, head.weight, head.bias)
            return ops.linear(feat, head.weight, head.bias)
        return head(feat if feat.is_contiguous() else feat.contiguous(), text, is_train, batch_max_length=self.opt.batch_max_length)

    def forward(self, image, text=None, is_train=True, frozen=None):
        """frozen: optional handle of frozen_prefetch(image) (same batch)"""
        feat = self._features(image, frozen)
        logits = self._head(self.Prediction, feat, text, is_train)
        aux = self._head(self.aux_Prediction, feat[:, :, -self.out_dim:], text, is_train)
        return {"logits": logits, "aux_logits": aux, "features": feat}

    def update_fc(self, hidden_size, nb_classes, device=None):
        dev = next(self.parameters()).device if len(self.model) else None
        self.model.append(Model_Extractor(self.opt))
        if len(self.model) > 1:
            self.model[-1].load_state_dict(self.model[-2].state_dict())
        if self.out_dim is None:
            self.out_dim = self.model[-1].SequenceModeling_output
        fc = nn.Linear(self.feature_dim if self.opt.Prediction == "CTC" else self.out_dim, nb_classes)
        if self.fc is not None:                       # keep the old classes' rows over the old extractors' columns
            old_w, old_b = self.fc.weight.data.cpu(), self.fc.bias.data.cpu()
            ncol = min(old_w.shape[1], self.feature_dim - self.out_dim, fc.weight.shape[1])
            fc.weight.data[:old_w.shape[0], :ncol] = old_w[:, :ncol]
            fc.bias.data[:old_b.shape[0]] = old_b
        self.fc = fc
        self.aux_fc = nn.Linear(self.out_dim, nb_classes)
        if dev is not None:
            self.to(dev)

    def build_prediction(self, opt, num_class):
        if opt.Prediction == "CTC":
            self.Prediction = self.fc
        elif opt.Prediction == "Attn":
            self.Prediction = Attention(self.feature_dim, opt.hidden_size, num_class, self.fc)
        else:
            raise Exception("Prediction is neither CTC or Attn")
        if len(self.model) > 1:
            self.Prediction.to(next(self.model[0].parameters()).device)

    def build_aux_prediction(self, opt, num_class):
        if opt.Prediction == "CTC":
            self.aux_Prediction = self.aux_fc
        elif opt.Prediction == "Attn":
            self.aux_Prediction = Attention(self.SequenceModeling_output, opt.hidden_size, num_class, self.aux_fc)
        else:
            raise Exception("Prediction is neither CTC or Attn")
        if len(self.model) > 1:
            self.aux_Prediction.to(next(self.model[0].parameters()).device)

    def freeze_conv(self):
        for p in self.model.parameters():
            p.requires_grad = False
        self.model.eval()

    def copy(self):
        group, self._group = self._group, None               # packed-weight caches / the side stream are not copied
        side, self._side = self._side, None
        try:
            return copy.deepcopy(self)
        finally:
            self._group = group
            self._side = side


class MRNNet(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.model = nn.ModuleList()
        self.out_dim = None
        self.fc = None
        self.opt = opt
        self.task_sizes = []
        self.patch = {"VGG": 63, "SVTR": 64, "ResNet": 65}[opt.FeatureExtraction]
        self.router = "dm-router"
        self.layer_num = 1
        self.beta = 1
        self.expert_streams = True          # run the frozen experts on separate HIP streams
        self.expert_grouping = True         # run the frozen experts' conv backbones in lock-step (modules/expert_group.py)
        self._stream_pool = []
        self._group = None
        self._heads = None
        self._halves = None
        # concurrent lock-step sub-groups: -1 = by backbone family (convolutional experts: ONE group; SVTR: three groups of >= 2 from six
        # experts on, two from four on), 0 = off (one group on the caller's stream), k >= 2 = exactly k
        self.expert_halves = int(os.environ.get("MRN_EXPERT_HALVES", "-1"))
        #   (measured on TRBA x 6, MI355X: 1 group 1.00, 2 halves on two streams 1.044, 3 thirds 1.015, staggered halves 1.035,
        #    halves with one high-priority stream 0.96; with the final conv kernel and the loop-B pipeline: one group on a
        #    side stream 0.992 of two halves)

    def _streams(self, n, device):
        while len(self._stream_pool) < n:
            self._stream_pool.append(ops.aux_stream(len(self._stream_pool), device))      # (process-wide: see ops.aux_stream)
        return self._stream_pool

    def _backbone_group(self):
        """BackboneGroup over the current experts when they can run in lock-step, else None (per-expert path)"""
        from . import expert_group
        if not self.expert_grouping:
            return None
        extractors = [e.model for e in self.model]
        key = tuple(id(e) for e in extractors)
        if self._group is None or self._group[0] != key:
            self._group = (key, expert_group.BackboneGroup(extractors))
        return self._group[1] if expert_group.supported(extractors) else None

    def _half_groups(self, is_train):
        """[(lo, hi, BackboneGroup, HeadsGroup)] per sub-group when the experts are split into `expert_halves` lock-step
        groups that run on separate streams, else None.  Fewer than two experts per sub-group: ONE group on one side stream,
        so that the loop-B software pipeline (experts_prefetch) still overlaps the router phase with the next batch's experts."""
        from . import expert_group
        I = len(self.model)
        k = self.expert_halves
        if k < 0:
            # default: ONE lock-step group of all experts on one side stream.  Sub-groups on separate streams (MRN_EXPERT_HALVES=2 / 3) were
            # the default while the convolution kernels left CU resources to share; the row-block Winograd kernel owns its CU (144 KiB of
            # LDS, 453 registers), so sub-groups only add tile-quantisation tails: same box 90.7 -> 89.7 ms for six TRBA experts, and every
            # launch has the GPU to itself (in-situ rate of the dominant kernel 0.25 -> 0.325 of peak).  SVTR's kernels share a CU, and its
            # experts keep their sub-groups (same box, six experts: one group 23.5-23.6 ms, three 22.4-22.6)
            svtr = I > 0 and self.model[0].model.stages.get("Feat") == "SVTR"
            k = (3 if I >= 6 else 2) if svtr else I
        if k < 2 or I < 2:
            return None
        if I < 2 * k:
            k = 1
        key = tuple(id(e) for e in self.model) + (k,)
        if self._halves is None or self._halves[0] != key:
            cuts = [I * i // k for i in range(k + 1)]
            parts = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                models = list(self.model)[lo:hi]
                parts.append((lo, hi, expert_group.BackboneGroup([m.model for m in models]), expert_group.HeadsGroup(models)))
            self._halves = (key, parts)
        return self._halves[1]

    def _heads_group(self, group, is_train):
        """HeadsGroup (SequenceModeling + Prediction of all experts in lock-step) when `group` exists and the heads allow it"""
        from .expert_group import HeadsGroup
        if group is None or not HeadsGroup.supported(list(self.model), is_train):
            return None
        if self._heads is None or self._heads[0] is not group:
            self._heads = (group, HeadsGroup(list(self.model)))
        return self._heads[1]

    @property
    def feature_dim(self):
        return 0 if self.out_dim is None else self.out_dim * len(self.model)

    def forward(self, image, cross=True, text=None, is_train=True, experts=None):
        """`experts`: optional handle from experts_prefetch() -- the frozen experts' outputs for THIS batch, issued earlier"""
        if cross == False:  # noqa: E712  (learners pass cross positionally, exactly as in the reference)
            features, index = self.model[-1](image, text, is_train)["predict"], None
        elif is_train == False:  # noqa: E712
            features, index = self.cross_forward_expert(image, text, is_train)
        else:
            features, index = self.cross_forward(image, text, is_train, experts=experts)
        return {"logits": features, "index": index, "aux_logits": None}

    # -- shared by both routed paths -------------------------------------------------------------------
    def experts_prefetch(self, image, text=None, is_train=True):
        """Issue the FROZEN experts' forward for a batch on the side streams and return a handle for
        forward(..., experts=handle) -- or None when the experts cannot run as lock-step half-groups.

        In the router phase the experts do not depend on the router being trained, so a learner can issue batch n+1's
        expert forward before batch n's router forward / backward / Adam: the side streams then run it while the main
        stream is busy with the router.  The side streams wait only for the inputs (an event recorded after the NHWC
        conversion), every output lives in the issuing stream's allocator pool, and the consumer waits on per-half
        events, so nothing orders the prefetch behind the router work of the previous step."""
        I = len(self.model)
        with torch.no_grad():
            group = self._backbone_group() if I > 1 else None
            halves = self._half_groups(is_train) if self._heads_group(group, is_train) is not None else None
            if halves is None:
                return None
            B, dev = image.shape[0], image.device
            img = to_nhwc(image).permute(0, 3, 1, 2)
            T_pred = self.patch if self.opt.Prediction == "CTC" else self.opt.batch_max_length + 1
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            # the heads on a stream of their own (see HEADS_STREAM) when the experts run as ONE lock-step group (convolutional experts;
            # measured, same box: CRNN x 3 11.23 -> 10.80 ms, TRBA x 6 unchanged -- the row-block kernel is power-bound and the
            # recurrences take their CUs from it; SVTR's three sub-groups already interleave: 21.4 -> 21.9 with six streams, not split)
            split = HEADS_STREAM and len(halves) == 1
            streams = self._streams(len(halves) * (2 if split else 1), dev)
            # the router's [B,P,I,C] feature tensor is allocated ONCE (issuing stream's pool) and every sub-group writes its experts'
            # slice of it: no concatenation launch when the router phase adopts the outputs
            feats = torch.empty(B, self.patch, I, self.out_dim, device=dev, dtype=torch.float32)
            parts = []
            for k, (lo, hi, bg, hg) in enumerate(halves):
                st = streams[k]
                sh = streams[len(halves) + k] if split else st
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    visual = bg.visual_all(img, as_act=True)
                    if split:
                        grown = torch.cuda.Event()
                        grown.record(st)
                if split:
                    sh.wait_event(grown)
                    for t in (visual.f32, visual.hl):
                        if t is not None:
                            t.record_stream(sh)          # (allocated in the backbone stream's pool, read by the heads' stream)
                with torch.cuda.stream(sh):
                    lg = [ops.padded_rows(B, T_pred, e.fc.out_features, dev) for e in list(self.model)[lo:hi]]
                    hg.run(visual, text, feats[:, :, lo:hi, :], lg)
                    done = torch.cuda.Event()
                    done.record(sh)
                for t in (img, image, text, feats):
                    if t is not None:
                        t.record_stream(st)
                        if split:
                            t.record_stream(sh)
                parts.append((lg, done))
            return {"parts": parts, "batch": B, "feats": feats}

    def _experts_and_gate(self, image, text, is_train, experts=None):
        I = len(self.model)
        B = image.shape[0]
        dev = image.device
        if experts is not None:
            # outputs of experts_prefetch(): wait for the two half-groups, adopt their buffers on this stream
            assert experts["batch"] == B
            main = torch.cuda.current_stream()
            logits = []
            for lg, done in experts["parts"]:
                main.wait_event(done)
                for l in lg:
                    l.record_stream(main)
                logits += lg
            feats = experts["feats"]
            feats.record_stream(main)
            r = self.dm_router[0].forward_l2(feats)
            r = LinearFn.apply(r.view(B * self.patch, I * self.out_dim), self.channel_route.weight, self.channel_route.bias)
            return logits, r.view(B, self.patch, I)
        image = to_nhwc(image).permute(0, 3, 1, 2)          # one NHWC conversion shared by all experts
        T_pred = self.patch if self.opt.Prediction == "CTC" else self.opt.batch_max_length + 1
        feats = torch.empty(B, self.patch, I, self.out_dim, device=dev, dtype=torch.float32)
        logits = [ops.padded_rows(B, T_pred, expert.fc.out_features, dev) for expert in self.model]
        with torch.no_grad():
            group = self._backbone_group() if I > 1 else None
            heads = self._heads_group(group, is_train)
            halves = self._half_groups(is_train) if heads is not None else None
            if halves is not None:
                # two lock-step half-groups on two HIP streams: while one half's convolutions keep the matrix pipes busy,
                # the other half's HBM-bound BatchNorm / pooling passes and latency-bound recurrences share the CUs
                main = torch.cuda.current_stream()
                streams = self._streams(len(halves), dev)
                for (lo, hi, bg, hg), st in zip(halves, streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        hg.run(bg.visual_all(image, as_act=True), text, feats[:, :, lo:hi, :], logits[lo:hi])
                for st in streams[:len(halves)]:
                    main.wait_stream(st)
                    image.record_stream(st)
            elif heads is not None:
                # backbones AND heads in lock-step: one grouped launch per conv layer / Linear / recurrence, one stream
                heads.run(group.visual_all(image, as_act=True), text, feats, logits)
            elif self.expert_streams and I > 1:
                # Phase 1, one stream: the conv backbones (grouped when the experts allow it, else one after the other).
                # Phase 2, one HIP stream per expert: BiLSTM / attention decoder are latency-bound launches of 16-32
                # workgroups each; the experts' recurrences run side by side instead of idling 90 % of the CUs.
                if group is not None:
                    stack = group.visual_all(image)                      # [I,B,T,C']
                    visuals = [stack[i] for i in range(I)]
                else:
                    visuals = None                                       # (SVTR: many small launches per expert -- whole
                main = torch.cuda.current_stream()                       #  experts run side by side, one stream each)
                streams = self._streams(I, dev)
                for i, expert in enumerate(self.model):
                    streams[i].wait_stream(main)
                    with torch.cuda.stream(streams[i]):
                        if visuals is None:
                            expert(image, text, is_train, feature_out=feats[:, :, i, :], predict_out=logits[i])
                        else:
                            expert.heads(visuals[i], text, is_train, feature_out=feats[:, :, i, :], predict_out=logits[i])
                            visuals[i].record_stream(streams[i])
                for st in streams[:I]:
                    main.wait_stream(st)
                    image.record_stream(st)
            else:
                for i, expert in enumerate(self.model):
                    expert(image, text, is_train, feature_out=feats[:, :, i, :], predict_out=logits[i])
        r = self.dm_router[0].forward_l2(feats)               # [B,P,I,C]
        r = LinearFn.apply(r.view(B * self.patch, I * self.out_dim), self.channel_route.weight, self.channel_route.bias)
        return logits, r.view(B, self.patch, I)

    def cross_forward(self, image, text=None, is_train=True, experts=None):
        for expert in self.model:
            if torch.is_grad_enabled() and any(p.requires_grad for p in expert.parameters()):
                raise NotImplementedError("cross_forward trains the router over FROZEN experts (il_modules/mrn.py:154-157,"
                                          "285-286); unfreeze is not supported on the HIP path")
        logits, r = self._experts_and_gate(image, text, is_train, experts=experts)
        w = GateTailFn.apply(r, self.route.weight, self.route.bias, float(self.beta))
        return FaninFn.apply(w, *logits), w

    def cross_forward_expert(self, image, text=None, is_train=True):
        with torch.no_grad():
            logits, r = self._experts_and_gate(image, text, is_train)
            _, index = ops.gate_tail_fwd(r.contiguous(), self.route.weight.view(-1), self.route.bias, float(self.beta), hard=True)
            return ops.select_expert(logits, index), index

    def build_fc(self, hidden_size, nb_classes):
        self.update_fc(hidden_size, nb_classes)

    def update_fc(self, hidden_size, nb_classes):
        dev = next(self.parameters()).device if len(self.model) else None
        self.model.append(Model(self.opt))
        self.model[-1].new_fc(hidden_size, nb_classes)
        if self.out_dim is None:
            self.out_dim = self.model[-1].SequenceModeling_output
        # the whole router is re-created for the new expert count (reference :437-452)
        self.route = nn.Linear(self.patch, 1)
        self.channel_route = nn.Linear(self.feature_dim, len(self.model))
        block = DM_Router(self.out_dim, self.out_dim * 2, self.patch, len(self.model))
        print("mlp {} has {} layers".format(block, self.layer_num))
        self.dm_router = nn.Sequential(*[block for _ in range(self.layer_num)])
        if dev is not None:
            self.to(dev)

    def build_prediction(self, opt, num_class):
        if opt.Prediction == "CTC" or opt.Prediction == "Attn":
            self.model[-1].build_prediction(opt, num_class)
            if len(self.model) > 1:
                self.model[-1].to(next(self.model[0].parameters()).device)
        else:
            raise Exception("Prediction is neither CTC or Attn")

    def copy(self):
        pool, self._stream_pool = self._stream_pool, []      # streams / packed-weight caches are not copied
        group, self._group = self._group, None
        heads, self._heads = self._heads, None
        halves, self._halves = self._halves, None
        try:
            return copy.deepcopy(self)
        finally:
            self._stream_pool = pool
            self._group = group
            self._heads = heads
            self._halves = halves

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        return self

    def softargmax1d(self, input, beta=5):
        """softmax(beta * input) over the last axis (reference :495-496).  cross_forward has it fused into the gate-tail kernel
        together with the `route` Linear; this stand-alone form runs the same kernel with a one-tap identity route."""
        shape = input.shape
        x = input.reshape(-1, 1, shape[-1])
        one = torch.ones(1, 1, device=input.device, dtype=torch.float32)
        zero = torch.zeros(1, device=input.device, dtype=torch.float32)
        return GateTailFn.apply(x, one, zero, float(beta)).view(shape)
