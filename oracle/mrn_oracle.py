"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (mrn_amd/*).

A functional, state-dict-driven restatement (plain torch CPU fp32 ops) of MRN's per-step recognition-and-routing
path, following the reference file:line cited on each function.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it, and only as the checker / reported baseline.

Parity pin: the reference repository contains no tests or golden vectors for this path, so
this oracle is pinned against outputs of the reference itself, generated in the build container by
tests/golden/make_golden.py (which imports /root/reference) and committed as tests/golden/*.npz;
tests/test_oracle_golden.py replays them.

Every function takes `sd`, a flat {state_dict key: tensor} mapping in the reference's key layout, and a key
prefix `pre` (e.g. "model.0.").  BatchNorm running statistics inside `sd` are updated in place in training mode,
exactly as the reference modules do.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# TPS (modules/transformation.py)
# ---------------------------------------------------------------------------------------------------------
def tps_constants(num_fid, size_hw):
    """inv_delta_C [F+3,F+3] and P_hat [H*W,F+3] as float32 (built in float64).
    Follows modules/transformation.py:148-202 (_build_C, _build_inv_delta_C, _build_P, _build_P_hat)."""
    H, W = size_hw
    half = num_fid // 2
    xs = np.linspace(-1.0, 1.0, half)
    C = np.concatenate([np.stack([xs, -np.ones(half)], 1), np.stack([xs, np.ones(half)], 1)], 0)  # :148-156
    d = np.linalg.norm(C[:, None, :] - C[None, :, :], axis=2)
    np.fill_diagonal(d, 1.0)
    hat_c = (d ** 2) * np.log(d)  # :160-167
    F3 = num_fid + 3
    delta = np.zeros((F3, F3))
    delta[:num_fid, 0] = 1.0
    delta[:num_fid, 1:3] = C
    delta[:num_fid, 3:] = hat_c
    delta[num_fid:num_fid + 2, 3:] = C.T
    delta[num_fid + 2, 3:] = 1.0  # :169-176
    inv_delta = np.linalg.inv(delta)
    gx = (np.arange(-W, W, 2) + 1.0) / W
    gy = (np.arange(-H, H, 2) + 1.0) / H
    P = np.stack(np.meshgrid(gx, gy), axis=2).reshape(-1, 2)  # :180-190
    r = np.linalg.norm(P[:, None, :] - C[None, :, :], ord=2, axis=2)
    rbf = np.square(r) * np.log(r + 1e-6)  # :199-200 (eps = 1e-6, :121)
    p_hat = np.concatenate([np.ones((P.shape[0], 1)), P, rbf], axis=1)
    return torch.tensor(inv_delta).float(), torch.tensor(p_hat).float()


def _bn(sd, pre, x, training, eps=1e-5, momentum=0.1):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        training, momentum, eps)


def _bn_tick(sd, pre, training):
    k = pre + "num_batches_tracked"
    if training and k in sd:
        sd[k] += 1


def localization_forward(sd, pre, x, training):
    """LocalizationNetwork.forward, modules/transformation.py:102-112 (+ layers :60-87)."""
    for ci, bi, pool in ((0, 1, True), (4, 5, True), (8, 9, True), (12, 13, False)):
        x = F.conv2d(x, sd[f"{pre}conv.{ci}.weight"], None, 1, 1)
        x = F.relu(_bn(sd, f"{pre}conv.{bi}.", x, training))
        _bn_tick(sd, f"{pre}conv.{bi}.", training)
        if pool:
            x = F.max_pool2d(x, 2, 2)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    x = F.relu(F.linear(x, sd[pre + "localization_fc1.0.weight"], sd[pre + "localization_fc1.0.bias"]))
    x = F.linear(x, sd[pre + "localization_fc2.weight"], sd[pre + "localization_fc2.bias"])
    return x.view(x.shape[0], -1, 2)


def tps_forward(sd, pre, img, training, num_fid=20, return_aux=False):
    """TPS_SpatialTransformerNetwork.forward, modules/transformation.py:30-50, build_P_prime :204-216."""
    B, _, H, W = img.shape
    cp = localization_forward(sd, pre + "LocalizationNetwork.", img, training)
    inv_delta, p_hat = tps_constants(num_fid, (H, W))
    cz = torch.cat([cp, torch.zeros(B, 3, 2)], 1)
    T = torch.bmm(inv_delta.repeat(B, 1, 1), cz)
    grid = torch.bmm(p_hat.repeat(B, 1, 1), T).reshape(B, H, W, 2)
    out = F.grid_sample(img, grid, padding_mode="border", align_corners=True)
    return (out, cp, grid) if return_aux else out


# ---------------------------------------------------------------------------------------------------------
# backbones (modules/feature_extraction.py)
# ---------------------------------------------------------------------------------------------------------
def vgg_forward(sd, pre, x, training):
    """VGG_FeatureExtractor, modules/feature_extraction.py:19-47 (Sequential indices are the state_dict keys)."""
    p = pre + "ConvNet."
    x = F.max_pool2d(F.relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)), 2, 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], 1, 1)), 2, 2)
    x = F.relu(F.conv2d(x, sd[p + "6.weight"], sd[p + "6.bias"], 1, 1))
    x = F.relu(F.conv2d(x, sd[p + "8.weight"], sd[p + "8.bias"], 1, 1))
    x = F.max_pool2d(x, (2, 1), (2, 1))
    x = F.relu(_bn(sd, p + "12.", F.conv2d(x, sd[p + "11.weight"], None, 1, 1), training))
    _bn_tick(sd, p + "12.", training)
    x = F.relu(_bn(sd, p + "15.", F.conv2d(x, sd[p + "14.weight"], None, 1, 1), training))
    _bn_tick(sd, p + "15.", training)
    x = F.max_pool2d(x, (2, 1), (2, 1))
    return F.relu(F.conv2d(x, sd[p + "18.weight"], sd[p + "18.bias"], 1, 0))


def _grcl(sd, p, u, training, num_iteration=5):
    """GRCL.forward + GRCL_unit.forward, modules/feature_extraction.py:112-161"""
    def bn(name, x):
        y = _bn(sd, p + name + ".", x, training)
        _bn_tick(sd, p + name + ".", training)
        return y
    pad = sd[p + "wf_u.weight"].shape[-1] // 2
    wgf_u = F.conv2d(u, sd[p + "wgf_u.weight"])
    wf_u = F.conv2d(u, sd[p + "wf_u.weight"], None, 1, pad)
    x = F.relu(bn("BN_x_init", wf_u))
    for i in range(num_iteration):
        q = f"GRCL.{i}."
        G = torch.sigmoid(bn(q + "BN_gfu", wgf_u) + bn(q + "BN_grx", F.conv2d(x, sd[p + "wgr_x.weight"])))
        x = F.relu(bn(q + "BN_fu", wf_u) + bn(q + "BN_Gx", bn(q + "BN_rx", F.conv2d(x, sd[p + "wr_x.weight"], None, 1, pad)) * G))
    return x


def rcnn_forward(sd, pre, x, training):
    """RCNN_FeatureExtractor, modules/feature_extraction.py:50-97"""
    p = pre + "ConvNet."
    x = F.max_pool2d(F.relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)), 2, 2)
    x = F.max_pool2d(_grcl(sd, p + "3.", x, training), 2, 2)
    x = F.max_pool2d(_grcl(sd, p + "5.", x, training), 2, (2, 1), (0, 1))
    x = F.max_pool2d(_grcl(sd, p + "7.", x, training), 2, (2, 1), (0, 1))
    x = F.relu(_bn(sd, p + "10.", F.conv2d(x, sd[p + "9.weight"], None, 1, 0), training))
    _bn_tick(sd, p + "10.", training)
    return x


RESNET_LAYERS = (1, 2, 5, 3)  # feature_extraction.py:105


def _conv_bn(sd, p, conv, bn, x, training, stride=1, padding=1, relu=True):
    x = _bn(sd, p + bn + ".", F.conv2d(x, sd[p + conv + ".weight"], None, stride, padding), training)
    _bn_tick(sd, p + bn + ".", training)
    return F.relu(x) if relu else x


def _basic_block(sd, p, x, training):
    """BasicBlock.forward, modules/feature_extraction.py:184-199."""
    out = _conv_bn(sd, p, "conv1", "bn1", x, training)
    out = _conv_bn(sd, p, "conv2", "bn2", out, training, relu=False)
    res = x
    if (p + "downsample.0.weight") in sd:
        res = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], None, 1, 0), training)
        _bn_tick(sd, p + "downsample.1.", training)
    return F.relu(out + res)


def resnet_forward(sd, pre, x, training):
    """ResNet.forward, modules/feature_extraction.py:318-352."""
    p = pre + "ConvNet."
    x = _conv_bn(sd, p, "conv0_1", "bn0_1", x, training)
    x = _conv_bn(sd, p, "conv0_2", "bn0_2", x, training)
    x = F.max_pool2d(x, 2, 2)
    for i in range(RESNET_LAYERS[0]):
        x = _basic_block(sd, f"{p}layer1.{i}.", x, training)
    x = _conv_bn(sd, p, "conv1", "bn1", x, training)
    x = F.max_pool2d(x, 2, 2)
    for i in range(RESNET_LAYERS[1]):
        x = _basic_block(sd, f"{p}layer2.{i}.", x, training)
    x = _conv_bn(sd, p, "conv2", "bn2", x, training)
    x = F.max_pool2d(x, 2, (2, 1), (0, 1))
    for i in range(RESNET_LAYERS[2]):
        x = _basic_block(sd, f"{p}layer3.{i}.", x, training)
    x = _conv_bn(sd, p, "conv3", "bn3", x, training)
    for i in range(RESNET_LAYERS[3]):
        x = _basic_block(sd, f"{p}layer4.{i}.", x, training)
    x = _conv_bn(sd, p, "conv4_1", "bn4_1", x, training, stride=(2, 1), padding=(0, 1))
    x = _conv_bn(sd, p, "conv4_2", "bn4_2", x, training, stride=1, padding=0)
    return x


# ---------------------------------------------------------------------------------------------------------
# SVTR backbone (modules/svtr.py)
# ---------------------------------------------------------------------------------------------------------
SVTR_DIMS, SVTR_DEPTH, SVTR_HEADS = (64, 128, 256), (3, 6, 3), (2, 4, 8)
SVTR_MIXER = ["Local"] * 6 + ["Global"] * 6           # svtr.py:321-322
SVTR_DROP_PATH = np.linspace(0, 0.1, 12)              # svtr.py:382 (drop_path_rate=0.1)


def svtr_local_mask(H, W, hk=7, wk=11):
    """additive attention mask of the Local mixer, svtr.py:117-128: 0 inside the hk x wk window, -inf outside"""
    mask = torch.ones(H * W, H + hk - 1, W + wk - 1)
    for h in range(H):
        for w in range(W):
            mask[h * W + w, h:h + hk, w:w + wk] = 0.0
    m = mask[:, hk // 2:H + hk // 2, wk // 2:W + wk // 2].flatten(1)
    return torch.where(m < 1, m, torch.full_like(m, float("-inf")))


def _svtr_block(sd, p, x, heads, mask, training, drop_prob, masks):
    """Block.forward svtr.py:200-204 (pre-norm attention + MLP, DropPath on both branches, LayerNorm eps 1e-6)"""
    B, N, C = x.shape

    def droppath(y):
        if drop_prob == 0.0 or not training:
            return y
        keep = 1 - drop_prob
        m = masks.pop(0).view(B, 1, 1).to(y.dtype)      # injected Bernoulli draw (svtr.py:17-22)
        return y * (m / keep)

    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "mixer.qkv.weight"], sd[p + "mixer.qkv.bias"]).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q.matmul(k.permute(0, 1, 3, 2))
    if mask is not None:
        attn = attn + mask
    attn = F.softmax(attn, dim=-1)
    h = attn.matmul(v).permute(0, 2, 1, 3).reshape(B, N, C)
    x = x + droppath(F.linear(h, sd[p + "mixer.proj.weight"], sd[p + "mixer.proj.bias"]))
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + droppath(h)


def svtr_forward(sd, pre, x, training, masks=None):
    """SVTR.forward_features, modules/svtr.py:500-528 (via SVTR_FeatureExtractor, feature_extraction.py:724-732).
    masks: list of [B] 0/1 tensors consumed by the DropPath sites in order (needed in training mode)."""
    p = pre + "ConvNet."
    masks = list(masks) if masks is not None else []
    B = x.shape[0]
    for ci, bi in ((0, 1), (3, 4)):                     # PatchEmbed :227-233
        x = F.conv2d(x, sd[f"{p}patch_embed.proj.{ci}.weight"], sd[f"{p}patch_embed.proj.{ci}.bias"], 2, 1)
        x = F.gelu(_bn(sd, f"{p}patch_embed.proj.{bi}.", x, training))
        _bn_tick(sd, f"{p}patch_embed.proj.{bi}.", training)
    H, W = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2) + sd[p + "pos_embed"]
    blk = 0
    for stage in range(3):
        hw = (H >> stage, W)
        for i in range(SVTR_DEPTH[stage]):
            mask = svtr_local_mask(*hw) if SVTR_MIXER[blk] == "Local" else None
            x = _svtr_block(sd, f"{p}blocks{stage + 1}.{i}.", x, SVTR_HEADS[stage], mask, training, float(SVTR_DROP_PATH[blk]), masks)
            blk += 1
        x = x.transpose(1, 2).reshape(B, SVTR_DIMS[stage], hw[0], hw[1])            # SubSample :298-305
        x = F.conv2d(x, sd[f"{p}sub_sample{stage + 1}.conv.weight"], sd[f"{p}sub_sample{stage + 1}.conv.bias"], (2, 1), 1)
        x = x.flatten(2).transpose(1, 2)
        C = x.shape[-1]
        x = F.layer_norm(x, (C,), sd[f"{p}sub_sample{stage + 1}.norm.weight"], sd[f"{p}sub_sample{stage + 1}.norm.bias"], 1e-5)
    return x.permute(0, 2, 1).reshape(B, -1, H // 8, W)


# ---------------------------------------------------------------------------------------------------------
# sequence modelling and heads
# ---------------------------------------------------------------------------------------------------------
def _lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.LSTM (gate order i,f,g,o), as used by modules/sequence_modeling.py:7-9."""
    B, T, _ = x.shape
    Hd = w_hh.shape[1]
    h = torch.zeros(B, Hd)
    c = torch.zeros(B, Hd)
    xs = F.linear(x, w_ih, b_ih)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xs[:, t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out, 1)


def bilstm_forward(sd, pre, x):
    """BidirectionalLSTM.forward, modules/sequence_modeling.py:12-22."""
    r = pre + "rnn."
    fw = _lstm_dir(x, sd[r + "weight_ih_l0"], sd[r + "weight_hh_l0"], sd[r + "bias_ih_l0"], sd[r + "bias_hh_l0"], False)
    bw = _lstm_dir(x, sd[r + "weight_ih_l0_reverse"], sd[r + "weight_hh_l0_reverse"], sd[r + "bias_ih_l0_reverse"],
                   sd[r + "bias_hh_l0_reverse"], True)
    return F.linear(torch.cat([fw, bw], 2), sd[pre + "linear.weight"], sd[pre + "linear.bias"])


def attention_forward(sd, pre, batch_H, text, is_train, batch_max_length, gen_w, gen_b, return_hidden=False):
    """Attention.forward + AttentionCell.forward, modules/prediction.py:38-88,102-118."""
    a = pre + "attention_cell."
    emb = sd[pre + "char_embeddings.weight"]
    num_class = emb.shape[0]
    B = batch_H.shape[0]
    S = batch_max_length + 1
    Hd = sd[a + "h2h.weight"].shape[0]
    h = torch.zeros(B, Hd)
    c = torch.zeros(B, Hd)
    Hproj = F.linear(batch_H, sd[a + "i2h.weight"])

    def cell(h, c, tok):
        tok = torch.where(tok >= num_class, 0, tok)  # cut_unknown :35-36
        e = F.linear(torch.tanh(Hproj + F.linear(h, sd[a + "h2h.weight"], sd[a + "h2h.bias"]).unsqueeze(1)),
                     sd[a + "score.weight"])
        alpha = F.softmax(e, dim=1)
        ctx = torch.bmm(alpha.permute(0, 2, 1), batch_H).squeeze(1)
        xin = torch.cat([ctx, emb[tok]], 1)
        g = F.linear(xin, sd[a + "rnn.weight_ih"], sd[a + "rnn.bias_ih"]) + F.linear(h, sd[a + "rnn.weight_hh"], sd[a + "rnn.bias_hh"])
        i, f, gg, o = g.chunk(4, 1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        return torch.sigmoid(o) * torch.tanh(c2), c2

    if is_train:
        hs = []
        for s in range(S):
            h, c = cell(h, c, text[:, s])
            hs.append(h)
        hid = torch.stack(hs, 1)
        probs = F.linear(hid, gen_w, gen_b)
        return (probs, hid) if return_hidden else probs
    targets = text[0].expand(B)
    probs = torch.zeros(B, S, gen_w.shape[0])
    for s in range(S):
        h, c = cell(h, c, targets)
        step = F.linear(h, gen_w, gen_b)
        probs[:, s] = step
        targets = step.max(1)[1]
    return probs


# ---------------------------------------------------------------------------------------------------------
# recogniser containers (modules/model.py)
# ---------------------------------------------------------------------------------------------------------
class Cfg:
    """The `opt` fields the path reads (modules/model.py:21-80,144)."""

    def __init__(self, Transformation="None", FeatureExtraction="VGG", SequenceModeling="BiLSTM", Prediction="CTC",
                 num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                 batch_max_length=25):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def extractor_forward(sd, pre, cfg, image, training, masks=None):
    """Model_Extractor.forward, modules/model.py:82-101."""
    x = image
    if cfg.Transformation == "TPS":
        x = tps_forward(sd, pre + "Transformation.", x, training, cfg.num_fiducial)
    if cfg.FeatureExtraction == "VGG":
        x = vgg_forward(sd, pre + "FeatureExtraction.", x, training)
    elif cfg.FeatureExtraction == "ResNet":
        x = resnet_forward(sd, pre + "FeatureExtraction.", x, training)
    elif cfg.FeatureExtraction == "RCNN":
        x = rcnn_forward(sd, pre + "FeatureExtraction.", x, training)
    elif cfg.FeatureExtraction == "SVTR":
        x = svtr_forward(sd, pre + "FeatureExtraction.", x, training, masks)
    else:
        raise NotImplementedError(cfg.FeatureExtraction)
    x = x.permute(0, 3, 1, 2)                       # [b,c,h,w] -> [b,w,c,h]
    x = F.adaptive_avg_pool2d(x, (x.shape[2], 1)).squeeze(3)
    if cfg.SequenceModeling == "BiLSTM":
        x = bilstm_forward(sd, pre + "SequenceModeling.0.", x)
        x = bilstm_forward(sd, pre + "SequenceModeling.1.", x)
    else:
        x = F.linear(x, sd[pre + "SequenceModeling.0.weight"], sd[pre + "SequenceModeling.0.bias"])
    return x


def model_forward(sd, pre, cfg, image, text=None, is_train=True, training=True, masks=None):
    """Model.forward, modules/model.py:133-148 -> {"predict", "feature"}."""
    feat = extractor_forward(sd, pre + "model.", cfg, image, training, masks)
    if cfg.Prediction == "CTC":
        pred = F.linear(feat, sd[pre + "fc.weight"], sd[pre + "fc.bias"])
    else:
        pred = attention_forward(sd, pre + "Prediction.", feat, text, is_train, cfg.batch_max_length,
                                 sd[pre + "fc.weight"], sd[pre + "fc.bias"])
    return {"predict": pred, "feature": feat}


def dernet_forward(sd, cfg, n_extractors, image, text=None, is_train=True, training=True, old_eval=True):
    """DERNet.forward, modules/model.py:223-254: N extractors -> concat -> main head (+ aux head on the newest 256).
    old_eval: old extractors run in eval mode during training (der.py:39-44 model_eval_and_train)."""
    feats = []
    for i in range(n_extractors):
        tr = training and not (old_eval and i < n_extractors - 1)
        feats.append(extractor_forward(sd, f"model.{i}.", cfg, image, tr))
    feat = torch.cat(feats, -1)
    hid = feats[-1].shape[-1]
    if cfg.Prediction == "CTC":
        logits = F.linear(feat, sd["fc.weight"], sd["fc.bias"])
        aux = F.linear(feat[:, :, -hid:], sd["aux_fc.weight"], sd["aux_fc.bias"])
    else:
        logits = attention_forward(sd, "Prediction.", feat, text, is_train, cfg.batch_max_length, sd["fc.weight"], sd["fc.bias"])
        aux = attention_forward(sd, "aux_Prediction.", feat[:, :, -hid:].contiguous(), text, is_train, cfg.batch_max_length,
                                sd["aux_fc.weight"], sd["aux_fc.bias"])
    return {"logits": logits, "aux_logits": aux, "features": feat}


def kd_loss(pred, soft, T=2.0):
    """_KD_loss, il_modules/lwf.py:111-114"""
    return -1 * torch.mul(torch.softmax(soft / T, dim=1), torch.log_softmax(pred / T, dim=1)).sum() / pred.shape[0]


def weight_align_gamma(fc_weight, increment):
    """Model.weight_align, modules/model.py:166-174"""
    new = torch.norm(fc_weight[-increment:, :], p=2, dim=1)
    old = torch.norm(fc_weight[:-increment, :], p=2, dim=1)
    return torch.mean(old) / torch.mean(new)


# ---------------------------------------------------------------------------------------------------------
# DM-Router and MRN fan-in (modules/dm_router.py, modules/model.py:361-423)
# ---------------------------------------------------------------------------------------------------------
def dm_router_forward(sd, pre, x):
    """DM_Router.forward, modules/dm_router.py:50-67; x [B, I, P, C]."""
    B, I, P, C = x.shape
    short = x
    h = F.layer_norm(x, (C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    h = F.gelu(F.linear(h, sd[pre + "proj_1.weight"], sd[pre + "proj_1.bias"]))
    h = h.reshape(B, I * P, -1)
    u, v = h.chunk(2, dim=-1)                                                      # SpatialDomainGating :11-17
    v = F.layer_norm(v, (v.shape[-1],), sd[pre + "spatial_gating.norm.weight"], sd[pre + "spatial_gating.norm.bias"])
    v = F.linear(v.permute(0, 2, 1), sd[pre + "spatial_gating.proj.weight"], sd[pre + "spatial_gating.proj.bias"]).permute(0, 2, 1)
    h = F.linear(u * v, sd[pre + "proj_2.weight"], sd[pre + "proj_2.bias"])
    h = h.reshape(B, I, P, C) + short
    z = h.permute(0, 1, 3, 2).reshape(B, I * C, P)                                 # 'b d p c -> b (d c) p'
    v = F.layer_norm(z, (P,), sd[pre + "channel_gating.norm.weight"], sd[pre + "channel_gating.norm.bias"])  # :26-33
    v = F.linear(v.permute(0, 2, 1), sd[pre + "channel_gating.proj.weight"], sd[pre + "channel_gating.proj.bias"]).permute(0, 2, 1)
    z = z * v
    h = z.reshape(B, I, C, P).permute(0, 1, 3, 2)                                  # 'b (d c) p -> b d p c'
    return F.linear(h, sd[pre + "proj_3.weight"], sd[pre + "proj_3.bias"]) + short


def gate_forward(sd, feats, beta=1.0, hard=False):
    """Gate part of MRNNet.cross_forward / cross_forward_expert, modules/model.py:399-406 / :368-377.
    feats: list of I tensors [B, P, C].  Returns softmax weights [B, I] (or argmax index when hard)."""
    r = torch.stack(feats, 1)
    r = dm_router_forward(sd, "dm_router.0.", r)
    B, I, P, C = r.shape
    r = r.permute(0, 2, 1, 3).reshape(B, P, I * C)                                 # 'b h w c -> b w (h c)'
    r = F.linear(r, sd["channel_route.weight"], sd["channel_route.bias"])
    s = F.linear(r.permute(0, 2, 1).contiguous(), sd["route.weight"], sd["route.bias"]).squeeze(-1)
    if hard:
        return s.max(-1)[1]
    return F.softmax(beta * s, dim=-1)


def fanin(logits, w):
    """Weighted fan-in with ones-padding, modules/model.py:361-364,410-423."""
    C = logits[-1].shape[-1]
    padded = [torch.cat([l, torch.ones(*l.shape[:2], C - l.shape[-1])], -1) for l in logits]
    stack = torch.stack(padded, 0)                                                 # [I,B,T,C]
    return torch.sum((stack.permute(2, 3, 1, 0) * w).permute(2, 0, 1, 3).contiguous(), -1)


def select_expert(logits, index):
    """Hard routing, modules/model.py:383-395."""
    C = logits[-1].shape[-1]
    padded = torch.stack([torch.cat([l, torch.ones(*l.shape[:2], C - l.shape[-1])], -1) for l in logits], 0)
    return torch.stack([padded[ix][i] for i, ix in enumerate(index)], 0).contiguous()


def mrn_forward(sd, cfg, n_experts, image, cross=True, text=None, is_train=True, training=True, masks=None):
    """MRNNet.forward, modules/model.py:343-359.  masks: per-expert lists of DropPath draws (SVTR experts in train mode)."""
    if not cross:
        out = model_forward(sd, f"model.{n_experts - 1}.", cfg, image, text, is_train, training,
                            masks[n_experts - 1] if masks else None)["predict"]
        return {"logits": out, "index": None, "aux_logits": None}
    outs = [model_forward(sd, f"model.{i}.", cfg, image, text, is_train, training, masks[i] if masks else None)
            for i in range(n_experts)]
    feats = [o["feature"] for o in outs]
    logits = [o["predict"] for o in outs]
    if not is_train:
        idx = gate_forward(sd, feats, hard=True)
        return {"logits": select_expert(logits, idx), "index": idx, "aux_logits": None}
    w = gate_forward(sd, feats)
    return {"logits": fanin(logits, w), "index": w, "aux_logits": None}


# ---------------------------------------------------------------------------------------------------------
# losses and the optimiser step (il_modules/base.py:128-137,242-262; il_modules/mrn.py:338-371)
# ---------------------------------------------------------------------------------------------------------
def ctc_loss(preds, labels_index, labels_length):
    """CTC site: log_softmax(2).permute(1,0,2) -> CTCLoss(mean, zero_infinity) (mrn.py:250-252)."""
    B, T, _ = preds.shape
    lp = preds.log_softmax(2).permute(1, 0, 2)
    return F.ctc_loss(lp, labels_index, torch.IntTensor([T] * B), labels_length, blank=0, reduction="mean",
                      zero_infinity=True)


def attn_ce_loss(preds, labels_index, pad_index=1):
    """CE site: CrossEntropyLoss(ignore_index=[PAD]) on preds vs labels[:, 1:] (mrn.py:254-258)."""
    target = labels_index[:, 1:]
    return F.cross_entropy(preds.reshape(-1, preds.shape[-1]), target.reshape(-1), ignore_index=pad_index)


def mrn_step_loss(out, labels_index, labels_length, domain, prediction, pi=15.0):
    """loss = pi * loss_clf + CE(softmax weights, domain) (mrn.py:342-360)."""
    taski = F.cross_entropy(out["index"], domain)
    clf = ctc_loss(out["logits"], labels_index, labels_length) if prediction == "CTC" else attn_ce_loss(out["logits"], labels_index)
    return pi * clf + taski, clf, taski


def one_cycle_lr(step, total_steps, max_lr, pct_start=0.3, div_factor=20.0, final_div_factor=1000.0):
    """torch.optim.lr_scheduler.OneCycleLR (cos anneal, two phases) as configured in base.py:97-104 / mrn.py:77-84.
    Returns the lr in effect for optimiser step number `step` (0-based)."""
    initial = max_lr / div_factor
    min_lr = initial / final_div_factor
    up_end = float(pct_start * total_steps) - 1
    down_end = total_steps - 1

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    if step <= up_end:
        return cos(initial, max_lr, step / up_end)
    return cos(max_lr, min_lr, (step - up_end) / (down_end - up_end))


def clip_and_adam(params, grads, state, lr, step, max_norm=5.0, betas=(0.9, 0.999), eps=1e-8):
    """clip_grad_norm_(max_norm) then torch.optim.Adam.step (base.py:255-262); in place on params/state."""
    total = torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for p, g, st in zip(params, grads, state):
        g = g * coef
        st["m"].mul_(betas[0]).add_(g, alpha=1 - betas[0])
        st["v"].mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** step
        bc2 = 1 - betas[1] ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)
    return total


# ---------------------------------------------------------------------------------------------------------
# integer label path (tools/utils.py:10-143)
# ---------------------------------------------------------------------------------------------------------
class CTCConverter:
    """CTCLabelConverter, tools/utils.py:10-76: 0 = blank, then [PAD]=1, [UNK]=2, ' '=3, characters..."""

    def __init__(self, character):
        self.character = ["[CTCblank]", "[PAD]", "[UNK]", " "] + list(character)
        self.dict = {}
        for i, ch in enumerate(self.character[1:]):
            self.dict[ch] = i + 1          # later duplicates overwrite, as in the reference's loop

    def encode(self, words, batch_max_length=25):
        idx = np.full((len(words), batch_max_length), self.dict["[PAD]"], dtype=np.int64)
        for i, w in enumerate(words):
            ids = [self.dict.get(ch, self.dict["[UNK]"]) for ch in w]
            idx[i, :len(ids)] = ids
        return torch.from_numpy(idx), torch.IntTensor([len(w) for w in words])

    def decode(self, word_index, word_length):
        out = []
        for row, n in zip(word_index, word_length):
            chars = []
            for i in range(int(n)):
                k = int(row[i])
                if k != 0 and not (i > 0 and int(row[i - 1]) == k):
                    chars.append(self.character[k])
            out.append("".join(chars))
        return out


class AttnConverter:
    """AttnLabelConverter, tools/utils.py:79-143: [UNK]=0 [PAD]=1 [SOS]=2 [EOS]=3 ' '=4, characters..."""

    def __init__(self, character):
        self.character = ["[UNK]", "[PAD]", "[SOS]", "[EOS]", " "] + list(character)
        self.dict = {}
        for i, ch in enumerate(self.character):
            self.dict[ch] = i

    def encode(self, words, batch_max_length=25):
        L = batch_max_length + 1
        idx = np.full((len(words), L + 1), self.dict["[PAD]"], dtype=np.int64)
        idx[:, 0] = self.dict["[SOS]"]
        for i, w in enumerate(words):
            ids = [self.dict.get(ch, self.dict["[UNK]"]) for ch in w] + [self.dict["[EOS]"]]
            idx[i, 1:1 + len(ids)] = ids
        return torch.from_numpy(idx), torch.IntTensor([len(w) + 1 for w in words])

    def decode(self, word_index, word_length):
        return ["".join(self.character[int(k)] for k in row[:int(n)]) for row, n in zip(word_index, word_length)]


# ---------------------------------------------------------------------------------------------------------
# BASELINE config 5 extras: EWC Fisher / penalty (il_modules/ewc.py:120-167), LwF step loss (il_modules/lwf.py:63-87)
# ---------------------------------------------------------------------------------------------------------
def fisher_diagonal(grad_lists, fishermax=1e-4):
    """getFisherDiagonal, il_modules/ewc.py:128-167: mean over iterations of the squared gradients, clipped at fishermax.
    grad_lists: one list of per-parameter gradients per iteration."""
    n = len(grad_lists)
    out = []
    for per_param in zip(*grad_lists):
        f = sum(g.pow(2) for g in per_param) / n
        out.append(torch.min(f, torch.tensor(fishermax)))
    return out


def fisher_blend(old, new, alpha=0.5):
    """the positional blend of il_modules/ewc.py:48-55: new[i][:len(old[i])] = alpha * old[i] + (1 - alpha) * new[i][:len(old[i])]"""
    out = [n.clone() for n in new]
    for i, o in enumerate(old):
        k = len(o)
        out[i][:k] = alpha * o + (1 - alpha) * out[i][:k]
    return out


def ewc_penalty(fisher, params, mean):
    """compute_ewc as written (il_modules/ewc.py:120-126) for matching keys: sum F * (p[:len(mean)] - mean)^2 / 2"""
    loss = torch.zeros(())
    for f, p, m in zip(fisher, params, mean):
        loss = loss + torch.sum(f * (p[:len(m)] - m).pow(2)) / 2
    return loss


def lwf_step_loss(new_logits, old_logits, labels_index, labels_length, prediction, known, T=2.0, lamda=3.0):
    """loss = lamda * KD(new[:, s:known], old[:, s:known]) + loss_clf with s = 0 (CTC) / 1 (Attn), il_modules/lwf.py:63-87"""
    s = 0 if prediction == "CTC" else 1
    kd = kd_loss(new_logits.reshape(-1, new_logits.shape[-1])[:, s:known], old_logits.reshape(-1, old_logits.shape[-1])[:, s:known], T)
    clf = ctc_loss(new_logits, labels_index, labels_length) if prediction == "CTC" else attn_ce_loss(new_logits, labels_index)
    return lamda * kd + clf, kd, clf
