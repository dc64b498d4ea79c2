"""validation(): greedy-decode evaluation with the reference's return convention (reference test.py:139-279).
Accuracy = exact match, norm_ED = ICDAR-2019 normalised edit distance; confidence = product of max-probabilities.
("next" row -- the forward/decoding path runs on the HIP kernels.)"""
import time

import torch

from . import functional as Fn
from . import ops


def edit_distance(a, b):
    """Levenshtein distance (the reference uses nltk.metrics.distance.edit_distance)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def validation(model, criterion, evaluation_loader, converter, opt, val_choose="val"):
    n_correct, norm_ED, length_of_data, infer_time = 0, 0.0, 0, 0.0
    loss_sum, loss_n = 0.0, 0
    preds_str_all, labels_all, conf_all = [], [], []
    dev = next(model.parameters()).device
    for image_tensors, labels in evaluation_loader:
        batch_size = image_tensors.size(0)
        length_of_data += batch_size
        image = image_tensors.to(dev)
        labels_index, labels_length = converter.encode(labels, batch_max_length=opt.batch_max_length)
        start = time.time()
        if "CTC" in opt.Prediction:
            out = model(image, False) if val_choose == "FF" else model(image, True, None, False) if val_choose == "TF" else model(image)
            preds = out["logits"] if isinstance(out, dict) and "logits" in out else out["predict"]
            infer_time += time.time() - start
            cost = Fn.ctc_loss(preds.contiguous() if preds.stride(-1) != 1 else preds, labels_index, labels_length)
            preds_index = ops.argmax_lastdim(preds)
            preds_str = converter.decode(preds_index.cpu().numpy(), [preds.size(1)] * batch_size)
        else:
            sos = torch.LongTensor(batch_size).fill_(converter.dict["[SOS]"]).to(dev)
            if val_choose == "FF":
                out = model(image, False, sos, False)
            elif val_choose == "TF":
                out = model(image, True, sos, False)
            else:
                out = model(image, sos, False)
            preds = out["logits"] if "logits" in out else out["predict"]
            infer_time += time.time() - start
            target = labels_index[:, 1:]
            cost = Fn.cross_entropy(preds, target, converter.dict["[PAD]"])
            preds_index = ops.argmax_lastdim(preds)
            preds_str = converter.decode(preds_index.cpu().numpy(), [preds.size(1)] * batch_size)
            labels = converter.decode(labels_index[:, 1:].cpu().numpy(), labels_length.cpu().numpy())
        loss_sum += float(cost)
        loss_n += 1
        probs = torch.softmax(preds.float(), dim=2).max(dim=2)[0].cpu()
        for gt, pd, pmax in zip(labels, preds_str, probs):
            if "Attn" in opt.Prediction:
                gt = gt[: gt.find("[EOS]")] if "[EOS]" in gt else gt
                cut = pd.find("[EOS]")
                pmax = pmax[:cut] if cut >= 0 else pmax
                pd = pd[:cut] if cut >= 0 else pd
            n_correct += int(pd == gt)
            if len(gt) == 0 or len(pd) == 0:
                norm_ED += 0
            elif len(gt) > len(pd):
                norm_ED += 1 - edit_distance(pd, gt) / len(gt)
            else:
                norm_ED += 1 - edit_distance(pd, gt) / len(pd)
            conf_all.append(float(pmax.cumprod(dim=0)[-1]) if len(pmax) else 0.0)
            preds_str_all.append(pd)
            labels_all.append(gt)
    accuracy = n_correct / float(max(length_of_data, 1)) * 100
    norm_ED = norm_ED / float(max(length_of_data, 1)) * 100
    valid_loss = loss_sum / max(loss_n, 1)
    return valid_loss, accuracy, norm_ED, preds_str_all, conf_all, labels_all, infer_time, length_of_data
