"""validation(): greedy-decode evaluation with the reference's return convention and scoring rules (reference
test.py:139-279).  Accuracy = exact match against the RAW label string (a predicted [UNK] never counts as correct,
test.py:232-236), norm_ED = ICDAR-2019 normalised edit distance, confidence = product of the per-step max-probabilities.
(the forward / decoding path runs on the HIP kernels; string scoring is host work as in the reference.)

Reference quirks reproduced on purpose (pinned by tests/golden/validation.npz):
  * attention head: the prediction is cut at prd.find("[EOS]"); when there is NO [EOS] find() returns -1, so the LAST
    character (and the last probability) is dropped (test.py:224-226);
  * the returned strings / confidences / labels are those of the LAST batch only (test.py:270-279);
  * an empty pruned prediction has confidence 0 (the reference's bare `except`, test.py:262-265).
"""
import time

import numpy as np
import torch

from . import functional as Fn
from . import ops


def edit_distance(a, b):
    """Levenshtein distance (the reference uses nltk.metrics.distance.edit_distance with its default unit costs)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _forward(model, image, opt, converter, val_choose):
    """the reference's call patterns (test.py:163-201): "FF" = newest expert, "TF" = routed ensemble, else a plain Model"""
    if "CTC" in opt.Prediction:
        if val_choose == "FF":
            out = model(image, cross=False, is_train=False)
        elif val_choose == "TF":
            out = model(image, cross=True, is_train=False)
        else:
            out = model(image, is_train=False)
    else:
        sos = torch.full((image.size(0),), converter.dict["[SOS]"], dtype=torch.long, device=image.device)
        if val_choose == "FF":
            out = model(image, cross=False, text=sos, is_train=False)
        elif val_choose == "TF":
            out = model(image, cross=True, text=sos, is_train=False)
        else:
            out = model(image, text=sos, is_train=False)
    return out["logits"] if "logits" in out else out["predict"]


def validation(model, criterion, evaluation_loader, converter, opt, val_choose="val", tqdm_position=1):
    n_correct, norm_ED, length_of_data, infer_time = 0, 0.0, 0, 0.0
    loss_sum, loss_n = 0.0, 0
    preds_str, confidence_score_list, labels = [], [], []
    params = list(model.parameters()) if hasattr(model, "parameters") else []
    dev = params[0].device if params else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    attn = "Attn" in opt.Prediction
    for image_tensors, labels in evaluation_loader:
        batch_size = image_tensors.size(0)
        length_of_data += batch_size
        image = image_tensors.to(dev)
        labels_index, labels_length = converter.encode(labels, batch_max_length=opt.batch_max_length)
        if image.is_cuda:
            torch.cuda.synchronize()                 # the launches are asynchronous: infer_time is forward time, not host issue time
        start = time.time()
        preds = _forward(model, image, opt, converter, val_choose)
        if image.is_cuda:
            torch.cuda.synchronize()
        infer_time += time.time() - start
        if criterion is not None:               # the learners' Criterion (il_modules/base.py): CTC or CE(ignore [PAD]) :178-207
            cost = criterion(preds, labels_index, labels_length)
        elif attn:
            cost = Fn.cross_entropy(preds, labels_index[:, 1:], converter.dict["[PAD]"])
        else:
            cost = Fn.ctc_loss(preds.contiguous() if preds.stride(-1) != 1 else preds, labels_index, labels_length)
        loss_sum += float(cost)
        loss_n += 1
        preds_index, preds_max_prob = ops.argmax_prob_lastdim(preds)                         # :211, :218-219
        preds_str = converter.decode(preds_index.cpu().numpy(), [preds.size(1)] * batch_size)
        probs = preds_max_prob.cpu().numpy()
        confidence_score_list = []
        for gt, prd, prd_max_prob in zip(labels, preds_str, probs):
            if attn:
                eos = prd.find("[EOS]")
                prd = prd[:eos]                 # find() == -1 (no [EOS]): drops the last character, as the reference does
                prd_max_prob = prd_max_prob[:eos]
            if getattr(opt, "NED", False):
                if len(gt) == 0 or len(prd) == 0:
                    norm_ED += 0
                elif len(gt) > len(prd):
                    norm_ED += 1 - edit_distance(prd, gt) / len(gt)
                else:
                    norm_ED += 1 - edit_distance(prd, gt) / len(prd)
            if prd == gt:
                n_correct += 1
            confidence_score_list.append(float(np.cumprod(prd_max_prob.astype(np.float32))[-1]) if len(prd_max_prob) else 0)
    ned_score = norm_ED / float(length_of_data) * 100 if getattr(opt, "NED", False) else None
    score = n_correct / float(length_of_data) * 100
    valid_loss = loss_sum / max(loss_n, 1)
    return valid_loss, score, ned_score, preds_str, confidence_score_list, labels, infer_time, length_of_data
