"""validation() (reference test.py:139-279) against fixtures produced by the reference's own validation():
hand-made logits that hit every scoring branch, and real CRNN / TRBA recognisers (forward + greedy decode + scoring)."""
import contextlib
import io
import types

import numpy as np
import pytest
import torch

from mrn_amd.tools import weights as W
from tests.helpers import DetLoader, crafted_validation_case, load_golden

pytestmark = pytest.mark.gpu


def make_opt(kind):
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25, NED=True)
    if kind == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    return o


def converter_and_criterion(kind, chars):
    from mrn_amd.il_modules.base import Criterion
    from mrn_amd.tools.utils import AttnLabelConverter, CTCLabelConverter
    with contextlib.redirect_stdout(io.StringIO()):
        conv = CTCLabelConverter(chars) if kind == "crnn" else AttnLabelConverter(chars)
    return conv, Criterion("CTC" if kind == "crnn" else "Attn", None if kind == "crnn" else conv.dict["[PAD]"])


def check(g, pre, result, conf_rtol=2e-4):
    loss, acc, ned, preds, conf, labels, _, n = result
    assert n == int(g[pre + "length"])
    assert abs(acc - float(g[pre + "accuracy"])) < 1e-9 and abs(ned - float(g[pre + "ned"])) < 1e-9
    assert abs(float(loss) - float(g[pre + "valid_loss"])) <= 1e-4 * max(1.0, abs(float(g[pre + "valid_loss"])))
    assert list(preds) == [str(s) for s in g[pre + "preds_last_batch"]]
    ref_conf = g[pre + "confidence_last_batch"]
    assert len(conf) == len(ref_conf)
    for a, b in zip(conf, ref_conf):
        assert abs(float(a) - float(b)) <= conf_rtol * abs(float(b)) + 1e-37, (a, b)


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_validation_scoring_rules_vs_reference(kind):
    from mrn_amd.test import validation
    g = load_golden("validation")
    chars, batches, logits = crafted_validation_case(kind)
    conv, crit = converter_and_criterion(kind, chars)
    opt = make_opt(kind)
    calls = iter(logits)
    stub = lambda image, *a, **k: {"predict": next(calls).cuda(), "feature": None}     # noqa: E731
    res = validation(stub, crit, batches, conv, opt)
    check(g, f"crafted/{kind}/", res)
    assert [str(s) for s in res[5]] == [str(s) for s in g[f"crafted/{kind}/labels_last_batch"]]
    for i in range(len(batches)):
        one = iter([logits[i]])
        r = validation(lambda image, *a, **k: {"predict": next(one).cuda(), "feature": None}, crit, [batches[i]], conv, opt)
        pre = f"crafted/{kind}/batch{i}/"
        assert abs(r[1] - float(g[pre + "accuracy"])) < 1e-9 and abs(r[2] - float(g[pre + "ned"])) < 1e-9
        assert list(r[3]) == [str(s) for s in g[pre + "preds"]]
        for a, b in zip(r[4], g[pre + "confidence"]):
            assert abs(float(a) - float(b)) <= 2e-4 * abs(float(b)) + 1e-37
    # NED switched off: the reference returns None for it (test.py:266-268)
    opt.NED = False
    calls = iter(logits)
    assert validation(stub, crit, batches, conv, opt)[2] is None


@pytest.mark.parametrize("kind", ["crnn", "trba"])
def test_validation_of_a_real_recogniser_vs_reference(kind):
    """a single recogniser (reference `Model`) in eval mode through validation(val_choose="val"): CTC head with greedy collapse,
    attention head with 26 greedy decoding steps; strings bit-exact, loss / confidences within fp32 tolerance"""
    from mrn_amd.modules.model import Model
    from mrn_amd.test import validation
    g = load_golden("validation")
    opt = make_opt(kind)
    classes, seed = (40, 51) if kind == "crnn" else (41, 52)
    chars = "".join(chr(0x4E00 + i) for i in range(36))
    with contextlib.redirect_stdout(io.StringIO()):
        net = Model(opt)
        net.update_fc(opt.hidden_size, classes)
        net.build_prediction(opt, classes)
    W.fill_state_dict(net.state_dict(), seed)
    with torch.no_grad():
        net.fc.weight *= 60.0
    net = net.cuda().eval()
    conv, crit = converter_and_criterion(kind, chars)
    loader = DetLoader(3, f"validation:{kind}", seed, oov=True, n_valid=2)
    loader.set_characters(chars)
    with torch.no_grad():
        res = validation(net, crit, loader.create_dataset(), conv, opt)
    check(g, f"model/{kind}/", res, conf_rtol=5e-3)
