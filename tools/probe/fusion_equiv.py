import sys, torch
sys.path.insert(0, '/root/repo')
import tests.test_model_gpu as T
from mrn_amd import functional as Fn, ops
kind, classes, B, seed = "svtr", (40,), 8, 9
g = T.load_golden("svtr_mrn3")
image, words, chars, _ = T.det_inputs(kind, classes, B, seed)
conv, li, ll = T.labels_for(kind, words, chars)
outs = []
for fused in (True, False, True, False):
    ops.TRAIN_OPERAND_FUSION = fused
    opt, net = T.build_net(kind, (40, 70, 97), g, 3)
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad = n.startswith("model.0.")
    T.set_drop_masks(net, kind, B, seed, "fusion", [0])
    preds = net.model[0](image.cuda(), None, True)["predict"]
    loss = Fn.ctc_loss(preds, li.cuda(), ll.cuda())
    loss.backward()
    outs.append((preds.detach().clone(), float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
def cmp(a, b, tag):
    worst = max(((float((a[2][n] - b[2][n]).abs().max()) / (float(b[2][n].abs().max()) + 1e-30)), n) for n in b[2])
    print(tag, 'logits', float((a[0] - b[0]).abs().max()) / float(b[0].abs().max()), 'loss', abs(a[1] - b[1]), 'worst grad', worst)
cmp(outs[0], outs[2], 'fused vs fused')
cmp(outs[1], outs[3], 'unfused vs unfused')
cmp(outs[0], outs[1], 'fused vs unfused')
