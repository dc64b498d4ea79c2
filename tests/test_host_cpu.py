"""CPU-only checks of the host side: C-ABI surface, converters, LR schedule, state_dict layout, data-parallel helpers
(gloo, world_size 2).  No kernel is launched here."""
import contextlib
import ctypes
import io
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from tests.helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mrn_amd import _lib
    from mrn_amd.build import build_library
    build_library(verbose=False)
    protos = _lib.parse_header()
    assert len(protos) >= 39
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/mrn_hip.h but not exported"
    _lib.LIB.load()
    assert _lib.LIB._dll.mrn_version() == 100


def test_ops_refuse_cpu_tensors():
    from mrn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))


def test_product_converters_match_reference():
    from mrn_amd.tools.utils import AttnLabelConverter, CTCLabelConverter
    g = load_golden("converters")
    words = [str(w) for w in g["words"]]
    chars = str(g["chars"])
    with contextlib.redirect_stdout(io.StringIO()):
        c, a = CTCLabelConverter(chars), AttnLabelConverter(chars)
    idx, ln = c.encode(words, 25)
    assert np.array_equal(idx.cpu().numpy(), g["ctc/encode_idx"]) and np.array_equal(ln.cpu().numpy(), g["ctc/encode_len"])
    assert c.decode(g["ctc/decode_in"], [16, 16]) == [str(s) for s in g["ctc/decode_out"]]
    idx, ln = a.encode(words, 25)
    assert np.array_equal(idx.cpu().numpy(), g["attn/encode_idx"]) and np.array_equal(ln.cpu().numpy(), g["attn/encode_len"])
    assert a.decode(idx.cpu().numpy()[:, 1:], ln.cpu().numpy()) == [str(s) for s in g["attn/decode_out"]]


def test_one_cycle_matches_torch_scheduler():
    from mrn_amd.optim import OneCycle
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=0.0005)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.0005, cycle_momentum=False, div_factor=20,
                                              final_div_factor=1000, total_steps=200)
    mine = OneCycle(0.0005, 200)
    for step in range(200):
        assert abs(opt.param_groups[0]["lr"] - mine.lr_at(step)) < 1e-15, step
        opt.step()
        if step < 199:
            sch.step()
    with pytest.raises(ValueError):
        mine.lr_at(200)


@pytest.mark.parametrize("name,kind,classes", [("crnn_mrn3", "crnn", (40, 70, 97)), ("trba_mrn3", "trba", (41, 71, 98)),
                                               ("svtr_mrn3", "svtr", (40, 70, 97))])
def test_state_dict_layout_matches_reference(name, kind, classes):
    from mrn_amd.modules.model import MRNNet
    from mrn_amd.parallel import ReplicaDataParallel
    o = types.SimpleNamespace(num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                              batch_max_length=25)
    if kind == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    elif kind == "svtr":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "SVTR", "None", "CTC"
    else:
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    with contextlib.redirect_stdout(io.StringIO()):
        net = MRNNet(o)
        for c in classes:
            net.update_fc(256, c)
            net.build_prediction(o, c)
    g = load_golden(name)
    ref = {str(k): str(s) for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    mine = {k: ",".join(map(str, v.shape)) for k, v in net.state_dict().items()}
    assert mine == ref
    # the learners' checkpoints carry the DataParallel prefix (il_modules/base.py:323-332)
    assert all(k.startswith("module.") for k in ReplicaDataParallel(net).state_dict())
    # fc and Prediction[.generator] alias one tensor, as in the reference (model.py:181,185-187)
    m = net.model[0]
    head = m.Prediction if kind != "trba" else m.Prediction.generator
    assert head.weight.data_ptr() == m.fc.weight.data_ptr()


WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["MRN_ROOT"])
from mrn_amd import parallel
rank, world, local = parallel.init_distributed(backend="gloo")
assert world == 2 and parallel.world_size() == 2
g = torch.full((1001,), float(rank + 1))
parallel.average_gradients(g)
assert torch.allclose(g, torch.full((1001,), 1.5)), g[:4]
p = torch.arange(10, dtype=torch.float32) * (rank + 1)
parallel.broadcast_parameters(p)
assert torch.equal(p, torch.arange(10, dtype=torch.float32))
# per-rank shards differ, the averaged gradient and therefore the replicas stay identical
torch.manual_seed(rank)
shard_grad = torch.randn(64)
parallel.average_gradients(shard_grad)
gathered = [torch.zeros(64) for _ in range(2)]
torch.distributed.all_gather(gathered, shard_grad)
assert torch.equal(gathered[0], gathered[1])
parallel.barrier()
open(os.path.join(os.environ["MRN_OUT"], f"ok_{rank}"), "w").write("ok")     # stdout of the two ranks can interleave
"""


def test_data_parallel_helpers_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MRN_ROOT=ROOT, MRN_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists(), r.stdout + r.stderr


WORKER_BUCKETS = r"""
import os, sys, torch
sys.path.insert(0, os.environ["MRN_ROOT"])
from mrn_amd import parallel
from mrn_amd.optim import FlatAdam
rank, world, local = parallel.init_distributed(backend="gloo")
torch.manual_seed(100 + rank)                       # replicas start DIFFERENT: broadcast_module must equalise them
net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.BatchNorm1d(64), torch.nn.ReLU(), torch.nn.Linear(64, 64),
                          torch.nn.ReLU(), torch.nn.Linear(64, 7))
net[1].running_mean.add_(rank + 1.0)
net[1].num_batches_tracked.add_(3 * rank + 1)
unused = torch.nn.Parameter(torch.randn(33))         # a trainable parameter that never gets a gradient (router at task 0)
net.register_parameter("unused", unused)
parallel.broadcast_module(net)
state = torch.cat([t.reshape(-1).double() for t in list(net.parameters()) + list(net.buffers())])
both = [torch.zeros_like(state) for _ in range(2)]
torch.distributed.all_gather(both, state)
assert torch.equal(both[0], both[1]), "parameters / buffers differ after broadcast_module"
assert int(net[1].num_batches_tracked) == 1
opt = FlatAdam(list(net.parameters()), lr=1e-3)
red = parallel.BucketedAllReduce(opt, bucket_bytes=4 * 2000)     # several buckets
assert len(red.buckets) >= 3 and red.buckets[0][1] == opt.grad.numel() and red.buckets[-1][0] == 0
for step in range(2):
    torch.manual_seed(7 + 10 * step + rank)         # per-rank shards
    x, y = torch.randn(16, 40), torch.randn(16, 7)
    opt.zero_grad()
    red.begin()
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    local_grad = None
    red.finish()
    assert red.launched_log == list(range(len(red.buckets))), red.launched_log      # in order, every bucket exactly once
    g = opt.grad.clone()
    both = [torch.zeros_like(g) for _ in range(2)]
    torch.distributed.all_gather(both, g)
    assert torch.equal(both[0], both[1]), "averaged gradients differ between ranks"
    # the average really is the mean of the two ranks' local gradients
    opt.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    mine = opt.grad.clone()
    both = [torch.zeros_like(mine) for _ in range(2)]
    torch.distributed.all_gather(both, mine)
    assert torch.allclose(g, (both[0] + both[1]) / 2, atol=1e-7)
    assert float(opt.view_of(g, [i for i, q in enumerate(opt.params) if q is unused][0]).abs().max()) == 0.0   # unused parameter: zeros travel
# direct accumulation (ops.direct_gradients(notify=red.param_ready)): a backward function ADDS the weight gradients of the Linear layers
# into .grad itself and returns None, so no post-accumulate hook fires for them -- the reducer is told through the notify callback,
# once per parameter after its LAST use (the last Linear is applied twice: a recurrent layer's weight)
from mrn_amd import ops
class DirectLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.params = (w, b)
        ctx.use_gen = ops.note_param_uses(ctx.params, any(ctx.needs_input_grad))
        return x @ w.t() + b
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        assert ops.GRAD_DIRECT
        ctx.params[0].grad.add_(dy.t() @ x)
        accumulated[id(ctx.params[0])] = accumulated.get(id(ctx.params[0]), 0) + 1
        ops.direct_done((ctx.params[0],), ctx.use_gen)
        return dy @ w, None, dy.sum(0)                 # (the bias goes through autograd: hooks and notifications mix inside a bucket)
accumulated = {}
def notify_checked(p):
    # a parameter is reported complete only after ALL its uses of this graph have been accumulated (the last Linear: two)
    assert accumulated.get(id(p), 0) == (2 if p is net[3].weight else 1), "parameter reported complete before its last accumulation"
    red.param_ready(p)
def forward(x):
    h = torch.relu(net[1](DirectLinear.apply(x, net[0].weight, net[0].bias)))
    h = torch.relu(DirectLinear.apply(h, net[3].weight, net[3].bias))
    h = torch.relu(DirectLinear.apply(h, net[3].weight, net[3].bias))        # the same weight again
    return DirectLinear.apply(h, net[5].weight, net[5].bias)
for step in range(2):
    torch.manual_seed(70 + 10 * step + rank)
    x, y = torch.randn(16, 40), torch.randn(16, 7)
    opt.zero_grad()
    if step == 1:
        # a recorded forward whose backward never runs under direct_gradients (a Fisher pass through torch.autograd.grad, a
        # grad-enabled validation) must not inflate the counts of the step that follows it; the data-parallel wrapper starts a new
        # generation at every grad-enabled forward
        dropped = forward(torch.randn(16, 40))
        ops.new_use_generation()
    red.begin()
    accumulated.clear()
    with ops.direct_gradients(notify=notify_checked):
        ((forward(x) - y) ** 2).mean().backward()
    assert ops._GRAD_NOTIFY[0] is None
    # every bucket but the one holding the never-used parameter completed DURING backward
    assert red.next >= len(red.buckets) - 1, (red.next, len(red.buckets))
    red.finish()
    assert red.launched_log == list(range(len(red.buckets))), red.launched_log
    g = opt.grad.clone()
    opt.zero_grad()
    with ops.direct_gradients():
        ((forward(x) - y) ** 2).mean().backward()
    mine = opt.grad.clone()
    both = [torch.zeros_like(mine) for _ in range(2)]
    torch.distributed.all_gather(both, mine)
    assert torch.allclose(g, (both[0] + both[1]) / 2, atol=1e-7), (g - (both[0] + both[1]) / 2).abs().max()
# interleaved graphs -- forward A, forward B, backward A, backward B -- with the twice-used weight: every backward ticks the counts of ITS
# OWN forward generation (ADVICE r4: the exit after backward A used to clear B's counts, and the weight was then reported complete after
# the first of its two accumulations)
torch.manual_seed(300 + rank)
xa, ya, xb, yb = torch.randn(16, 40), torch.randn(16, 7), torch.randn(16, 40), torch.randn(16, 7)
loss_a = ((forward(xa) - ya) ** 2).mean()
ops.new_use_generation()
loss_b = ((forward(xb) - yb) ** 2).mean()
for loss, (xx, yy) in ((loss_a, (xa, ya)), (loss_b, (xb, yb))):
    opt.zero_grad()
    red.begin()
    accumulated.clear()
    with ops.direct_gradients(notify=notify_checked):
        loss.backward()
    assert red.next >= len(red.buckets) - 1, (red.next, len(red.buckets))
    red.finish()
    g = opt.grad.clone()
    opt.zero_grad()
    with ops.direct_gradients():
        ((forward(xx) - yy) ** 2).mean().backward()
    mine = opt.grad.clone()
    both = [torch.zeros_like(mine) for _ in range(2)]
    torch.distributed.all_gather(both, mine)
    assert torch.allclose(g, (both[0] + both[1]) / 2, atol=1e-7), (g - (both[0] + both[1]) / 2).abs().max()
# forward A, forward B (the data-parallel wrapper starts a generation for each), ONE backward of the summed loss: a parameter counts
# in two tables and is reported once, after the last accumulation of BOTH (ADVICE r05: it used to be reported twice, the first time
# after half of its accumulations -- the bucket then travelled before its gradient was complete)
ops.new_use_generation()
out_a = forward(xa)
ops.new_use_generation()
out_b = forward(xb)
loss_ab = ((out_a - ya) ** 2).mean() + ((out_b - yb) ** 2).mean()
assert len(ops.graph_generations((loss_ab,))) == 2
opt.zero_grad()
red.begin()
accumulated.clear()
reported = []
def notify_both(p):
    assert accumulated.get(id(p), 0) == (4 if p is net[3].weight else 2), "reported before the last accumulation of both forwards"
    reported.append(id(p))
    red.param_ready(p)
    red.param_ready(p)                                   # idempotent per step
with ops.direct_gradients(notify=notify_both, roots=(loss_ab,)):
    loss_ab.backward()
assert len(reported) == len(set(reported)) == 3, reported
assert red.next >= len(red.buckets) - 1, (red.next, len(red.buckets))
red.finish()
assert red.launched_log == list(range(len(red.buckets))), red.launched_log
g = opt.grad.clone()
opt.zero_grad()
with ops.direct_gradients():
    (((forward(xa) - ya) ** 2).mean() + ((forward(xb) - yb) ** 2).mean()).backward()
mine = opt.grad.clone()
both = [torch.zeros_like(mine) for _ in range(2)]
torch.distributed.all_gather(both, mine)
assert torch.allclose(g, (both[0] + both[1]) / 2, atol=1e-7), (g - (both[0] + both[1]) / 2).abs().max()
# a backward whose forward generation has been dropped (or was never counted) reports nothing: finish() launches what is left
loss_c = ((forward(xa) - ya) ** 2).mean()
for _ in range(ops._USE_GEN_KEEP + 1):
    ops.new_use_generation()
opt.zero_grad()
red.begin()
with ops.direct_gradients(notify=notify_checked):
    loss_c.backward()
red.finish()
assert red.launched_log == list(range(len(red.buckets))), red.launched_log
parallel.barrier()
open(os.path.join(os.environ["MRN_OUT"], f"bok_{rank}"), "w").write("ok")
"""


def test_bucketed_all_reduce_and_module_broadcast_gloo_world2(tmp_path):
    """parallel.BucketedAllReduce (gradient buckets launched from post-accumulate hooks, in bucket order on every rank) and
    parallel.broadcast_module (parameters AND buffers from rank 0) on two CPU ranks"""
    script = tmp_path / "worker_buckets.py"
    script.write_text(WORKER_BUCKETS)
    env = dict(os.environ, MRN_ROOT=ROOT, MRN_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "bok_0").exists() and (tmp_path / "bok_1").exists(), r.stdout + r.stderr


def _opt(trans, feat, seq, pred):
    return types.SimpleNamespace(Transformation=trans, FeatureExtraction=feat, SequenceModeling=seq, Prediction=pred,
                                 num_fiducial=20, imgH=32, imgW=256, input_channel=4, output_channel=512, hidden_size=256,
                                 batch_max_length=25)


def test_lockstep_grouping_policy():
    """host logic of modules/expert_group.py: which sets of experts may run in lock-step, and the tile choice of the
    grouped conv (no kernel is launched)"""
    from mrn_amd import ops
    from mrn_amd.modules import expert_group
    from mrn_amd.modules.model import Model, MRNNet
    with contextlib.redirect_stdout(io.StringIO()):
        trba = [Model(_opt("TPS", "ResNet", "BiLSTM", "Attn")) for _ in range(3)]
        crnn = [Model(_opt("None", "VGG", "BiLSTM", "CTC")) for _ in range(2)]
        net = MRNNet(_opt("None", "VGG", "BiLSTM", "CTC"))
        for c in (20, 30, 40, 50, 60):
            net.update_fc(256, c)
            net.build_prediction(net.opt, c)
    ext = [m.model for m in trba]
    assert expert_group.supported(ext)
    assert not expert_group.supported(ext[:1])                                     # a single expert: nothing to group
    assert not expert_group.supported([trba[0].model, crnn[0].model])              # different architectures
    trba[1].model.FeatureExtraction.ConvNet.bn1.eval()                             # one BatchNorm in another mode
    assert not expert_group.supported(ext)
    trba[1].model.train()
    assert expert_group.supported(ext)
    saved = ops.CONV_PRECISION
    try:
        ops.CONV_PRECISION = "f32"                                                 # exact mode keeps the per-expert kernels
        assert not expert_group.supported(ext)
    finally:
        ops.CONV_PRECISION = saved
    for m in crnn:
        m.build_prediction(m.opt, 30) if m.fc is not None else None
    assert expert_group.HeadsGroup.supported(trba, True) and not expert_group.HeadsGroup.supported(trba, False)   # greedy Attn decode: per expert
    assert expert_group.HeadsGroup.supported(crnn, False)                          # CTC heads have no feedback loop
    # SVTR experts (Trans None, Seq None, CTC): lock-step with the fused attention kernel, all in the same train / eval mode
    with contextlib.redirect_stdout(io.StringIO()):
        svtr = [Model(_opt("None", "SVTR", "None", "CTC")) for _ in range(2)]
    sext = [m.model for m in svtr]
    assert expert_group.supported(sext) and expert_group.HeadsGroup.supported(svtr, True) and expert_group.HeadsGroup.supported(svtr, False)
    svtr[1].model.eval()                                                           # DropPath / BatchNorm mode differs
    assert not expert_group.supported(sext)
    svtr[1].model.train()
    saved_attn = ops.SVTR_FUSED_ATTENTION
    try:
        ops.SVTR_FUSED_ATTENTION = False                                           # per-head GEMM attention: per-expert path
        assert not expert_group.supported(sext)
    finally:
        ops.SVTR_FUSED_ATTENTION = saved_attn
    assert not expert_group.supported([svtr[0].model, crnn[0].model])
    # tile selection of the grouped conv
    assert ops.x3_tile(512, 4608) == (256, 256) and ops.x3_tile(64, 288) == (256, 64)
    assert ops.x3_tile(128, 576) == (128, 128) and ops.x3_tile(128, 2304) == (256, 128)
    # one expert's 4x65 layer (loop A): 520 tiles of 256x256 = 2.03 rounds on 256 CUs -> smaller tiles; six experts: 12.2 rounds
    assert ops.x3_tile(512, 4608, M=66560, G=1) == (128, 128) and ops.x3_tile(512, 4608, M=66560, G=6) == (256, 256)
    # default: ONE lock-step group of all experts on one side stream; MRN_EXPERT_HALVES / expert_halves = 2: five experts -> two
    # sub-groups (2 + 3) on two streams, never for fewer than four
    assert [(lo, hi) for lo, hi, _, _ in net._half_groups(True)] == [(0, 5)]
    net.expert_halves = 2
    parts = net._half_groups(True)
    assert [(lo, hi) for lo, hi, _, _ in parts] == [(0, 2), (2, 5)]
    net.expert_halves = 0
    assert net._half_groups(True) is None
    # two or three experts: ONE lock-step group on a side stream, so the loop-B pipeline still overlaps the router phase
    net.expert_halves = 2
    saved_models = net.model
    net.model = torch.nn.ModuleList(list(saved_models)[:3])
    assert [(lo, hi) for lo, hi, _, _ in net._half_groups(True)] == [(0, 3)]
    net.model = saved_models


def test_converter_encode_array_fill_equals_the_per_word_loop():
    """CTCLabelConverter / AttnLabelConverter.encode build the index batch with one array fill; the reference fills a padded tensor word by
    word (tools/utils.py:35-58,102-131).  Same tensors on random batches (empty words, unknown characters, full-length words, an empty
    batch), and a word that does not fit raises as the reference's slice assignment does"""
    from mrn_amd.tools.utils import AttnLabelConverter, CTCLabelConverter
    chars = "abcdefghij 0123"
    with contextlib.redirect_stdout(io.StringIO()):
        c, a = CTCLabelConverter(chars), AttnLabelConverter(chars)
    rng = np.random.default_rng(3)
    alphabet = list(chars) + ["?", "Z"]                 # two characters outside the dictionary -> [UNK]
    for trial in range(20):
        n = int(rng.integers(0, 9))
        words = ["".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=int(rng.integers(0, 26)))) for _ in range(n)]
        if trial == 0:
            words = []
        idx, ln = c.encode(words, 25)
        ref = torch.full((len(words), 25), c.dict["[PAD]"], dtype=torch.long)
        for i, w in enumerate(words):
            ids = [c.dict[ch] if ch in c.dict else c.dict["[UNK]"] for ch in w]
            ref[i][:len(ids)] = torch.LongTensor(ids)
        assert idx.dtype == torch.long and tuple(idx.shape) == (len(words), 25) and torch.equal(idx.cpu(), ref)
        assert ln.dtype == torch.int32 and ln.tolist() == [len(w) for w in words]
        idx, ln = a.encode(words, 25)
        ref = torch.full((len(words), 27), a.dict["[PAD]"], dtype=torch.long)
        if len(words):
            ref[:, 0] = a.dict["[SOS]"]
        for i, w in enumerate(words):
            ids = [a.dict[ch] if ch in a.dict else a.dict["[UNK]"] for ch in list(w) + ["[EOS]"]]
            ref[i][1:1 + len(ids)] = torch.LongTensor(ids)
        assert idx.dtype == torch.long and tuple(idx.shape) == (len(words), 27) and torch.equal(idx.cpu(), ref)
        assert ln.tolist() == [len(w) + 1 for w in words]
    with pytest.raises(RuntimeError):
        c.encode(["a" * 26], 25)
    with pytest.raises(RuntimeError):
        a.encode(["a" * 26], 25)


def test_bench_compact_line_is_bounded_and_strict_json():
    """bench.compact_line (what the driver parses): headline + roofline numbers + cpu_baseline + {value, ms_per_step} per extra line, never
    the detail record's prose; under 4 KB whatever the record holds (optional objects are dropped first), non-finite numbers as null"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    res = {"metric": "m", "value": 3000.123456, "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 83.3333333,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "arithmetic": "x" * 500,
           "config": {"workload": "w", "per_gpu_batch": 256, "global_batch": 256, "parallelism": "dp1", "classes": [1, 2], "parity": "p" * 900},
           "roofline": {"bound": "mfma", "kernel": "wino_rows_kernel F(4,3)", "achieved": 750.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.3,
                        "traffic": float("nan"), "measured": "prose " * 200, "power_probe": {"a": 1}, "isolated": {"achieved": 800.0, "frac": 0.32, "avg_launch_ms": 1.7, "measured": "z" * 300}},
           "roofline_other_kernels": [{"kernel": "k" * 300}] * 20,
           "cpu_baseline": {"value": 14.0, "unit": "images/s", "cores": 16, "kind": "port", "cpu": "c", "sample": "s", "index_agreement": {"greedy_index_agreement": 1.0}},
           "extra": {"loop_a": {"value": 1.0, "ms_per_step": 2.0, "roofline": {"kernel": "y" * 2000}, "metric": "q" * 300}}}
    line = bench.compact_line(res, "bench_detail.json")
    assert len(line) < 1500 and "\n" not in line and "prose" not in line and "NaN" not in line
    d = json.loads(line)
    assert d["value"] == 3000.1 and d["roofline"]["frac"] == 0.3 and d["roofline"]["traffic"] is None and "measured" not in d["roofline"]
    assert d["roofline"]["isolated"] == {"achieved": 800.0, "frac": 0.32, "avg_launch_ms": 1.7} and d["cpu_baseline"]["index_agreement"]["greedy_index_agreement"] == 1.0
    assert d["extra"] == {"loop_a": {"value": 1.0, "ms_per_step": 2.0}} and d["detail"] == "bench_detail.json" and "roofline_other_kernels" not in d
    # an oversized record sheds its optional objects instead of growing past the driver's tail
    res["extra"] = {"line%d" % i: {"value": float(i), "ms_per_step": 1.0} for i in range(200)}
    line = bench.compact_line(res, "bench_detail.json")
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and "extra" not in d and d["roofline"]["frac"] == 0.3 and d["cpu_baseline"]["value"] == 14.0


def test_reduced_mode_winograd_eligibility():
    """host logic of the reduced-precision mode's Winograd form (no kernel is launched): which layers take the plain-fp16 (d16) operands"""
    from mrn_amd import ops
    saved = (ops.X3_PRODUCTS, ops.WINO_DENSE)
    try:
        ops.X3_PRODUCTS = 3
        assert not ops.wino_dense() and ops.wino_eligible((3, 3), (1, 1), (1, 1), 512, 512) and ops.wino_eligible((3, 3), (1, 1), (1, 1), 160, 96)
        assert not ops.wino_eligible((3, 3), (1, 1), (1, 1), 64, 128) and not ops.wino_eligible((3, 3), (2, 1), (1, 1), 512, 512)      # narrow / strided
        ops.X3_PRODUCTS = 1
        assert ops.wino_dense() and ops.wino_eligible((3, 3), (1, 1), (1, 1), 512, 512)
        assert not ops.wino_eligible((3, 3), (1, 1), (1, 1), 160, 96)                   # Cin % 64 != 0: no d16 form
        assert ops.wino_eligible((3, 3), (1, 1), (1, 1), 160, 96, products=3)           # a trained layer of a parity-mode step asks for its own mode
        assert ops.wino_dense(1) and not ops.wino_dense(3)
        ops.WINO_DENSE = False
        assert not ops.wino_dense() and not ops.wino_eligible((3, 3), (1, 1), (1, 1), 512, 512)      # MRN_WINO_DENSE=0: the round-5 form of the mode
    finally:
        ops.X3_PRODUCTS, ops.WINO_DENSE = saved

