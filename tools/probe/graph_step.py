"""feasibility probe: one loop-A training step captured as a HIP graph (torch.cuda.graph) and replayed"""
import contextlib, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from mrn_amd import ops  # noqa
from mrn_amd.data.synthetic import SyntheticTextLines, synthetic_characters  # noqa
from mrn_amd.il_modules.mrn import MRN  # noqa

model = sys.argv[1] if len(sys.argv) > 1 else "svtr"
torch.cuda.set_device(0)
opt = bench.make_opt(model, 256)
with contextlib.redirect_stdout(io.StringIO()):
    learner = MRN(opt)
    learner.character = synthetic_characters(2086)
    learner.converter = learner.build_converter()
    learner.criterion = learner.build_criterion()
    learner.build_model()
    learner.build_optimizer(learner.count_param())
data = SyntheticTextLines(opt, seed=111)
data.set_characters(learner.character)
image, labels = data.get_batch()
if os.environ.get("NOSIDE") == "1":
    ops.WGRAD_SIDE_STREAM = False
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
torch.cuda.set_stream(s)                 # everything from here on this stream: the parameters' AccumulateGrad nodes belong to it
for _ in range(3):
    learner.train_step(image, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    loss = learner.train_step(image, labels)
torch.cuda.synchronize()
print(f"eager (prepack on): {(time.perf_counter() - t0) * 100:.1f} ms/step loss {float(loss):.4f}", flush=True)
ops.TRAIN_PREPACK = False
ops._PREPACKED.clear()
for _ in range(3):
    learner.train_step(image, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    loss = learner.train_step(image, labels)
torch.cuda.synchronize()
print(f"eager (packs inline): {(time.perf_counter() - t0) * 100:.1f} ms/step loss {float(loss):.4f}", flush=True)
g = torch.cuda.CUDAGraph()
static_image = image.clone()
li, ll = learner.converter.encode(labels, batch_max_length=opt.batch_max_length)
torch.cuda.synchronize()
stage = os.environ.get("STAGE", "all")
if os.environ.get("NOSIDE") == "1":
    ops.WGRAD_SIDE_STREAM = False
mt = contextlib.nullcontext() if os.environ.get("MT", "1") == "1" else torch.autograd.set_multithreading_enabled(False)
del loss
with mt, torch.cuda.graph(g, stream=s):
    if stage == "fwd_nograd":
        with torch.no_grad():
            preds = learner._forward_train(static_image, None if "CTC" in opt.Prediction else li[:, :-1])
        static_loss = preds.sum()
    else:
        preds = learner._forward_train(static_image, None if "CTC" in opt.Prediction else li[:, :-1])
        static_loss = learner.criterion(preds, li, ll)
        if stage == "bwd":
            learner.optimizer.zero_grad()
            with ops.direct_gradients():
                static_loss.backward()
        elif stage == "all":
            learner.backward_and_step(static_loss)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) * 50:.1f} ms/step loss {float(static_loss):.4f}", flush=True)
