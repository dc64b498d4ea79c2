#!/usr/bin/env python3
"""bench.py -- MRN loop B (router phase, il_modules/mrn.py:323-371) on synthetic 32x256 crops.

One "step" = one routing_step of the MRN learner on a per-GPU batch of 256 crops: forward of all frozen experts
(train-mode BatchNorm, as in the reference's step 1), DM-Router forward, fused fan-in, losses, backward through the
router, gradient all-reduce (N > 1), global-norm clip + Adam.  Default workload = BASELINE.json's metric
configuration: TRBA (TPS+ResNet+BiLSTM+Attn) x 6 experts.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLASSES_MLT19 = (2086, 220, 1728, 1160, 73, 102)      # README.md:103, per-task class counts
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0                         # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s achievable on a float4 copy)


PMC_SOURCE = "profiles/r06_pmc.json (tools/pmc_pass.sh: separate rocprofv3 --pmc passes of this command; 2 x FETCH_SIZE + WRITE_SIZE)"


def pmc_traffic(kernel_name, section="kernels"):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/r01_pmc.json, produced by
    tools/pmc_pass.sh on this same command): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE,
    both KiB counters.  None when the kernel has no entry (counters cannot be read from inside the timed run)."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return None
    with open(path) as f:
        pmc = json.load(f)
    key = kernel_name.split(" ")[0]          # ("wino_rows_kernel F(4,3)" -> wino_rows_kernel; template arguments carry no spaces here)
    kernels = pmc.get(section)
    if kernels is None:          # (no pass of that workload in this file: no figure rather than another workload's)
        return None
    kernels = {k.replace(" ", ""): v for k, v in kernels.items()}
    ent = kernels.get(key)
    if ent is None:          # (template arguments the name leaves out: bn_apply_wino_grouped_kernel<4>)
        ent = next((v for k, v in kernels.items() if k.startswith(key + "<")), None)
    return None if ent is None else ent["hbm_bytes_per_launch"]


def make_opt(model, batch):
    o = types.SimpleNamespace(
        exp_name="bench", il="mrn", memory="random", memory_num=2000, batch_max_length=25, imgH=32, imgW=256,
        manual_seed=111, start_task=0, num_fiducial=20, input_channel=4, output_channel=512, hidden_size=256,
        schedule="super", optimizer="adam", lr=0.0005, batch_size=batch, num_iter=10000, val_interval=5000, grad_clip=5,
        lan_list=["Chinese", "Latin", "Japanese", "Korean", "Arabic", "Bangla"], NED=True, workers=0)
    if model == "trba":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    elif model == "crnn":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "VGG", "BiLSTM", "CTC"
    elif model == "svtr":
        o.Transformation, o.FeatureExtraction, o.SequenceModeling, o.Prediction = "None", "SVTR", "None", "CTC"
    else:
        raise SystemExit(f"unknown model {model}")
    return o


def build_learner(opt, n_experts, quiet=True):
    from mrn_amd.data.synthetic import synthetic_characters
    from mrn_amd.il_modules.mrn import MRN
    sink = io.StringIO() if quiet else sys.stdout
    with contextlib.redirect_stdout(sink):
        learner = MRN(opt)
        total = 0
        for taski in range(n_experts):
            total += CLASSES_MLT19[taski]
            learner.character = synthetic_characters(total)
            learner.converter = learner.build_converter()
            if taski == 0:
                learner.criterion = learner.build_criterion()
                learner.build_model()
            else:
                learner.change_model()
        learner.freeze_experts(n_experts)
        learner.model.train()       # reference: every "frozen" expert runs train-mode BatchNorm through step 1 (mrn.py:107,401)
        learner.prepare_routing(total_steps=10 ** 9)
    return learner


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(learner, opt, n_experts, batch=32, iters=3, max_threads=32, budget_s=40.0):
    """The CPU oracle (oracle/mrn_oracle.py, validated against the reference by tests/test_oracle_golden.py) on a bounded
    sample of the same workload, as SURVEY.md section 8d planned it: loop B at batch 32, 1 warm-up + 3 timed iterations, all
    host threads (capped: more threads only add synchronisation overhead at this size)."""
    from oracle import mrn_oracle as O
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools.utils import host_cpu_budget
    torch.set_num_threads(max(1, min(host_cpu_budget(), max_threads)))   # (affinity capped by the cgroup quota; more only adds sync overhead)
    sd = {k[len("module."):]: v.detach().cpu().clone() for k, v in learner.model.state_dict().items()}
    cfg = O.Cfg(opt.Transformation, opt.FeatureExtraction, opt.SequenceModeling, opt.Prediction)
    names = [n for n, p in learner.model.module.named_parameters() if p.requires_grad]
    params = [sd[n].requires_grad_(True) for n in names]
    state = [{"m": torch.zeros_like(p), "v": torch.zeros_like(p)} for p in params]
    o2 = types.SimpleNamespace(**vars(opt))
    o2.batch_size = batch
    data = SyntheticTextLines(o2, device=torch.device("cpu"))
    data.set_characters(learner.character)
    conv = O.CTCConverter(learner.character) if opt.Prediction == "CTC" else O.AttnConverter(learner.character)
    times, spent = [], 0.0
    for it in range(iters + 1):
        image, labels, idx = data.get_batch2()
        domain = torch.LongTensor(idx).squeeze()
        li, ll = conv.encode(labels, 25)
        t0 = time.time()
        text = None if opt.Prediction == "CTC" else li[:, :-1]
        masks = None
        if opt.FeatureExtraction == "SVTR":      # DropPath draws of the train-mode experts, two sites per block (svtr.py:17-22)
            masks = [[torch.bernoulli(torch.full((batch,), 1.0 - float(dp))) for dp in O.SVTR_DROP_PATH if dp > 0 for _ in range(2)]
                     for _ in range(n_experts)]
        out = O.mrn_forward(sd, cfg, n_experts, image, True, text, True, training=True, masks=masks)
        loss, _, _ = O.mrn_step_loss(out, li, ll, domain, opt.Prediction)
        if it == 0:          # the oracle as the checker: index agreement of the HIP path on this bench's own input distribution
            agreement = index_agreement(learner, image, text, masks, out)
        grads = torch.autograd.grad(loss, params)
        with torch.no_grad():
            O.clip_and_adam(params, grads, state, 2.5e-5, it + 1)
        dt = time.time() - t0
        spent += dt
        if it > 0:
            times.append(dt)
        elif dt > budget_s:      # keep the default run within minutes: a slow host reports its (cold) first iteration
            times.append(dt)
            break
        if spent + dt > budget_s and times:
            break
    sec = sum(times) / len(times)
    return {"value": batch / sec, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model_string(), "sample": f"{len(times)} iters of loop B at batch {batch}, fp32 torch-CPU oracle",
            "index_agreement": agreement}


def index_agreement(learner, image, text, masks, ref):
    """the HIP path's train-mode loop-B forward on the batch the CPU oracle just ran (same weights, same DropPath draws): how many
    greedy indices of the fused logits and how many routing decisions (argmax of the gate weights) are equal, and the largest
    differences -- measured on U(-1,1) noise crops, the distribution the headline is timed on"""
    dev = learner.device
    net = learner.model.module
    if masks is not None:
        from mrn_amd.modules.svtr import DropPath
        for e, ms in enumerate(masks):
            for j, m in enumerate(mod for mod in net.model[e].modules() if isinstance(mod, DropPath)):
                m.forced_masks = [ms[2 * j].clone(), ms[2 * j + 1].clone()]
    with torch.no_grad():
        got = learner.model(image.to(dev), True, None if text is None else text.to(dev), True)
    gl, gw = got["logits"].float().cpu(), got["index"].float().cpu()
    rl, rw = ref["logits"].detach(), ref["index"].detach()
    return {"samples": int(rl.shape[0]), "positions": int(rl.shape[0] * rl.shape[1]),
            "greedy_index_agreement": (gl.argmax(2) == rl.argmax(2)).float().mean().item(),
            "routing_argmax_agreement": (gw.argmax(1) == rw.argmax(1)).float().mean().item(),
            "max_abs_logit_diff": (gl - rl).abs().max().item(), "max_abs_logit": rl.abs().max().item(),
            "max_abs_gate_weight_diff": (gw - rw).abs().max().item()}


def power_probe(ops, R, launches=3):
    """The dominant layer shape (6 experts x 256 crops, 4x65 maps, 512 -> 512 3x3) alone, `launches` launches each on random
    operands, on zero activations and on zero activations AND weights: identical instruction stream, geometry and memory traffic,
    so the differences are what the chip's power management does with the clock (MI355X_MICROARCH.md, DVFS give-back).
    R > 0: the Winograd F(R,3) form the step runs; 0: the direct split-fp16 x3 kernel."""
    G, B, H, W, C = 6, 256, 4, 65, 512
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev).manual_seed(7)
    ypre = torch.randn(G, B, H, W, C, device=dev, generator=gen)
    ws = [(torch.rand(C, 3, 3, C, device=dev, generator=gen) * 2 - 1) * 0.02 for _ in range(G)]
    out = {}
    for name, za, zw in (("random_ms", False, False), ("zero_act_ms", True, False), ("zero_all_ms", True, True)):
        y_ = torch.zeros_like(ypre) if za else ypre
        ws_ = [torch.zeros_like(w) for w in ws] if zw else ws
        if R:
            u_hl, u_scale = ops.pack_weights_wino(ws_, R)
            _, _, v = ops.bn_apply_wino_grouped(y_, None, None, R, relu=True)
            fn = lambda: ops.conv2d_x3_wino(v, G, False, B, H, W, C, u_hl, u_scale, C, R, want_stats=True)      # noqa: E731
        else:
            w_hl, w_scale = ops.pack_weights_hl32(ws_)
            _, hl = ops.bn_apply_grouped(y_.clone(), None, None, relu=True, want_f32=False, want_hl=True)
            fn = lambda: ops.conv2d_x3(hl, G, False, B, H, W, C, w_hl, w_scale, C, (3, 3), (1, 1), (1, 1), want_stats=True)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / launches
        del fn
    gflop = 2.0 * G * B * H * W * C * 9 * C / 1e9
    out.update({"shape": "G6 B256 4x65 512->512 3x3 (%s)" % ("Winograd F(%d,3)" % R if R else "direct"), "launches_each": launches,
                "algorithmic_gflop_per_launch": gflop, "random_tflops": gflop / out["random_ms"], "zero_all_tflops": gflop / out["zero_all_ms"],
                "measured": "live, behind the timed region: HIP events around back-to-back launches"})
    return out


def comm_telemetry(reducer, world, dev, step_ms):
    """N > 1: what the gradient exchange looked like -- did RCCL see `world` ranks, how many bytes per step, what the exchange costs
    standalone, and how much of it the compute stream actually waited for (events around BucketedAllReduce.finish())"""
    import torch.distributed as dist
    ids = [torch.zeros(1, device=dev, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([dist.get_rank()], device=dev, dtype=torch.int64))
    out = {"backend": dist.get_backend(), "ranks_seen": sorted(int(t.item()) for t in ids)}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:          # (gloo-only builds)
        out["rccl_version"] = f"unavailable ({type(e).__name__})"
    if reducer is not None:
        out["allreduce_bytes_per_step"] = reducer.bytes_per_step()
        out["buckets"] = len(reducer.buckets)
        exposed = [a.elapsed_time(b) for a, b in reducer.exposed]
        out["allreduce_ms_per_step"] = reducer.standalone_ms()
        if exposed:
            out["exposed_ms_per_step"] = sum(exposed) / len(exposed)
            out["overlap_frac"] = max(0.0, 1.0 - out["exposed_ms_per_step"] / max(out["allreduce_ms_per_step"], 1e-9))
        out["measured"] = "standalone: buckets back to back on an idle GPU; exposed: HIP events around the wait inside the timed steps"
    return out


def cpu_baseline_loop_a(learner, opt, batch=32, max_threads=32, budget_s=45.0):
    """loop A (il_modules/mrn.py:225-279: forward, loss, backward, clip, Adam of ONE expert) on the CPU oracle at batch 32:
    1 warm-up + up to 2 timed iterations inside the time budget (TRBA: ~10 s per iteration)"""
    from oracle import mrn_oracle as O
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools.utils import host_cpu_budget
    torch.set_num_threads(max(1, min(host_cpu_budget(), max_threads)))
    sd = {k[len("module."):]: v.detach().cpu().clone() for k, v in learner.model.state_dict().items()}
    cfg = O.Cfg(opt.Transformation, opt.FeatureExtraction, opt.SequenceModeling, opt.Prediction)
    names = [n for n, p in learner.model.module.named_parameters() if p.requires_grad and n.startswith("model.0.")]
    params = [sd[n].requires_grad_(True) for n in names]
    state = [{"m": torch.zeros_like(p), "v": torch.zeros_like(p)} for p in params]
    o2 = types.SimpleNamespace(**vars(opt))
    o2.batch_size = batch
    data = SyntheticTextLines(o2, device=torch.device("cpu"))
    data.set_characters(learner.character)
    conv = O.CTCConverter(learner.character) if opt.Prediction == "CTC" else O.AttnConverter(learner.character)
    times, spent = [], 0.0
    for it in range(3):
        image, labels = data.get_batch()
        li, ll = conv.encode(labels, 25)
        t0 = time.time()
        masks = None
        if opt.FeatureExtraction == "SVTR":
            masks = [torch.bernoulli(torch.full((batch,), 1.0 - float(dp))) for dp in O.SVTR_DROP_PATH if dp > 0 for _ in range(2)]
        out = O.model_forward(sd, "model.0.", cfg, image, None if opt.Prediction == "CTC" else li[:, :-1], True, training=True, masks=masks)
        loss = O.ctc_loss(out["predict"], li, ll) if opt.Prediction == "CTC" else O.attn_ce_loss(out["predict"], li)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            keep = [(p, g, s_) for p, g, s_ in zip(params, grads, state) if g is not None]
            O.clip_and_adam([k[0] for k in keep], [k[1] for k in keep], [k[2] for k in keep], 2.5e-5, it + 1)
        dt = time.time() - t0
        spent += dt
        if it > 0 or dt > budget_s / 2:
            times.append(dt)
        if spent + dt > budget_s:
            break
    sec = sum(times) / len(times)
    return {"value": batch / sec, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model_string(),
            "sample": f"{len(times)} iters of loop A (one {opt.FeatureExtraction} expert) at batch {batch}, fp32 torch-CPU oracle"}


def build_loop_a_learner(opt, quiet=True):
    """MRN task 0: ONE expert, everything trainable -- loop A (il_modules/mrn.py:225-279), the full forward + backward step"""
    from mrn_amd.data.synthetic import synthetic_characters
    from mrn_amd.il_modules.mrn import MRN
    with contextlib.redirect_stdout(io.StringIO() if quiet else sys.stdout):
        learner = MRN(opt)
        learner.character = synthetic_characters(CLASSES_MLT19[0])
        learner.converter = learner.build_converter()
        learner.criterion = learner.build_criterion()
        learner.build_model()
        learner.build_optimizer(learner.count_param(), total_steps=10 ** 9)
    return learner


def train_dtype():
    """f32: every product is a 22-bit split-fp16 x3 MFMA term or exact fp32, fp32 accumulate and storage; fp16: ONE fp16 product per term
    (--precision fp16, the reduced mode of BASELINE configs 2 / 5)"""
    from mrn_amd import ops
    return ("bf16" if ops.REDUCED_BF16 else "fp16") if ops.TRAIN_PRODUCTS == 1 else "f32"


def describe_kernel(kind):
    """KernelTimer kind -> (kernel name, peak TFLOP/s of its MFMA dtype, MFMA flops executed per algorithmic flop).
    What each kernel does is DESIGN.md's kernel table, not the bench line's business."""
    if kind == "f32":
        return ("gemm_f32_kernel<2,2,2,2,16,true>", FP32_MFMA_PEAK_TFLOPS, 1)
    if kind.startswith("gemm_f32"):
        return ("gemm_f32_kernel", FP32_MFMA_PEAK_TFLOPS, 1)
    arith, staging = kind.split("/")
    if arith == "f32mfma":
        return (staging, FP32_MFMA_PEAK_TFLOPS, 1)
    nprod = 3 if arith == "fp16x3" else 1
    if staging.startswith("winorows"):
        R = int(staging[8])
        return (f"wino_rows_kernel F({R},3)", BF16_MFMA_PEAK_TFLOPS, nprod * (R + 2) / (3.0 * R))
    if staging.startswith("wino"):
        R = int(staging[4])
        return (f"conv_x3_kernel<2,4,2,1,false,3,{R}>", BF16_MFMA_PEAK_TFLOPS, 3.0 * (R + 2) / (3 * R))
    if staging.startswith("x3g"):
        tile = staging[3:]
        targs = {"256x256": "4,4,2,2", "256x128": "4,2,2,2", "128x128": "4,2,1,2", "256x64": "8,1,1,2", "128x64": "4,1,1,2", "64x64": "2,2,1,1"}[tile]
        return (f"conv_x3_kernel<{targs}{',false,1' if nprod == 1 else ''}>", BF16_MFMA_PEAK_TFLOPS, nprod)
    if staging.startswith("patch"):
        cin = int(staging[5:7])
        return (f"conv_patch_x3_kernel<{cin // 32},{cin // 16},2{',true' if staging.endswith('pool') else ''}>", BF16_MFMA_PEAK_TFLOPS, 3)
    svtr = {"svtrmlp": "svtr_mlp_kernel", "svtrmixer": "svtr_mixer_kernel", "svtrattn": "svtr_mixer_kernel<256,ATTN>", "svtrblock3": "svtr_tail_kernel<256>"}
    if staging in svtr:
        return (svtr[staging], BF16_MFMA_PEAK_TFLOPS, 3)
    raise ValueError(f"unknown KernelTimer kind {kind!r}")


def rl_allow_hbm(kind):
    """the dominant Winograd kernels always stay MFMA entries (their intensity is far above the ridge anyway)"""
    return "wino" not in kind


def roofline_entries(kinds, steps, elapsed_s, pmc_section="kernels"):
    """KernelTimer.summary() -> (MFMA-bound entries sorted by share of the step, HBM-bound entries); pmc_section: which workload's PMC
    passes of the committed profile the `traffic` figures come from (loop A launches G = 1 kernels: its own passes)"""
    rl, hbm = [], []
    for kind, s_ in kinds.items():
        if kind.startswith("hbm/"):
            # HBM-bound pass: algorithmic bytes / union of its launch intervals against the 8 TB/s HBM3E peak
            gbs = s_["total_bytes"] / (s_["union_ms"] * 1e-3) / 1e9
            what = {"bn_apply_grouped": "BatchNorm-apply + residual + ReLU over all experts, fp32 in, HL32 split-fp16 operand out",
                    "bn_apply_wino_grouped": "BatchNorm-apply + residual + ReLU + Winograd input transform B^T over all experts, fp32 in, "
                                             "transformed HL32 operand (6 components per 4 columns) [+ plain HL32] out",
                    "conv_first_kernel": "first 3x3 convolution of the stacks on the Cin = 4 crops, all experts, exact-fp32 MFMA; <NT, true>: "
                                         "the 2x2 max-pool taken in the epilogue by BatchNorm-weight sign, a quarter of the map out"
                    }.get(kind[4:].split("<")[0], kind[4:])
            kfull = kind[4:] if "_kernel" in kind else kind[4:] + "_kernel"
            hbm.append({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": pmc_traffic(kfull, pmc_section), "kernel": f"{kfull} ({what})",
                        "algorithmic_bytes_per_launch": s_["total_bytes"] / s_["launches"],
                        "launches_per_step": s_["launches"] / steps, "avg_launch_ms": s_["union_ms"] / s_["launches"],
                        "avg_launch_ms_raw_event": s_["total_ms"] / s_["launches"],
                        "kernel_share_of_step": s_["union_ms"] / (elapsed_s * 1e3)})
            continue
        per_launch = s_["total_flops"] / s_["launches"]
        raw_ms = s_["total_ms"] / s_["launches"]          # per-launch HIP-event time (what rocprofv3 reports)
        # lock-step sub-groups run on separate streams, so launches of this kernel share the GPU most of the
        # time: the rate is taken over the UNION of the launch intervals (= raw time when nothing overlaps)
        avg_ms = s_["union_ms"] / s_["launches"]
        ach = per_launch / (avg_ms * 1e-3) / 1e12
        kname, peak, per_flop = describe_kernel(kind)
        nbytes = s_.get("total_bytes", 0.0)
        # a launch family whose MFMA work per algorithmic byte lies under the ridge (peak flop/s over peak B/s) is priced against HBM:
        # the small-K / narrow x3 launches (1 x 1 shortcuts, the first-stage convolutions, heads) are output-write-bound, not MFMA-bound
        if nbytes > 0 and s_["total_flops"] * per_flop / nbytes < peak * 1e12 / (HBM_PEAK_GBS * 1e9) and rl_allow_hbm(kind):
            gbs = nbytes / (s_["union_ms"] * 1e-3) / 1e9
            hbm.append({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": pmc_traffic(kname, pmc_section), "kernel": kname,
                        "algorithmic_bytes_per_launch": nbytes / s_["launches"], "algorithmic_gflop_per_launch": per_launch / 1e9,
                        "mfma_flop_per_algorithmic_byte": s_["total_flops"] * per_flop / nbytes,
                        "mfma_issue_frac": ach * per_flop / peak,
                        "launches_per_step": s_["launches"] / steps, "avg_launch_ms": avg_ms, "avg_launch_ms_raw_event": raw_ms,
                        "kernel_share_of_step": s_["union_ms"] / (elapsed_s * 1e3)})
            continue
        rl.append({"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                   "traffic": pmc_traffic(kname, pmc_section), "traffic_source": PMC_SOURCE,
                   "algorithmic_bytes_per_launch": s_.get("total_bytes", 0.0) / s_["launches"],
                   "kernel": kname, "mfma_flops_per_algorithmic_flop": per_flop,
                   "mfma_issue_frac": ach * per_flop / peak,
                   "launches_per_step": s_["launches"] / steps, "avg_launch_ms": avg_ms,
                   "avg_launch_ms_raw_event": raw_ms, "launch_overlap": s_["total_ms"] / s_["union_ms"],
                   "algorithmic_gflop_per_launch": per_launch / 1e9,
                   "kernel_share_of_step": s_["union_ms"] / (elapsed_s * 1e3)})
    rl.sort(key=lambda r: -r["kernel_share_of_step"])
    return rl, hbm


def time_loop_a(args, opt, rank, world, steps, warmup, with_cpu_baseline=False):
    """loop A on the same synthetic crops: forward, loss, backward (bucketed all-reduce when N > 1), clip, Adam"""
    from mrn_amd import ops, parallel
    from mrn_amd.data.synthetic import SyntheticTextLines
    learner = build_loop_a_learner(opt, quiet=not args.verbose)
    data = SyntheticTextLines(opt, seed=211 + rank)
    data.set_characters(learner.character)
    for _ in range(warmup):
        learner.train_step(*data.get_batch())
    reducer = getattr(learner, "reducer", None)
    if reducer is not None:
        reducer.record, reducer.exposed = True, []
    if not args.no_kernel_timer:
        ops.CONV_TIMER = ops.KernelTimer()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    side0 = ops.DIRECT_STATS["parameters"]
    for _ in range(steps):
        loss = learner.train_step(*data.get_batch())
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    side_params = (ops.DIRECT_STATS["parameters"] - side0) / max(steps, 1)
    timer, ops.CONV_TIMER = ops.CONV_TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], device=learner.device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    n_params = learner.optimizer.flat.numel()
    res = {"metric": f"text-line images/sec, loop A (train the newest {args.model.upper()} expert: fwd + bwd + clip + Adam) at 32x256",
           "value": world * args.batch * steps / elapsed, "unit": "images/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "warmup": warmup, "per_gpu_batch": args.batch, "trainable_parameters": n_params, "loss": float(loss.detach()),
           "dtype": train_dtype()}
    if timer is not None and timer.spans:
        rl, hbm = roofline_entries(timer.summary(), steps, elapsed, pmc_section="kernels_loop_a" if args.model == "trba" else "none")
        if rl:
            res["roofline"] = rl[0]
            res["roofline"]["measured"] = "timed region, HIP events per launch on the launch stream; rate over the union of the launch intervals"
            if len(rl) > 1 or hbm:
                res["roofline_other_kernels"] = rl[1:4] + hbm
    if world > 1:
        res["comm"] = comm_telemetry(reducer, world, learner.device, elapsed / steps * 1e3)
        res["comm"]["side_stream_parameters_per_step"] = side_params      # parameter gradients accumulated on the second stream (as at N = 1)
    if with_cpu_baseline and world == 1 and rank == 0:
        res["cpu_baseline"] = cpu_baseline_loop_a(learner, opt)
    del learner
    torch.cuda.empty_cache()
    return res


def time_il_step(args, opt, rank, world, steps, warmup):
    """one task-1 iteration of LwF (lwf.py:52-95: new network forward + backward, frozen previous network forward,
    KD term on the old classes) or EWC (ewc.py:73-104: classification loss + the quadratic Fisher penalty -- run with the penalty
    the code intends, i.e. more work than the reference's identically-zero one) on one expert of `--model`"""
    from mrn_amd import parallel
    from mrn_amd.data.synthetic import SyntheticTextLines, synthetic_characters
    from mrn_amd.il_modules.ewc import EWC
    from mrn_amd.il_modules.lwf import LwF
    data = SyntheticTextLines(opt, seed=411 + rank)
    with contextlib.redirect_stdout(io.StringIO() if not args.verbose else sys.stdout):
        learner = (LwF if args.loop == "lwf" else EWC)(opt)
        learner.character = synthetic_characters(CLASSES_MLT19[0])
        learner.converter = learner.build_converter()
        learner.criterion = learner.build_criterion()
        learner.build_model()
        learner.build_optimizer(learner.count_param(), total_steps=10 ** 9)
        data.set_characters(learner.character)
        if args.loop == "ewc":
            learner.reference_prefix_bug = False
            learner.fisher_iterations = 1
            learner.fisher = learner.getFisherDiagonal(data)
            learner.mean = {n: p.clone().detach() for n, p in learner.model.named_parameters() if p.requires_grad}
        learner.after_task()
        learner.character = synthetic_characters(CLASSES_MLT19[0] + CLASSES_MLT19[1])
        learner.converter = learner.build_converter()
        learner.change_model()
        learner.build_optimizer(learner.count_param(), total_steps=10 ** 9)
        data.set_characters(learner.character)
    step = learner.kd_step if args.loop == "lwf" else learner.ewc_step
    for _ in range(warmup):
        step(*data.get_batch())
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(*data.get_batch())
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=learner.device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = out[0] if isinstance(out, tuple) else out
    what = "LwF step (new network fwd + bwd, frozen previous network fwd, KD on the old classes)" if args.loop == "lwf" else \
        "EWC step (fwd + bwd + Fisher penalty over all parameters)"
    return {"metric": f"text-line images/sec, {what}, {args.model.upper()}, task 1, at 32x256",
            "value": world * args.batch * steps / elapsed, "unit": "images/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "warmup": warmup, "per_gpu_batch": args.batch, "trainable_parameters": learner.optimizer.flat.numel(),
            "loss": float(loss.detach()), "dtype": train_dtype()}


def time_der_step(args, opt, rank, world, steps, warmup):
    """BASELINE config 5: DER (il_modules/der.py:208-290) on DERNet with `--experts` TRBA extractors -- the older ones frozen and in
    eval mode (lock-step grouped forward), the newest one trained, main attention head over the 256 * N-wide concatenation,
    auxiliary head on the newest 256 channels; loss = loss_clf."""
    from mrn_amd import parallel
    from mrn_amd.data.synthetic import SyntheticTextLines, synthetic_characters
    from mrn_amd.il_modules.der import DER
    with contextlib.redirect_stdout(io.StringIO() if not args.verbose else sys.stdout):
        learner = DER(opt)
        total = 0
        for taski in range(args.experts):
            total += CLASSES_MLT19[taski]
            learner.character = synthetic_characters(total)
            learner.converter = learner.build_converter()
            if taski == 0:
                learner.criterion = learner.build_criterion()
                learner.build_model()
            else:
                learner.model = learner.model.module
                learner._known_classes = learner._total_classes
                learner.change_model()
        for i in range(args.experts - 1):
            for p in learner.model.module.model[i].parameters():
                p.requires_grad = False
        learner.build_optimizer(learner.count_param(), total_steps=10 ** 9)
        learner.model_eval_and_train(args.experts - 1)
    data = SyntheticTextLines(opt, seed=311 + rank)
    data.set_characters(learner.character)
    # one batch of look-ahead, as DER._update runs it: the frozen extractors of batch n+1 are issued (side stream) before batch n
    # trains the newest extractor; every timed step issues exactly one frozen forward and one training step
    pending = []

    def fetch():
        image, labels = data.get_batch()
        return image, labels, (learner.prefetch_frozen(image) if not args.serial else None)

    def step():
        if not pending:
            pending.append(fetch())
        image, labels, pre = pending.pop(0)
        pending.append(fetch())
        return learner.der_step(image, labels, prefetched=pre)

    for _ in range(warmup):
        step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, aux = step()
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=learner.device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"metric": f"text-line images/sec, DER step ({args.model.upper()} x {args.experts} extractors, newest trained: fwd + bwd + clip + Adam) at 32x256",
            "value": world * args.batch * steps / elapsed, "unit": "images/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "warmup": warmup, "per_gpu_batch": args.batch, "trainable_parameters": learner.optimizer.flat.numel(),
            "loss_clf": float(loss.detach()), "loss_aux": float(aux.detach()),
            "dtype": train_dtype()}


def time_loop_b_short(args, opt, steps, warmup):
    """the headline workload (loop B, pipelined) for a few steps under the CURRENT arithmetic switches: the reduced-precision line"""
    from mrn_amd import ops
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools.utils import to_device
    learner = build_learner(opt, args.experts, quiet=not args.verbose)
    data = SyntheticTextLines(opt, seed=111)
    data.set_characters(learner.character)

    def fetch():
        image, labels, idx = data.get_batch2()
        indexs = to_device(torch.LongTensor(idx).squeeze())
        pre = learner.prefetch_experts(image, labels)
        return image, labels, indexs, pre if (pre is not None and pre[0] is not None) else None
    pending = [fetch()]

    def step():
        image, labels, indexs, pre = pending.pop()
        pending.append(fetch())
        return learner.routing_step(image, labels, indexs, prefetched=pre)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pending.clear()
    del learner
    return {"metric": "text-line images/sec (fwd+bwd) at 32x256, TRBA+MRN 6 experts, REDUCED precision (not the headline)",
            "value": args.batch * steps / elapsed, "unit": "images/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "host_issue_ms_per_step": host / steps * 1e3,
            "dtype": "fp16" if ops.X3_PRODUCTS == 1 else "f32"}


RL_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "algorithmic_gflop_per_launch",
           "launches_per_step", "avg_launch_ms", "kernel_share_of_step", "mfma_flops_per_algorithmic_flop")
CPU_KEYS = ("value", "unit", "cores", "kind", "cpu", "sample", "index_agreement")
COMM_KEYS = ("backend", "ranks_seen", "rccl_version", "allreduce_bytes_per_step", "buckets", "allreduce_ms_per_step", "exposed_ms_per_step",
             "overlap_frac")
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
LINE_LIMIT = 4096            # the driver keeps an 8 KB tail of stdout: the line must fit with room to spare


def _sig(x, digits=5):
    """numbers of the compact line at 5 significant digits (the detail file keeps them in full)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None      # (NaN / inf are not JSON)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(res, detail_path=None):
    """the ONE stdout line: headline fields, the dominant kernel's roofline (numbers + kernel name), cpu_baseline, comm (N > 1) and
    {value, ms_per_step} per extra workload -- no prose.  Everything else stays in the detail file."""
    out = {k: res[k] for k in HEAD_KEYS if k in res}
    cfg = res.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "per_gpu_batch", "global_batch", "parallelism", "classes") if k in cfg}
    if "roofline" in res:
        out["roofline"] = {k: res["roofline"][k] for k in RL_KEYS if k in res["roofline"]}
        iso = res["roofline"].get("isolated")
        if iso:
            out["roofline"]["isolated"] = {k: iso[k] for k in ("achieved", "frac", "avg_launch_ms") if k in iso}
    if "cpu_baseline" in res:
        out["cpu_baseline"] = {k: res["cpu_baseline"][k] for k in CPU_KEYS if k in res["cpu_baseline"]}
    if "comm" in res:
        out["comm"] = {k: res["comm"][k] for k in COMM_KEYS if k in res["comm"]}
    if "extra" in res:
        out["extra"] = {name: {k: line[k] for k in ("value", "ms_per_step") if k in line} for name, line in res["extra"].items()}
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(_sig(out), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:          # never let the line outgrow the driver again: drop the optional objects, loudest last
        for k in ("extra", "comm"):
            out.pop(k, None)
            line = json.dumps(_sig(out), separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    assert len(line) < LINE_LIMIT and "\n" not in line, len(line)
    return line


def emit(res):
    """rank 0: write the full record to bench_detail.json (repo root, and gpurun_out/ so it travels back from a gpurun call), then
    print the compact line as the LAST line of stdout"""
    rel = "bench_detail.json"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, rel), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            rel = None if d == ROOT else rel
    sys.stdout.flush()
    print(compact_line(res, rel), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (how the driver calls it): run the same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` as a child
    process and return its exit code.  Called before mrn_amd or any torch.cuda function is touched."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # (dmabuf IPC only on this host driver: RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="trba", choices=["trba", "crnn", "svtr"])
    ap.add_argument("--experts", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (reference default 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-all", action="store_true", help="HIP events around EVERY timed launch inside the timed region (default: the dominant "
                    "kernel's launches only; the other kernels are measured in a short pass behind it)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "fp16x3", "fp16", "bf16"],
                    help="arithmetic of the frozen experts' convolutions / Linear layers: auto = split-fp16 x3 (22-bit products, keeps "
                         "the 1e-4 parity band: the headline); fp16 = ONE fp16 product per term on the same grouped kernels (the "
                         "reduced-precision mode of BASELINE configs 2 and 5: a separate line, never the headline); f32: "
                         "the per-expert exact-fp32 MFMA kernel (mrn_amd/ops.py: CONV_PRECISION); bf16: the fp16 mode with the Winograd layers' 16-bit operands as "
                         "bfloat16 (comparison line: BASELINE config 2 says \"bf16\")")
    ap.add_argument("--no-streams", action="store_true", help="run the experts sequentially on one stream")
    ap.add_argument("--no-pipeline", action="store_true", help="do not issue batch n+1's expert forward before batch n's router phase")
    ap.add_argument("--serial", action="store_true", help="one lock-step group on one stream, no look-ahead (every kernel runs alone)")
    ap.add_argument("--no-isolated-pass", action="store_true", help="skip the 2 extra serialized steps that measure the dominant kernel alone")
    ap.add_argument("--loop", default="b", choices=["a", "b", "der", "lwf", "ewc"], help="b (default): the router phase over frozen experts, BASELINE's "
                    "metric workload; a: train one expert (full forward + backward); der: BASELINE config 5's DER step over --experts "
                    "extractors; lwf / ewc: config 5's auxiliary-loss learners, one task-1 step -- printed as the main line instead")
    ap.add_argument("--no-extra", action="store_true", help="do not append the short loop-A measurement under \"extra\"")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the live zero-operand probe of the dominant layer shape")
    ap.add_argument("--no-reduced", action="store_true", help="skip the short reduced-precision lines (extra.fp16_loop_b / extra.fp16_der)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # The driver's bare `python bench.py --gpus N ...`: start one rank per GPU as CHILD processes (never exec: nothing in this
        # process has touched the GPU yet, and nothing will) and hand back the launcher's return code.  Rank 0's JSON line goes
        # straight to our stdout.
        raise SystemExit(self_launch(args.gpus))

    from mrn_amd import ops, parallel
    from mrn_amd.data.synthetic import SyntheticTextLines
    from mrn_amd.tools.utils import to_device
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    torch.cuda.set_device(local)
    from mrn_amd.tools.utils import host_cpu_budget
    torch.set_num_threads(max(1, host_cpu_budget() // max(1, world)))     # (PyTorch sizes its pools from the affinity mask, not the cgroup quota)
    torch.manual_seed(111)

    if args.precision in ("fp16", "bf16"):
        ops.X3_PRODUCTS = 1
        ops.TRAIN_PRODUCTS = 1           # (--loop a / der / lwf / ewc: the trained convolutions too)
        ops.REDUCED_BF16 = args.precision == "bf16"      # (the Winograd layers' 16-bit operands as bfloat16: comparison line of config 2's "bf16")
    else:
        ops.CONV_PRECISION = args.precision
    opt = make_opt(args.model, args.batch)
    if args.loop in ("a", "der", "lwf", "ewc"):
        fn = {"a": time_loop_a, "der": time_der_step, "lwf": time_il_step, "ewc": time_il_step}[args.loop]
        kw = {"with_cpu_baseline": not args.no_cpu_baseline} if args.loop == "a" else {}
        res = fn(args, opt, rank, world, args.steps, args.warmup, **kw)
        if rank == 0:
            what = {"a": f"MRN loop A: one {args.model.upper()} expert trained",
                    "der": f"DER step: DERNet over {args.experts} {args.model.upper()} extractors, newest trained",
                    "lwf": f"LwF task-1 step on one {args.model.upper()} network", "ewc": f"EWC task-1 step on one {args.model.upper()} network"}[args.loop]
            res.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                        "config": {"workload": f"{what} on 32x256x4 crops, random-init weights",
                                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}"}})
            emit(res)
        parallel.barrier()
        return
    learner = build_learner(opt, args.experts, quiet=not args.verbose)
    learner.model.module.expert_streams = not args.no_streams
    if args.serial:
        learner.model.module.expert_halves = 0
        args.no_pipeline = True
    data = SyntheticTextLines(opt, seed=111 + rank)
    data.set_characters(learner.character)
    dev = learner.device

    # Software pipeline of loop B (il_modules/mrn.py::_update_representation): the frozen experts' forward of batch n+1 is
    # issued (side streams) before batch n's router forward / backward / Adam (main stream).  Every timed step still
    # consists of one expert forward of all experts + one router step; --no-pipeline runs them strictly in sequence.
    def fetch():
        image, labels, idx = data.get_batch2()
        indexs = to_device(torch.LongTensor(idx).squeeze())
        pre = learner.prefetch_experts(image, labels) if not args.no_pipeline else None
        return image, labels, indexs, pre if (pre is not None and pre[0] is not None) else None

    pending = [fetch()]

    def step():
        image, labels, indexs, pre = pending.pop()
        pending.append(fetch())
        return learner.routing_step(image, labels, indexs, prefetched=pre)

    for _ in range(args.warmup):
        step()
    reducer = getattr(learner, "reducer", None)
    if reducer is not None:
        reducer.record, reducer.exposed = True, []
    sites = None
    if not args.no_kernel_timer:
        # the timed region brackets the DOMINANT kernel's launches only (site "wino": the Winograd convolutions of the TRBA / CRNN experts:
        # 26 launches a step); HIP events around every timed launch cost ~0.9 ms of an 84 ms step (A/B, round 5).  The other kernels'
        # figures of the detail file come from a short pass of its own behind the timed region.  SVTR has no such site: every launch.
        only = os.environ.get("MRN_TIMER_ONLY")
        sites = set(only.split(",")) if only else ({"wino"} if (args.model in ("trba", "crnn") and not args.timer_all) else None)
        ops.CONV_TIMER = ops.KernelTimer(sites)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_clf, loss_t = step()
    host_elapsed = time.perf_counter() - t0          # time the host needed to ISSUE the timed steps (it runs ahead of the GPU)
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = time.perf_counter() - t0
    timer = ops.CONV_TIMER
    ops.CONV_TIMER = None
    timer_steps, timer_elapsed = args.steps, elapsed
    if timer is not None and sites is not None and not timer.spans:       # (no launch of the dominant site -- e.g. --precision f32 -- : every launch, below)
        timer = None
    # detail pass (outside the timed region, every rank): the same pipelined steps with events around EVERY timed launch
    detail = None
    if not args.no_kernel_timer and sites is not None:
        ops.CONV_TIMER = ops.KernelTimer()
        torch.cuda.synchronize()
        td = time.perf_counter()
        n_detail = min(3, args.steps)
        for _ in range(n_detail):
            step()
        torch.cuda.synchronize()
        detail = (ops.CONV_TIMER, n_detail, time.perf_counter() - td)
        ops.CONV_TIMER = None
        if timer is None:
            timer, detail = detail[0], None
            timer_steps, timer_elapsed = n_detail, time.perf_counter() - td
    # Isolated pass (outside the timed region): the same workload with ONE lock-step group on ONE stream and no
    # look-ahead, so every launch of the dominant kernel has the GPU to itself -- its own rate, next to the in-situ
    # rate of the timed region where the experts' stream(s) and the router phase share the chip.
    isolated = None
    if timer is not None and not args.no_isolated_pass:     # (every rank: routing_step holds the gradient all-reduce)
        net = learner.model.module
        saved, net.expert_halves = net.expert_halves, 0
        pending.clear()
        image, labels, idx = data.get_batch2()
        indexs = to_device(torch.LongTensor(idx).squeeze())
        learner.routing_step(image, labels, indexs)            # (re-packs nothing: same weights; warms the G = 6 path)
        ops.CONV_TIMER = ops.KernelTimer()
        for _ in range(2):
            learner.routing_step(image, labels, indexs)
        isolated = ops.CONV_TIMER.summary()
        ops.CONV_TIMER = None
        net.expert_halves = saved
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    comm = comm_telemetry(reducer, world, dev, elapsed / args.steps * 1e3) if world > 1 else None
    # live power probe of the dominant layer shape (rank 0, TRBA x3 mode only: the shape is TRBA's)
    probe = None
    if rank == 0 and timer is not None and not args.no_power_probe and args.model == "trba" and ops.X3_PRODUCTS == 3 \
            and ops.CONV_PRECISION in ("auto", "fp16x3"):
        probe = power_probe(ops, ops.WINO_R if ops.WINO_R in (2, 4) else 0)

    if rank == 0:
        res = {
            "metric": "text-line images/sec (fwd+bwd) at 32x256, TRBA+MRN 6 experts" if (args.model, args.experts) == ("trba", 6)
            else f"text-line images/sec (fwd+bwd) at 32x256, {args.model.upper()}+MRN {args.experts} experts",
            "value": world * args.batch * args.steps / elapsed,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "host_issue_ms_per_step": host_elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "skipped_optimizer_steps": learner.optimizer.skipped_steps(),      # non-finite gradient norms (csrc/optim.hip): 0 or the line is suspect
            "dtype": ("bf16" if ops.REDUCED_BF16 else "fp16") if ops.X3_PRODUCTS == 1 else "f32",
            "arithmetic": "one fp16 MFMA product per term, fp32 accumulate (reduced mode, not the headline)" if ops.X3_PRODUCTS == 1 else
            {"auto": "split-fp16 x3 MFMA products (22-bit), fp32 accumulate", "f32": "exact fp32 MFMA",
             "fp16x3": "split-fp16 x3 MFMA products (22-bit), fp32 accumulate"}[ops.CONV_PRECISION],
            "data": "synthetic",
            "config": {"workload": f"MRN loop B: {args.model.upper()}x{args.experts} frozen experts fwd + DM-Router fwd/bwd + clip + Adam, 32x256x4 crops",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "classes": [sum(CLASSES_MLT19[:i + 1]) + (5 if args.model == "trba" else 4) for i in range(args.experts)],
                       "loss_clf": loss_clf.detach().item(), "loss_taski": loss_t.detach().item(),
                       "parity": "tests/test_model_gpu.py (bands and index-agreement rates: DESIGN.md section 2)"},
        }
        if timer is not None and timer.spans:
            pmc_sec = "kernels" if (args.model, args.experts, args.batch) == ("trba", 6, 256) else ("kernels_svtr" if args.model == "svtr" else "none")
            rl, hbm = roofline_entries(timer.summary(), timer_steps, timer_elapsed, pmc_section=pmc_sec)
            if detail is not None:          # every other timed kernel: from the detail pass (its own steps and wall time)
                rl2, hbm2 = roofline_entries(detail[0].summary(), detail[1], detail[2], pmc_section=pmc_sec)
                lead = rl[0]["kernel"] if rl else None
                rl, hbm = rl[:1] + [r for r in rl2 if r["kernel"] != lead], hbm2
            if not rl:                      # (every timed family under the ridge: the largest HBM-bound one leads)
                hbm.sort(key=lambda r: -r["kernel_share_of_step"])
                rl, hbm = hbm[:1], hbm[1:]
            res["roofline"] = rl[0]
            if probe is not None:
                res["roofline"]["power_probe"] = probe
            net = learner.model.module
            groups = net._half_groups(True) if hasattr(net, "_half_groups") else None
            res["roofline"]["measured"] = ("timed region, HIP events per launch on the launch stream(s), rate over the union of the launch "
                                           "intervals; %d lock-step sub-group(s)" % (len(groups) if groups else 1))
            if isolated:
                k = max((k for k in isolated if not k.startswith("hbm/")), key=lambda k: isolated[k]["total_ms"])
                i_ = isolated[k]
                ms = i_["total_ms"] / i_["launches"]
                ach = i_["total_flops"] / i_["launches"] / (ms * 1e-3) / 1e12
                kname, peak, per_flop = describe_kernel(k)
                res["roofline"]["isolated"] = {
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "mfma_issue_frac": ach * per_flop / peak,
                    "avg_launch_ms": ms, "launches_per_step": i_["launches"] / 2, "kernel": kname.split(" (")[0],
                    "measured": "2 serialized steps behind the timed region (python bench.py --serial reproduces it)"}
            if len(rl) > 1 or hbm:
                res["roofline_other_kernels"] = rl[1:] + hbm
        if comm is not None:
            res["comm"] = comm
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(learner, opt, args.experts)
    extra = None
    reduced = {}
    if not args.no_extra:                  # (every rank: loop A's train_step holds the gradient all-reduce)
        del learner
        pending.clear()
        torch.cuda.empty_cache()
        extra = time_loop_a(args, opt, rank, world, steps=5, warmup=2, with_cpu_baseline=not args.no_cpu_baseline)
        if world == 1 and args.model == "trba" and args.precision == "auto" and not args.no_reduced:
            # the reduced-precision mode BASELINE configs 2 ("bf16") and 5 ("fp16 MFMA") name, driver-timed in the default run: ONE fp16
            # product per term (fp16 keeps 11 significand bits where bf16 keeps 8; same MFMA rate), fp32 accumulate -- short lines
            # BASELINE config 5 in the PARITY mode (split-fp16 x3 products everywhere): the DER step over six TRBA extractors, short line
            reduced["der"] = {k: v for k, v in time_der_step(args, opt, rank, world, 5, 3).items()
                              if k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "trainable_parameters")}
            torch.cuda.empty_cache()
            saved = (ops.X3_PRODUCTS, ops.TRAIN_PRODUCTS)
            ops.X3_PRODUCTS = ops.TRAIN_PRODUCTS = 1
            try:
                reduced["fp16_loop_b"] = time_loop_b_short(args, opt, steps=5, warmup=3)
                torch.cuda.empty_cache()
                reduced["fp16_der"] = {k: v for k, v in time_der_step(args, opt, rank, world, 5, 3).items()
                                       if k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype")}
            finally:
                ops.X3_PRODUCTS, ops.TRAIN_PRODUCTS = saved
            torch.cuda.empty_cache()
            # the other recogniser families of BASELINE.json's configs (2: CRNN x 3 experts, 4: SVTR), same loop B, short lines:
            # driver-timed evidence for the kernels only they exercise (VGG stack, fused SVTR mixing blocks, CTC heads)
            for key, model, n_exp in (("crnn3_loop_b", "crnn", 3), ("svtr6_loop_b", "svtr", 6)):
                a2 = argparse.Namespace(**vars(args))
                a2.model, a2.experts = model, n_exp
                line = time_loop_b_short(a2, make_opt(model, args.batch), steps=20, warmup=6)      # (10-22 ms steps: 20 of them cost half a second)
                line["metric"] = f"text-line images/sec (fwd+bwd) at 32x256, {model.upper()}+MRN {n_exp} experts (not the headline)"
                reduced[key] = line
                torch.cuda.empty_cache()
    if rank == 0:
        if extra is not None:
            res["extra"] = {"loop_a": extra}
            res["extra"].update(reduced)
        emit(res)
    parallel.barrier()


if __name__ == "__main__":
    main()
