#!/usr/bin/env python3
"""Streaming vs weight-stationary LSTM layer of the frozen experts (rnn.hip): bit-equality and time per launch.
    python tools/bench_lstm.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrn_amd import ops  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    H = 256
    SHAPES = ((1, 256, 65), (3, 256, 65), (3, 256, 63), (4, 256, 65), (6, 256, 65), (3, 100, 65))
    for (G, B, T) in SHAPES[:int(os.environ.get("NSHAPES", "99"))]:
        torch.manual_seed(G * 1000 + B)
        xproj = torch.randn(G, B, T, 2 * 4 * H, device="cuda")
        ws = [[torch.randn(4 * H, H, device="cuda") * 0.06 for _ in range(2)] for _ in range(G)]
        packs = [[ops.pack_fragment_major_h(w) for w in pair] for pair in ws]
        w_h = torch.stack([torch.stack([d[0] for d in p]) for p in packs]).contiguous()
        w_inv = torch.stack([torch.cat([d[1] for d in p]) for p in packs]).contiguous()
        b_hh = torch.randn(G, 2 * 4 * H, device="cuda") * 0.1
        ref = ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, H, 2)
        ms_old = timeit(lambda: ops.lstm_layer_x3_grouped(xproj, w_h, w_inv, b_hh, H, 2), reps)
        spk = [ops.pack_lstm_steps_weights(pair) for pair in ws]
        w_s, w_sinv = torch.stack([p[0] for p in spk]).contiguous(), torch.stack([p[1] for p in spk]).contiguous()
        out_s = ops.lstm_layer_x3_steps(xproj, w_s, w_sinv, b_hh, H, 2)
        ms_steps = timeit(lambda: ops.lstm_layer_x3_steps(xproj, w_s, w_sinv, b_hh, H, 2), reps)
        print(f"G{G} B{B} T{T}: streaming {ms_old:.3f} ms ({ms_old * 1e3 / T:.1f} us/step) | step kernels from a graph {ms_steps:.3f} ms "
              f"({ms_steps * 1e3 / T:.1f} us/step), max |diff| {(out_s - ref).abs().max().item():.2e}", flush=True)
        if "--steps-only" in sys.argv:
            continue
        if ops.call("mrn_lstm_cluster_workgroups", G, B, 2) > 256:
            print(f"G{G} B{B} T{T}: streaming {ms_old:.3f} ms | cluster: does not fit")
            continue
        out = ops.lstm_layer_x3_cluster(xproj, w_h, w_inv, b_hh, H, 2)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        err = (out - ref).abs().max().item()
        ms_new = timeit(lambda: ops.lstm_layer_x3_cluster(xproj, w_h, w_inv, b_hh, H, 2), reps)
        if os.environ.get("CL_PROFILE"):      # what-if build (tools/build_probe.sh MRN_CL_PROFILE rnn.hip): phase clocks of workgroup 0
            nbytes = ops.call("mrn_lstm_cluster_workspace_bytes", G, B, 2)
            ws = torch.zeros(nbytes, device="cuda", dtype=torch.uint8)
            o = torch.empty(G, B, T, 2 * H, device="cuda")
            ops.call("mrn_lstm_layer_fwd_x3_cluster", ops._ptr_array([xproj[g].data_ptr() for g in range(G)]),
                     ops._ptr_array([w_h[g].data_ptr() for g in range(G)]), ops._ptr_array([w_inv[g].data_ptr() for g in range(G)]),
                     ops._ptr_array([b_hh[g].data_ptr() for g in range(G)]), ops._ptr_array([o[g].data_ptr() for g in range(G)]), G, B, T,
                     H, 2, ops._p(ws), nbytes, ops._stream())
            torch.cuda.synchronize()
            clk = ws[512:576].view(torch.int64).tolist()
            print("   phase clocks per step (0 wait+barrier, 1 loads+mfma, 2 final barrier, 3 signal, 4 pointwise, 5 stores, 6 release):",
                  [round(c / (T - 1)) for c in clk])
        print(f"G{G} B{B} T{T}: streaming {ms_old:.3f} ms ({ms_old / T * 1e3:.1f} us/step) | cluster {ms_new:.3f} ms "
              f"({ms_new / T * 1e3:.1f} us/step) | bit-identical {same} (max diff {err:.2e}, nan {bool(torch.isnan(out).any())})", flush=True)


if __name__ == "__main__":
    main()
