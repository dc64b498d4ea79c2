#!/bin/bash
# usage (build container, repo root, after `bash tools/refresh_evidence.sh r06` on the GPU box merged gpurun_out/): copy the judged
# summaries into profiles/ under the round's names -- gpurun_out/ is scratch, profiles/ is tracked
r=${1:-r06}
put() {  # put <tag> <profiles name>
  [ -f gpurun_out/$1/step.md ] || { echo "missing gpurun_out/$1/step.md"; return; }
  { cat gpurun_out/$1/step.md; printf '\n---\n\n## whole trace (warm-up included)\n\n'; cat gpurun_out/$1/summary.md 2>/dev/null; } > profiles/$2
}
put ${r}_trba6 ${r}a_trba6_loop_b_kernel_trace.md
put ${r}_trba6_serial ${r}b_trba6_loop_b_serial_kernel_trace.md
put ${r}_loopa ${r}c_trba_loop_a_kernel_trace.md
put ${r}_svtr6_serial ${r}d_svtr6_loop_b_serial_kernel_trace.md
put ${r}_crnn3 ${r}e_crnn3_loop_b_kernel_trace.md
put ${r}_der ${r}f_trba6_der_step_kernel_trace.md
put ${r}_trba6_fp16_serial ${r}g_trba6_fp16_loop_b_serial_kernel_trace.md
put ${r}_svtr_loopa ${r}h_svtr_loop_a_kernel_trace.md
cp gpurun_out/${r}_bench_default_line.json profiles/${r}_bench_default_line.json
cp gpurun_out/${r}_bench_default_detail.json profiles/${r}_bench_default_detail.json
cp gpurun_out/${r}_bench_lines.jsonl profiles/${r}_bench_lines.jsonl
python3 tools/pmc_to_profile.py gpurun_out/${r}_pmc/pmc_summary.json gpurun_out/${r}_pmc_loopa/pmc_summary.json \
  kernels_svtr=gpurun_out/${r}_pmc_svtr/pmc_summary.json kernels_svtr_loop_a=gpurun_out/${r}_pmc_svtr_a/pmc_summary.json > profiles/${r}_pmc.json
ls -la profiles/${r}*
