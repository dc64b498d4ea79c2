#!/bin/bash
# usage (GPU box, repo root): bash tools/bench_all_lines.sh <outfile.jsonl>   -- every bench line quoted in DESIGN.md section 5, one JSON object per line
out=${1:-gpurun_out/bench_lines.jsonl}
: > $out
for args in "" "--precision fp16" "--loop a" "--loop a --model crnn" "--loop a --model svtr" "--loop der" "--precision fp16 --loop a" "--precision fp16 --loop der" "--loop lwf" "--loop ewc" "--model crnn --experts 3" "--model svtr"; do
  extra="--no-cpu-baseline --no-extra"
  [ -z "$args" ] && extra=""
  timeout 900 python bench.py --steps 8 --warmup 3 $extra $args 2>/dev/null | tail -1 >> $out
done
wc -l $out
