#!/bin/bash
# the decoder kernels' samples-per-workgroup (forward MRN_ATTN_VB, backward MRN_ATTN_BWD_VB) inside the DER step and loop A (GPU box, repo root)
for cfg in "1 1" "2 2" "4 4" "4 2" "2 4"; do set -- $cfg
  for loop in der a; do
    v=$(MRN_ATTN_VB=$1 MRN_ATTN_BWD_VB=$2 python bench.py --loop $loop --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "fwd_vb $1 bwd_vb $2 loop $loop: $v"
  done
done
