#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace rocpd database: for the LAST bench step (kernels after the last
`adam_kernel` but one), busy/idle time of the GPU and the phase boundaries (backbone convs / heads / router).

    python tools/rocprof_timeline.py gpurun_out/x/trace_results.db
"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) >= 2:
        rows = rows[adam[-2] + 1: adam[-1] + 1]
    t0 = rows[0][1]
    span = (rows[-1][2] - t0) / 1e6
    busy, end = 0, t0
    gaps = []
    for i, (name, s, e, st) in enumerate(rows):
        if s > end:
            gaps.append((s - end, rows[i - 1][0][:50] if i else "", name[:50], (end - t0) / 1e6))
            busy += e - s
        else:
            busy += max(0, e - max(end, s))
        end = max(end, e)
    print(f"last step: {len(rows)} kernels, span {span:.2f} ms, GPU busy (union) {busy / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms")
    print("largest idle gaps (us, at ms, after -> before):")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"  {g[0] / 1e3:8.1f} us at {g[3]:7.2f} ms   {g[1]} -> {g[2]}")
    # phases
    def last_end(pred):
        es = [e for n, s, e, st in rows if pred(n)]
        return (max(es) - t0) / 1e6 if es else float("nan")
    def first_start(pred):
        ss = [s for n, s, e, st in rows if pred(n)]
        return (min(ss) - t0) / 1e6 if ss else float("nan")
    print(f"backbone convs end at {last_end(lambda n: 'conv_x3' in n or 'conv_bf16' in n):.2f} ms; "
          f"lstm {first_start(lambda n: 'lstm_layer' in n):.2f}..{last_end(lambda n: 'lstm_layer' in n):.2f} ms; "
          f"attn decoder ..{last_end(lambda n: 'attn_decoder' in n):.2f} ms; router from {first_start(lambda n: 'layernorm_fwd' in n):.2f} ms; "
          f"step end {span:.2f} ms")
    t_heads = max([e for n, s_, e, st in rows if "attn_decoder" in n] or [t0])
    tail = {}
    for n, s_, e, st in rows:
        if s_ >= t_heads:
            d = tail.setdefault(n[:80], [0, 0])
            d[0] += 1
            d[1] += e - s_
    print("after the last attention decoder (router fwd/bwd, losses, optimiser):")
    for n, v in sorted(tail.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {v[1] / 1e3:8.1f} us  x{v[0]:<3d} {n}")
    streams = {}
    for n, s, e, st in rows:
        d = streams.setdefault(st, [0, 0])
        d[0] += 1
        d[1] += e - s
    print("per stream: " + ", ".join(f"{k}: {v[0]} kernels {v[1] / 1e6:.1f} ms" for k, v in sorted(streams.items(), key=lambda kv: str(kv[0]))))


if __name__ == "__main__":
    main()
