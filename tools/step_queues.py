#!/usr/bin/env python3
"""Per-queue view of the ordered launch list `tools/rocprof_step.py --list` writes (step.md): kernel time per queue and kernel,
and the busy fraction of every queue per millisecond of the step.
    python tools/step_queues.py gpurun_out/<tag>/step.md [top]"""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    rows = []
    for line in open(path):
        m = re.match(r"\s*([\d.]+) us\s+\+\s*([\d.]+)\s+q(\d+)\s+(.*)", line)
        if m:
            rows.append((float(m[1]), float(m[2]), int(m[3]), m[4]))
    queues = sorted({q for _, _, q, _ in rows})
    for q in queues:
        agg = collections.defaultdict(lambda: [0, 0.0])
        for _, d, k, n in rows:
            if k == q:
                a = agg[n.split("(")[0][:70]]
                a[0] += 1
                a[1] += d
        print(f"--- queue {q}: {sum(a[1] for a in agg.values()) / 1e3:.2f} ms in {sum(a[0] for a in agg.values())} launches")
        for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"{t / 1e3:7.2f} ms x{c:4d} {name}")
    T = int(max(s + d for s, d, _, _ in rows) / 1000) + 1
    busy = {q: [0.0] * T for q in queues}
    for s, d, q, _ in rows:
        a, e = s, s + d
        while a < e:
            b = int(a / 1000)
            nxt = min(e, (b + 1) * 1000)
            busy[q][b] += nxt - a
            a = nxt
    print("--- busy % per ms: " + "  ".join(f"q{q}" for q in queues))
    for b in range(T):
        print(f"{b:3d}  " + "  ".join(f"{busy[q][b] / 10:4.0f}" for q in queues))


if __name__ == "__main__":
    main()
