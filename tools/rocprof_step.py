#!/usr/bin/env python3
"""Steady-state view of a rocprofv3 --kernel-trace database (rocpd sqlite): the dispatches between the last two launches of a marker
kernel (default: adam_kernel = one optimiser step), i.e. ONE step without warm-up work (weight packing, first-use allocations).

    python tools/rocprof_step.py trace_results.db [marker] [--list]   ->  per-kernel table of that step (+ the ordered launch list)
"""
import collections
import sqlite3
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "adam_kernel"
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(cur.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print(f"fewer than two launches of {marker}")
        return
    step = rows[marks[-2] + 1: marks[-1] + 1]
    t0, t1 = step[0][1], step[-1][2]
    agg = collections.OrderedDict()
    for r in step:
        name = r[0].replace("(anonymous namespace)::", "")
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    total = sum(a[1] for a in agg.values())
    print(f"# one steady-state step of `{path}` (between the last two `{marker}` launches)\n")
    print(f"{len(step)} dispatches, kernel time {total / 1e6:.2f} ms summed over the streams, wall {(t1 - t0) / 1e6:.2f} ms\n")
    print("| kernel | launches | total ms | % | avg us |")
    print("|---|---:|---:|---:|---:|")
    for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 100 else name[:97] + "..."
        print(f"| `{short.replace('|', chr(92) + '|')}` | {n} | {tot / 1e6:.3f} | {100.0 * tot / total:.1f} | {tot / n / 1e3:.1f} |")
    if "--list" in sys.argv:
        print("\n```")
        for r in step:
            q = r[3] if qcol else ""
            print(f"{(r[1] - t0) / 1e3:10.1f} us  +{(r[2] - r[1]) / 1e3:8.1f}  q{q}  {r[0].replace('(anonymous namespace)::', '')[:110]}")
        print("```")


if __name__ == "__main__":
    main()
