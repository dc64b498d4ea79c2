#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite, the default output of ROCm 7.2's rocprofv3)
into the per-kernel table that `--stats` prints: calls, total/avg/min/max duration, share.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_name.md
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                            "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of `{path}`\n")
    print(f"total kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f" ({total / 1e6 / steps:.2f} ms per step over {steps:g} steps incl. warm-up)" if steps else "") + "\n")
    print("| kernel | calls | total ms | % | avg us | min us | max us | vgpr | agpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
        name = name.replace("(anonymous namespace)::", "").replace("|", "\\|")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"| `{name}` | {n} | {tot / 1e6:.2f} | {100.0 * tot / total:.1f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {vg} | {ag} | {lds} |")


if __name__ == "__main__":
    main()
