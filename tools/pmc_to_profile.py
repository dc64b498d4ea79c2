#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_summary.json (tools/pmc_pass.sh) -> profiles/r01_pmc.json: per kernel, HBM bytes per launch
(FETCH_SIZE x 2 + WRITE_SIZE, KiB counters; the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md), MFMA
utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)) and L2 hit rate.

    python tools/pmc_to_profile.py gpurun_out/r1l/pmc_summary.json "bench.py --steps 1 --warmup 1" > profiles/r01_pmc.json
"""
import json
import sys


def kernels_of(d):
    names = set(d.get("FETCH_SIZE", {})) | set(d.get("WRITE_SIZE", {}))
    out = {"kernels": {}}
    for k in sorted(names):
        f, w = d.get("FETCH_SIZE", {}).get(k), d.get("WRITE_SIZE", {}).get(k)
        if f is None or w is None:
            continue
        ent = {"launches": f["n"], "fetch_size_kib": f["sum"], "write_size_kib": w["sum"],
               "hbm_bytes_per_launch": (2.0 * f["sum"] / f["n"] + w["sum"] / w["n"]) * 1024.0}
        g, m = d.get("GRBM_GUI_ACTIVE", {}).get(k), d.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get(k)
        if g and m and g["sum"] > 0:
            ent["mfma_util"] = m["sum"] / (g["sum"] / 8.0 * 1024.0)
        lc, la = d.get("SQ_LDS_BANK_CONFLICT", {}).get(k), d.get("SQ_LDS_IDX_ACTIVE", {}).get(k)
        if lc and la and la["sum"] > 0:
            ent["lds_bank_conflict_frac"] = lc["sum"] / la["sum"]
        h, ms = d.get("TCC_HIT_sum", {}).get(k), d.get("TCC_MISS_sum", {}).get(k)
        if h and ms and h["sum"] + ms["sum"] > 0:
            ent["l2_hit_rate"] = h["sum"] / (h["sum"] + ms["sum"])
        out["kernels"][k.replace("void ", "").split("(")[0].replace(", false, 3, 0>", ">").replace(", false, 3>", ">").replace(", false>", ">")] = ent      # (HL_OUT = false, NPROD = 3, WINO = 0: the default variant)
    return out["kernels"]


def main():
    """pmc_to_profile.py <loop-B pmc_summary.json> [<loop-A pmc_summary.json>] [section=<pmc_summary.json> ...]
    (e.g. kernels_svtr=gpurun_out/r06_pmc_svtr/pmc_summary.json: the SVTR x 6 loop-B step's kernels)"""
    pos = [a for a in sys.argv[1:] if "=" not in a]
    named = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
    out = {"command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra "
                      "--no-power-probe [--loop a] [--model svtr]  (one run per counter group, tools/pmc_pass.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches (the x2 is the gfx950 "
                    "FETCH_SIZE correction of MI355X_MICROARCH.md); Infinity-Cache hits are counted; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
           "kernels": kernels_of(json.load(open(pos[0])))}
    if len(pos) > 1:
        out["kernels_loop_a"] = kernels_of(json.load(open(pos[1])))
    for name, path in named:
        out[name] = kernels_of(json.load(open(path)))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
