#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --precondition-ms 0`
into the two small JSON files bench.py reads for its roofline block:

    python legged-robot-movability-cuda_amd/tools/summarize_profiles.py PMC_DIR POINTS MODE OUT_PREFIX

PMC_DIR holds one sub-directory per pass (collected separately, each with --kernel-trace --pmc <counters> only, as
MI355X_MICROARCH.md prescribes): fetch/ (FETCH_SIZE), write/ (WRITE_SIZE), sq/ (SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
SQ_BUSY_CYCLES).  Writes OUT_PREFIX_hbm_traffic.json and OUT_PREFIX_valu.json.

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024: gfx950 tallies a 128-byte read request as 64 bytes in FETCH_SIZE
(MI355X_MICROARCH.md, HBM section).  A "step" of a mode is every kernel its launch issues (tolerance mode: the
tolerance kernel + its fix-up)."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from srchash import kernel_src_sha  # noqa: E402  (the tree the passes were collected from: run this on the same snapshot)

STEP_KERNELS = {
    # whichever main kernel the build launches (the second template argument is the layout: false = one array per component)
    "tol": ["dist_tab_kernel<2, false, false>", "dist_tab_kernel<2, false>", "dist_tol_staged_kernel<2, false>", "dist_tol_staged_kernel<2>", "dist_tol_kernel<2>", "tol_fixup_kernel<2, false, 8, 128>", "tol_fixup_kernel<2, false, 8>", "tol_fixup_kernel<2, false>", "tol_fixup_kernel<2>"],
    "tol_rel": ["dist_tab_kernel<2, false, true>", "tol_fixup_kernel<2, false, 8, 128>"],
    "fast": ["dist_xtab_kernel<2, false>", "tol_fixup_kernel<2, false, 8, 128>", "dist_soa_kernel<2, true>"],
    "strict": ["dist_soa_kernel<2, false>"],
}


def per_kernel(directory, counter):
    acc = {}
    for path in glob.glob(directory + "/**/*_counter_collection.csv", recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                name = re.sub(r"\(.*$", "", name).strip()
                acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return acc


def main():
    pmc, points, mode, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    fetch, write = per_kernel(pmc + "/fetch", "FETCH_SIZE"), per_kernel(pmc + "/write", "WRITE_SIZE")
    kernels = {}
    for name in fetch:
        if name not in write:
            continue
        f, w = sum(fetch[name]) / len(fetch[name]), sum(write[name]) / len(write[name])
        kernels[name] = {"FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "dispatches": len(fetch[name]),
                         "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    step = [k for k in STEP_KERNELS[mode] if k in kernels]
    traffic = {
        "points_per_launch": points, "mode": mode, "step_kernels": step, "kernel_src_sha": kernel_src_sha(),
        "step_hbm_bytes": sum(kernels[k]["hbm_bytes_per_launch"] for k in step),
        "algorithmic_bytes": 25 * points + points // 8,
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                   f"--mode {mode} --steps 3 --warmup 1 --no-cpu-baseline --no-extras --precondition-ms 0",
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 counts 128-B read requests as 64 B)",
        "kernels": {k: kernels[k] for k in kernels if "dist" in k or "tol" in k or "reach" in k},
    }
    json.dump(traffic, open(prefix + "_hbm_traffic.json", "w"), indent=1)
    valu, salu, waves = per_kernel(pmc + "/sq", "SQ_INSTS_VALU"), per_kernel(pmc + "/sq", "SQ_INSTS_SALU"), per_kernel(pmc + "/sq", "SQ_WAVES")
    per = {}
    for name in valu:
        if not ("dist" in name or "tol" in name or "reach" in name):
            continue
        per[name] = {"SQ_INSTS_VALU_mean": sum(valu[name]) / len(valu[name]),
                     "SQ_INSTS_SALU_mean": sum(salu.get(name, [0])) / max(len(salu.get(name, [0])), 1),
                     "SQ_WAVES_mean": sum(waves.get(name, [0])) / max(len(waves.get(name, [0])), 1)}
    step_valu = sum(per[k]["SQ_INSTS_VALU_mean"] for k in step if k in per)
    out = {"mode": mode, "points_per_launch": points, "step_kernels": step, "kernel_src_sha": kernel_src_sha(),
           "valu_insts_per_eval": step_valu * 64.0 / points,  # SQ_INSTS_VALU counts wave instructions: 64 lanes each
           "salu_insts_per_eval": sum(per[k]["SQ_INSTS_SALU_mean"] for k in step if k in per) * 64.0 / points,
           "note": "wave-level VALU instructions issued per evaluated point (all launches of one step); the kernel's own issue "
                   "floor is this x points / 64 / 1024 SIMDs x 2 cycles at 2.4 GHz (every instruction in the full-rate class)",
           "kernels": per}
    json.dump(out, open(prefix + "_valu.json", "w"), indent=1)
    print(json.dumps({"step_hbm_bytes": traffic["step_hbm_bytes"], "valu_insts_per_eval": out["valu_insts_per_eval"]}))


if __name__ == "__main__":
    main()
