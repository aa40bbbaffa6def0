#!/bin/bash
# Evidence for BASELINE config 3 (the pair kernel reach_any_wave_kernel) on the GPU box, from the repository root:
#   tools/c3_profile.sh r03     -> gpurun_out/r03_c3/: kernel stats, SQ counters, evaluated-pair counts, r03_c3_evidence.json
# Needs csrc/build/variants/liblrm_count.so (make variant NAME=count FLAGS=-DLRM_PAIR_COUNT) for the counts.
set -o pipefail
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}_c3
tool=$root/legged-robot-movability-cuda_amd/tools/c3_evidence.py
mkdir -p $out
python3 $tool --reps 100 > $out/timing.json
LRM_LIB_PATH=$root/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_count.so python3 $tool --count > $out/counts.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $tool --reps 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- python3 $tool --reps 3 --warm 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc_sq2 -- python3 $tool --reps 3 --warm 1 > /dev/null 2>&1
cd $root
python3 - <<PY
import csv, glob, json, sys
sys.path.insert(0, "$root/legged-robot-movability-cuda_amd")
from srchash import kernel_src_sha
out = "$out"
timing, counts = json.load(open(out + "/timing.json")), json.load(open(out + "/counts.json"))
stats = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reach_any_wave_kernel" in r["Name"]:
            stats = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"])}
pmc = {}
for f in glob.glob(out + "/pmc_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "reach_any_wave_kernel" in r["Kernel_Name"]:
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
pmc = {k: sum(v) / len(v) for k, v in pmc.items()}
ms = stats.get("average_ns", timing["ms"] * 1e6) * 1e-6
ev = counts["pairs_evaluated"]
valu_lane_ops = pmc.get("SQ_INSTS_VALU", 0.0) * 64.0
rec = {"kernel": "reach_any_wave_kernel<true>", "kernel_src_sha": kernel_src_sha(), **timing, **counts,
       "rocprofv3_kernel_ms": ms, "rocprofv3": stats, "pmc_per_launch": pmc,
       "evals_per_s": ev / (ms * 1e-3), "pairs_answered_per_s": counts["pairs_answered"] / (ms * 1e-3),
       "valu_insts_per_eval": valu_lane_ops / ev if ev else None,
       # FP32 vector peak: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = 78.6e12 lane-ops / s
       "frac_valu": (valu_lane_ops / (ms * 1e-3)) / 78.6e12,
       "note": "pairs_evaluated = full lrm_reachable_rotate_leg evaluations of one launch (a -DLRM_PAIR_COUNT build); valu_insts_per_eval = "
               "64 x SQ_INSTS_VALU / pairs_evaluated: ALL of the kernel's vector instructions (box walks, sphere culls, queueing) per full evaluation"}
json.dump(rec, open(out + "/${tag}_c3_evidence.json", "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("rocprofv3_kernel_ms", "pairs_evaluated", "evals_per_s", "valu_insts_per_eval", "frac_valu")}))
PY
