#!/bin/bash
# Per-kernel averages (us, rocprofv3 kernel trace, the first 150 launches of every kernel left out) and VALU / SALU
# wave-instructions per point (a separate PMC pass) of tools/bench_modes.py in one arithmetic mode:
#   tools/kprof.sh TAG MODE [POINTS]     (env: LRM_LIB_PATH, LRM_XTAB, LRM_TOL_TABLE ... are passed through)
# -> gpurun_out/kprof_TAG_MODE.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; mode=$2; pts=${3:-10000000}
d=$root/gpurun_out/kprof_${tag}_${mode}
rm -rf $d; mkdir -p $d
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d/trace -- python3 $root/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes $mode --reps 400 --points $pts > $d/bench.json 2> /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $d/pmc -- python3 $root/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes $mode --reps 3 --points $pts > /dev/null 2>&1
python3 - <<PY > $root/gpurun_out/kprof_${tag}_${mode}.txt
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob("$d/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$d/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$tag $mode $pts points:", open("$d/bench.json").read().strip()[:400])
for k, v in sorted(acc.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v = [d for _, d in sorted(v)]
    if len(v) < 200: continue
    w = v[150:]
    line = "%-70s %5d launches  avg %8.2f us  min %8.2f" % (k[:70], len(v), sum(w) / len(w) / 1e3, min(w) / 1e3)
    if k in pmc:
        c = pmc[k]
        valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]); salu = sum(c["SQ_INSTS_SALU"]) / len(c["SQ_INSTS_SALU"])
        line += "  VALU/pt %7.1f  SALU/pt %6.1f  waves %d" % (valu * 64 / $pts, salu * 64 / $pts, sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]))
    print(line)
PY
cat $root/gpurun_out/kprof_${tag}_${mode}.txt
