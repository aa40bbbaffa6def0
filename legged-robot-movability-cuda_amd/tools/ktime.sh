#!/bin/bash
# rocprofv3 averages (us) of the tolerance mode's two kernels for library variants on ONE box:
#   tools/ktime.sh NAME...   (csrc/build/variants/liblrm_NAME.so; "default" = the in-tree liblrm.so), 600 launches each, the
#   first 300 (clock ramp) left out.
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  if [ "$a" != "default" ]; then export LRM_LIB_PATH=$GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so; else unset LRM_LIB_PATH; fi
  d=$GRAFT_REPO_ROOT/gpurun_out/ktime_$a
  rm -rf $d
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol --reps 600 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$d/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = "main" if ("dist_tol" in k or "dist_tab" in k) else "fix" if "fixup" in k else None
        if name: acc[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
out=[]
for k in ("main","fix"):
    v=[d for _,d in sorted(acc[k])][300:]
    out.append("%s %.2f (min %.2f, %d launches)" % (k, sum(v)/max(1,len(v))/1e3, min(v)/1e3 if v else 0, len(v)))
print("$a", "; ".join(out))
PY
done
