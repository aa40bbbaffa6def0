"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_hbm_traffic.json.

    python legged-robot-movability-cuda_amd/tools/summarize_pmc.py \
        gpurun_out/r1c/pmc_fetch gpurun_out/r1c/pmc_write 10000000 fast > profiles/r01c_hbm_traffic.json

The passes are collected separately, each with `--kernel-trace --pmc <counter>` only (MI355X_MICROARCH.md,
HBM / rocprofv3 section). gfx950 counts a 128-byte read request as 64 bytes in FETCH_SIZE, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024. bench.py reads the newest such file for `roofline.traffic`.
"""
import csv
import glob
import json
import re
import sys


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
    fetch_dir, write_dir, points, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in fetch:
        if name not in write or "warmup" in name:
            continue
        f = sum(fetch[name]) / len(fetch[name])
        w = sum(write[name]) / len(write[name])
        kernels[name] = {
            "FETCH_SIZE_KB_mean": f, "FETCH_SIZE_dispatches": len(fetch[name]),
            "WRITE_SIZE_KB_mean": w, "WRITE_SIZE_dispatches": len(write[name]),
            "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
        }
    print(json.dumps({
        "points_per_launch": points, "mode": mode,
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv "
                   "-- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950: FETCH_SIZE counts 128-B read "
                      "requests as 64 B; MI355X_MICROARCH.md, HBM)",
        "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
