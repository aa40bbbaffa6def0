"""sha256 over the kernel sources a profile was taken with.

tools/summarize_profiles.py writes it into profiles/r*_hbm_traffic.json / r*_valu.json and bench.py prints those
PMC figures only when the hash equals the tree's: a kernel change without a fresh PMC pass shows up as "stale"
instead of silently mixing rounds."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# every file the timed kernels are compiled from (the .hip translation units and the headers they include)
KERNEL_SOURCES = ("lrm_tol_kernels.hip", "lrm_point_tol.h", "lrm_point_xtab.h", "lrm_kernels.hip", "lrm_point_fast.h", "lrm_point.h",
                  "lrm_exact_math.h", "lrm_types.h", "lrm_launch.h", "lrm_toltab.cpp", "lrm_toltab_build.h", "lrm_toltab_dev.hip", "Makefile")


def kernel_src_sha():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode() + b"\0")
        with open(os.path.join(_CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
