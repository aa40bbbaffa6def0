#!/usr/bin/env python3
"""Where the tolerance mode's fix-up launch spends its time: builds nothing, needs the variant library made with
`make variant NAME=trace FLAGS=-DLRM_FIX_TRACE` (LRM_LIB_PATH points at it).  Prints, in microseconds relative to the
end of the last workgroup of the main kernel: start of the fix-up waves, tables + counts in LDS, prefix, each batch of
64 queued points, and the number of queued points per wave."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import lrm_amd
    n = 10_000_000
    rng = np.random.default_rng(42)
    lo, hi = np.array([-200, -500, -500], np.float32), np.array([700, 500, 300], np.float32)
    host = (rng.random((n, 3), dtype=np.float32) * (hi - lo) + lo).T.copy()
    cloud = torch.from_numpy(host).cuda()
    leg = lrm_amd.get_M2_leg(0.0)
    mask = torch.empty(n, dtype=torch.uint8, device="cuda")
    field = torch.empty((3, n), dtype=torch.float32, device="cuda")
    bits = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
    lrm_amd.set_mode(lrm_amd.MODE_TOL_REL if "--rel" in sys.argv else lrm_amd.MODE_TOL)  # --rel: the headline mode's launches
    for _ in range(300):  # steady clocks
        lrm_amd.device.reach_dist(cloud[0], cloud[1], cloud[2], leg, None, mask=mask, out=field, bits=bits)
    torch.cuda.synchronize()
    lib = C.CDLL(os.environ["LRM_LIB_PATH"])
    fix = np.zeros(8 * 4096, np.uint64)
    mn = np.zeros(2 * 32768, np.uint64)
    assert lib.lrm_dbg_fix_trace(fix.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p)) == 0
    fix = fix.reshape(4096, 8)
    mn_raw = mn.copy()
    mn = mn.reshape(32768, 2)
    ext = mn[16384:]  # table kernel: entry and end of the last wave of each workgroup
    mn = mn[:16384]
    ext = ext[mn[:, 1] > 0]
    mn = mn[mn[:, 1] > 0]
    fix = fix[fix[:, 0] > 0]
    t_end = mn[:, 1].max()
    us = lambda t: (t.astype(np.int64) - np.int64(t_end)) / 100.0
    print("main kernel: %d workgroups, first start %.2f us, last start %.2f us, first end %.2f us (all relative to the last end)"
          % (len(mn), us(mn[:, 0]).min(), us(mn[:, 0]).max(), us(mn[:, 1]).min()))
    dur = (mn[:, 1].astype(np.int64) - mn[:, 0].astype(np.int64)) / 100.0
    en = us(mn[:, 1])
    print("  workgroup durations (us): min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f;  ends: 5%% %.1f  median %.1f  95%% %.1f"
          % (dur.min(), np.percentile(dur, 5), np.median(dur), np.percentile(dur, 95), dur.max(), np.percentile(en, 5), np.median(en), np.percentile(en, 95)))
    if len(ext) and ext[:, 1].max() > 0:
        t_last = ext[:, 1].max()
        rel = lambda t: (t.astype(np.int64) - np.int64(t_last)) / 100.0
        print("  table kernel, relative to the end of its last wave: entry min %.1f max %.1f; staged min %.1f median %.1f max %.1f; last wave of a workgroup ends 5%% %.1f median %.1f 95%% %.1f"
              % (rel(ext[:, 0]).min(), rel(ext[:, 0]).max(), rel(mn[:, 0]).min(), np.median(rel(mn[:, 0])), rel(mn[:, 0]).max(),
                 np.percentile(rel(ext[:, 1]), 5), np.median(rel(ext[:, 1])), np.percentile(rel(ext[:, 1]), 95)))
        hw = mn_raw[49152:49152 + len(ext)]
        xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
        cu = (xcc.astype(np.int64) << 16) | ((hwid >> 8) & 0xff).astype(np.int64)  # CU_ID[11:8] SH_ID[12] SE_ID[15:13]
        dur_all = (ext[:, 1].astype(np.int64) - mn[:, 0].astype(np.int64)) / 100.0
        import collections
        per = collections.defaultdict(list)
        for c, d in zip(cu, dur_all):
            per[int(c)].append(float(d))
        hist = collections.Counter(len(v) for v in per.values())
        print("  CUs by number of workgroups they ran:", dict(hist), "; distinct CUs", len(per))
        for k in sorted(hist):
            ds = [d for v in per.values() if len(v) == k for d in v]
            print("    CUs with %d workgroup(s): duration of a workgroup median %.1f min %.1f max %.1f" % (k, np.median(ds), min(ds), max(ds)))
        for xc in range(8):
            ds = dur_all[xcc == xc]
            if len(ds):
                print("    XCC %d: %d workgroups, median %.1f max %.1f" % (xc, len(ds), np.median(ds), ds.max()))
        t_end = t_last
        us = lambda t: (t.astype(np.int64) - np.int64(t_end)) / 100.0
    print("fix-up: %d waves; queued points per wave: mean %.1f max %d" % (len(fix), fix[:, 7].mean(), fix[:, 7].max()))
    for k, name in [(0, "wave start"), (1, "tables + counts in LDS"), (2, "prefix done")]:
        t = us(fix[:, k])
        print("  %-26s min %7.2f  median %7.2f  max %7.2f" % (name, t.min(), np.median(t), t.max()))
    for p in range(4):
        sel = fix[:, 3 + p] > 0
        if sel.any():
            t = us(fix[sel, 3 + p])
            print("  batch %d done (%4d waves)   min %7.2f  median %7.2f  max %7.2f" % (p + 1, sel.sum(), t.min(), np.median(t), t.max()))
    d1 = (fix[:, 3].astype(np.int64) - fix[:, 2].astype(np.int64)) / 100.0
    print("  first batch duration: min %.2f median %.2f max %.2f us" % (d1[fix[:, 3] > 0].min(), np.median(d1[fix[:, 3] > 0]), d1[fix[:, 3] > 0].max()))


if __name__ == "__main__":
    main()
