#!/usr/bin/env python3
"""Fused reach + distance against distance-only launches (with / without validity bytes) in every arithmetic mode, config-2 sized random cloud: HIP-event ms per call."""
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np, lrm_amd as lrm
n = 10_000_000
leg = lrm.get_M2_leg(0.0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
lo, hi = (np.array([-300., -700, -700]), np.array([900., 700, 500]))
c = torch.rand((3, n), device="cuda", generator=g) * torch.tensor(hi - lo, device="cuda", dtype=torch.float32).view(3, 1) + torch.tensor(lo, device="cuda", dtype=torch.float32).view(3, 1)
x, y, z = c[0].contiguous(), c[1].contiguous(), c[2].contiguous()
field = torch.empty((3, n), device="cuda"); mask = torch.empty(n, dtype=torch.uint8, device="cuda"); valid = torch.empty(n, dtype=torch.uint8, device="cuda")
bits = torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")
def t(f, reps=500):
    for _ in range(100): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
only = sys.argv[1] if len(sys.argv) > 1 else None
if only in ("maskbits", "maskonly"):
    lrm.set_mode(lrm.MODE_TOL_REL)
    print(only, t((lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field, bits=bits)) if only == "maskbits" else (lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field))))
    sys.exit(0)
if only == "bits":
    for mode in ("MODE_TOL_REL", "MODE_TOL"):
        lrm.set_mode(getattr(lrm, mode))
        for rep in range(3):
            print(mode, "mask+bits %.5f" % t(lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field, bits=bits)),
                  "mask only %.5f" % t(lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field)),
                  "bits only %.5f" % t(lambda: lrm.device.reach_dist(x, y, z, leg, None, out=field, bits=bits)), flush=True)
    sys.exit(0)
if only:
    lrm.set_mode(lrm.MODE_TOL_REL)
    print(only, t({"novalid": lambda: lrm.device.dist(x, y, z, leg, out=field, want_valid=False), "valid": lambda: lrm.device.dist(x, y, z, leg, out=field, valid=valid)}[only]))
    sys.exit(0)
for mode in ("MODE_TOL_REL", "MODE_TOL", "MODE_FAST"):
    lrm.set_mode(getattr(lrm, mode))
    print(mode, "fused(mask+bits)", t(lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field, bits=bits)),
          "dist(valid)", t(lambda: lrm.device.dist(x, y, z, leg, out=field, valid=valid)), "dist(no valid)", t(lambda: lrm.device.dist(x, y, z, leg, out=field, want_valid=False)), "fused(mask only)", t(lambda: lrm.device.reach_dist(x, y, z, leg, None, mask=mask, out=field)), flush=True)
