#!/usr/bin/env python3
"""bench.py -- headline benchmark of the reach+distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works from this plain command (the parent starts N ranks under torch.distributed.run and stays off the GPU) and
from `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / WORLD_SIZE in the environment).

A "step" is ONE pass of the fused reach+distance launch (lrm_reach_dist_bits_dev: reach mask as bytes and ballot
bit words + 3-component distance field) over a synthetic cloud that is already resident in HBM as SoA float32.

  N = 1   BASELINE.json config 2: 1e7 uniform-random targets in [-200,700]x[-500,500]x[-500,300] mm (seed 42),
          M2 leg, identity orientation.
  N > 1   BASELINE.json config 4 (north star): ONE 1e8-point cloud (same distribution, seeded per 1e6-point chunk so
          that a rank generates only its own shard), rank r owns lrm_amd.shard.shard_bounds(1e8, N, r); a step =
          the fused launch on the shard + the RCCL all-gather of the bit-packed reach mask (side stream,
          overlapping the next step's kernel).  Strong scaling: the total work is fixed.

The headline runs in LRM_MODE_TOL_REL, the literal text of BASELINE.json: reach mask bit-exact and |d - d_ref| <= 1e-5 |d_ref|
for EVERY distance vector (vectors shorter than max(19 mm, 2250 decision bands) come from the table-guided bit-exact chain of
csrc/lrm_point_xtab.h, all others from the tolerance arithmetic).  The line carries the error statistics of the very output
it timed ("tolerance_check").  Timed next to it and reported under "modes" with their own roofline fractions:
LRM_MODE_FAST (every float of the distance field identical to the reference's host path: tolerance 0; since round 4 the
table-guided kernel), the filtered kernel it replaced (LRM_XTAB=0), and LRM_MODE_TOL (a FLOORED reading of the tolerance:
faster, and NOT the contract).
value = leg-target evaluations per second over the whole job, one evaluation = reachability AND distance vector
of one (leg, target) pair.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_EVAL = {"reach": 13, "dist": 24, "reach_dist": 25}
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 VALU op issues over 2 cycles
LO = np.array([-200, -500, -500], np.float32)
HI = np.array([700, 500, 300], np.float32)
CHUNK = 1_000_000


def committed_profile(points, mode, profiles_dir=None):
    """Figures that cannot be read inside the timed run (PMC counters need their own rocprofv3 passes): the latest
    committed profiles/r*_hbm_traffic.json / r*_valu.json recorded for this workload and mode -- used only when the
    kernel sources they were taken with (kernel_src_sha, lrm_amd/srchash.py) are the tree's; otherwise null + "stale"."""
    import glob
    from lrm_amd.srchash import kernel_src_sha
    sha = kernel_src_sha()
    out = {"traffic": None, "traffic_source": None, "valu_insts_per_eval": None, "valu_source": None, "kernel_src_sha": sha}
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    for path in sorted(glob.glob(os.path.join(profiles_dir, "r*_hbm_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("points_per_launch") == points and rec.get("mode") == mode and "step_hbm_bytes" in rec:
            if rec.get("kernel_src_sha") == sha:
                out["traffic"], out["traffic_source"] = rec["step_hbm_bytes"], os.path.basename(path)
            else:
                out["traffic_source"] = f"stale: {os.path.basename(path)} was taken with other kernel sources"
            break
    for path in sorted(glob.glob(os.path.join(profiles_dir, "r*_valu.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("mode") == mode and "valu_insts_per_eval" in rec:
            if rec.get("kernel_src_sha") == sha:
                out["valu_insts_per_eval"], out["valu_source"] = rec["valu_insts_per_eval"], os.path.basename(path)
            else:
                out["valu_source"] = f"stale: {os.path.basename(path)} was taken with other kernel sources"
            break
    return out


def make_cloud(n, seed):
    """config 2: one generator, seed 42 (the cloud the tests and the committed profiles use)"""
    rng = np.random.default_rng(seed)
    out = np.empty((3, n), np.float32)
    for s in range(0, n, 2 * CHUNK):
        e = min(n, s + 2 * CHUNK)
        out[:, s:e] = (rng.random((e - s, 3), dtype=np.float32) * (HI - LO) + LO).T
    return out


def make_shard(lo, hi, seed=42):
    """points [lo, hi) of the chunk-seeded cloud: chunk c = default_rng([seed, c]).random((CHUNK, 3))"""
    out = np.empty((3, hi - lo), np.float32)
    c = lo // CHUNK
    while c * CHUNK < hi:
        pts = np.random.default_rng([seed, c]).random((CHUNK, 3), dtype=np.float32) * (HI - LO) + LO
        a, b = max(lo, c * CHUNK), min(hi, (c + 1) * CHUNK)
        out[:, a - lo:b - lo] = pts[a - c * CHUNK:b - c * CHUNK].T
        c += 1
    return out


def cpu_baseline(sample_points, leg):
    """CPU figures on this box's cores, same cloud, bounded samples.
    Primary: the product's own CPU entry points lrm_reach_cpu + lrm_dist_cpu (the apply_reach_cpu / apply_dist_cpu
    drop-ins, cross_compiled.cu:163-181; bit-identical to the reference's host path), one thread as the reference
    runs them and all cores with a static split.  Beside it: the RBDL-equivalent LM position IK (apply_RBDL's work,
    parity unpinned).  (The reference's own host build, oracle/_ref/libref.so, no longer travels to the GPU box: its single-thread
    rate in the build container is recorded in DESIGN.md section 5.)"""
    from concurrent.futures import ThreadPoolExecutor
    import lrm_amd
    pts = np.ascontiguousarray(sample_points.T)  # AoS, as the reference's Array<float3>
    n = len(pts)

    def both(sl):
        lrm_amd.apply_reach_cpu(sl, leg)
        lrm_amd.apply_dist_cpu(sl, leg)

    n1 = min(n, 4_000_000)
    t0 = time.perf_counter()
    both(pts[:n1])
    single = n1 / (time.perf_counter() - t0)
    cores = min(os.cpu_count() or 1, 16)
    parts = [pts[i[0]:i[-1] + 1] for i in np.array_split(np.arange(n), cores)]
    dts = []
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL during the C loops
        for _ in range(3):                 # three passes over the sample (~10 CPU-seconds), median
            t0 = time.perf_counter()
            list(ex.map(both, parts))
            dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    out = {"value": n / dt, "unit": "leg-target evaluations/s (reach+dist)", "cores": cores, "kind": "port",
           "impl": "liblrm.so lrm_reach_cpu + lrm_dist_cpu (the apply_reach_cpu / apply_dist_cpu drop-ins; bit-identical "
                   "to the reference's host path; NOT the RBDL baseline, which is rbdl_equivalent below)",
           "sample": f"first {n} points of the same cloud, reach loop + distance loop, {cores} threads (static split), median of 3 passes; "
                     f"single thread on {n1} points: {single:.3e}/s",
           "single_thread_value": single}
    # apply_RBDL's work (rbdl_benchmark.cpp:18-111), 3 repeats as setting_bench.h:7, on <= 1e5 points
    nr = min(n, 100_000)
    times = []
    for _ in range(3):
        _, ms = lrm_amd.apply_rbdl_equiv(pts[:nr], leg)
        times.append(ms)
    ms = float(np.median(times))
    out["rbdl_equivalent"] = {
        "value": nr / (ms * 1e-3), "unit": "targets/s (position IK solved or given up)", "ns_per_point": ms * 1e6 / nr,
        "cores": 1, "sample": f"first {nr} points, median of 3 repeats",
        "note": "RBDL-equivalent Levenberg-Marquardt position IK (same chain incl. /400, max_steps 10, <= 5 starts) with "
                "closed-form kinematics; RBDL itself is an external unpinned dependency that is absent: parity unpinned, "
                "timing baseline only; published RBDL figure: 14 610 ns/point on an i5-12600K (bdata/pc/rbdl.csv)"}
    return out


def config5_block(torch, lrm_amd, n=12_500_000, depth=6):
    """BASELINE config 5 at ONE GPU's share outside the timed headline: the octree-culled positionability (lrm_apply_oct_dev) on
    1.25e7 synthetic footholds of a 10 m x 10 m relief (the N = 8 share of the 1e8-point cloud), the reference's settings (root box
    +-5000 mm, min box 100 mm, 27 orientations below 50 mm), 4 legs, stability 3, depth 6.  kernel_ms = the level kernels of the second
    call (the first one also builds and caches the 108 plane tables)."""
    rng = np.random.default_rng(5)
    xy = rng.uniform(-5000, 5000, (n, 2)).astype(np.float32)
    z = (400 * np.sin(xy[:, 0] / 900) * np.cos(xy[:, 1] / 700) + 60 * np.sin(xy[:, 0] / 130) + rng.normal(0, 5, n) - 200).astype(np.float32)
    t = torch.from_numpy(np.ascontiguousarray(np.column_stack([xy, z]).astype(np.float32).T)).cuda()
    del xy, z
    dim = lrm_amd.get_M2_leg(0.0)
    st = lrm_amd.octree_default_settings()
    st.max_depth = depth
    st.leg_number_for_stab = 3
    t0 = time.perf_counter()
    leaves, ms_first = lrm_amd.device.apply_oct(t[0], t[1], t[2], dim, st)
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    leaves, ms = lrm_amd.device.apply_oct(t[0], t[1], t[2], dim, st)
    wall = time.perf_counter() - t0
    return {"workload": f"apply_oct, {n} device-resident footholds on a 10 m x 10 m relief, depth {depth}, 4 legs, stability 3 (one GPU's share of BASELINE config 5)",
            "data": "synthetic", "kernel_ms": ms, "wall_ms": wall * 1e3, "first_call_wall_ms": first * 1e3, "leaves": int(len(leaves)),
            "parity": "pinned by composition only (tests/test_gpu_octree.py against tests/octree_oracle.py): the reference's call site is dead code"}


def config3_block(torch, lrm_amd):
    """BASELINE config 3 outside the timed headline: 6-leg positionability, one launch of lrm_reach_any_dev on the
    reference's own terrain and near-ground body lattice (tests/golden/terrain_ground.npz) in Morton order."""
    from lrm_amd import workloads
    path = os.path.join(ROOT, "tests", "golden", "terrain_ground.npz")
    if os.path.exists(path):
        t = np.load(path)
        ground, bodies, src = t["ground"], t["bodies"], "reference maps.py output (tests/golden/terrain_ground.npz)"
    else:
        ground = workloads.terrain(256)
        bodies = workloads.body_lattice(ground, 100_000)
        src = "lrm_amd.workloads.terrain (own generator)"
    ground = ground[lrm_amd.morton_order(ground)]
    bodies = bodies[lrm_amd.morton_order(bodies)]
    legs = workloads.hexapod(lrm_amd.get_M2_leg, 6)
    tb = torch.from_numpy(np.ascontiguousarray(bodies.T)).cuda()
    tt = torch.from_numpy(np.ascontiguousarray(ground.T)).cuda()
    out = torch.empty((6, len(bodies)), dtype=torch.uint8, device="cuda")
    alll = torch.empty(len(bodies), dtype=torch.uint8, device="cuda")
    run = lambda: lrm_amd.device.reach_any(tb[0], tb[1], tb[2], tt[0], tt[1], tt[2], legs, None, out=out, all_legs=alll)
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    block = {"workload": f"{len(bodies)} body poses x {len(ground)} terrain points x 6 M2 legs, identity orientation, Morton order"
                         + (" (the reference's real terrain: 59 % of the nominal 1e5 x 1e5 pair count)" if os.path.exists(path) else ""),
             "data": src, "ms": ms, "pairs_per_s": float(len(bodies)) * len(ground) * 6 / (ms * 1e-3),
             "pairs_per_s_note": "pairs ANSWERED per second: most are decided by bounding boxes and spheres, see pairs_evaluated",
             "positionable_fraction": float(alll.float().mean().item())}
    # how many pairs one launch really evaluates, and the kernel's VALU figures: a counting build and PMC passes of the same
    # launch (tools/c3_profile.sh), used only when taken with the tree's kernel sources
    import glob
    from lrm_amd.srchash import kernel_src_sha
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_evidence.json")), reverse=True):
        try:
            ev = json.load(open(pth))
        except (OSError, ValueError):
            continue
        if ev.get("kernel_src_sha") != kernel_src_sha() or ev.get("pairs_answered") != int(len(bodies)) * int(len(ground)) * 6:
            block["evidence_source"] = f"stale: {os.path.basename(pth)} was taken with other kernel sources"
            break
        lane_ops = ev["pmc_per_launch"]["SQ_INSTS_VALU"] * 64.0
        block.update({"pairs_evaluated": ev["pairs_evaluated"], "evals_per_s": ev["pairs_evaluated"] / (ms * 1e-3),
                      "valu_insts_per_eval": ev["valu_insts_per_eval"], "frac_valu": lane_ops / (ms * 1e-3) / 78.6e12,
                      "frac_valu_peak": "78.6e12 FP32 lane-ops/s (256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz)",
                      "evidence_source": os.path.basename(pth), "kernel": ev["kernel"]})
        break
    return block


def cache_and_scaling_block(torch, lrm_amd, leg, cloud, mask, field, time_kernel):
    """What the headline figure owes to the 256 MiB Infinity Cache, and the two ends of the curve `--gpus N` draws -- all in
    the headline's mode, on one GPU, outside the timed region:
      copy_soa            the box's achieved copy bandwidth: a (3, n) float32 tensor copied device to device (read + write)
      fused_rotating      the config-2 step over FOUR distinct 1e7-point clouds round-robin, each with its own outputs: 1 GB is
                          touched between two uses of a buffer, so nothing a step reads or writes is still in the cache
      fused_shard_1.25e7  one GPU's share of the 1e8-point cloud at N = 8
      fused_1e8_one_gpu   the whole 1e8-point cloud on one GPU (the base of the strong-scaling curve)"""
    n = cloud.shape[1]
    out = {}
    dst = torch.empty_like(cloud)
    ms = time_kernel(lambda: dst.copy_(cloud), reps=200)
    copy_gbs = 2 * cloud.numel() * 4 / (ms * 1e-3) / 1e9
    out["copy_soa"] = {"ms": ms, "GBs": copy_gbs, "bytes": 2 * cloud.numel() * 4,
                       "what": f"torch device-to-device copy of the (3, {n}) float32 cloud, read + write bytes over HIP-event time"}
    del dst
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    lo_t, span = torch.tensor(LO, device="cuda").view(3, 1), torch.tensor(HI - LO, device="cuda").view(3, 1)

    def rand_cloud(m):
        return (torch.rand((3, m), device="cuda", generator=g) * span + lo_t).contiguous()

    def fused(c, m_, f_, w_):
        lrm_amd.device.reach_dist(c[0], c[1], c[2], leg, None, mask=m_, out=f_, bits=w_)

    sets = []
    for _ in range(4):
        sets.append((rand_cloud(n), torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty((3, n), dtype=torch.float32, device="cuda"),
                     torch.empty((n + 63) // 64, dtype=torch.int64, device="cuda")))
    state = {"k": 0}

    def rot():
        c, m_, f_, w_ = sets[state["k"] & 3]
        state["k"] += 1
        fused(c, m_, f_, w_)

    ms = time_kernel(rot, reps=200)
    out["fused_rotating"] = {"kernel_ms": ms, "roofline_frac": BYTES_PER_EVAL["reach_dist"] * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "ps_per_point": ms * 1e9 / n,
                             "what": "four distinct clouds of the config-2 size with their own outputs, round-robin: 1 GB touched between two uses of any buffer"}
    del sets
    for key, m in (("fused_shard_1.25e7", 12_500_000), ("fused_1e8_one_gpu", 100_000_000)):
        c = rand_cloud(m)
        m_, f_ = torch.empty(m, dtype=torch.uint8, device="cuda"), torch.empty((3, m), dtype=torch.float32, device="cuda")
        w_ = torch.empty((m + 63) // 64, dtype=torch.int64, device="cuda")
        ms = time_kernel(lambda: fused(c, m_, f_, w_), reps=50 if m > 20_000_000 else 200)
        out[key] = {"points": m, "kernel_ms": ms, "ps_per_point": ms * 1e9 / m,
                    "roofline_frac": BYTES_PER_EVAL["reach_dist"] * m / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del c, m_, f_, w_
    torch.cuda.empty_cache()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU needs ~50 ms of sustained load to reach its steady clocks
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--points", type=int, default=None, help="N = 1: targets of the config-2 cloud (default 1e7)")
    ap.add_argument("--total-points", type=int, default=100_000_000, help="N > 1: points of the one sharded cloud")
    ap.add_argument("--mode", choices=["tol", "tol_rel", "fast", "strict"], default="tol_rel", help="arithmetic mode of the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed secondary figures (other mode, reach/dist only, brackets, config 3)")
    ap.add_argument("--no-tolerance-check", action="store_true", help="skip the untimed comparison of the timed output with lrm_dist_cpu")
    ap.add_argument("--precondition-ms", type=float, default=100.0,
                    help="untimed GPU load before the W warm-up steps so that short runs are also measured at the "
                         "steady clocks (the first ~50 ms after idle run ~15 %% slower); 0 disables it")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: this process becomes the launcher.  It must
    never touch the GPU (a process that has initialised HIP may not start GPU children by exec on this pool, and a
    parent holding the device would be a ninth process on an 8-GPU node): no torch import, no lrm_amd.load() here.
    The N ranks run as children of `python -m torch.distributed.run`; rank 0's single JSON line is relayed on stdout,
    everything else the children print goes to stderr, and the exit code is theirs."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in proc.stdout.splitlines():
        try:
            rec = json.loads(ln) if ln.startswith("{") else None
        except ValueError:
            rec = None
        if isinstance(rec, dict) and "metric" in rec:
            lines.append((ln, rec))
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed with exit code {proc.returncode}")
    if len(lines) != 1:
        raise SystemExit(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}")
    ln, rec = lines[0]
    if rec.get("n_gpus") != args.gpus or rec.get("rccl_ranks") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the collective saw {rec.get('rccl_ranks')} (n_gpus {rec.get('n_gpus')})")
    print(ln, flush=True)


def count_ranks(torch, dist, device):
    """the number of ranks the collective backend really connects: an all-reduce of ones"""
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def dry_run(args, world, rank):
    """LRM_BENCH_DRYRUN=1: rehearsal of the multi-rank CONTROL FLOW where there is no GPU (the CPU test of the
    self-launch path): process group (gloo), shard bounds, BitsGatherLoop's step loop with its all-gather, barriers,
    max-over-ranks timing, one JSON line from rank 0.  Nothing is evaluated -- the local "computation" writes a word
    pattern -- so the line carries "dry_run": true and value null: it is not a measurement and not a CPU fallback."""
    import torch
    import torch.distributed as dist
    from lrm_amd import shard
    if world > 1:
        dist.init_process_group("gloo")
    ranks = count_ranks(torch, dist, "cpu") if world > 1 else 1
    if ranks != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks} ranks")
    n_total = args.total_points if world > 1 else (args.points or 10_000_000)
    loop = shard.BitsGatherLoop(n_total, device="cpu", time_gather=True)

    def compute(words, lo, hi):
        words.copy_(torch.arange(lo // 64, lo // 64 + words.numel(), dtype=torch.int64))

    for k in range(args.warmup):
        loop.step(k, compute)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        loop.step(args.warmup + k, compute)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last = loop.result((args.warmup + args.steps - 1) & 1)
    ok = bool((last == torch.arange(last.numel(), dtype=torch.int64)).all())  # every rank holds every shard's words
    if rank == 0:
        print(json.dumps({"metric": "leg-target evaluations/sec (reach+dist)", "value": None, "dry_run": True,
                          "unit": "evaluations/s", "n_gpus": world, "rccl_ranks": ranks, "backend": "gloo",
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                          "gather_ms": loop.gather_ms(), "gathered_words_ok": ok,
                          "config": {"points_total": n_total, "points_per_gpu": loop.hi - loop.lo}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: the gathered words are not the shards' words")


_REFERENCE = {}  # n -> (mask, field) of the CPU entry points: computed once per run


def tolerance_check(lrm_amd, host, leg, mask_dev, field_dev, limit=10_000_000):
    """Error statistics of the output the timed run left behind (untimed), against the product's own bit-exact CPU
    entry points lrm_reach_cpu / lrm_dist_cpu (apply_reach_cpu / apply_dist_cpu drop-ins; NOT the oracle), all cores.
    Reports the frozen metric of include/lrm.h next to the literal relative error |d - d_ref| / |d_ref|."""
    from concurrent.futures import ThreadPoolExecutor
    n = min(host.shape[1], limit)
    pts = np.ascontiguousarray(host[:, :n].T)
    cores = min(os.cpu_count() or 1, 16)
    parts = np.array_split(np.arange(n), cores)
    if n in _REFERENCE:
        ref_m, ref_d = _REFERENCE[n]
    else:
        ref_m, ref_d = np.empty(n, np.uint8), np.empty((n, 3), np.float32)

        def one(idx):
            a, b = int(idx[0]), int(idx[-1]) + 1
            ref_m[a:b] = lrm_amd.apply_reach_cpu(pts[a:b], leg)[0]
            ref_d[a:b] = lrm_amd.apply_dist_cpu(pts[a:b], leg)[0]

        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, [p for p in parts if len(p)]))
        _REFERENCE[n] = (ref_m, ref_d)
    got_m = mask_dev[:n].cpu().numpy()
    got_d = field_dev[:, :n].cpu().numpy().T
    stats = {"points": n, "reference": "liblrm.so lrm_reach_cpu + lrm_dist_cpu (bit-identical to the reference's host path)",
             "mask_mismatches": int((got_m != ref_m).sum()), "bit_identical_vectors": 0,
             "max_metric": 0.0, "max_abs_mm": 0.0, "max_plain_rel": 0.0, "frac_plain_rel_gt_1e-5": 0.0,
             "max_plain_rel_dref_ge_16mm": 0.0,
             "metric": "|d - d_ref| / max(|d_ref|, (|p| + body) / 8) <= 1e-5 (include/lrm.h, frozen); plain_rel = |d - d_ref| / |d_ref| "
                       "over the points with |d_ref| > 1e-2 mm"}
    body = abs(float(np.asarray(leg, np.float64).reshape(-1)[1]))
    ident = 0
    for a in range(0, n, 1_000_000):  # float64 in slices: bounded host memory
        b = min(n, a + 1_000_000)
        d, r, p = got_d[a:b].astype(np.float64), ref_d[a:b].astype(np.float64), pts[a:b].astype(np.float64)
        ident += int((got_d[a:b].view(np.uint32) == ref_d[a:b].view(np.uint32)).all(axis=1).sum())
        err = np.linalg.norm(d - r, axis=1)
        nref = np.linalg.norm(r, axis=1)
        metric = err / np.maximum(nref, (np.linalg.norm(p, axis=1) + body) / 8.0)
        metric = np.where(np.isfinite(metric), metric, np.inf)
        sel = nref > 1.0e-2
        plain = err[sel] / nref[sel]
        stats["max_metric"] = max(stats["max_metric"], float(metric.max(initial=0.0)))
        stats["max_abs_mm"] = max(stats["max_abs_mm"], float(np.nan_to_num(err, nan=np.inf).max(initial=0.0)))
        stats["max_plain_rel"] = max(stats["max_plain_rel"], float(plain.max(initial=0.0)))
        stats["frac_plain_rel_gt_1e-5"] += float((plain > 1.0e-5).sum())
        big = nref >= 16.0
        stats["max_plain_rel_dref_ge_16mm"] = max(stats["max_plain_rel_dref_ge_16mm"], float((err[big] / nref[big]).max(initial=0.0)))
    stats["frac_plain_rel_gt_1e-5"] /= max(n, 1)
    stats["bit_identical_vectors"] = ident
    return stats


def bracket_clouds(torch, lrm_amd, n, leg):
    """the two ends of the lane divergence: only points beyond the workspace / only reachable points (resampled cubes)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    lo_t, span = torch.tensor(LO, device="cuda").view(3, 1), torch.tensor(HI - LO, device="cuda").view(3, 1)
    far = torch.rand((3, n), device="cuda", generator=g) * span + lo_t
    far[0] += 900.0
    parts, have = [], 0
    prev = lrm_amd.get_mode()
    lrm_amd.set_mode(lrm_amd.MODE_FAST)
    try:
        while have < n:
            c = (torch.rand((3, n), device="cuda", generator=g) * span + lo_t).contiguous()
            m = lrm_amd.device.reach(c[0], c[1], c[2], leg)
            keep = c[:, m.bool()]
            parts.append(keep)
            have += keep.shape[1]
            if keep.shape[1] == 0:
                break
    finally:
        lrm_amd.set_mode(prev)
    near = torch.cat(parts, dim=1)[:, :n].contiguous()
    return {"all_unreachable": far.contiguous(), "all_reachable": near}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)  # before anything that could initialise the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("LRM_BENCH_DRYRUN") == "1":
        return dry_run(args, world, rank)

    import torch
    import lrm_amd
    from lrm_amd import shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count()  # rehearsals with more ranks than GPUs share devices
    torch.cuda.set_device(device_index)
    dist = None
    # LRM_BENCH_BACKEND=gloo rehearses the multi-rank control flow where RCCL cannot run (several ranks on one
    # GPU): the bit words then travel through host memory
    backend = os.environ.get("LRM_BENCH_BACKEND", "nccl")
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
        ranks_seen = count_ranks(torch, dist, "cuda" if backend == "nccl" else "cpu")
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the {backend} process group connects {ranks_seen} ranks")
    modes = {"tol": lrm_amd.MODE_TOL, "tol_rel": lrm_amd.MODE_TOL_REL, "fast": lrm_amd.MODE_FAST, "strict": lrm_amd.MODE_STRICT}
    lrm_amd.set_mode(modes[args.mode])

    leg = lrm_amd.get_M2_leg(0.0)
    if world == 1:
        n_total = args.points or 10_000_000
        loop = shard.BitsGatherLoop(n_total, device="cuda")
        host = make_cloud(n_total, seed=42)
    else:
        n_total = args.total_points
        loop = shard.BitsGatherLoop(n_total, device="cuda", host_staging=(backend != "nccl"), time_gather=True)
        host = make_shard(loop.lo, loop.hi)
    n = loop.hi - loop.lo  # this rank's points
    cloud = torch.from_numpy(host).cuda()
    x, y, z = cloud[0], cloud[1], cloud[2]
    mask = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
    field = torch.empty((3, max(n, 1)), dtype=torch.float32, device="cuda")

    def compute(words, lo, hi):
        lrm_amd.device.reach_dist(x, y, z, leg, None, mask=mask[:n], out=field[:, :n], bits=words)

    # the plane table of this (leg, orientation) is built by the first call: timed here, cold, outside the steps
    lrm_amd.release_workspaces()
    torch.cuda.synchronize()
    t_tab = time.perf_counter()
    lrm_amd.tol_prepare(leg, None, n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    table_build_ms = {"first_call_prepare_ms": (time.perf_counter() - t_tab) * 1e3, "table_ms": lrm_amd.last_table_build_ms()}

    def full_sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import gc
    gc.collect()
    gc.disable()  # a generation-2 collection inside the timed loop costs tens of ms of launch-queue starvation

    def timed_run(steps, warmup, precondition_ms):
        if precondition_ms > 0:  # clock conditioning: the same launch, results discarded, before the warm-up
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < precondition_ms:
                for _ in range(20):
                    compute(loop.words[0][:loop.local_words], loop.lo, loop.hi)
                torch.cuda.synchronize()
        for k in range(warmup):
            loop.step(k, compute)
        full_sync()
        loop.reset_gather_timing()
        # HIP events on the launch stream (torch's current stream is the one the C ABI launches on).  One GPU: ONE pair
        # around the K back-to-back steps -- an event between two steps costs a few microseconds of dispatch pipeline
        # per step (0.124 ms per step with per-step pairs against 0.116 without) and the steps contain nothing but
        # the launches.  Several GPUs: a pair around the launches of every step, so that the gather stays outside.
        per_step = world > 1
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps if per_step else 1)]

        cur = {}

        def compute_timed(words, lo, hi):
            cur["ev"][0].record()
            compute(words, lo, hi)
            cur["ev"][1].record()

        t0 = time.perf_counter()
        if per_step:
            for k in range(steps):
                cur["ev"] = ev[k]
                loop.step(warmup + k, compute_timed)
        else:
            ev[0][0].record()
            for k in range(steps):
                loop.step(warmup + k, compute)
            ev[0][1].record()
        full_sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            elapsed = reduce_max(elapsed)
        if per_step:
            return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return elapsed, ev[0][0].elapsed_time(ev[0][1]) / max(steps, 1)

    elapsed, kernel_ms = timed_run(args.steps, args.warmup, args.precondition_ms)
    gather_ms = loop.gather_ms()
    gc.enable()

    # the gathered words of the last step: every rank holds the mask of the whole cloud
    last = loop.result((args.warmup + args.steps - 1) & 1)
    reachable_fraction = float(sum(bin(int(w) & (2**64 - 1)).count("1") for w in last[:4096].cpu().tolist()) / (64 * min(4096, last.numel())))

    # error statistics of the output the timed steps left in `mask` / `field` (before anything overwrites them)
    tol_check = None
    if rank == 0 and world == 1 and not args.no_tolerance_check:
        tol_check = tolerance_check(lrm_amd, host, leg, mask, field)
        tol_check["mode"] = args.mode

    def time_kernel(fn, reps=100):
        # warm-up by TIME: the checks between the timed sections leave the GPU idle for seconds, and it needs some 50 ms of load to be
        # back at its steady clocks (20 launches of a 0.1 ms step were not: the side modes read 6 % slow against tools/bench_modes.py)
        t_warm = time.perf_counter()
        while True:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            if time.perf_counter() - t_warm >= 0.08:
                break
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def frac_of(ms, what="reach_dist"):
        return BYTES_PER_EVAL[what] * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    extra, other_modes, configs = {}, {}, {}
    if rank == 0 and not args.no_extras:
        words = loop.words[0][:loop.local_words]
        contracts = {
            "fast": "meets north_star at tolerance 0: reach mask AND every float of the distance field bit-identical to the reference's host "
                    "path (since round 4 the table-guided kernel dist_xtab_kernel + the filtered fix-up of its doubtful points)",
            "fast_filtered": "as fast, by the filtered kernel of rounds 1-3 (LRM_XTAB=0; still what small clouds and legs without a table run)",
            "tol_rel": "meets north_star literally: reach mask bit-identical, |d - d_ref| <= 1e-5 |d_ref| for every vector",
            "tol": "NOT the contract: reach mask bit-identical, distance within 1e-5 of max(|d|, (|p| + body)/8) -- a floored reading",
        }
        for name in ("tol_rel", "fast", "fast_filtered", "tol"):  # the other arithmetic modes, same launch, untimed region
            if name == args.mode:
                continue
            lrm_amd.set_mode(modes["fast" if name == "fast_filtered" else name])
            if name == "fast_filtered":
                os.environ["LRM_XTAB"] = "0"
            try:
                ms = time_kernel(lambda: compute(words, loop.lo, loop.hi), reps=200)
                other_modes[name] = {"kernel_ms": ms, "evals_per_s_per_gpu": n / (ms * 1e-3), "roofline_frac": frac_of(ms), "contract": contracts[name]}
                if world == 1 and not args.no_tolerance_check:  # the output this mode just left behind
                    c = tolerance_check(lrm_amd, host, leg, mask, field)
                    other_modes[name]["tolerance_check"] = {k: c[k] for k in ("points", "mask_mismatches", "bit_identical_vectors", "max_plain_rel",
                                                                                "frac_plain_rel_gt_1e-5", "max_abs_mm")}
                    try:
                        npts, nq, nover = lrm_amd.dbg_tol_queue_counts()
                        other_modes[name]["queued_fraction"] = nq / max(npts, 1)
                    except lrm_amd.LrmError:
                        pass
            finally:
                os.environ.pop("LRM_XTAB", None)
        lrm_amd.set_mode(modes[args.mode])
        ms_reach = time_kernel(lambda: lrm_amd.device.reach(x, y, z, leg, out=mask[:n], bits=words))
        valid = torch.empty(n, dtype=torch.uint8, device="cuda")
        ms_dist = time_kernel(lambda: lrm_amd.device.dist(x, y, z, leg, out=field[:, :n], valid=valid))
        extra = {
            "reach_only": {"evals_per_s": n / (ms_reach * 1e-3), "ms": ms_reach,
                           "hbm_GBs": BYTES_PER_EVAL["reach"] * n / (ms_reach * 1e-3) / 1e9,
                           "roofline_frac": frac_of(ms_reach, "reach")},
            "dist_only": {"evals_per_s": n / (ms_dist * 1e-3), "ms": ms_dist,
                          "hbm_GBs": BYTES_PER_EVAL["dist"] * n / (ms_dist * 1e-3) / 1e9,
                          "roofline_frac": frac_of(ms_dist, "dist")},
        }
        if world == 1:
            # divergence brackets: the config-2 cube is mostly unreachable points; the same
            # launch on a cloud of only unreachable / only reachable points
            del valid
            for bname, c in bracket_clouds(torch, lrm_amd, n, leg).items():
                row = {}
                for name in ("tol_rel", "fast"):
                    lrm_amd.set_mode(modes[name])
                    ms = time_kernel(lambda: lrm_amd.device.reach_dist(c[0], c[1], c[2], leg, None, mask=mask[:n], out=field[:, :n], bits=words), reps=200)
                    row[name] = {"kernel_ms": ms, "roofline_frac": frac_of(ms)}
                extra["fused_" + bname] = row
                del c
            lrm_amd.set_mode(modes[args.mode])
            extra.update(cache_and_scaling_block(torch, lrm_amd, leg, cloud, mask, field, time_kernel))
            configs["c3_positionability"] = config3_block(torch, lrm_amd)
            configs["c5_octree_share"] = config5_block(torch, lrm_amd)
    if world > 1:
        dist.barrier()

    if rank == 0:
        total_evals = float(n_total) * args.steps
        achieved = BYTES_PER_EVAL["reach_dist"] * n / (kernel_ms * 1e-3) / 1e9
        kname = {"tol": "dist_tab_kernel<2, false, false> + tol_fixup_kernel<2, false, 8, 128> (one step = both launches)",
                 "tol_rel": "dist_tab_kernel<2, false, true> (tolerance evaluation + strict replay of its short vectors) + tol_fixup_kernel<2, false, 8, 128> (one step = both launches)",
                 "fast": "dist_xtab_kernel<2, false> + tol_fixup_kernel<2, false, 8, 128> (one step = both launches)",
                 "strict": "dist_soa_kernel<2, false>"}[args.mode]
        prof = committed_profile(n, args.mode)
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": prof["traffic"], "traffic_source": prof["traffic_source"],
            "kernel": kname + " (fused reach+distance)", "kernel_ms": kernel_ms,
            "kernel_ms_method": ("HIP events on the launch stream: one pair around the K back-to-back steps, divided by K"
                                 if world == 1 else "HIP events on the launch stream: mean of one pair per step around its launches"),
            "algorithmic_bytes_per_eval": BYTES_PER_EVAL["reach_dist"],
            "algorithmic_bytes_per_launch": BYTES_PER_EVAL["reach_dist"] * n,
            "kernel_src_sha": prof["kernel_src_sha"],
            "valu_insts_per_eval": prof["valu_insts_per_eval"], "valu_source": prof["valu_source"],
        }
        if extra.get("copy_soa"):
            roofline["measured_copy_GBs"] = extra["copy_soa"]["GBs"]
            roofline["frac_of_measured_copy"] = achieved / extra["copy_soa"]["GBs"]
        if prof["valu_insts_per_eval"]:
            # the kernel is VALU-issue bound: its own floor = wave-instructions / SIMDs x 2 cycles (full-rate class)
            floor_ms = prof["valu_insts_per_eval"] * n / 64.0 / N_SIMD * 2.0 / CLOCK_HZ * 1e3
            roofline.update({"valu_floor_ms": floor_ms, "frac_valu": floor_ms / kernel_ms})
        line = {
            "metric": "leg-target evaluations/sec (reach+dist)",
            "value": total_evals / elapsed,
            "unit": "evaluations/s",
            "n_gpus": world,
            "rccl_ranks": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "precondition_ms": args.precondition_ms,
            "ms_per_step": elapsed / args.steps * 1e3,
            "gather_ms": gather_ms,
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE config 2: single M2 leg, reach+distance on {n_total} uniform-random 3-D targets (seed 42), "
                             f"identity orientation, SoA resident in HBM") if world == 1 else
                            (f"BASELINE config 4: single M2 leg, reach+distance on ONE cloud of {n_total} uniform-random 3-D targets "
                             f"(chunk-seeded), sharded contiguously over {world} GPUs ({n} points on rank 0), SoA resident in HBM"),
                "points_total": n_total, "points_per_gpu": n, "mode": args.mode,
                "table_build_ms": table_build_ms,
                "table_build_note": "the plane table of this (leg, orientation): built once, on the first call, before the timed region "
                                    "(lrm_tol_prepare); its cost is this figure, per (leg, orientation), not per step",
                "mode_contract": {"tol": "reach mask bit-exact; distance within 1e-5 of max(|d|, (|p| + body)/8): a FLOORED reading of "
                                         "BASELINE's '1e-5 relative' (include/lrm.h; literal relative error in tolerance_check)",
                                  "tol_rel": "reach mask bit-exact; |d - d_ref| <= 1e-5 |d_ref| for every vector (BASELINE's text without a floor: vectors "
                                             "shorter than 19 mm come from the bit-exact code)",
                                  "fast": "mask and every float of the distance field bit-identical to the reference's host path (table-guided kernel)",
                                  "strict": "as fast, reference operation order"}[args.mode],
                "exchange": "none" if world == 1 else f"{'RCCL' if backend == 'nccl' else backend} all-gather of the "
                                                         "bit-packed reach mask per step, overlapped on a side stream",
                "reachable_fraction_sampled": reachable_fraction,
            },
            "roofline": roofline,
            "tolerance_check": tol_check,
            "modes": other_modes,
            "kernels": extra,
            "configs": configs,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host, leg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
