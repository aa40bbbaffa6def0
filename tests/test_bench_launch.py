"""`python bench.py --gpus N` must start from the plain command the driver uses (no torch.distributed.run around it):
the parent launches the ranks itself and never touches the GPU.  Rehearsed here without a GPU through
LRM_BENCH_DRYRUN=1 (gloo, nothing evaluated: the step loop, the gathers, the rank count and the JSON relay)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_self_launch_two_ranks_one_json_line():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--total-points", "300000"], {"LRM_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dry_run"] is True and rec["value"] is None
    assert rec["gathered_words_ok"] is True and rec["steps"] == 4
    assert rec["config"]["points_per_gpu"] == 150016  # ceil(3e5 / 2) rounded up to whole 64-point words
    assert rec["gather_ms"] is not None and rec["gather_ms"] > 0


def test_three_ranks_ragged():
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "1", "--total-points", "1000"], {"LRM_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip())
    assert rec["rccl_ranks"] == 3 and rec["gathered_words_ok"] is True


def test_parent_stays_off_torch():
    """the launcher process imports neither torch nor the HIP library: it could not start GPU children otherwise"""
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--total-points', '6400'];"
            f"sys.path.insert(0, {ROOT!r}); import bench; bench.main();"
            "bad = [m for m in ('torch', 'lrm_amd', 'lrm_amd._capi') if m in sys.modules];"
            "assert not bad, bad; print('PARENT_CLEAN', file=sys.stderr)")
    env = dict(os.environ, LRM_BENCH_DRYRUN="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "PARENT_CLEAN" in r.stderr
    assert json.loads(r.stdout.strip())["rccl_ranks"] == 2


def test_rank_count_mismatch_fails_loudly():
    """under torch.distributed.run with a WORLD_SIZE that is not --gpus the bench refuses to run"""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"LRM_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
