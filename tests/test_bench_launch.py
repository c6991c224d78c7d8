"""CPU tier: `python bench.py --gpus 2` must start two ranks by itself (the driver's command shape) -- checked with
`--dry-run`: gloo rendezvous on 127.0.0.1, then bench.run_workload ITSELF (the function the timed GPU run uses: sharded
loader, warm-up, timed loop between fences, max over ranks, both gathers) around a stand-in compute.  Also: the launcher refuses to pretend when the node has fewer devices than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_dry_run_spawns_two_ranks():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    # the contract is ONE JSON line on stdout: what libraries print to descriptor 1 ("[Gloo] Rank 0 is connected to ...", from
    # every rank) goes to stderr in the rank processes
    assert r.stdout.strip() == lines[0].strip(), r.stdout[:400]
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["ok"] is True
    # the ranks went through bench.run_workload itself (sharded loader, timed loop, fences, max over ranks, the plain
    # gather and the chunk-overlapped point-to-point one), not a look-alike
    assert d["timed_path"] == "run_workload" and d["gather_error"] is None and d["gather_ms"] is not None and d["ms_per_step"] > 0


def test_bench_gpus8_dry_run_is_the_drivers_command_shape():
    """`python bench.py --gpus 8 --dry-run`: the exact command shape the driver uses on the 8-GPU node, eight ranks through
    run_workload on gloo (the widest rehearsal was four).  Rank 0's line says so: eight ranks joined and
    finished the timed region, weak scaling, and rank r drives local device r -- one process per GPU, as launch_ranks /
    torchrun set LOCAL_RANK -- and both gathers returned every rank's rows."""
    r = _run("--gpus", "8", "--dry-run", "--batch", "3", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["ranks_joined"] == 8 and d["ranks_alive"] == 8 and d["scaling"] == "weak"
    assert sorted(d["rank_to_local_device"]) == [[i, i] for i in range(8)]
    assert d["ok"] is True and d["timed_path"] == "run_workload" and d["gather_error"] is None and d["gather_ms"] is not None


def test_bench_refuses_more_gpus_than_devices():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run("--gpus", str(have + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "HIP device(s) visible" in (r.stderr + r.stdout)


def test_watchdog_of_the_gather_legs():
    """bench.guarded: the (untimed, secondary) RCCL legs run under a watchdog, so that a collective that hangs cannot cost
    the headline line -- the handler fires once for a function that overstays, never for one that returns in time."""
    import threading
    import time
    sys.path.insert(0, ROOT)
    import bench
    fired = []
    assert bench.guarded(lambda: 7, 5.0, lambda: fired.append("early")) == 7
    time.sleep(0.05)
    assert fired == []
    release = threading.Event()
    t0 = time.perf_counter()
    bench.guarded(lambda: release.wait(2.0), 0.1, lambda: (fired.append("late"), release.set()))
    assert fired == ["late"] and time.perf_counter() - t0 < 1.5


def test_algorithmic_bytes_are_survey_8d():
    """bench.algorithmic_bytes -- the numerator of `roofline.achieved` -- is SURVEY.md section 8d's figure: per grid V*C*4
    (one float32 per voxel-channel) + N*(12 + 4*C) (coords and per-channel sigmas read once).  cfg2: 8 388 608 + 2 200 000
    bytes per grid (2 710.7 MB per 256-grid launch, the number DESIGN.md quotes); cfg3: (442 368 + 2 640) per pose."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    p, _, nv = bench.make_workload("cfg2", 2, seed=7)
    assert [int(v) for v in nv] == [64, 64, 64] and int(p["atom_offsets"][-1]) == 100_000
    assert bench.algorithmic_bytes(p, nv) == 2 * (8_388_608 + 2_200_000)
    assert abs(256 * (8_388_608 + 2_200_000) / 1e6 - 2710.7) < 0.05
    p, _, nv = bench.make_workload("cfg3", 5, seed=3)
    assert bench.algorithmic_bytes(p, nv) == 5 * (442_368 + 2_640)
    # ragged items (cfg5): the atoms are counted, not assumed
    p, _, nv = bench.make_workload("cfg5", 7, seed=5)
    n = int(p["atom_offsets"][-1])
    assert 7 * 20 <= n <= 7 * 50 and bench.algorithmic_bytes(p, nv) == 7 * 24 ** 3 * 8 * 4 + n * 44
    # the roofline object divides by the kernel time measured between the events
    r = bench.roofline_of({"k_ms": 4.0, "k_n": 2, "alg": 2 * 10_588_608}, "cfg2", 2, 0)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - 2 * 10_588_608 / 2e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4


def test_rank_0_prints_its_line_when_another_rank_dies():
    """Round 4: `python bench.py --gpus N` must leave a JSON line from rank 0 even if another rank dies in the timed region
    (nothing has run with N > 1 on RCCL yet).  Dry run, gloo: rank 1 exits hard right after its timed steps
    (MKAMD_BENCH_KILL_RANK) -- rank 0's closing fence fails or the launcher's SIGTERM arrives, whichever is first; either
    way its line comes out, says `ranks_alive` = 1 and why, and the launcher reports failure."""
    env = dict(os.environ, MKAMD_BENCH_KILL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
    d = lines[0]
    assert d["ok"] is False and d["n_gpus"] == 2 and d["ranks_alive"] == 1 and d["degraded"]


def test_bench_dist_workload_shards_frames_over_the_ranks_dry_run():
    """`python bench.py --workload dist --gpus N --dry-run`: N ranks, each with its own frames of the
    trajectory in a moleculekit_amd.distributed.ShardedDistances, the timed loop without a collective, rows left sharded, then
    both gathers checked row by row (a stand-in compute whose rows are a function of the frame alone)."""
    for n in (2, 8):
        r = _run("--workload", "dist", "--gpus", str(n), "--dry-run", "--batch", "3", timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and r.stdout.strip() == lines[0].strip(), r.stdout[:400]
        d = json.loads(lines[0])
        assert d["dry_run"] is True and d["n_gpus"] == n and d["ranks_alive"] == n and d["scaling"] == "weak"
        assert d["ok"] is True and d["gather_ok"] is True and d["gather_ms"] is not None and d["value"] > 0
        assert "frames per rank" in d["config"]["workload"] and "no collective" in d["config"]["sharding"]
