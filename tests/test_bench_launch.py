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
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["ok"] is True
    # the ranks went through bench.run_workload itself (sharded loader, timed loop, fences, max over ranks, the plain
    # gather and the chunk-overlapped point-to-point one), not a look-alike
    assert d["timed_path"] == "run_workload" and d["gather_error"] is None and d["gather_ms"] is not None and d["ms_per_step"] > 0


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
