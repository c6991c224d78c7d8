"""Process plumbing of `bench.py --gpus N`: starting the ranks, the watchdog of the gather legs, and what keeps rank 0's line alive when
another rank dies."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

from .workloads import ROOT

def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _json_only_stdout():
    """The contract is ONE JSON line on stdout, and libraries write there too (gloo: "[Gloo] Rank n is connected to ..." from every
    rank; RCCL with NCCL_DEBUG set).  From here on file descriptor 1 IS stderr for everything below Python; `print` keeps the real
    stdout through a duplicate.  (Rank processes only: the launcher's children inherit its descriptors.)"""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(keep, "w", buffering=1)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves -- one process per
    GPU under torch.distributed.run on this node (rendezvous on 127.0.0.1) -- and let rank 0 print the line."""
    import subprocess
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not (os.environ.get("MKAMD_BENCH_SHARE_DEVICES", "0") == "1" and have > 0):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + argv
    return subprocess.call(cmd, env=env)


def guarded(fn, seconds, on_timeout):
    """fn() under a watchdog: when it has not returned after `seconds`, on_timeout() is called from another thread (fn
    itself keeps running: a collective that hangs cannot be cancelled, only left behind)."""
    done = threading.Event()

    def watchdog():
        if not done.wait(seconds):
            on_timeout()

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        return fn()
    finally:
        done.set()



# ---- a rank that dies must not cost rank 0 its line ----------------------------------------------------------------------
# Nothing here has run with N > 1 on RCCL where this file was written.  Every collective BETWEEN the ranks (the gloo fences,
# the max over ranks, the gather legs) goes through _collective(): the first failure is remembered, nothing is attempted
# after it, and rank 0 reports what it measured itself with `ranks_alive` (ranks that finished the timed region, read
# from the rendezvous store, which lives in the launcher) and `degraded` on the line.  torchrun answers a dead worker by
# sending the others SIGTERM: rank 0 turns that into its line too (term_reporter: a wake-up fd and a thread, so that it
# works while the main thread sits inside a collective).
_RANKS = {"broken": None, "emit": None}


def _collective(fn, default=None):
    if _RANKS["broken"]:
        return default
    try:
        return fn()
    except Exception as e:                                    # noqa: BLE001 -- reported on the line
        _RANKS["broken"] = f"{type(e).__name__}: {e}"[:200]
        return default


def _store():
    try:
        import torch.distributed as dist
        return dist.distributed_c10d._get_default_store() if dist.is_initialized() else None
    except Exception:                                         # noqa: BLE001
        return None


def mark_done(rank):
    st = _store()
    if st is not None:
        try:
            st.set(f"mkamd_bench_timed_{rank}", "1")
        except Exception:                                     # noqa: BLE001
            pass


def ranks_done(world):
    st = _store()
    if st is None:
        return world
    n = 0
    for r in range(world):
        try:
            n += bool(st.check([f"mkamd_bench_timed_{r}"]))
        except Exception:                                     # noqa: BLE001
            pass
    return n


def term_reporter():
    """SIGTERM -> whatever _RANKS['emit'] holds is called (rank 0's line, as far as it got), then the process ends."""
    import select
    import signal
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.signal(signal.SIGTERM, lambda *_: None)            # (a Python-level handler must exist for the wake-up fd to fire)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def watch():
        while True:
            select.select([r], [], [])
            if signal.SIGTERM in os.read(r, 64):
                _RANKS["broken"] = _RANKS["broken"] or "SIGTERM: the launcher is taking the job down (a rank failed)"
                try:
                    if _RANKS["emit"]:
                        _RANKS["emit"]()
                finally:
                    os._exit(1)

    threading.Thread(target=watch, daemon=True).start()


def _max_over_ranks(x, world):
    """MAX of a host scalar over the ranks through a CPU tensor (the gloo side of the process group): nothing in or around
    the timed region touches RCCL -- once an RCCL communicator exists in the process every kernel of the step runs 3-7 %
    slower (measured with one rank, profiles/r3_torchrun_probe.txt), so it is first created by the gather legs, after
    everything that is timed."""
    if world > 1 or "RANK" in os.environ:
        import torch
        import torch.distributed as dist
        tt = torch.tensor([x], dtype=torch.float64)
        return _collective(lambda: (dist.all_reduce(tt, op=dist.ReduceOp.MAX), float(tt.item()))[1], default=x)
    return x
