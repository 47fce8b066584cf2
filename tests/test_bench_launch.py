"""bench.py --gpus N from a plain shell (no torchrun around it): the parent starts the ranks itself, relays rank 0's JSON line and the
worst exit code.  Runs here without a GPU through --dry-run, which exercises everything of the N-rank path except the kernels: the
rendezvous (gloo, 127.0.0.1), block-cyclic ownership, the double-buffered gather into rank 0, the assembly, and the per-phase clocks
the N > 1 line carries (render_ms_max, gather_ms, assemble_ms, gather_bytes)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("n,extra", [(2, []), (3, ["--block-cols", "32"])])
def test_plain_shell_launch_of_n_ranks(n, extra):
    W, H, B = 256, 136, 3
    r = _run("--gpus", str(n), "--backend", "gloo", "--steps", "2", "--warmup", "1", "--dry-run", "--width", str(W), "--height", str(H), "--frames", str(B), *extra)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["value"] is None and "dry_run" in d
    assert d["assembled_frames_ok"] is True
    for k in ("render_ms_max", "gather_ms", "assemble_ms"):
        assert d[k] > 0.0, k
    assert d["gather_bytes"] > 0 and d["gather_bytes"] % (n - 1) == 0
    assert d["gather_bytes"] // (n - 1) >= B * (H // n) * W * 3 * 0.9          # one padded tile set per peer


def test_a_failing_rank_fails_the_launch():
    """Without --dry-run the ranks need a GPU; here they exit with an error, and the parent must report it (not hang, not return 0)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    r = _run("--gpus", "2", "--backend", "gloo", "--steps", "1", "--no-pmc", "--no-cpu-baseline", timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
