"""The launch structure bench.py relies on, pinned: renders through the C ABI captured into a hipGraph (torch.cuda.CUDAGraph = HIP
stream capture) and replayed give the eager frames bit for bit.  Runs in a subprocess: torch must initialise HIP before the
library does (two HIP users in one process), which the other GPU tests' process cannot guarantee."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_renders_captured_into_a_hip_graph_replay_bit_exact():
    r = subprocess.run([sys.executable, os.path.join(HERE, "graph_capture_case.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "graph capture case: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_randomised_scenes_shipped_pipelines_against_the_oracle():
    """tools/fuzz_gpu.py, 16 seeds: random small scenes / sizes / light-sample counts / shares, the non-counting pipelines and the
    batch call against the oracle (own process: the tool prints one line per configuration)."""
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "--seeds", "16"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz: 0 mismatching" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
