"""bench.py and __graft_entry__.py at the repo root: importable / runnable on a host without a GPU as far as they can go."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--lanes", "--inflight", "--interval-optimization", "--no-cpu-baseline"):
        assert flag in out.stdout


def test_bench_refuses_to_run_without_a_gpu():
    """the product has no CPU path: on a host without a HIP device bench.py stops with a message instead of falling back"""
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0
    assert "MI355X" in (out.stderr + out.stdout)


def test_graft_entry_exposes_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
