"""Randomised sweeps (tests/fuzz_filterbank.py, tests/fuzz_fold.py, tests/fuzz_misc.py: scripts, not collected) with fixed seeds: filterbank geometries against the
float64 oracle, fold shapes and bin plans against the CPU loop, pipeline configurations fused against Detection + Fold; code paths that must agree bit for bit (fuzz_misc.py), the search epilogue of the inverse pass against Detection + TScrunch bit for bit (fuzz_search.py).
(The long sweeps that found the round-2 staging overflow run from the command line; these are the regression-sized ones.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,ncases,seed", [("fuzz_filterbank.py", 40, 11), ("fuzz_fold.py", 36, 12), ("fuzz_misc.py", 24, 13), ("fuzz_search.py", 40, 14)])
def test_randomised_sweep(tool, ncases, seed):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", tool), str(ncases), str(seed)], capture_output=True, text=True,
                       timeout=600)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-15:])
    assert p.returncode == 0 and "GPU core dump" not in p.stdout + p.stderr, tail
    assert "0 failures" in p.stdout, tail
