"""-m gpu: a short run of the differential fuzzer (tools/fuzz_parity.py) -- random combinations of the robot-layer, solver and
terrain options, both lane mappings, per-robot ETG parameters / dynamic rows / strength ratios / pushes -- through the C-ABI
against the fp64 and fp32 oracles built from the env's own EtgConfig, plus the fused tape kernel and the fused closed loop against
stepping and step(auto_reset) against manual resets.  The long
runs (hundreds of trials, other seeds) are a tool invocation; their last result is profiles/r04_fuzz.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_option_combinations_match_the_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--trials", "24", "--seed", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "24 trials, 0 failed" in r.stdout
