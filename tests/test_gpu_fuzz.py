"""A short run of tools/fuzz_parity.py: random Tanner graphs (regular, irregular, empty and heavy
nodes), channel probabilities, iteration caps, ragged batches, both BP kernels with random geometry
/ hand-off options and BP-OTS, every result compared with the CPU oracles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_parity_fuzz(gpu, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "8", str(seed)],
                         capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
