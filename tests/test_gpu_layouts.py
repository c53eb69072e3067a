"""The GloVe-width Euclidean kernels exist in two global-memory layouts (row-aligned `euclid_pair32_kernel`,
workgroup-dense `euclid_block_kernel`; simcross_elementwise.hip).  The library picks one per kind of launch; the
other GPU tests therefore see forward = row-aligned, backward = dense, fused = row-aligned.  This one runs the
opposite choice for all three (a process-wide dev switch, hence the subprocess) against the CPU oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("fwd,bwd,fused", [("block", "pair", "block"), ("wave", "wave", "wave"), ("pair", "block", "pair")])
def test_every_layout_is_bitexact(fwd, bwd, fused, hiplib, oracle):
    env = dict(os.environ, MMS_EUCLID_LAYOUT_FWD=fwd, MMS_EUCLID_LAYOUT_BWD=bwd, MMS_EUCLID_LAYOUT_FUSED=fused)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "layout_gpu_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "mismatching outputs 0" in out.stdout
