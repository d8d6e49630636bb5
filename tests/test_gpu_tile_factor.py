"""The serial core of the tile Cholesky on its own (tools/tile_factor_bench.hip includes rsba_amd/csrc/cholesky.hip): W = chol(D)^-1 of one
48 x 48 tile by one workgroup — the MFMA-pivot LDL^T on two waves that the DIAG tasks run, and the round-1 lane-per-row form —
against a host factorisation of the same tile."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "tile_factor_bench")


@pytest.mark.gpu
def test_tile_factor_and_inverse_match_a_host_factorisation():
    if not os.path.exists(TOOL):
        pytest.skip("tools/tile_factor_bench is not built (python __graft_entry__.py builds it)")
    out = subprocess.run([TOOL], capture_output=True, text=True, timeout=120).stdout
    rows = re.findall(r"^(.*?): *([0-9.]+) us per tile .*?\|W A W\^T - I\| = ([0-9.e+-]+), \|W - W_host\| = ([0-9.e+-]+) \(\|W\| = ([0-9.e+-]+)\), (.*)$", out, re.M)
    assert len(rows) >= 2, out
    for name, us, err_id, err_ref, wmax, status in rows:
        assert status.strip() == "no error", (name, status)
        assert float(err_id) <= 1e-13, (name, err_id)                 # W D W^T = I: backward stable on a tile of condition ~1e2
        assert float(err_ref) <= 1e-13 * max(1.0, float(wmax)), (name, err_ref)
