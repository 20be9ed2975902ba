"""CPU tier, build container only: the reference's own localize_3D / fit2D / identify / zfit executed with
``picasso_amd.localize.install()`` applied (tests/golden/check_install_callsites.py).  Needs the read-only
reference tree; skipped where it does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/picasso"), reason="reference tree not present")
def test_reference_call_sites_accept_the_rebound_workers():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "check_install_callsites.py")],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "call sites ok" in out.stdout
