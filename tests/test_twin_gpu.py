"""The shipped code object against its forced-wait twin (round-3 review item 8): `libtha4_hip_wait0.so` is the SAME source compiled with
`-mllvm -amdgpu-waitcnt-forcezero=1` - every `s_waitcnt` waits for everything - so both libraries execute identical arithmetic and any
byte that differs between them is a missing wait / hazard / race in the shipped one (this comparison is what exposed the faulty level-2
geometry in round 3, HISTORY.md B §4).  The disassembly gate (`test_library_holds_no_packed_fp32_instructions`) names ONE known hazard
class; this test does not need to know the class.  Built by `__graft_entry__.build()` (`tha4_amd._build.build_forced_wait_twin`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _twin():
    import tha4_amd  # noqa: F401
    from tha4_amd import _build
    if not os.path.exists(_build.TWIN):
        pytest.skip("forced-wait twin not built (python -c 'import __graft_entry__ as g; g.build()')")
    if os.path.getmtime(_build.TWIN) < os.path.getmtime(_build.LIB):
        pytest.skip("forced-wait twin is older than the library it should mirror: rebuild both with __graft_entry__.build()")
    return _build.TWIN


def test_student_library_equals_its_forced_wait_twin():
    """Both characters, 32 poses of the config-2 stream in batches of 8 + single frames, all six outputs hashed: 0 differing."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_libs.py"), "default", _twin(), "32"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "differing: 0" in r.stdout


def test_full_model_library_equals_its_forced_wait_twin():
    """Handles for 1 / 4 / 8 frames (three launch plans incl. the four-wave convolution tiles), all 33 outputs hashed, steady and cold."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_libs.py"), "default", _twin(), "4", "--full"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "differing: 0" in r.stdout
