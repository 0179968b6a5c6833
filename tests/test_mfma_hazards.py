"""The kernels issue their MFMAs through asm statements (AGPR accumulators updated in place), which hides them from the
compiler's hazard recogniser: a VGPR written by a VALU instruction must not be read by a v_mfma within the next two wait
states.  tools/check_mfma_hazards.py compiles the f16 kernel file to gfx950 assembly and scans every kernel for that pattern
(the 2 x 4 wave tile of conv_h3g_kernel had it once: a zero-select sunk next to its MFMA, velocity off by 30 %)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_no_valu_write_feeds_an_asm_mfma_too_early():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 hazard(s) in" in r.stdout


def test_scanner_flags_the_pattern():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_hazards as C
    asm = """
_ZN3nbe4demoEv: ; @demo
\tv_cndmask_b32_e64 v16, v56, 0, s[8:9]
\tv_mfma_f32_16x16x32_f16 a[56:59], v[16:19], v[12:15], a[56:59]
\ts_endpgm
_ZN3nbe5demo2Ev: ; @demo2
\tv_cndmask_b32_e64 v16, v56, 0, s[8:9]
\ts_nop 1
\tv_mfma_f32_16x16x32_f16 a[56:59], v[16:19], v[12:15], a[56:59]
\ts_endpgm
_ZN3nbe5demo3Ev: ; @demo3
\tv_cndmask_b32_e64 v16, v56, 0, s[8:9]
\tv_mfma_f32_16x16x32_f16 a[0:3], v[20:23], v[12:15], a[0:3]
\tv_mfma_f32_16x16x32_f16 a[56:59], v[16:19], v[12:15], a[56:59]
\ts_endpgm
"""
    found = C.scan(asm)
    assert [f[0] for f in found] == ["_ZN3nbe4demoEv", "_ZN3nbe5demo3Ev"]
