"""Static resource checks of the gfx950 code objects (hipcc cross-compiles; no GPU): register budgets that the kernels' residency,
and with it their measured times, rest on."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_lighting_kernel_fits_five_waves_per_simd(tmp_path):
    """k_lighting<2, ...> (two pixels per lane, the form every even-sized frame takes) runs five waves per SIMD: 512 registers / 5 = 102, in
    allocation steps of 8 = 96 at most, nothing spilled.  Round 6 met 98 twice while adding the wide-window path (the per-pixel ranges kept alive
    into the walk; a scalar slot base): four waves per SIMD and 173-175 us instead of 167-170 (profiles/r06_scenes_depth_split_hot_spot.txt).
    The one-pixel form is far below the seven waves its launcher asks for."""
    out = tmp_path / "lighting.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "granite_amd", "csrc", "lighting.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*k_lightingILi(\d)\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) == 8, [k[0] for k in kernels]
    for name, px, vgprs, spilled in kernels:
        assert int(spilled) == 0, (name, spilled)
        assert int(vgprs) <= (96 if px == "2" else 72), (name, vgprs)
