"""Guards, on the device assembly hipcc emits here (no GPU needed), for two load patterns that round 6 found and removed
(tools/isa_wait_scan.py): the WPE lag-product kernel's prefetch of the next tile and the register solver's tile
loads must be BATCHES of loads in flight, not one global round trip per load."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_wait_scan as scan  # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def wpe_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "wpe_kernels.s"
    csrc = os.path.join(ROOT, "distant_speech_recognition_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                    "-I", csrc, os.path.join(csrc, "wpe_kernels.hip"), "-o", str(out)], check=True, capture_output=True)
    return dict(scan.kernels(out.read_text()))


def _kernel(asm, fragment):
    hits = [body for name, body in asm.items() if fragment in name]
    assert len(hits) == 1, (fragment, [n for n in asm if fragment in n])
    return hits[0]


def test_lagprod_prefetch_is_one_batch_of_loads(wpe_asm):
    body = _kernel(wpe_asm, "wpe_lagprod16_w2_kernel")
    # the weights of the next tile are global_load_dword: none of them may be waited for alone (eight in a row per tile until round 6)
    assert scan.serial_of(body, "global_load_dword") == 0
    # 8 sample loads + 8 weight loads + the tile's exponent in flight together: before the tile loop and inside it, in each of the four
    # row-block instantiations of the task
    assert scan.batches(body, 16) >= 8


def test_register_solver_loads_tiles_in_groups(wpe_asm):
    body = _kernel(wpe_asm, "wpe_solve_reg_kernel")
    assert scan.batches(body, 16) >= 5          # twenty tile slots, four at a time (a branch per slot: twenty round trips)


def test_scan_counts_a_serial_load():
    body = "\n".join(["global_load_dword v1, v[2:3], off", "s_waitcnt vmcnt(0)", "v_mul_f32_e32 v1, v1, v4",
                      "global_load_dword v1, v[5:6], off", "s_waitcnt vmcnt(0)", "v_add_f32_e32 v0, v0, v1",
                      "global_load_dwordx2 v[8:9], v[5:6], off", "global_load_dwordx2 v[10:11], v[5:6], off offset:8", "s_waitcnt vmcnt(1)"])
    assert scan.scan(body) == (2, 2, 4)
    assert scan.serial_of(body, "global_load_dword") == 2
    assert scan.batches(body, 2) == 1
