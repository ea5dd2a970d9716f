"""Static checks of the compiled kernels (no GPU: hipcc cross-compiles gfx950 here).

The vector-memory counter of a wave is in order and counts stores, so a wait the compiler places behind a store
drains that store as well (DESIGN findings 59, 63).  These tests compile one kernel file to assembly and run the
scanners that found the cases fixed in round 3, so that an innocent-looking edit of an epilogue does not bring them back."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "localdiffusion-hallucination_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(name, tmp_path, extra=()):
    out = str(tmp_path / (name + ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-S", "--cuda-device-only",
           os.path.join(CSRC, name + ".hip"), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=CSRC)
    return out


def _run_tool(name, path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name + ".py"), path], check=True, capture_output=True, text=True)
    return r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_conv3x3_epilogues_have_no_wait_behind_a_store(tmp_path):
    # conv3x3.hip is compiled without -amdgpu-mfma-vgpr-form (csrc/build.sh)
    out = _run_tool("scan_store_waits", _asm("conv3x3", tmp_path))
    assert " 0 kernels with a drain behind a store" in out.splitlines()[0], out[:2000]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fused_linear_attention_has_no_wait_behind_a_store(tmp_path):
    out = _run_tool("scan_store_waits", _asm("linattn_fused", tmp_path, extra=("-mllvm", "-amdgpu-mfma-vgpr-form")))
    assert " 0 kernels with a drain behind a store" in out.splitlines()[0], out[:2000]
