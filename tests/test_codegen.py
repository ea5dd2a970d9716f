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


def test_asm_write_through_stores_never_read_a_raw_accumulator():
    """store16_out (csrc/common.hip.h) is an inline-asm `global_store_dwordx4 ... sc1`: hipcc cannot guard an MFMA-write ->
    VMEM-read hazard in front of it, so its registers' last writer must be a VALU instruction.  The scanner is checked on two
    synthetic listings, then run on the device code of the library the tests load."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_asm_store_sources as sc
    bad = """
0000000000001000 <_Z4kernPv>:
	v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[12:15], v[4:7]
	s_nop 4
	global_store_dwordx4 v[0:1], v[4:7], off sc1
"""
    good = """
0000000000001000 <_Z4kernPv>:
	v_mfma_f32_16x16x32_bf16 v[4:7], v[8:11], v[12:15], v[4:7]
	v_add_f32_e32 v20, v4, v30
	v_add_f32_e32 v21, v5, v31
	v_cvt_pk_bf16_f32 v22, v20, v21
	v_permlane16_swap_b32_e32 v22, v23
	global_store_dwordx4 v[0:1], v[20:23], off sc1
"""
    assert sc.scan_text(bad)[:2] == (1, 1) and sc.scan_text(good)[:2] == (1, 0)
    lib = os.path.join(CSRC, "liblocaldiff_hip.so")
    if not os.path.exists(lib) or not os.path.exists(sc.OBJDUMP):
        pytest.skip("library or llvm-objdump missing")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_asm_store_sources.py"), lib], capture_output=True, text=True)
    first = r.stdout.splitlines()[0]
    assert r.returncode == 0 and int(first.split()[0]) > 1000 and "; 0 read" in first, r.stdout[:3000]
