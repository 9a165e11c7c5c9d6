"""The gfx950 hazard of DESIGN.md 3.4 must not be compiled into any kernel of the library: a 64-bit vector shift
whose amount register is the last of the wave's allocation (tools/k4_fault_repro.md).  CPU-only: hipcc
cross-compiles to assembly."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_64bit_shift_reads_the_last_register_of_its_allocation():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa_shift64.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout
    assert " 0 with the amount in the last register" in r.stdout, r.stdout
    # the lint looks at the code objects inside the built library (what ships), not at a separate compilation
    assert "shipped code objects of csolve_amd/libcsolve_hip.so" in r.stdout, r.stdout


def test_the_source_mode_of_the_lint_uses_the_makefiles_flags():
    """--from-source (no built library) must compile with the Makefile's HIPFLAGS: register allocation depends on them"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa_shift64 as lint
    flags = lint.makefile_hipflags()
    assert "--offload-arch=gfx950" in flags and "-O3" in flags
    mk = open(os.path.join(ROOT, "csolve_amd", "csrc", "Makefile")).read()
    for f in flags:
        assert f in mk or f == "--offload-arch=gfx950"


def test_the_lint_recognises_the_failing_shape(tmp_path):
    """the shape of the failing round-1 build: amount in v47 of a 48-register kernel is flagged, v46 is not"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa_shift64 as lint
    def asm(amount, nfv):
        return (f"k: ; @k\n\tv_lshlrev_b64 v[44:45], v{amount}, 1\n\ts_endpgm\n"
                f"\t.amdhsa_kernel k\n\t\t.amdhsa_next_free_vgpr {nfv}\n\t.end_amdhsa_kernel\n")
    assert len(lint.check(asm(47, 48), "x")[2]) == 1
    assert len(lint.check(asm(46, 48), "x")[2]) == 0
    assert len(lint.check(asm(47, 45), "x")[2]) == 1   # 45 used -> 48 owned: v47 is still the last one
    assert len(lint.check(asm(44, 45), "x")[2]) == 0
    assert len(lint.check(asm(55, 56), "x")[2]) == 1
