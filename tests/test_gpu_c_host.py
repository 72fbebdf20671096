"""A plain C99 program (tests/c_host/c_host_smoke.c: no HIP headers, no C++, only include/kvz_hip.h) hosts the library
the way Kvazaar would: the CPU test checks that it compiles and links against every symbol it uses, the GPU test runs it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_c_host_builds_and_links():
    import __graft_entry__ as g
    exe = g.build_c_host()
    assert os.access(exe, os.X_OK)
    # every undefined kvz_hip_* symbol of the program is exported by the library it will load
    need = {l.split()[-1] for l in subprocess.check_output(["nm", "-u", exe], text=True).splitlines() if "kvz_hip_" in l}
    have = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so")],
                                                            text=True).splitlines()}
    assert need and need <= have, need - have


@pytest.mark.gpu
def test_c_host_runs():
    exe = os.path.join(ROOT, "tests", "c_host", "c_host_smoke")
    if not os.access(exe, os.X_OK):
        import __graft_entry__ as g
        exe = g.build_c_host()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "c_host_smoke ok" in r.stdout, r.stdout
