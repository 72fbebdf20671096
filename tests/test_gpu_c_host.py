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


@pytest.mark.gpu
@pytest.mark.parametrize("threads,per_thread,extra,tune", [
    (4, 1500, (), ""), (16, 1500, (), ""), (48, 700, (), ""),
    (16, 1500, (), "service_workers=0"),                                  # a launch per batch instead of the resident workers
    (16, 1500, (), "service_push=0"),                                     # resident workers that read the units from host memory (no large BAR)
    (16, 3000, ("640", "384", "4", "40"), "service_linger_us=30,service_life_ms=1"),    # workers come and go all the time
    (16, 600, ("640", "384", "4", "0", "16"), ""),                        # the exhaustive search on the whole workgroup
])
def test_service_stress_from_c_threads(threads, per_thread, extra, tune):
    """tests/c_host/service_stress.c: pthread workers post searches concurrently; every answer must equal the one the same request
    got single-threaded (any interleaving, batching, ring wrap-around; more workers than the box has cores for the 48-thread case)"""
    exe = os.path.join(ROOT, "tests", "c_host", "service_stress")
    if not os.access(exe, os.X_OK):
        import __graft_entry__ as g
        g.build_c_host()
    env = dict(os.environ)
    if tune:
        env["KVZ_HIP_TUNE"] = tune
    r = subprocess.run([exe, str(threads), str(per_thread)] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240, env=env)
    print(r.stdout)
    assert r.returncode == 0 and "service_stress ok" in r.stdout, r.stdout
