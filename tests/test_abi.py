"""CPU: the C-ABI shared library loads and exports every symbol include/kvz_hip.h
declares (no compute calls -- there is no GPU here), the symbol hygiene rule of
the reference holds (tests/test_external_symbols.sh:7: every extern is kvz_
prefixed), and the product never routes through the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kvz_hip.h")
LIB = os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"KVZ_HIP_API\s+[^;(]*?\b(kvz_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return LIB


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 40
    for s in ("kvz_hip_sad_nxn_batch", "kvz_hip_satd_nxn_batch", "kvz_hip_transform_batch", "kvz_hip_quant_batch",
              "kvz_hip_sample_luma_batch", "kvz_hip_search_frac_batch", "kvz_strategy_register_picture_hip",
              "kvz_strategy_register_dct_hip", "kvz_strategy_register_quant_hip", "kvz_strategy_register_ipol_hip"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib_path):
    L = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_python_binding_covers_the_header(lib_path):
    from kvazaar_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()        # binds all of them; no GPU touched


def test_exported_symbols_are_kvz_prefixed(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    bad = []
    for line in out.splitlines():
        parts = line.split()
        if len(parts) == 3 and parts[1] in "TDB":
            name = parts[2]
            if not name.startswith("kvz_") and not name.startswith("_Z") and not name.startswith("__hip"):
                bad.append(name)
    # C++ template instantiations of kernels (_Z...) are device-stub symbols; the C surface must be kvz_ only
    assert not bad, bad


def test_no_device_fails_loudly(lib_path):
    """in this container there is no GPU: init must fail with an error, not fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kvazaar_amd import _lib
    L = _lib.load()
    assert L.kvz_hip_init(0) != 0
    assert b"no HIP device" in L.kvz_hip_last_error() or L.kvz_hip_last_error()
    with pytest.raises(_lib.KvzHipError):
        _lib.init(0)
    # registration hooks refuse too (=> kvz_strategyselector_init would fail, strategyselector.c:54-95)
    assert L.kvz_strategy_register_picture_hip(None, 8) == 0


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "kvazaar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "kvz_oracle" not in text and "libkvzref" not in text, f


def test_build_is_free_of_v_ashr_pk_u8_i32():
    """hipcc 7.2 selects v_ashr_pk_u8_i32 for "two (x >> 16) clamped to bytes" and consumes the result as if the upper half
    of the destination were zero; on gfx950 it is left untouched, which corrupted a reconstruction kernel (quant.hip).
    Guard: no translation unit of the product may contain the instruction."""
    import concurrent.futures
    import glob
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    srcs = sorted(glob.glob(os.path.join(ROOT, "kvazaar_amd", "csrc", "*.hip")))
    tmp = tempfile.mkdtemp(prefix="kvz_isa_")

    def compile_one(src):
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return src, open(out).read().count("v_ashr_pk_u8_i32")
    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
            hits = [(s, n) for s, n in ex.map(compile_one, srcs) if n]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert not hits, hits


def test_python_record_layouts_match_the_header(tmp_path):
    """the numpy record types the tests, tools and bench.py build descriptors with (tests/patterns.py) against sizeof / offsetof of
    the structs of include/kvz_hip.h as a C compiler lays them out"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import patterns as P
    pairs = [("kvz_hip_me_pu", P.ME_PU), ("kvz_hip_me_params", P.ME_PARAMS), ("kvz_hip_me_result", P.ME_RESULT), ("kvz_hip_me_cabac", P.ME_CABAC),
             ("kvz_hip_cu_info", P.CU_INFO), ("kvz_hip_deblock_params", P.DEBLOCK_PARAMS), ("kvz_hip_inter_params", P.INTER_PARAMS),
             ("kvz_hip_merge_cand", P.MERGE_CAND), ("kvz_hip_me_request", P.ME_REQUEST), ("kvz_hip_me_service_config", P.ME_SERVICE_CONFIG),
             ("kvz_hip_me_service_stats", P.ME_SERVICE_STATS)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "kvz_hip.h"', 'int main(void) {']
    for cname, dt in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for field in dt.names:
            if field in ("pad", "reserved"):
                continue
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ['  printf("kvz_hip_intra_ref %zu\\n", sizeof(kvz_hip_intra_ref));', '  printf("kvz_hip_intra_pos %zu\\n", sizeof(kvz_hip_intra_pos));',
              '  printf("kvz_hip_block_pair %zu\\n", sizeof(kvz_hip_block_pair));', '  printf("kvz_hip_bipred_cand %zu\\n", sizeof(kvz_hip_bipred_cand));',
              '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, dt in pairs:
        assert int(got[cname]) == dt.itemsize, cname
        for field in dt.names:
            if field in ("pad", "reserved"):
                continue
            assert int(got["%s.%s" % (cname, field)]) == dt.fields[field][1], "%s.%s" % (cname, field)
    assert (int(got["kvz_hip_intra_ref"]), int(got["kvz_hip_intra_pos"]), int(got["kvz_hip_block_pair"]), int(got["kvz_hip_bipred_cand"])) == (130, 8, 24, 24)


def _init_with_env_device(value):
    """kvz_hip_init(-1) in a child process whose $KVZ_HIP_DEVICE is `value` -> (exit code, rc, error text)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from kvazaar_amd import _lib; L = _lib.load(); rc = L.kvz_hip_init(-1); "
            "print(rc, (L.kvz_hip_last_error() or b'').decode())" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, KVZ_HIP_DEVICE=value), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    out = r.stdout.strip().split(" ", 1)
    return r.returncode, int(out[0]) if out and out[0].lstrip("-").isdigit() else None, out[1] if len(out) > 1 else ""


def test_negative_env_device_is_refused_not_indexed():
    """$KVZ_HIP_DEVICE=-1 used to reach the context table with index -1: now an error code, never a crash.  Without a
    GPU the device count fails first (KVZ_HIP_ERR_NO_DEVICE); the GPU variant below sees KVZ_HIP_ERR_INVALID."""
    code, rc, _ = _init_with_env_device("-1")
    assert code == 0 and rc in (-1, -2)


@pytest.mark.gpu
def test_negative_env_device_is_refused_on_a_gpu_box():
    code, rc, text = _init_with_env_device("-1")
    assert code == 0 and rc == -2 and "out of range" in text
