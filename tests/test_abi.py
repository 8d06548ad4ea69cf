"""CPU: the C-ABI library loads and exports every symbol include/mtr.h declares (no compute calls
without a GPU), the header compiles as plain C, and the FFI structs have the sizes the reference pins
(PrimitiveInfo = 0x38, src/rmodel.rs:489)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "mtr.h")


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mt_renderer_amd import api
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(api.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libmtr.so does not export {n}"
    assert sorted(api.EXPORTED_SYMBOLS) == names
    assert api.lib.mtr_abi_version() == 2


def test_header_is_plain_c_and_struct_sizes():
    prog = r'''
#include "mtr.h"
_Static_assert(sizeof(mtr_primitive) == 0x38, "PrimitiveInfo is 0x38 bytes (src/rmodel.rs:489)");
_Static_assert(sizeof(mtr_element) == 8, "mtr_element");
_Static_assert(sizeof(mtr_layout) == 4 + 8 * 8, "mtr_layout");
_Static_assert(sizeof(mtr_frame_stats) == 80, "mtr_frame_stats");
int main(void) { return 0; }
'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o",
                               os.path.join(td, "t.o")])


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mt_renderer_amd import api
    with pytest.raises(api.MtrError) as e:
        api.Device(0)
    assert e.value.code == api.MTR_E_HIP


def test_host_crc32_matches_reference_kat():
    from mt_renderer_amd import api
    assert api.crc32(b"MtObject") == 0x2EA10CEB  # src/util/crc.rs:55
    assert api.crc32(b"rModel") & 0x7FFFFFFF == 1486968918  # src/dti.txt, rule src/dti.rs:174


def test_shard_bytes_matches_host_index_math():
    from mt_renderer_amd import api, sharding
    for (w, h, n) in [(1920, 1080, 1), (1920, 1080, 8), (3840, 2160, 4), (333, 171, 3)]:
        assert api.lib.mtr_shard_bytes(w, h, n) == sharding.shard_bytes(w, h, n)


def test_product_does_not_touch_the_oracle():
    """The product path (package + C sources + public header) must not import, link or include oracle/."""
    bad = []
    for base in ("mt_renderer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle\b|oracle/|mtr_oracle|liboracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def _build_cpp_demo(td):
    exe = os.path.join(td, "cube_demo")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "cube_demo.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "mt_renderer_amd"), "-lmtr", "-Wl,-rpath," + os.path.join(ROOT, "mt_renderer_amd")])
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu():
    import torch
    with tempfile.TemporaryDirectory() as td:
        exe = _build_cpp_demo(td)
        if torch.cuda.is_available():
            pytest.skip("GPU present: covered by the gpu-marked test")
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 77, (r.returncode, r.stderr)  # MTR_E_HIP surfaced as mtr::Error


@pytest.mark.gpu
def test_cpp_mirror_renders_the_cube():
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([_build_cpp_demo(td)], capture_output=True, text=True)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
