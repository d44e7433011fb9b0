"""The C-ABI library loads and exports every symbol include/orbgpu.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def glib():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "orb_slam2_map_amd", "liborbgpu.so")):
        ge.build()
    from orb_slam2_map_amd import lib
    return lib


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "orbgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbgpu_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(glib):
    L = glib.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(glib.ABI_SYMBOLS) == syms, "lib.ABI_SYMBOLS out of sync with the header"
    assert L.orbgpu_abi_version() == 1


def test_struct_layouts(glib):
    assert glib.KEYPOINT_DTYPE.itemsize == 28  # cv::KeyPoint
    assert glib.POINT_DTYPE.itemsize == 16
    assert C.sizeof(glib.ExtractorParams) == 28


def test_argument_validation_needs_no_gpu(glib):
    L = glib.lib()
    h = C.c_void_p()
    p = glib.ExtractorParams(1000, 1.2, 40, 20, 7, 0, 1)  # nlevels out of range
    assert L.orbgpu_extractor_create(C.byref(p), C.byref(h)) == glib.EINVAL
    assert b"nlevels" in L.orbgpu_last_error_string()
    assert L.orbgpu_extractor_create(None, C.byref(h)) == glib.EINVAL
    cs = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(4, np.int32)
    x = np.array([5.0, 635.0, 700.0, 320.0], np.float32)
    y = np.array([5.0, 475.0, 100.0, 240.0], np.float32)
    rc = L.orbgpu_assign_features_to_grid(4, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 0.0, 0.0,
                                          0.1, 0.1, cs.ctypes.data_as(C.c_void_p), items.ctypes.data_as(C.c_void_p))
    assert rc == 0 and cs[-1] == 2  # (635,475) rounds to cell 64 -> dropped like PosInGrid; 700 is outside


def test_no_cpu_fallback(glib):
    if glib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(glib.OrbGpuError) as ei:
        glib.ORBextractor(1000)
    assert ei.value.status == glib.EHIP and "no CPU fallback" in str(ei.value)
    a = np.zeros((4, 32), np.uint8)
    with pytest.raises(glib.OrbGpuError):
        glib.ORBmatcher.DescriptorDistance(a, a)
    with pytest.raises(glib.OrbGpuError):
        glib.PointCloudMapping(0.01)


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "orb_slam2_map_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_py" not in src and "orb_oracle" not in src and "liborb_oracle" not in src, f
    out = os.popen("ldd %s" % os.path.join(pkg, "liborbgpu.so")).read()
    assert "oracle" not in out


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/orbgpu.h compiles as C99 with -pedantic -Werror, and a C program links against the
    library and calls it (no device needed for the version query)."""
    import subprocess
    src = tmp_path / "abi_c99.c"
    src.write_text('#include "orbgpu.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    orbgpu_extractor_params p = {1000, 1.2f, 8, 20, 7, 0, 1};\n'
                   '    orbgpu_keypoint k; orbgpu_point_xyzrgba q; (void)p; (void)k; (void)q;\n'
                   '    printf("%d %d %d\\n", orbgpu_abi_version(), (int)sizeof(orbgpu_keypoint), (int)sizeof(orbgpu_point_xyzrgba));\n'
                   '    return orbgpu_abi_version() == ORBGPU_ABI_VERSION ? 0 : 1;\n}\n')
    pkg = os.path.join(ROOT, "orb_slam2_map_amd")
    exe = str(tmp_path / "abi_c99")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src),
                    "-o", exe, "-L" + pkg, "-lorbgpu", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.split()[1:] == ["28", "16"], r.stdout
