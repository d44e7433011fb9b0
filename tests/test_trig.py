"""cos / sin of computeOrbDescriptor (reference src/ORBextractor.cc:112-113).

`using namespace cv; using namespace std;` (:66-67) makes `cos(angle)` on a float resolve to std::cos(float) = libm's cosf,
not the double function rounds 1-3 restated.  cosf is not correctly rounded, so the reference's value is a property of its
host's libm; the oracle now calls this host's cosf / sinf and liborbgpu reproduces them on the device (a fixed IEEE
double sequence + a table of the arguments where the host differs, csrc/trig_base.h, csrc/trig.hip).  CPU part: the
overload resolution, the exhaustive scan, the residual against the correctly rounded value.  GPU part: device == host."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def scan():
    """tests/trig_test.cpp: static_asserts on the overload resolution + the exhaustive scan of [0, 6.2832]."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "trig_test")
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-pthread",
                            "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "csrc"),
                            os.path.join(ROOT, "tests", "trig_test.cpp"), "-o", exe, "-lm"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout  # includes: decltype(cos(1.0f)) is float under `using namespace std`
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)


def test_reference_expression_is_cosf(scan):
    # the expression of ORBextractor.cc:113 compiled under `using namespace std` returns cosf's bits everywhere
    assert scan["resolved_call_differs_from_cosf"] == 0
    # merging the two calls into sincosf (gcc does at -O2) changes nothing on this libm
    assert scan["sincosf_differs"] == 0


def test_base_is_the_correctly_rounded_value(scan):
    # the fixed double sequence equals (float)cos((double)x) / (float)sin((double)x) of this libm for every argument
    assert scan["base_vs_rounded_double"] == 0
    assert scan["values"] > 1_086_000_000


def test_host_table_matches_the_scan(scan, oracle):
    """The library's table (clang-compiled base, cosf / sinf through pointers) has exactly the entries the g++ scan
    counts, and evaluating through it returns the host's cosf / sinf."""
    from orb_slam2_map_amd import lib as G
    O = oracle
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(0, 6.2831855, 20000).astype(np.float32),
                         np.array([0.0, 1e-30, 0.5, 0.7853982, 1.5707964, 3.1415927, 4.712389, 6.2831855, 6.2832], np.float32)])
    hits = 0
    for x in xs:
        c, s, t = G.trig_host_eval(x)
        a, b = O.descriptor_trig(x)
        assert c.tobytes() == a.tobytes() and s.tobytes() == b.tobytes(), float(x)
        hits += t
    n, ms = G.trig_table_info()
    assert n == scan["entries"]
    assert max(scan["cos_differs"], scan["sin_differs"]) <= n <= scan["cos_differs"] + scan["sin_differs"]
    assert hits > 100  # ~2.6 % of uniformly drawn angles are table entries
    print("trig table: %d entries, %.0f ms" % (n, ms))


def test_residual_rounded_double_vs_host_libm(oracle):
    """over the golden frames and a 100 k key-point sweep, how many a / b values and descriptor bits
    differ between (float)cos((double)x) and this host's cosf / sinf.  The numbers in DESIGN.md section 2 come from
    tools/trig_residual.py; here they only have to stay in the same region."""
    import trig_residual as T
    g = T.residual(T.golden_frames())
    s = T.residual(T.sweep_frames(100000))
    print("goldens:", g, "sweep:", s)
    assert g["keypoints"] >= 2000 and s["keypoints"] >= 100000
    assert 0 < s["a_or_b_differs"] < 0.05 * s["keypoints"]  # ~2.6 % with glibc 2.35
    assert s["descriptor_bits_differ"] <= 64 and g["descriptor_bits_differ"] == 0  # far below one per 10 k key points
    assert oracle.lib().ora_get_trig_mode() == oracle.TRIG_LIBM_FLOAT


@pytest.mark.gpu
def test_device_trig_equals_host_libm(gpu, oracle):
    """Every table entry's neighbourhood and 4 M random arguments: the device's values are this host's cosf / sinf."""
    rng = np.random.default_rng(9)
    assert gpu.get_trig_mode() == gpu.TRIG_HOST_LIBM
    x = np.concatenate([rng.uniform(0, 6.2831855, 4_000_000).astype(np.float32),
                        rng.integers(0, 0x40C90FDB, 1_000_000, dtype=np.uint32).view(np.float32),  # uniform over bit patterns
                        np.array([0.0, 6.2831855, 6.2832], np.float32)])
    c, s = gpu.trig_eval(x)
    a, b = oracle.descriptor_trig_array(x)  # this host's cosf / sinf (numpy's float32 functions are not libm's)
    assert c.tobytes() == a.tobytes() and s.tobytes() == b.tobytes()
    n, ms = gpu.trig_table_info()
    assert n > 0
    # the rounded-double mode: no table, the correctly rounded values
    gpu.set_trig_mode(gpu.TRIG_ROUNDED_DOUBLE)
    try:
        c1, s1 = gpu.trig_eval(x)
    finally:
        gpu.set_trig_mode(gpu.TRIG_HOST_LIBM)
    assert np.array_equal(c1, np.cos(x.astype(np.float64)).astype(np.float32))
    assert np.array_equal(s1, np.sin(x.astype(np.float64)).astype(np.float32))
    frac = float(((c1 != c) | (s1 != s)).mean())
    assert 0.001 < frac < 0.05
    print("device: %d table entries (%.0f ms scan); %.2f %% of the sampled arguments corrected" % (n, ms, 100 * frac))


@pytest.mark.gpu
def test_rounded_double_mode_extraction(gpu, oracle, stream640):
    """ORBGPU_TRIG_ROUNDED_DOUBLE against the oracle in the same mode (the default mode is what every other extractor
    test runs)."""
    g = np.stack([stream640.frame(t)[0] for t in (3, 4)])
    gpu.set_trig_mode(gpu.TRIG_ROUNDED_DOUBLE)
    oracle.set_trig_mode(oracle.TRIG_ROUNDED_DOUBLE)
    try:
        ext = gpu.ORBextractor(1000, max_batch=2)
        ks, ds = ext.extract_batch(g)
        oe = oracle.Extractor(1000)
        for i in range(2):
            ok, od = oe.extract(g[i])
            assert ks[i].tobytes() == ok.tobytes() and np.array_equal(ds[i], od)
    finally:
        gpu.set_trig_mode(gpu.TRIG_HOST_LIBM)
        oracle.set_trig_mode(oracle.TRIG_LIBM_FLOAT)
