"""Oracle vs constants the reference's sources imply."""
import hashlib
import struct

import numpy as np


def test_quotas_and_tables(oracle):
    e = oracle.Extractor(1000, 1.2, 8, 20, 7)  # Examples/RGB-D/TUM1.yaml:42-55
    assert list(e.quotas()) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(oracle.Extractor(2000).quotas()) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(e.umax()) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = e.scale_factors()
    expect = np.array([1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638,
                       3.5831816196], np.float32)
    assert np.array_equal(sf, expect)
    assert np.array_equal(e.sigma2(), sf * sf)
    assert np.array_equal(e.inv_scale_factors(), np.float32(1) / sf)
    assert sum(e.quotas()) == 1000


def test_pyramid_sizes(oracle):
    e = oracle.Extractor(1000)
    e.extract(np.zeros((480, 640), np.uint8))
    sizes = [(e.pyramid_level(l).shape[1] - 38, e.pyramid_level(l).shape[0] - 38) for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    e2 = oracle.Extractor(2000)
    e2.extract(np.zeros((960, 1280), np.uint8))
    sizes = [(e2.pyramid_level(l).shape[1] - 38, e2.pyramid_level(l).shape[0] - 38) for l in range(8)]
    assert sizes == [(1280, 960), (1067, 800), (889, 667), (741, 556), (617, 463), (514, 386), (429, 322), (357, 268)]


def test_pattern_fingerprint(oracle):
    p = oracle.pattern().astype(np.int32)
    assert len(p) == 1024
    assert list(p[:8]) == [8, -3, 9, 5, 4, 2, 7, -12] and list(p[-8:]) == [7, 0, 12, -2, -1, -6, 0, -11]
    assert p.sum() == -406 and np.abs(p).sum() == 6854
    assert max(int(p[i]) ** 2 + int(p[i + 1]) ** 2 for i in range(0, 1024, 2)) == 338
    assert hashlib.sha256(struct.pack("<1024i", *p)).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"


def test_gaussian_taps(oracle):
    """Impulse response of the blur = outer product of the 8-bit taps [18,34,49,55,49,34,18] (A3)."""
    img = np.zeros((31, 31), np.uint8)
    img[15, 15] = 255
    out = oracle.gauss7(img).astype(np.int64)
    taps = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    expect = (255 * np.outer(taps, taps) + 32768) >> 16
    assert np.array_equal(out[12:19, 12:19], expect)
    assert out.sum() == expect.sum()
    # taps sum to 257 (not renormalised): a saturated image stays 255, a flat image gains 257^2/65536
    assert np.all(oracle.gauss7(np.full((20, 20), 255, np.uint8)) == 255)
    assert np.all(oracle.gauss7(np.full((20, 20), 100, np.uint8)) == (100 * 257 * 257 + 32768) >> 16)


def test_border_reflect101(oracle):
    img = np.arange(5 * 7, dtype=np.uint8).reshape(5, 7)
    b = oracle.border101(img, 3)
    assert np.array_equal(b, np.pad(img, 3, mode="reflect"))


def test_keypoint_sizes(oracle, stream640):
    k, _ = oracle.Extractor(1000).extract(stream640.frame(0)[0])
    expect = [31, 37, 44, 53, 64, 77, 92, 111]
    for l in range(8):
        assert np.all(k["size"][k["octave"] == l] == expect[l])
    assert np.all(k["class_id"] == -1)
    # E3': at most quota + 2 per level (first round may give 4*nIni)
    q = oracle.Extractor(1000).quotas()
    for l in range(8):
        assert (k["octave"] == l).sum() <= max(q[l] + 2, 4)
