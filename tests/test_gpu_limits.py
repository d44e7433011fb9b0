"""The configurations the GPU path refuses ("hard limits"): each returns ORBGPU_EINVAL with a message
of its own through orbgpu_last_error_string -- a status, never a surprise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_too_many_features_on_one_level(gpu):
    """The quadtree keeps a level's node list in LDS: > ~2 300 key points on ONE level are refused
    (an 8-level extractor reaches that at ~10 000 features)."""
    img = np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8)
    ge = gpu.ORBextractor(60000, 1.2, 8, 20, 7)
    with pytest.raises(RuntimeError, match="nfeatures too large for the quadtree kernel"):
        ge(img)
    # the same image with the reference's own setting is fine afterwards (the handle stays usable after a refusal)
    k, d = gpu.ORBextractor(1000)(img)
    assert len(k) > 500


def test_image_side_limit_and_too_small_levels(gpu):
    big = np.zeros((64, 4200), np.uint8)
    with pytest.raises(RuntimeError, match="outside the supported range"):
        gpu.ORBextractor(500, 1.2, 1, 20, 7)(big)
    small = np.zeros((100, 100), np.uint8)  # level 3 of 8 is 58 px wide: no room for a 30-px FAST cell inside the 16-px border
    with pytest.raises(RuntimeError, match="pyramid level 3 too small"):
        gpu.ORBextractor(500)(small)


def test_portrait_image_whose_aspect_rounds_to_zero(gpu):
    """nIni = round(width / height) = 0: the reference divides by zero there (ORBextractor.cc:542-544)."""
    img = np.zeros((1000, 300), np.uint8)
    with pytest.raises(RuntimeError, match="aspect ratio unsupported"):
        gpu.ORBextractor(500, 1.2, 2, 20, 7)(img)


def test_brute_force_row_limit(gpu):
    d = np.zeros((4097, 32), np.uint8)
    a = np.zeros(4097, np.float32)
    with pytest.raises(RuntimeError, match=r"na/nb must be in \[0,4096\]"):
        gpu.ORBmatcher(0.7, True).MatchBruteForce(d, a, d[:10], a[:10])
    with pytest.raises(RuntimeError, match=r"cap in \[1,4096\]"):
        gpu.BatchMatcher(2, 5000)


def test_projection_frame_limit(gpu):
    n = 16385
    z = np.zeros(n, np.float32)
    with pytest.raises(RuntimeError, match="frame key point count out of range"):
        f = gpu.Frame(z, z, np.zeros(n, np.int32), z, z, np.zeros((n, 32), np.uint8), 640, 480, np.ones(8, np.float32))
        gpu.DeviceFrame().upload(f)
