"""Oracle vs independent definitional implementations (numpy / pure Python), SURVEY.md 8c."""
import numpy as np

from conftest import corners_to_array

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_definitional(img, t):
    """FAST-9/16 by the definition: corner iff 9 contiguous ring pixels are all > v+t or all < v-t;
    score = largest threshold for which the pixel is still a corner; strict 3x3 NMS."""
    h, w = img.shape
    im = img.astype(np.int32)

    def is_corner(y, x, th):
        v = im[y, x]
        ring = [im[y + dy, x + dx] for dx, dy in RING]
        for sign in (1, -1):
            flags = [(p - v) * sign > th for p in ring]
            ext = flags + flags[:8]
            run = 0
            for f in ext:
                run = run + 1 if f else 0
                if run >= 9:
                    return True
        return False

    score = np.zeros((h, w), np.int32)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(y, x, t):
                corner[y, x] = True
                s = t
                while s < 255 and is_corner(y, x, s + 1):
                    s += 1
                score[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not corner[y, x]:
                continue
            s = score[y, x]
            nb = [score[y + dy, x + dx] if corner[y + dy, x + dx] else 0
                  for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
            if all(s > n for n in nb):
                out.append((x, y, s))
    return np.array(out, np.int32).reshape(-1, 3)


def test_fast_vs_definition(oracle):
    rng = np.random.default_rng(0)
    tex = (rng.integers(0, 256, (48, 64)) // 64 * 64 + rng.integers(0, 20, (48, 64))).astype(np.uint8)
    blobs = np.full((60, 70), 90, np.int64)
    for _ in range(25):
        x, y = rng.integers(5, 60), rng.integers(5, 50)
        blobs[y:y + rng.integers(2, 9), x:x + rng.integers(2, 9)] += rng.integers(15, 120)
    blobs = np.clip(blobs, 0, 255).astype(np.uint8)
    for img in (tex, blobs):
        for t in (7, 20, 40):
            got = corners_to_array(oracle.fast(img, t))
            ref = fast_definitional(img, t)
            assert got.shape == ref.shape and np.array_equal(got, ref), (t, len(got), len(ref))


def test_fast_tiny_and_flat(oracle):
    assert len(oracle.fast(np.zeros((6, 6), np.uint8), 10)) == 0
    assert len(oracle.fast(np.full((40, 40), 77, np.uint8), 0)) == 0


def test_hamming_vs_bigint(oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    for i in range(2000):
        x = int.from_bytes(a[i].tobytes(), "little") ^ int.from_bytes(b[i].tobytes(), "little")
        assert oracle.descriptor_distance(a[i], b[i]) == bin(x).count("1")
    z = np.zeros(32, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, np.full(32, 255, np.uint8)) == 256


def test_resize_close_to_exact_bilinear(oracle):
    """Fixed-point bilinear (A2) stays within 1 grey level of the float64 definition."""
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (100, 120), dtype=np.uint8)
    dw, dh = 100, 83
    got = oracle.resize_linear(src, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (120 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (100 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 119)
    y0 = np.clip(np.floor(sy).astype(int), 0, 99)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)
    x1 = np.minimum(x0 + 1, 119)
    y1 = np.minimum(y0 + 1, 99)
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    ref = top * (1 - fy)[:, None] + bot * fy[:, None]
    assert np.max(np.abs(got - ref)) <= 1.0
    # identity resize reproduces the image
    assert np.array_equal(oracle.resize_linear(src, 120, 100), src)


def test_blur_close_to_float_gaussian(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (50, 60), dtype=np.uint8)
    got = oracle.gauss7(img).astype(np.float64)
    x = np.arange(-3, 4)
    k = np.exp(-x * x / 8.0)
    k /= k.sum()
    p = np.pad(img.astype(np.float64), 3, mode="reflect")
    tmp = sum(k[i] * p[:, i:i + 60] for i in range(7))
    ref = sum(k[i] * tmp[i:i + 50, :] for i in range(7))
    assert np.max(np.abs(got - ref)) <= 2.5  # 257/256 gain of the integer taps + rounding


def test_fast_atan2(oracle):
    rng = np.random.default_rng(4)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        a = oracle.lib().ora_fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360
        assert 0 <= a <= 360 and (abs(a - ref) < 0.02 or abs(abs(a - ref) - 360) < 0.02)
    assert oracle.lib().ora_fast_atan2(0.0, 0.0) == 0.0


def distribute_reference_python(keys, min_x, max_x, min_y, max_y, N):
    """Independent list-based transcription of the quadtree rules in Python."""
    import math
    W, H = max_x - min_x, max_y - min_y
    n_ini = int(math.floor(W / H + 0.5))
    hx = np.float32(W) / np.float32(n_ini)
    seq = [0]

    def node(ulx, uly, brx, bry, ks):
        seq[0] += 1
        return {"b": (ulx, uly, brx, bry), "k": ks, "s": seq[0]}

    nodes = [node(int(hx * np.float32(i)), 0, int(hx * np.float32(i + 1)), H, []) for i in range(n_ini)]
    for i, k in enumerate(keys):
        nodes[int(np.float32(k[0]) / hx)]["k"].append(i)
    lst = [n for n in nodes if n["k"]]

    def divide(n):
        ulx, uly, brx, bry = n["b"]
        hx_ = int(math.ceil(np.float32(brx - ulx) / 2))
        hy_ = int(math.ceil(np.float32(bry - uly) / 2))
        mx, my = ulx + hx_, uly + hy_
        ch = [[], [], [], []]
        for i in n["k"]:
            x, y = keys[i][0], keys[i][1]
            ch[(0 if x < mx else 1) + (0 if y < my else 2)].append(i)
        bs = [(ulx, uly, mx, my), (mx, uly, brx, my), (ulx, my, mx, bry), (mx, my, brx, bry)]
        return [node(*bs[q], ch[q]) for q in range(4) if ch[q]]

    finish = False
    while not finish:
        prev = len(lst)
        new_front, keep, expand_list, n_exp = [], [], [], 0
        for n in lst:
            if len(n["k"]) == 1:
                keep.append(n)
                continue
            for c in divide(n):
                new_front.insert(0, c)
                if len(c["k"]) > 1:
                    n_exp += 1
                    expand_list.append(c)
        lst = new_front + keep
        if len(lst) >= N or len(lst) == prev:
            finish = True
        elif len(lst) + 3 * n_exp > N:
            while not finish:
                prev = len(lst)
                cur = sorted(expand_list, key=lambda n: (len(n["k"]), n["s"]))
                expand_list = []
                for n in reversed(cur):
                    for c in divide(n):
                        lst.insert(0, c)
                        if len(c["k"]) > 1:
                            expand_list.append(c)
                    lst.remove(n)
                    if len(lst) >= N:
                        break
                if len(lst) >= N or len(lst) == prev:
                    finish = True
    out = []
    for n in lst:
        best = n["k"][0]
        for i in n["k"][1:]:
            if keys[i][2] > keys[best][2]:
                best = i
        out.append(keys[best])
    return np.array(out, np.int32).reshape(-1, 3)


def test_distribute_vs_python_model(oracle):
    rng = np.random.default_rng(5)
    for (W, H, n, N) in ((608, 448, 3000, 217), (608, 448, 150, 217), (501, 368, 800, 60), (1248, 928, 9000, 434),
                         (300, 100, 500, 40), (147, 102, 5, 60), (608, 448, 1, 10), (608, 448, 0, 10)):
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
        keys = np.array([(x, y, int(rng.integers(7, 120))) for x, y in sorted(pts, key=lambda p: (p[1] // 30, p[0] // 30, p[1], p[0]))],
                        np.int32).reshape(-1, 3)
        rec = np.zeros(len(keys), oracle.CORNER_DTYPE)
        if len(keys):
            rec["x"], rec["y"], rec["response"] = keys[:, 0], keys[:, 1], keys[:, 2]
        got = corners_to_array(oracle.distribute(rec, 16, 16 + W, 16, 16 + H, N))
        ref = distribute_reference_python([tuple(k) for k in keys], 16, 16 + W, 16, 16 + H, N)
        assert got.shape == ref.shape and np.array_equal(got, ref), (W, H, n, N, len(got), len(ref))
        assert len(got) <= max(N + 2, 4 * round(W / H))


def test_match_bf_vs_python(oracle):
    rng = np.random.default_rng(6)
    pool = rng.integers(0, 256, (12, 32), dtype=np.uint8)
    for trial in range(6):
        na, nb = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        a = pool[rng.integers(0, 12, na)] ^ (rng.random((na, 32)) < 0.03).astype(np.uint8)
        b = pool[rng.integers(0, 12, nb)] ^ (rng.random((nb, 32)) < 0.03).astype(np.uint8)
        ratio = [0.6, 0.9, 1.5][trial % 3]
        n, mb = oracle.match_bf(a, np.zeros(na, np.float32), b, np.zeros(nb, np.float32), nnratio=ratio,
                                check_orientation=False)
        ref = -np.ones(nb, np.int32)
        cnt = 0
        dist = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
        for i in range(na):
            b1, j1, b2 = 256, -1, 256
            for j in range(nb):
                if ref[j] >= 0:
                    continue
                d = int(dist[i, j])
                if d < b1:
                    b2, b1, j1 = b1, d, j
                elif d < b2:
                    b2 = d
            if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                ref[j1] = i
                cnt += 1
        assert n == cnt and np.array_equal(mb, ref)


def test_rotation_histogram_filter(oracle):
    """Matches whose rotation bin is not among the three largest are removed (ORBmatcher.cc:262-285)."""
    rng = np.random.default_rng(7)
    n = 200
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ang_a = np.zeros(n, np.float32)
    ang_b = np.zeros(n, np.float32)
    ang_b[:140] = 350.0     # rot = 10 deg -> bin round(10/30) = 0
    ang_b[140:180] = 300.0  # rot 60  -> bin 2
    ang_b[180:198] = 240.0  # rot 120 -> bin 4 (18 >= 0.1*140: kept as third maximum)
    ang_b[198:] = 180.0     # rot 180 -> bin 6 (fourth largest: removed)
    n_m, mb = oracle.match_bf(d, ang_a, d, ang_b, nnratio=0.9, check_orientation=True)
    assert n_m == 198 and np.all(mb[198:] == -1) and np.array_equal(mb[:198], np.arange(198))
    # ComputeThreeMaxima drops maxima below 10 % of the first (ORBmatcher.cc:1632-1641)
    ang_b[180:198] = 350.0
    ang_b[180:188] = 240.0  # bin 4 now holds 8 < 0.1*150 -> also removed
    n_m, mb = oracle.match_bf(d, ang_a, d, ang_b, nnratio=0.9, check_orientation=True)
    assert n_m == 190 and np.all(mb[180:188] == -1) and np.all(mb[198:] == -1)


def test_features_in_area_vs_bruteforce(oracle):
    rng = np.random.default_rng(8)
    n = 1500
    x = (rng.random(n) * 640).astype(np.float32)
    y = (rng.random(n) * 480).astype(np.float32)
    octv = rng.integers(0, 8, n).astype(np.int32)
    sf = oracle.Extractor(1000).scale_factors()
    f = oracle.Frame(x, y, octv, np.zeros(n, np.float32), -np.ones(n, np.float32), np.zeros((n, 32), np.uint8), 640,
                     480, sf)
    for _ in range(200):
        cx, cy, r = float(rng.random() * 700 - 30), float(rng.random() * 540 - 30), float(rng.random() * 60 + 1)
        lo, hi = int(rng.integers(-1, 8)), int(rng.integers(-1, 8))
        got = f.features_in_area(cx, cy, r, lo, hi)
        check = (lo > 0) or (hi >= 0)
        sel = (np.abs(x - np.float32(cx)) < np.float32(r)) & (np.abs(y - np.float32(cy)) < np.float32(r))
        if check:
            sel &= octv >= lo
            if hi >= 0:
                sel &= octv <= hi
        # key points outside the grid (PosInGrid false) are never returned
        px = np.round((x - 0) * f.inv_w)
        py = np.round((y - 0) * f.inv_h)
        sel &= (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
        # the cell window of the reference can miss candidates whose cell lies outside
        # [floor((x-r)*inv), ceil((x+r)*inv)] only through round() vs floor() cell assignment
        assert set(got) <= set(np.nonzero(sel)[0])
        missing = set(np.nonzero(sel)[0]) - set(got)
        for m in missing:  # every miss must be explained by the round()-based cell placement
            c0 = max(0, int(np.floor((np.float32(cx) - np.float32(r)) * f.inv_w)))
            c1 = min(63, int(np.ceil((np.float32(cx) + np.float32(r)) * f.inv_w)))
            r0 = max(0, int(np.floor((np.float32(cy) - np.float32(r)) * f.inv_h)))
            r1 = min(47, int(np.ceil((np.float32(cy) + np.float32(r)) * f.inv_h)))
            assert not (c0 <= px[m] <= c1 and r0 <= py[m] <= r1)


def test_voxel_filter_vs_numpy(oracle):
    rng = np.random.default_rng(9)
    p = rng.normal(0, 0.5, (20000, 3)).astype(np.float32)
    pts = np.zeros(len(p), oracle.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = p[:, 0], p[:, 1], p[:, 2]
    pts["rgba"] = rng.integers(0, 1 << 24, len(p))
    leaf = np.float32(0.05)
    out, ov = oracle.voxel_filter(pts, float(leaf))
    assert not ov
    inv = np.float32(1) / leaf
    ijk = np.floor(p * inv).astype(np.int64)
    mn = ijk.min(0)
    div = ijk.max(0) - mn + 1
    idx = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    uniq, start, cnt = np.unique(idx[order], return_index=True, return_counts=True)
    assert len(out) == len(uniq)
    sums = np.add.reduceat(p[order].astype(np.float64), start, axis=0) / cnt[:, None]
    got = np.stack([out["x"], out["y"], out["z"]], 1).astype(np.float64)
    assert np.max(np.abs(got - sums)) < 1e-5
    r = ((pts["rgba"] >> 16) & 255).astype(np.float64)
    rm = np.add.reduceat(r[order], start) / cnt
    assert np.all(np.abs(((out["rgba"] >> 16) & 255) - np.floor(rm)) <= 1)
    assert np.all(out["rgba"] >> 24 == 0)


def test_backproject_vs_numpy(oracle, stream640):
    _, rgb, depth = stream640.frame(0)
    fx, fy, cx, cy = (np.float32(v) for v in (stream640.fx, stream640.fy, stream640.cx, stream640.cy))
    pts = oracle.backproject(depth, rgb, float(fx), float(fy), float(cx), float(cy))
    d = depth[::3, ::3]
    m, n = np.mgrid[0:480:3, 0:640:3]
    ok = ~((d.astype(np.float64) < 0.01) | (d > 10))
    z = d[ok]
    x = (n[ok].astype(np.float32) - cx) * z / fx
    y = (m[ok].astype(np.float32) - cy) * z / fy
    assert len(pts) == ok.sum() <= 214 * 160
    assert np.array_equal(pts["x"], x) and np.array_equal(pts["y"], y) and np.array_equal(pts["z"], z)
    c = rgb[::3, ::3][ok].astype(np.uint32)
    assert np.array_equal(pts["rgba"], c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16))


def test_pose_inverse(oracle):
    import scenario
    T = scenario.rigid(0.3, -0.2, 0.5, (0.4, -0.1, 0.9))
    R, t = oracle.pose_inverse(T)
    R = R.reshape(3, 3)
    Td = T.astype(np.float64)
    assert np.allclose(R, Td[:3, :3].T, atol=1e-6) and np.allclose(t, -Td[:3, :3].T @ Td[:3, 3], atol=1e-6)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)  # re-normalised through the quaternion


def test_undistort_points_inverts_the_brown_model(oracle):
    """cv::undistortPoints is the (5-step fixed point) inverse of the forward distortion model: distorting the
    undistorted point again must land on the input pixel; zero coefficients are the identity."""
    fx, fy, cx, cy = 517.306408, 516.469215, 318.643040, 255.313989
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314])  # Examples/RGB-D/TUM1.yaml
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(120, 520, 2000), rng.uniform(90, 390, 2000)], 1).astype(np.float32)
    un = oracle.undistort_points(pts, fx, fy, cx, cy, dist).astype(np.float64)
    x, y = (un[:, 0] - np.float32(cx)) / np.float32(fx), (un[:, 1] - np.float32(cy)) / np.float32(fy)
    k1, k2, p1, p2, k3 = (float(np.float32(v)) for v in dist)
    r2 = x * x + y * y
    cd = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([xd * np.float32(fx) + np.float32(cx), yd * np.float32(fy) + np.float32(cy)], 1)
    assert np.abs(back - pts).max() < 0.05  # five iterations: a few hundredths of a pixel in the image centre region
    same = oracle.undistort_points(pts, fx, fy, cx, cy, np.zeros(5))
    assert np.abs(same - pts).max() < 1e-4


def test_distinctive_descriptor_vs_numpy(oracle):
    """Least median distance (MapPoint.cc:289-301): sorted row [floor((N-1)/2)], first minimum."""
    rng = np.random.default_rng(12)
    for n in (1, 2, 5, 8, 31):
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        bits = np.unpackbits(d, axis=1).astype(np.int32)
        dist = (bits[:, None, :] != bits[None, :, :]).sum(2)
        med = np.sort(dist, axis=1)[:, (n - 1) // 2]
        assert oracle.distinctive_descriptor(d) == int(np.argmin(med))
    assert oracle.distinctive_descriptor(np.zeros((0, 32), np.uint8)) == -1
