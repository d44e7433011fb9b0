"""Seeded inputs of the golden fixtures that are too large to commit as bytes: they are regenerated from a seed and
checked against a CRC stored next to the expected outputs (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import zlib

import numpy as np

import scenario


def hamming_kat_inputs(n=4096, seed=0x9E3779B9):
    """n descriptor pairs from a xorshift32 stream written out here (no dependence on a library's generator): every
    third pair differs in few bits, every third in about half, the rest are independent."""
    out = np.zeros((2, n, 32), np.uint8)
    x = seed & 0xFFFFFFFF
    words = np.zeros(2 * n * 8 + n, np.uint32)
    for i in range(len(words)):
        x ^= (x << 13) & 0xFFFFFFFF
        x ^= x >> 17
        x ^= (x << 5) & 0xFFFFFFFF
        words[i] = x
    a = words[:n * 8].reshape(n, 8)
    b = words[n * 8:2 * n * 8].reshape(n, 8).copy()
    sel = words[2 * n * 8:]
    for i in range(n):
        if i % 3 == 0:    # near duplicates: a with a few bits flipped
            b[i] = a[i]
            for k in range(int(sel[i] % 9)):
                bit = int((sel[i] >> (3 + 7 * (k % 4))) + 31 * k) % 256
                b[i, bit // 32] ^= np.uint32(1 << (bit % 32))
        elif i % 3 == 1:  # complement in the lower half
            b[i, :4] = ~a[i, :4]
            b[i, 4:] = a[i, 4:]
    out[0] = a.view(np.uint8).reshape(n, 32)
    out[1] = b.view(np.uint8).reshape(n, 32)
    return out[0], out[1]


def popcount_reference(a, b):
    """ORBmatcher::DescriptorDistance by definition: bits set in a XOR b, with Python integers."""
    return np.array([bin(int.from_bytes(bytes(x), "little") ^ int.from_bytes(bytes(y), "little")).count("1")
                     for x, y in zip(a, b)], np.int32)


def c3_projection_scenario(oracle):
    """C3's matcher problem (BASELINE.json configs[2]): frame 12 of the seeded 1280x960 stream, 2000 features, against the
    local map made of the key points of frames 11..7 (about 10 k points), mTrack* filled by the oracle's isInFrustum,
    300 key points already associated.  Built from the ORACLE's extraction, so it can be replayed without a GPU."""
    from orb_slam2_map_amd.synth import Stream
    rng = np.random.default_rng(5678)
    st = Stream(1280, 960, 1234)
    t_cur = 12
    e = oracle.Extractor(2000)
    g, _, depth = st.frame(t_cur)
    kc, dc = e.extract(g)
    sf = e.scale_factors()
    Tcw = scenario.rigid()
    ox, oy = st.offset(t_cur)
    wp, dsc, octv = [], [], []
    for t in (11, 10, 9, 8, 7):
        gp, _, dp = st.frame(t)
        k, d = oracle.Extractor(2000).extract(gp)
        px, py = st.offset(t)
        P, _ = scenario.world_points_from_prev(k, dp, (ox - px, oy - py), st, Tcw, rng)
        wp.append(P), dsc.append(d), octv.append(k["octave"])
    wp, dsc, octv = np.concatenate(wp), np.concatenate(dsc), np.concatenate(octv)
    mp = scenario.local_map(oracle, st, Tcw, wp, dsc, octv, sf, rng, obs_zero_frac=0.1, vary=True)
    frame = scenario.make_frame(oracle, kc, dc, depth, st, sf)
    k0 = np.full(frame.n, -1, np.int32)
    pre = rng.choice(frame.n, 300, replace=False)
    k0[pre[:150]] = rng.integers(0, len(wp), 150)
    k0[pre[150:]] = -2
    crc = 0
    for a in (wp, mp["desc"], mp["in_view"], mp["bad"], mp["obs_pos"], mp["level"], mp["view_cos"], mp["proj_x"], mp["proj_y"],
              mp["proj_xr"], frame.kp_x, frame.kp_y, frame.octave, frame.u_right, frame.desc, k0):
        crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    return {"stream": st, "frame": frame, "mp": mp, "world_pos": wp, "Tcw": Tcw, "k0": k0, "sf": sf, "th": 3.0,
            "nnratio": 0.8, "inputs_crc": crc}
