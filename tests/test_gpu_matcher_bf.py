"""GPU parity: Hamming distance and the brute-force matcher vs the oracle. Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def low_entropy(rng, n, pool, flip=0.02):
    base = pool[rng.integers(0, len(pool), n)].copy()
    noise = (rng.random((n, 32)) < flip).astype(np.uint8) << rng.integers(0, 8, (n, 32)).astype(np.uint8)
    return base ^ noise


def test_hamming_vs_popcount(gpu):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    a[:3] = 0
    b[0], b[1], b[2] = 0, 255, 1
    ref = np.unpackbits(a ^ b, axis=1).sum(1)
    out = gpu.ORBmatcher.DescriptorDistance(a, b)
    assert np.array_equal(out, ref)
    assert out[1] == 256


def test_bf_on_extracted_frames(gpu, oracle, stream640):
    """C2: consecutive synthetic frames, extract on the GPU, match on GPU vs oracle."""
    ge = gpu.ORBextractor(1000, max_batch=4)
    imgs = stream640.gray_batch(0, 4)
    k, d = ge.extract_batch(imgs)
    for ratio, ori in ((0.7, True), (0.9, True), (0.6, False)):
        m = gpu.ORBmatcher(ratio, ori)
        for f in range(3):
            ng, mg = m.MatchBruteForce(d[f], k[f]["angle"], d[f + 1], k[f + 1]["angle"])
            no, mo = oracle.match_bf(d[f], k[f]["angle"], d[f + 1], k[f + 1]["angle"], nnratio=ratio,
                                     check_orientation=ori)
            assert ng == no and np.array_equal(mg, mo), "ratio %.1f ori %s frame %d" % (ratio, ori, f)
            assert ng > 100, "the synthetic stream must produce real matches"


@pytest.mark.parametrize("seed,na,nb,ratio,th", [(5, 900, 1000, 0.6, 50), (6, 1000, 900, 0.9, 50),
                                                (7, 1016, 1016, 1.5, 80), (8, 50, 2000, 2.0, 256),
                                                (9, 2016, 2016, 1.2, 100)])
def test_bf_conflict_stress(gpu, oracle, seed, na, nb, ratio, th):
    """Few distinct descriptors -> many A rows compete for the same B row, so the greedy claim
    order (ORBmatcher.cc:209-210,232) decides the result; the sweeps must reach the same fixpoint."""
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    a, b = low_entropy(rng, na, pool), low_entropy(rng, nb, pool)
    aa = (rng.random(na) * 360).astype(np.float32)
    ab = (rng.random(nb) * 360).astype(np.float32)
    valid = (rng.random(na) > 0.1).astype(np.uint8)
    m = gpu.ORBmatcher(ratio, True)
    ng, mg = m.MatchBruteForce(a, aa, b, ab, valid_a=valid, th_low=th)
    no, mo = oracle.match_bf(a, aa, b, ab, valid_a=valid, th_low=th, nnratio=ratio)
    assert ng == no and np.array_equal(mg, mo)


def test_bf_adversarial_chain(gpu, oracle):
    """A chain of dependent claims: A row i prefers B row i-1 unless it is taken -> needs ~n sweeps,
    far more than the typical 2-7."""
    n = 120
    rng = np.random.default_rng(2)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a = np.zeros((n, 32), np.uint8)
    a[0] = b[0]
    for i in range(1, n):
        # equidistant (1 bit each) to b[i-1] and b[i]... tie -> lower index wins, so row i takes b[i-1]
        # only if it is free; build descriptor half-way between the two
        a[i] = b[i - 1]
        diff = np.unpackbits(b[i - 1] ^ b[i])
        idx = np.nonzero(diff)[0]
        half = idx[: len(idx) // 2]
        bits = np.unpackbits(a[i])
        bits[half] ^= 1
        a[i] = np.packbits(bits)
    ang = np.zeros(n, np.float32)
    m = gpu.ORBmatcher(10.0, False)
    ng, mg = m.MatchBruteForce(a, ang, b, ang, th_low=256)
    no, mo = oracle.match_bf(a, ang, b, ang, th_low=256, nnratio=10.0, check_orientation=False)
    assert ng == no and np.array_equal(mg, mo)


def test_bf_edge_cases(gpu, oracle):
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    ang = np.zeros(10, np.float32)
    m = gpu.ORBmatcher(0.7, True)
    n, mb = m.MatchBruteForce(a[:0], ang[:0], a, ang)
    assert n == 0 and np.all(mb == -1)
    n, mb = m.MatchBruteForce(a, ang, a[:0], ang[:0])
    assert n == 0 and len(mb) == 0
    # identical sets: every row matches itself (distance 0 < ratio * second)
    n, mb = m.MatchBruteForce(a, ang, a, ang)
    no, mo = oracle.match_bf(a, ang, a, ang, nnratio=0.7)
    assert n == no == 10 and np.array_equal(mb, mo) and np.array_equal(mb, np.arange(10))
    # single B row: second distance stays 256
    n, mb = m.MatchBruteForce(a, ang, a[:1], ang[:1])
    no, mo = oracle.match_bf(a, ang, a[:1], ang[:1], nnratio=0.7)
    assert n == no and np.array_equal(mb, mo)


def test_bf_more_rows_than_candidates_at_threshold_256(gpu, oracle):
    """Once every B row is taken a later A row has no candidate left: best distance stays 256 and the best index -1.  The
    reference's TH_LOW = 50 rejects that row before it looks at the index; with th_low = 256 (this entry point takes the
    threshold as a parameter) the row must simply not match (found by tools/fuzz_bf.py: the oracle wrote match[-1])."""
    rng = np.random.default_rng(14)
    base = rng.integers(0, 256, (100, 32), dtype=np.uint8)
    a, b = base[rng.integers(0, 100, 683)], base[rng.integers(0, 100, 86)]
    aa, ab = (rng.random(683) * 360).astype(np.float32), (rng.random(86) * 360).astype(np.float32)
    for ori in (True, False):
        ng, mg = gpu.ORBmatcher(10.0, ori).MatchBruteForce(a, aa, b, ab, th_low=256)
        no, mo = oracle.match_bf(a, aa, b, ab, th_low=256, nnratio=10.0, check_orientation=ori)
        assert ng == no and np.array_equal(mg, mo)
        assert ori or ng == 86  # every B row ends up taken


def test_bf_batched_device_path(gpu, oracle, stream640):
    """The device-resident batched entry point the benchmark times (pairs of consecutive frames)."""
    torch = pytest.importorskip("torch")
    B = 6
    imgs = torch.from_numpy(stream640.gray_batch(20, B)).cuda()
    ge = gpu.ORBextractor(1000, max_batch=B)
    cap = ge.max_keypoints(640, 480)
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ge.extract_batch_device(imgs.data_ptr(), B, 640, 480, 640, 640 * 480, kps.data_ptr(), desc.data_ptr(), cap,
                            nout.data_ptr(), stream)
    bm = gpu.BatchMatcher(B - 1, cap)
    match_b = torch.full((B - 1, cap), -7, dtype=torch.int32, device="cuda")
    nm = torch.zeros(B - 1, dtype=torch.int32, device="cuda")
    bm.match(B - 1, cap, desc.data_ptr(), kps.data_ptr() + 12, None, nout.data_ptr(), desc.data_ptr() + cap * 32,
             kps.data_ptr() + cap * 28 + 12, nout.data_ptr() + 4, 28, 50, 0.7, True, match_b.data_ptr(),
             nm.data_ptr(), stream)
    torch.cuda.synchronize()
    n = nout.cpu().numpy()
    k = kps.cpu().numpy()
    d = desc.cpu().numpy()
    mb = match_b.cpu().numpy()
    sweeps = bm.last_sweeps(B - 1)
    assert np.all(sweeps > 0) and np.all(sweeps <= 64), sweeps
    for p in range(B - 1):
        no, mo = oracle.match_bf(d[p, :n[p]], k[p, :n[p], 3], d[p + 1, :n[p + 1]], k[p + 1, :n[p + 1], 3], nnratio=0.7)
        assert int(nm[p]) == no
        assert np.array_equal(mb[p, :n[p + 1]], mo)
        assert np.all(mb[p, n[p + 1]:] == -1)


def test_bf_self_match_is_identity_at_full_size(gpu):
    """Property at the largest supported row count (4096 x 4096, no oracle run needed): matching a descriptor set
    against itself pairs every row with itself (distance 0 passes any ratio test against a non-zero second
    distance); rows with an exact duplicate are the only ones allowed to differ."""
    rng = np.random.default_rng(21)
    n = 4096
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d[100] = d[7]  # one exact duplicate pair: the second distance is 0 for both, so neither may match
    ang = rng.uniform(0, 360, n).astype(np.float32)
    nm, mb = gpu.ORBmatcher(0.7, True).MatchBruteForce(d, ang, d, ang)
    expect = np.arange(n)
    expect[[7, 100]] = -1
    assert np.array_equal(mb, expect) and nm == n - 2


def test_distinctive_descriptors(gpu, oracle):
    """MapPoint::ComputeDistinctiveDescriptors batched over map points: group sizes 0, 1, 2 (even: lower median),
    odd / even sizes, many ties (duplicated descriptors), one group beyond a workgroup's 256 rows."""
    rng = np.random.default_rng(8)
    groups = []
    for n in (0, 1, 2, 3, 4, 7, 16, 33, 64, 100, 257, 600):
        base = rng.integers(0, 256, (1, 32), dtype=np.uint8)
        d = np.repeat(base, n, 0)
        if n:
            flips = rng.integers(0, 256, (n, 40))
            nflip = rng.integers(0, 40, n)
            for i in range(n):
                for b in flips[i, :nflip[i]]:
                    d[i, b // 8] ^= np.uint8(1 << (b % 8))
        groups.append(d)
    dup = groups[6].copy()
    dup[5] = dup[2]
    dup[9] = dup[2]      # exact duplicates: equal medians, the first index must win
    groups.append(dup)
    got = gpu.distinctive_descriptors(groups)
    want = [oracle.distinctive_descriptor(g) for g in groups]
    assert list(got) == want, (list(got), want)
    assert got[0] == -1 and got[1] == 0
    assert len(gpu.distinctive_descriptors([])) == 0


def test_argument_errors_are_statuses_not_crashes(gpu):
    """Bad arguments come back as ORBGPU_EINVAL with a message (the shim turns them into exceptions); nothing is
    launched and nothing crashes."""
    import ctypes as C
    L = gpu.lib()
    a = np.zeros((4, 32), np.uint8)
    out = np.zeros(4, np.int32)
    n = C.c_int32()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    assert L.orbgpu_hamming256(None, p(a), 4, p(out), 0) == gpu.EINVAL
    assert L.orbgpu_match_bf(p(a), None, None, 5000, p(a), None, 4, 50, 0.7, 0, p(out), C.byref(n), 0) == gpu.EINVAL  # na > 4096
    assert L.orbgpu_match_bf(p(a), None, None, 4, p(a), None, 4, 50, 0.7, 1, p(out), C.byref(n), 0) == gpu.EINVAL     # angles missing
    assert L.orbgpu_distinctive_descriptors(1, None, p(a), p(out), 0) == gpu.EINVAL
    off = np.array([0, 3000], np.int32)  # more descriptors in a group than supported
    assert L.orbgpu_distinctive_descriptors(1, p(off), p(a), p(out), 0) == gpu.EINVAL
    assert b"2048" in L.orbgpu_last_error_string()
    h = C.c_void_p()
    assert L.orbgpu_matcher_create(0, 0, 100, C.byref(h)) == gpu.EINVAL
    assert L.orbgpu_cloud_create(0.0, 0, C.byref(h)) == gpu.EINVAL  # resolution must be positive


def test_hamming_kat_on_gpu(gpu):
    """The committed 4096-pair known-answer test of DescriptorDistance (tests/golden/hamming_kat_4096.npz) on the device."""
    import os
    import zlib
    import golden_scenarios as GS
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hamming_kat_4096.npz"))
    a, b = GS.hamming_kat_inputs()
    assert zlib.crc32(a.tobytes() + b.tobytes()) == int(g["inputs_crc"][0])
    assert np.array_equal(gpu.ORBmatcher.DescriptorDistance(a, b), g["dist"].astype(np.int32))
