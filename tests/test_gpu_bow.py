"""GPU parity: vocabulary-tree transform (Frame::ComputeBoW) and the node-wise SearchByBoW against the oracle's
restatement of DBoW2 / ORBmatcher.cc:159-288, through the C ABI.  The vocabulary is synthetic (the reference's
ORBvoc.bin is a missing blob): seeded trees of the same shape, including irregular ones and stopped words."""
import numpy as np
import pytest

import scenario

pytestmark = pytest.mark.gpu


def _features(v, n, seed, p=0.06):
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(v["is_leaf"])[0]
    base = v["desc"][rng.choice(leaves, n)]
    return np.packbits(np.unpackbits(base, axis=1) ^ (rng.random((n, 256)) < p).astype(np.uint8), axis=1)


def _same(g, o):
    assert np.array_equal(g["word_id"], o["word_id"])
    assert g["weight"].tobytes() == o["weight"].tobytes()
    live = o["weight"] > 0
    assert np.array_equal(g["node_id"][live], o["node_id"][live]) and np.all(g["node_id"][~live] == -1)
    assert np.array_equal(g["bow_ids"], o["bow_ids"])
    assert g["bow_vals"].tobytes() == o["bow_vals"].tobytes(), "BowVector values must agree bit for bit (double)"
    for k in ("fv_nodes", "fv_start", "fv_items"):
        assert np.array_equal(g[k], o[k]), k


@pytest.mark.parametrize("k,L,irregular,stop,weighting,scoring",
                         [(10, 4, False, 0.0, 0, 0), (10, 3, True, 0.1, 0, 0), (20, 2, False, 0.0, 1, 1),
                          (3, 7, True, 0.05, 2, 5), (17, 3, True, 0.3, 3, 2)])
def test_bow_transform(gpu, oracle, k, L, irregular, stop, weighting, scoring):
    v = scenario.synthetic_vocabulary(k, L, 11 * k + L, irregular, stop)
    gv = gpu.ORBVocabulary(k, L, v["parent"], v["is_leaf"], v["desc"], v["weight"], weighting, scoring)
    ov = oracle.Vocabulary(k, L, v["parent"], v["is_leaf"], v["desc"], v["weight"], weighting, scoring)
    assert gv.size() == ov.size() == int(v["is_leaf"].sum())
    desc = _features(v, 1500, 5)
    desc[7] = desc[3]  # identical features share a word: addWeight accumulates
    for levelsup in (0, 1, 4, L, L + 2):
        _same(gv.transform(desc, levelsup), ov.transform(desc, levelsup))
    _same(gv.transform(desc[:1], 1), ov.transform(desc[:1], 1))
    e = gv.transform(np.zeros((0, 32), np.uint8), 4)
    assert len(e["word_id"]) == 0 and len(e["bow_ids"]) == 0 and len(e["fv_nodes"]) == 0
    gv.close()


def test_bow_transform_orbvoc_shape(gpu, oracle):
    """k = 10, L = 6 like ORBvoc (1.1 M nodes, 35 MB of node descriptors) on extracted descriptors of a synthetic
    frame; levelsup = 4 as Frame::ComputeBoW."""
    from orb_slam2_map_amd.synth import Stream
    v = scenario.synthetic_vocabulary(10, 6, 2024)
    assert len(v["parent"]) == 1 + sum(10 ** i for i in range(1, 7))
    gv = gpu.ORBVocabulary(10, 6, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(10, 6, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    st = Stream(640, 480, 1234)
    ge = gpu.ORBextractor(1000)
    _, d = ge(st.frame(3)[0])
    _same(gv.transform(d, 4), ov.transform(d, 4))
    gv.close()


def test_bow_transform_batch_device(gpu, oracle):
    torch = pytest.importorskip("torch")
    from orb_slam2_map_amd.synth import Stream
    v = scenario.synthetic_vocabulary(10, 4, 77, stop_frac=0.05)
    gv = gpu.ORBVocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    st = Stream(640, 480, 1234)
    B = 3
    ge = gpu.ORBextractor(1000, max_batch=B)
    cap = ge.max_keypoints(640, 480)
    img = torch.from_numpy(np.stack([st.frame(t)[0] for t in range(B)])).cuda()
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    nout = torch.zeros(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ge.extract_batch_device(img.data_ptr(), B, 640, 480, 640, 640 * 480, kps.data_ptr(), desc.data_ptr(), cap, nout.data_ptr(), s)
    word = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    node = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    wgt = torch.zeros((B, cap), dtype=torch.float64, device="cuda")
    gv.transform_batch_device(desc.data_ptr(), B, cap, nout.data_ptr(), 2, word.data_ptr(), wgt.data_ptr(), node.data_ptr(), s)
    torch.cuda.synchronize()
    for b in range(B):
        n = int(nout[b])
        o = ov.transform(desc[b, :n].cpu().numpy(), 2)
        assert np.array_equal(word[b, :n].cpu().numpy(), o["word_id"])
        assert wgt[b, :n].cpu().numpy().tobytes() == o["weight"].tobytes()
        live = o["weight"] > 0
        gn = node[b, :n].cpu().numpy()
        assert np.array_equal(gn[live], o["node_id"][live]) and np.all(gn[~live] == -1)
        assert np.all(word[b, n:].cpu().numpy() == -7), "rows beyond the frame's count are not touched"
    gv.close()


@pytest.mark.parametrize("levelsup,check_ori,ratio,valid_frac", [(2, True, 0.7, 0.85), (3, False, 0.9, 1.0),
                                                                  (1, True, 0.6, 0.5), (4, True, 0.7, 0.9)])
def test_search_by_bow_on_extracted_frames(gpu, oracle, levelsup, check_ori, ratio, valid_frac):
    """TrackReferenceKeyFrame (Tracking.cc:1044-1051): two frames of the synthetic stream, both transformed, then
    SearchByBoW(KF, F).  levelsup 4 on L = 4 puts everything under the root: the brute-force case."""
    from orb_slam2_map_amd.synth import Stream
    rng = np.random.default_rng(levelsup)
    v = scenario.synthetic_vocabulary(10, 4, 5 + levelsup, stop_frac=0.02)
    gv = gpu.ORBVocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(10, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    st = Stream(640, 480, 1234)
    ge = gpu.ORBextractor(1000, max_batch=2)
    (k0, k1), (d0, d1) = ge.extract_batch(np.stack([st.frame(40)[0], st.frame(41)[0]]))
    valid = (rng.random(len(k0)) < valid_frac).astype(np.uint8)
    g0, g1 = gv.transform(d0, levelsup), gv.transform(d1, levelsup)
    o0, o1 = ov.transform(d0, levelsup), ov.transform(d1, levelsup)
    _same(g0, o0), _same(g1, o1)
    ng, mg = gpu.search_by_bow(d0, k0["angle"], valid, g0["node_id"], d1, k1["angle"], g1["node_id"], 50, ratio, check_ori)
    no, mo = oracle.search_by_bow(d0, k0["angle"], valid, o0, d1, k1["angle"], o1, 50, ratio, check_ori)
    assert ng == no and np.array_equal(mg, mo), "%d vs %d matches, %d entries differ" % (ng, no, int((mg != mo).sum()))
    assert no > (20 if levelsup < 4 else 100)
    if levelsup == 4:  # one node (the root): identical to the brute-force matcher
        nb, mb = gpu.ORBmatcher(ratio, check_ori).MatchBruteForce(d0, k0["angle"], d1, k1["angle"], valid)
        keep = g1["node_id"] >= 0
        assert nb >= ng and np.array_equal(mb[keep] >= 0, mb[keep] >= 0)
    gv.close()


def test_search_by_bow_conflicts_within_nodes(gpu, oracle):
    """Many near-duplicate features under few nodes: the greedy claim order inside a node decides."""
    rng = np.random.default_rng(9)
    v = scenario.synthetic_vocabulary(4, 2, 1)
    gv = gpu.ORBVocabulary(4, 2, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    ov = oracle.Vocabulary(4, 2, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    base = _features(v, 40, 1, p=0.02)
    dk = np.repeat(base, 20, axis=0)
    dk = np.packbits(np.unpackbits(dk, axis=1) ^ (rng.random((800, 256)) < 0.01).astype(np.uint8), axis=1)
    df = np.packbits(np.unpackbits(dk[rng.permutation(800)], axis=1) ^ (rng.random((800, 256)) < 0.01).astype(np.uint8), axis=1)
    ak, af = rng.uniform(0, 360, 800).astype(np.float32), rng.uniform(0, 360, 800).astype(np.float32)
    gk, gf = gv.transform(dk, 1), gv.transform(df, 1)
    ok, of = ov.transform(dk, 1), ov.transform(df, 1)
    ng, mg = gpu.search_by_bow(dk, ak, None, gk["node_id"], df, af, gf["node_id"], 50, 0.95, False)
    no, mo = oracle.search_by_bow(dk, ak, None, ok, df, af, of, 50, 0.95, False)
    assert ng == no and np.array_equal(mg, mo)
    assert no > 100
    gv.close()
