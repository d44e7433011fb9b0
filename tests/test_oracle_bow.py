"""Oracle BoW (oracle/orb_oracle_bow.c) against an independent, definition-level Python model of the DBoW2 code
it restates (std::map semantics with dicts, big-int Hamming): TemplatedVocabulary::transform
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1274), BowVector / FeatureVector (BowVector.cpp:34-90,
FeatureVector.cpp:32-46) and ORBmatcher::SearchByBoW(KeyFrame*, Frame&) (src/ORBmatcher.cc:159-288)."""
import math

import numpy as np
import pytest

import scenario


def _ham(a, b):
    return bin(int.from_bytes(bytes(a), "little") ^ int.from_bytes(bytes(b), "little")).count("1")


def py_transform(v, desc, levelsup, weighting=0, scoring=0):
    n_nodes = len(v["parent"])
    children = [[] for _ in range(n_nodes)]
    for i in range(1, n_nodes):
        children[v["parent"][i]].append(i)
    word_of = {}
    for i in range(1, n_nodes):
        if v["is_leaf"][i]:
            word_of[i] = len(word_of)
    bow, fv = {}, {}
    per = []
    for idx, f in enumerate(desc):
        nid_level = v["L"] - levelsup
        nid = 0 if nid_level <= 0 else None
        node, lvl = 0, 0
        while children[node]:
            lvl += 1
            best, bd = None, None
            for c in children[node]:
                d = _ham(f, v["desc"][c])
                if bd is None or d < bd:
                    best, bd = c, d
            node = best
            if lvl == nid_level:
                nid = node
        if nid is None:
            nid = node
        w = float(v["weight"][node])
        per.append((word_of[node], w, nid))
        if w > 0:
            wid = word_of[node]
            if weighting in (0, 1):
                bow[wid] = bow[wid] + w if wid in bow else w
            elif wid not in bow:
                bow[wid] = w
            fv.setdefault(nid, []).append(idx)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    must, l2 = scoring != 5, scoring == 1
    if weighting in (0, 1) and ids and not must:
        vals = [x / float(len(ids)) for x in vals]
    if must:
        norm = 0.0
        for x in vals:
            norm += x * x if l2 else abs(x)
        if l2:
            norm = math.sqrt(norm)
        if norm > 0.0:
            vals = [x / norm for x in vals]
    return per, ids, vals, {k: fv[k] for k in sorted(fv)}


def py_search_by_bow(dkf, akf, valid, fvk, df, af, fvf, th_low, ratio, check_ori):
    match = [-1] * len(df)
    hist = [[] for _ in range(30)]
    n = 0
    for node in sorted(set(fvk) & set(fvf)):
        for ik in fvk[node]:
            if not valid[ik]:
                continue
            b1, bi, b2 = 256, -1, 256
            for jf in fvf[node]:
                if match[jf] >= 0:
                    continue
                d = _ham(dkf[ik], df[jf])
                if d < b1:
                    b2, b1, bi = b1, d, jf
                elif d < b2:
                    b2 = d
            if b1 <= th_low and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                match[bi] = ik
                if check_ori:
                    rot = np.float32(akf[ik]) - np.float32(af[bi])
                    if rot < 0.0:
                        rot = np.float32(rot + np.float32(360.0))
                    # round(): half away from zero, rot >= 0 here
                    b = int(math.floor(float(np.float32(rot * (np.float32(1.0) / np.float32(30)))) + 0.5))
                    if b == 30:
                        b = 0
                    hist[b].append(bi)
                n += 1
    if check_ori:
        sizes = [len(h) for h in hist]
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i, s in enumerate(sizes):
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for i in range(30):
            if i in (i1, i2, i3):
                continue
            for j in hist[i]:
                match[j] = -1
                n -= 1
    return n, match


@pytest.mark.parametrize("k,L,irregular,stop,levelsup,weighting,scoring",
                         [(10, 3, False, 0.0, 1, 0, 0), (5, 4, True, 0.1, 2, 0, 0), (4, 3, True, 0.2, 4, 1, 1),
                          (3, 5, False, 0.1, 0, 2, 5), (6, 2, False, 0.0, 1, 3, 2)])
def test_oracle_transform_matches_definition(oracle, k, L, irregular, stop, levelsup, weighting, scoring):
    v = scenario.synthetic_vocabulary(k, L, 100 + k + L, irregular, stop)
    rng = np.random.default_rng(k * L)
    leaves = np.nonzero(v["is_leaf"])[0]
    base = v["desc"][rng.choice(leaves, 300)]
    noise = (rng.random((300, 256)) < 0.06).astype(np.uint8)
    desc = np.packbits(np.unpackbits(base, axis=1) ^ noise, axis=1)
    ov = oracle.Vocabulary(k, L, v["parent"], v["is_leaf"], v["desc"], v["weight"], weighting, scoring)
    assert ov.size() == int(v["is_leaf"].sum())
    r = ov.transform(desc, levelsup)
    per, ids, vals, fv = py_transform(v, desc, levelsup, weighting, scoring)
    assert [p[0] for p in per] == r["word_id"].tolist()
    assert [p[1] for p in per] == r["weight"].tolist()
    assert [p[2] for p in per] == r["node_id"].tolist()
    assert ids == r["bow_ids"].tolist()
    assert vals == r["bow_vals"].tolist(), "double arithmetic in the reference's order must agree bit for bit"
    assert list(fv) == r["fv_nodes"].tolist()
    for t, node in enumerate(fv):
        assert fv[node] == r["fv_items"][r["fv_start"][t]:r["fv_start"][t + 1]].tolist()
    if scoring == 0 and len(vals):
        assert abs(sum(vals) - 1.0) < 1e-12


@pytest.mark.parametrize("check_ori,ratio", [(True, 0.7), (False, 0.9)])
def test_oracle_search_by_bow_matches_definition(oracle, check_ori, ratio):
    v = scenario.synthetic_vocabulary(6, 4, 7)
    rng = np.random.default_rng(3)
    leaves = np.nonzero(v["is_leaf"])[0]
    base = v["desc"][rng.choice(leaves, 250)]
    def noisy(p):
        return np.packbits(np.unpackbits(base, axis=1) ^ (rng.random((250, 256)) < p).astype(np.uint8), axis=1)
    dkf, df = noisy(0.03), noisy(0.03)[rng.permutation(250)]
    akf = rng.uniform(0, 360, 250).astype(np.float32)
    af = ((akf[rng.permutation(250)] + rng.normal(0, 4, 250)) % 360).astype(np.float32)
    valid = (rng.random(250) < 0.85).astype(np.uint8)
    ov = oracle.Vocabulary(6, 4, v["parent"], v["is_leaf"], v["desc"], v["weight"])
    tk, tf = ov.transform(dkf, 2), ov.transform(df, 2)
    n, m = oracle.search_by_bow(dkf, akf, valid, tk, df, af, tf, 50, ratio, check_ori)
    fvk = {int(nd): tk["fv_items"][tk["fv_start"][t]:tk["fv_start"][t + 1]].tolist() for t, nd in enumerate(tk["fv_nodes"])}
    fvf = {int(nd): tf["fv_items"][tf["fv_start"][t]:tf["fv_start"][t + 1]].tolist() for t, nd in enumerate(tf["fv_nodes"])}
    pn, pm = py_search_by_bow(dkf, akf, valid, fvk, df, af, fvf, 50, ratio, check_ori)
    assert n == pn and m.tolist() == pm
    assert n > 40
