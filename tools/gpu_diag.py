"""Stage-by-stage GPU-vs-oracle diagnostic (development aid; prints, never asserts)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as O
from orb_slam2_map_amd import lib as G
from orb_slam2_map_amd.synth import Stream

def diag(w, h, nfeat, batch=2):
    print("=== %dx%d nfeat=%d batch=%d" % (w, h, nfeat, batch), flush=True)
    st = Stream(w, h, 1234)
    imgs = st.gray_batch(0, batch)
    ge = G.ORBextractor(nfeat, max_batch=batch)
    oe = O.Extractor(nfeat)
    print("quotas", ge.quotas(), oe.quotas())
    t = time.time()
    gk, gd = ge.extract_batch(imgs)
    print("gpu extract_batch %.1f ms" % ((time.time() - t) * 1e3), flush=True)
    for f in range(batch):
        ok, od = oe.extract(imgs[f])
        for l in range(8):
            raw, pitch = ge.debug_read(G.DBG_PYRAMID_PADDED, f, l)
            gp = raw.reshape(-1, pitch)
            op = oe.pyramid_level(l)
            hh, ww = op.shape
            d = (gp[:hh, :ww] != op)
            braw, _ = ge.debug_read(G.DBG_BLURRED_PADDED, f, l)
            gb = braw.reshape(-1, pitch)[19:hh - 19, 19:ww - 19]
            ob = oe.blurred_level(l)
            db = (gb != ob).sum() if ob is not None else -1
            gc, _ = ge.debug_read(G.DBG_CANDIDATES, f, l)
            oc = oe.level_candidates(l)
            oc3 = np.stack([oc['x'], oc['y'], oc['response']], 1) if len(oc) else np.zeros((0, 3), np.int32)
            gs, _ = ge.debug_read(G.DBG_SELECTED, f, l)
            os_ = oe.level_selected(l)
            os3 = np.stack([os_['x'], os_['y'], os_['response']], 1) if len(os_) else np.zeros((0, 3), np.int32)
            ceq = gc.shape == oc3.shape and (gc == oc3).all()
            seq = gs.shape == os3.shape and (gs == os3).all()
            print("f%d L%d pyr_mismatch=%d blur_mismatch=%d cand %d/%d eq=%s sel %d/%d eq=%s" % (
                f, l, d.sum(), db, len(gc), len(oc3), ceq, len(gs), len(os3), seq), flush=True)
            if not ceq and len(gc) and len(oc3):
                sg = set(map(tuple, gc)); so = set(map(tuple, oc3))
                print("   cand only-gpu", sorted(sg - so)[:5], "only-oracle", sorted(so - sg)[:5])
            if ceq and not seq:
                sg = set(map(tuple, gs)); so = set(map(tuple, os3))
                print("   sel set-equal=%s only-gpu %s only-oracle %s" % (sg == so, sorted(sg - so)[:5], sorted(so - sg)[:5]))
        n = min(len(ok), len(gk[f]))
        print("f%d nk gpu=%d oracle=%d" % (f, len(gk[f]), len(ok)))
        if n:
            for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
                a, b = gk[f][fld][:n], ok[fld][:n]
                print("   %s mismatches %d maxabs %g" % (fld, (a.view(np.uint32) != b.view(np.uint32)).sum(),
                      np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))
            print("   desc rows differing %d of %d" % ((gd[f][:n] != od[:n]).any(1).sum(), n))
    # BF match of frames 0,1
    if batch >= 2:
        m = G.ORBmatcher(0.7, True)
        t = time.time()
        ng, mg = m.MatchBruteForce(gd[0], gk[0]['angle'], gd[1], gk[1]['angle'])
        print("gpu bf %.1f ms" % ((time.time() - t) * 1e3))
        no, mo = O.match_bf(gd[0], gk[0]['angle'], gd[1], gk[1]['angle'], nnratio=0.7)
        print("bf nmatches gpu=%d oracle=%d mismatching=%d" % (ng, no, (mg != mo).sum()))
    ge.set_profiling(True)
    gk, gd = ge.extract_batch(imgs)
    print("stage ms", ge.stage_times())

if __name__ == "__main__":
    print("devices", G.device_count())
    diag(640, 480, 1000, 2)
    diag(1280, 960, 2000, 2)
    # random-descriptor BF stress (forces conflicts: low-entropy descriptors)
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    a = base[rng.integers(0, 40, 900)].copy(); b = base[rng.integers(0, 40, 1000)].copy()
    a ^= (rng.random((900, 32)) < 0.02).astype(np.uint8) << rng.integers(0, 8, (900, 32)).astype(np.uint8)
    b ^= (rng.random((1000, 32)) < 0.02).astype(np.uint8) << rng.integers(0, 8, (1000, 32)).astype(np.uint8)
    aa = rng.random(900).astype(np.float32) * 360; ab = rng.random(1000).astype(np.float32) * 360
    for ratio in (0.6, 0.9, 1.5):
        m = G.ORBmatcher(ratio, True)
        ng, mg = m.MatchBruteForce(a, aa, b, ab)
        no, mo = O.match_bf(a, aa, b, ab, nnratio=ratio)
        print("stress ratio %.1f: gpu=%d oracle=%d mismatching=%d" % (ratio, ng, no, (mg != mo).sum()))
    x = rng.integers(0, 256, (4096, 32), dtype=np.uint8); y = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    hd = G.ORBmatcher.DescriptorDistance(x, y)
    ref = np.unpackbits(x ^ y, axis=1).sum(1)
    print("hamming mismatches", (hd != ref).sum())
