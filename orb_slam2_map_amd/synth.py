"""Deterministic synthetic RGB-D streams.

No dataset ships with the reference (its TUM association lists point at images that are not in
the tree) and there is no network, so every test and benchmark runs on these streams:

* gray  = 3 octaves of value noise (periods 64/16/4 px, amplitudes 60/40/25) + random filled
          rectangles/disks with intensity steps >= 40 (FAST corners well above quota) + per-frame
          +-3 uniform noise, clamped to u8;
* rgb   = gray with per-channel offsets (BGR byte order as the reference's KeyFrame::mImRGB);
* depth = float32 metres, 2.0 + 1.5*valuenoise(period 128), 5 % invalid (0) pixels;
* frame t = a window of one big canvas moved by an integer triangle-wave offset (<= 5 px/frame)
            so consecutive frames overlap and matching is meaningful.

Only integer / exactly-rounded float64 arithmetic is used so that the build container and the
GPU box generate identical bytes.
"""
import numpy as np

# Examples/RGB-D/TUM1.yaml:8-11 (intrinsics), :42-55 (extractor), scaled by width/640
TUM1_FX, TUM1_FY, TUM1_CX, TUM1_CY = 517.306408, 516.469215, 318.643040, 255.313989
TUM1_BF = 40.0

MARGIN = 112


def _value_noise(rng, h, w, period):
    gh, gw = h // period + 2, w // period + 2
    lat = rng.random((gh, gw))
    ys = np.arange(h) / period
    xs = np.arange(w) / period
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = ys - y0
    fx = xs - x0
    fy = fy * fy * (3 - 2 * fy)
    fx = fx * fx * (3 - 2 * fx)
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    top = a + (b - a) * fx[None, :]
    bot = c + (d - c) * fx[None, :]
    return top + (bot - top) * fy[:, None]


def _tri(k, period):
    k = k % (2 * period)
    return k if k < period else 2 * period - k


class Stream:
    """Synthetic RGB-D sequence. frame(t) -> (gray u8 HxW, rgb u8 HxWx3, depth f32 HxW)."""

    def __init__(self, width=640, height=480, seed=1234, n_shapes=None, flat_fraction=0.0):
        """flat_fraction > 0: that share of the canvas is covered by large low-texture regions (walls, table tops, a
        monitor: rectangles of 90-320 px with a gentle shading and +-1 sensor noise instead of the +-3 elsewhere), the way a
        real indoor sequence such as TUM fr1/desk has them -- the default stream (0.0) is textured everywhere, which is the
        worst case for any early-out of the FAST score.  The default stream's bytes do not depend on this option."""
        self.w, self.h, self.seed = width, height, seed
        self.flat_fraction = float(flat_fraction)
        s = width / 640.0
        self.fx, self.fy = np.float32(TUM1_FX * s), np.float32(TUM1_FY * s)
        self.cx, self.cy = np.float32(TUM1_CX * s), np.float32(TUM1_CY * s)
        self.bf = np.float32(TUM1_BF * s)
        rng = np.random.Generator(np.random.PCG64(seed))
        ch, cw = height + MARGIN, width + MARGIN
        base = 100.0 + 60 * (_value_noise(rng, ch, cw, 64) - 0.5) + 40 * (_value_noise(rng, ch, cw, 16) - 0.5) \
            + 25 * (_value_noise(rng, ch, cw, 4) - 0.5)
        if n_shapes is None:
            n_shapes = int(round(400 * (cw * ch) / (640.0 * 480.0)))
        yy, xx = np.mgrid[0:ch, 0:cw]
        for _ in range(n_shapes):
            kind = rng.integers(0, 2)
            cx = int(rng.integers(0, cw))
            cy = int(rng.integers(0, ch))
            step = float(rng.integers(40, 90)) * (1 if rng.integers(0, 2) else -1)
            if kind == 0:
                hw = int(rng.integers(4, 28))
                hh = int(rng.integers(4, 28))
                y0, y1 = max(0, cy - hh), min(ch, cy + hh)
                x0, x1 = max(0, cx - hw), min(cw, cx + hw)
                base[y0:y1, x0:x1] += step
            else:
                r = int(rng.integers(4, 20))
                y0, y1 = max(0, cy - r), min(ch, cy + r + 1)
                x0, x1 = max(0, cx - r), min(cw, cx + r + 1)
                m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
                base[y0:y1, x0:x1][m] += step
        self.canvas = base
        dep = 2.0 + 1.5 * _value_noise(rng, ch, cw, 128)
        holes = rng.random((ch, cw)) < 0.05
        dep[holes] = 0.0
        self.depth_canvas = dep.astype(np.float32)
        self.flat_mask = None
        if self.flat_fraction > 0:
            frng = np.random.Generator(np.random.PCG64([seed, 424242]))  # a generator of its own: the draws above stay as they are
            mask = np.zeros((ch, cw), bool)
            shade = 10.0 * (_value_noise(frng, ch, cw, 128) - 0.5)
            target = self.flat_fraction * ch * cw
            for _ in range(10000):
                if mask.sum() >= target:
                    break
                rw, rh = int(frng.integers(90, 321)), int(frng.integers(90, 321))
                x0, y0 = int(frng.integers(-rw // 2, cw - rw // 2)), int(frng.integers(-rh // 2, ch - rh // 2))
                x1, y1 = min(cw, x0 + rw), min(ch, y0 + rh)
                x0, y0 = max(0, x0), max(0, y0)
                level = float(frng.integers(50, 200))
                self.canvas[y0:y1, x0:x1] = level + shade[y0:y1, x0:x1]
                mask[y0:y1, x0:x1] = True
            self.flat_mask = mask

    def offset(self, t):
        return _tri(5 * t, MARGIN), _tri(3 * t + 17, MARGIN)

    def frame(self, t):
        ox, oy = self.offset(t)
        rng = np.random.Generator(np.random.PCG64([self.seed, 7919, t]))
        noise = rng.integers(-3, 4, (self.h, self.w))
        if self.flat_mask is not None:  # low-texture regions carry less sensor noise than the +-3 of the textured canvas
            noise = np.where(self.flat_mask[oy:oy + self.h, ox:ox + self.w], np.clip(noise, -1, 1), noise)
        g = self.canvas[oy:oy + self.h, ox:ox + self.w] + noise
        gray = np.clip(np.rint(g), 0, 255).astype(np.uint8)
        rgb = np.empty((self.h, self.w, 3), np.uint8)
        rgb[..., 0] = np.clip(gray.astype(np.int32) - 10, 0, 255)
        rgb[..., 1] = gray
        rgb[..., 2] = np.clip(gray.astype(np.int32) + 12, 0, 255)
        depth = np.ascontiguousarray(self.depth_canvas[oy:oy + self.h, ox:ox + self.w])
        return gray, rgb, depth

    def gray_batch(self, t0, n):
        return np.stack([self.frame(t0 + i)[0] for i in range(n)])


def random_descriptors(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)
