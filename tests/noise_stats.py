"""Statistics that pin the reference's randomly seeded Perlin marble (texture/noise.rs:26-33, :45-47, :98-109).

The reference draws Perlin's 256 gradient vectors from thread_rng on every run, so no render can reproduce the marble
of assets/noise_and_textures.png value for value.  What a render with ANY table must reproduce is the marble's
statistics: its mean colour and how its contrast is spread over spatial scales (standard deviation of cell means at
cell sizes 5, 10, 20 and 35 pixels) inside the 140x140-pixel square inscribed in the sphere's disc.  The fixture
(tests/golden/reference_assets.json: marble_patch) holds those numbers of the screenshot; the tests render the scene
with K different Perlin seeds, take the envelope of each statistic over the seeds and require the screenshot inside it.

What this does and does not pin (measured with the oracle, tests/test_oracle_reference_vectors.py asserts the first
two): dropping the sine's z phase (scale 0) moves the mean from 0.43 to 0.63 and the contrast from 0.22 to 0.08 — far
outside; a wrong scale (12 instead of 4) moves the contrast ratio std5/std35 from 1.3 to 2.0 — outside; fewer than
two octaves is visible in the ratio for some seeds; octaves 3-7 (weights 1/4 ... 1/64) are below what a 107-pixel
sphere shows and stay unpinned by any reference output (they are held by the oracle's line-by-line restatement)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_assets.json")))["marble_patch"]
CELLS = (5, 10, 20, 35)


def patch_stats(q):
    """q: [600, 600, 3] quantised frame in [0, 1] -> vector of mean (3), cell stds (4 x 3), std5/std35 (3)."""
    x0, y0, n = PATCH["x0"], PATCH["y0"], PATCH["size"]
    p = q[y0:y0 + n, x0:x0 + n]
    stds = [p.reshape(n // c, c, n // c, c, 3).mean(axis=(1, 3)).std(axis=(0, 1)) for c in CELLS]
    return np.concatenate([p.mean(axis=(0, 1))] + stds + [stds[0] / stds[-1]])


def reference_stats():
    stds = [np.array(PATCH["cell_std"][str(c)]) for c in CELLS]
    return np.concatenate([np.array(PATCH["mean"])] + stds + [stds[0] / stds[-1]])


def inside_envelope(x, samples, widen=0.25):
    """x within [min - widen * range, max + widen * range] of `samples` (rows), per component."""
    lo, hi = samples.min(axis=0), samples.max(axis=0)
    pad = widen * (hi - lo)
    return (x >= lo - pad) & (x <= hi + pad)


def noise_session(host, seed, depth=None, scale=None):
    """noise_and_textures.yml under the reference's default config (600x600); the Perlin table comes from `seed`."""
    s = host.Session(os.path.join(ROOT, "scenes", "config_ref.yml"), scene=os.path.join(ROOT, "scenes", "noise_and_textures.yml"), seed=seed)
    for i in range(s.desc.n_textures):
        t = s.desc.textures[i]
        if t.kind == 3:  # RT_TEX_NOISE
            if depth is not None:
                t.depth = depth
            if scale is not None:
                t.scale = scale
    return s
