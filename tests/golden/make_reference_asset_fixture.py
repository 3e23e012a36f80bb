#!/usr/bin/env python3
"""Extracts the only reference-PRODUCED data points into a small JSON fixture.

The reference cannot be built or run here (Rust, no toolchain; windowed;
unseeded), so its own outputs are limited to the 600x600 `SavePng` screenshots
under /root/reference/assets/.  This script reads those PNG files as DATA with
PIL (nothing from the reference is executed) and stores

  * a grid of sky pixels of three_balls.png / noise_and_textures.png (RGBA8):
    sky pixels do not depend on the random jitter beyond rounding, so they pin
    camera basis, pixel->(u,v) mapping, Sky, sqrt gamma, tone map None and the
    truncating x255 quantisation;
  * 4x4 and 8x8 block means of three_balls.png / cornell_box.png /
    noise_and_textures.png / clown.png / emissive.png (statistical goldens: a block is
    150x150 or 75x75 pixels of a 200-spp image, so its mean carries next to no Monte Carlo
    noise; clown.yml and three_balls.yml are fully deterministic scenes, noise_and_textures.yml
    outside its randomly seeded Perlin sphere too);
  * for noise_and_textures.png additionally the CONTRAST of every 8x8 block (standard deviation of its 5x5-pixel cell
    means): with the block mean it pins the statistics of the randomly seeded Perlin marble (octaves, turbulence weight,
    sine phase), whose individual values no fixed table can reproduce;
  * for emissive.png, whose lights are brighter than the shipped emissive.yml's (the screenshot
    predates the scene file, like cornell_box.png) and whose every diffuse surface is a randomly
    seeded Noise texture: where its two lights ARE — per-row and per-column counts of saturated
    pixels — and a grid of pure-background pixels;
  * a 16x8 grid of texels of resources/images/earthmap.jpg as decoded by PIL
    (libjpeg-turbo), to pin the library's own baseline-JPEG decoder.

Run in the build container (needs /root/reference): writes
tests/golden/reference_assets.json.
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {"source": "sakarias88/racer-tracer assets/*.png and resources/images/earthmap.jpg (MIT), read with PIL",
           "sky_pixels": {}, "block_means": {}, "earthmap_texels": {}}
    for name in ("three_balls", "noise_and_textures"):
        img = np.array(Image.open(os.path.join(REF, "assets", name + ".png")).convert("RGBA"))
        assert img.shape == (600, 600, 4)
        pts = []
        ys = (0, 50, 100, 150, 190) if name == "three_balls" else (0, 20, 40)
        for y in ys:
            for x in (0, 150, 300, 450, 599):
                pts.append({"x": x, "y": y, "rgba": [int(v) for v in img[y, x]]})
        out["sky_pixels"][name] = pts
    out["block_means_8x8"] = {}
    for name in ("three_balls", "cornell_box", "noise_and_textures", "clown", "emissive"):
        img = np.array(Image.open(os.path.join(REF, "assets", name + ".png")).convert("RGB")).astype(np.float64) / 255.0
        assert img.shape == (600, 600, 3)
        bm = img.reshape(4, 150, 4, 150, 3).mean(axis=(1, 3))
        out["block_means"][name] = np.round(bm, 5).tolist()
        bm8 = img.reshape(8, 75, 8, 75, 3).mean(axis=(1, 3))
        out["block_means_8x8"][name] = np.round(bm8, 5).tolist()
    # texture CONTRAST inside a block, for the randomly seeded marble of noise_and_textures.png: the standard deviation,
    # per channel, of the 15x15 means of 5x5-pixel cells of each 75x75 block (cell means carry 1/5 of a pixel's Monte
    # Carlo noise, so what is left is the texture's own variation)
    img = np.array(Image.open(os.path.join(REF, "assets", "noise_and_textures.png")).convert("RGB")).astype(np.float64) / 255.0
    cells = img.reshape(8, 15, 5, 8, 15, 5, 3).mean(axis=(2, 5))           # [block y, cell y, block x, cell x, rgb]
    out["block_cell_stds_8x8"] = {"noise_and_textures": np.round(cells.std(axis=(1, 3)), 5).tolist()}
    # ... and of the 140x140-pixel square inscribed in the marble sphere's disc (projected centre (319.5, 273.8), radius
    # 107 px; the square's half diagonal is 99): mean and contrast at four cell sizes — how the variation is spread
    # over spatial frequencies is what the octave count and the noise scale decide
    x0, y0, size = 250, 204, 140
    patch = img[y0:y0 + size, x0:x0 + size]
    out["marble_patch"] = {"x0": x0, "y0": y0, "size": size, "mean": np.round(patch.mean(axis=(0, 1)), 5).tolist(),
                           "cell_std": {str(c): np.round(patch.reshape(size // c, c, size // c, c, 3).mean(axis=(1, 3)).std(axis=(0, 1)), 5).tolist()
                                        for c in (5, 10, 20, 35)}}
    em = np.array(Image.open(os.path.join(REF, "assets", "emissive.png")).convert("RGB")).astype(int)
    lit = (em >= 250).all(axis=-1)            # both lights saturate in the screenshot: (255, 255, 254) / (255, 255, 255)
    black = (em == 0).all(axis=-1)
    grid = [(x, y) for y in range(12, 600, 25) for x in range(12, 600, 25)]
    out["emissive_layout"] = {
        "lit_threshold": 250,
        "lit_pixels": int(lit.sum()),
        "lit_row_counts": lit.sum(axis=1).tolist(),
        "lit_col_counts": lit.sum(axis=0).tolist(),
        # grid points whose whole 9x9 neighbourhood is background in the screenshot
        "black_grid": [[x, y] for x, y in grid if black[y - 4:y + 5, x - 4:x + 5].all()],
    }
    earth = np.array(Image.open(os.path.join(REF, "resources", "images", "earthmap.jpg")).convert("RGBA"))
    out["earthmap_texels"] = {"width": int(earth.shape[1]), "height": int(earth.shape[0]), "step": 64,
                              "offset": 17,
                              "rgba": earth[17::64, 17::64].reshape(-1, 4).tolist(),
                              "sum_rgb": [int(v) for v in earth[..., :3].reshape(-1, 3).sum(axis=0)]}
    with open(os.path.join(HERE, "reference_assets.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote reference_assets.json")


if __name__ == "__main__":
    main()
