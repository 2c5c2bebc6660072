"""Generates tests/golden/cornell_box_png_blocks.npy from the reference's own
example render, /root/reference/examples/cornell_box.png (600x600, 8-bit RGB,
written by src/main.rs:134-160 at an unrecorded sample count).

The fixture is data derived from a result file the reference ships: the mean of
every 20x20 pixel block of the PNG, shape (30, 30, 3), float32.  It is the one
artefact of the reference that pins the CPU oracle (the reference has no tests
or golden vectors).  Run from the repository root inside the build container,
where /root/reference exists; the GPU box never needs the PNG itself.
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/examples/cornell_box.png"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_box_png_blocks.npy")

img = np.asarray(Image.open(SRC).convert("RGB")).astype(np.float64)
assert img.shape == (600, 600, 3), img.shape
blocks = img.reshape(30, 20, 30, 20, 3).mean(axis=(1, 3)).astype(np.float32)
np.save(DST, blocks)
print("wrote", DST, blocks.shape, float(blocks.mean()))

# The finer fixture: per 10x10 block the mean, and the pixel noise variance estimated from horizontally adjacent
# pixels (half the mean squared difference of the five disjoint pairs of each row, which a smooth gradient barely
# enters).  The noise level is what lets a test estimate the PNG's unrecorded sample count.
DST10 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_box_png_blocks10.npz")
mean10 = img.reshape(60, 10, 60, 10, 3).mean(axis=(1, 3)).astype(np.float32)
pairs = img.reshape(600, 300, 2, 3)
half_sq = 0.5 * (pairs[:, :, 0, :] - pairs[:, :, 1, :]) ** 2          # (600, 300, 3)
noise10 = half_sq.reshape(60, 10, 60, 5, 3).mean(axis=(1, 3)).astype(np.float32)
np.savez_compressed(DST10, mean=mean10, noise_var=noise10)
print("wrote", DST10, mean10.shape, float(noise10.mean()))
