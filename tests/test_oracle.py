"""The CPU oracle against the only things that can pin it without a Rust
toolchain: Philox known-answer vectors, the reference's own example render, and
its own internal consistency (BVH vs list traversal, f32 vs f64)."""
import os

import numpy as np

from conftest import GOLDEN, cornell_ray_batch
from raytracing_amd import make_params


def test_philox_known_answers(orc):
    # Known-answer vectors of Philox4x32-10 (Random123 kat_vectors; Salmon et al., SC'11).
    assert list(orc.philox([0, 0, 0, 0], [0, 0])) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_oracle_matches_reference_example_render(orc, cornell):
    """examples/cornell_box.png is the one output of the reference that exists.
    Its 20x20 block means (tests/golden/cornell_box_png_blocks.npy) must be
    reproduced by the f64 oracle after the reference's tonemap.  The PNG's sample
    count is unknown (its noise level suggests about 1000), so the comparison is
    on block means and the tolerance is in 8-bit levels: sqrt-then-quantise is
    biased by noise, more at our 64 spp than in the PNG."""
    blocks_ref = np.load(os.path.join(GOLDEN, "cornell_box_png_blocks.npy")).astype(np.float64)
    spp = 64
    params = make_params(600, 600, spp, seed=0x5EED0001)
    s, _ = orc.render(cornell.flat, cornell.camera.c, params, mode=orc.ORC_F64, n_threads=8)
    img = orc.tonemap(s, spp).astype(np.float64)
    blocks = img.reshape(30, 20, 30, 20, 3).mean(axis=(1, 3))
    diff = blocks - blocks_ref
    assert np.abs(diff).max() < 4.0, np.abs(diff).max()
    assert np.sqrt((diff ** 2).mean()) < 1.0
    assert abs(diff.mean()) < 0.75
    # the black frame around the box opening (vfov 40 from z=-800 sees past it onto the black sky)
    assert img[:10].max() == 0.0 and img[:, :10].max() == 0.0


def test_oracle_matches_reference_example_render_at_fine_blocks(orc, cornell):
    """The same PNG at 10x10 blocks, with its sample count estimated from its own
    noise (about 800: the reference's thread split makes it a multiple of 8) and the
    tonemap's noise bias allowed for, plus the colour of the two painted walls,
    which pins the albedos and the interreflection independently of the exposure."""
    from conftest import reference_png_fine
    spp = 64
    s, s2 = orc.render(cornell.flat, cornell.camera.c, make_params(600, 600, spp, seed=0x5EED0001), mode=orc.ORC_F64, n_threads=8)
    n_hat, diff, ratios = reference_png_fine(s, s2, spp, orc.tonemap(s, spp))
    assert 400 < n_hat < 1600, n_hat
    assert np.abs(diff).max() < 9.0 and np.sqrt((diff ** 2).mean()) < 1.3 and abs(diff.mean()) < 0.5, (
        np.abs(diff).max(), np.sqrt((diff ** 2).mean()), diff.mean())
    assert ratios["blocks"][0] > 30 and ratios["blocks"][1] > 300
    for key in ("green_wall_g_over_r", "red_wall_r_over_g"):
        png, ours = ratios[key]
        assert abs(ours / png - 1.0) < 0.015, (key, png, ours)


def test_oracle_bvh_equals_list_traversal(orc, cornell):
    """BvhNode::hit (bvh.rs:197-228) and a flat HitableList over the same
    leaves (main.rs:109, commented alternative) must find the same nearest hit
    except where the BVH's own box tests reject a ray the list would accept
    (rays that touch a box face with zero width, bvh.rs:56)."""
    o, d = cornell_ray_batch(cornell, 20000, seed=11)
    pb, tb = orc.trace_closest(cornell.flat, o, d, mode=orc.ORC_F64, traversal=orc.ORC_TRAVERSE_BVH)
    pl, tl = orc.trace_closest(cornell.flat, o, d, mode=orc.ORC_F64, traversal=orc.ORC_TRAVERSE_LIST)
    differ = pb != pl
    # the list can only ever see MORE than the BVH
    assert np.all((pb[differ] == -1) | (tl[differ] <= tb[differ]))
    assert differ.mean() < 0.01
    same = ~differ & (pb >= 0)
    assert np.array_equal(tb[same], tl[same])


def test_oracle_f32_tracks_f64(orc, cornell):
    """fp32 restatements against the f64 reference arithmetic: same primitive
    except at edges, t within 1e-5 relative."""
    n = 30000
    o, d = cornell_ray_batch(cornell, n, seed=12)
    n_generic = 2 * (n // 3) + n // 12  # camera, interior and axis-aligned rays; the rest are aimed at edges
    p64, t64 = orc.trace_closest(cornell.flat, o, d, mode=orc.ORC_F64)
    for mode in (orc.ORC_F32, orc.ORC_F32_FUSED):
        p32, t32 = orc.trace_closest(cornell.flat, o, d, mode=mode)
        same = p32 == p64
        # rays in general position: rounding never changes the primitive
        assert same[:n_generic].mean() > 0.9999
        # rays aimed within ulps of box corners and rect edges: rounding decides which side they pass
        assert same[n_generic:].mean() > 0.98
        hit = same & (p64 >= 0)
        assert_t_close(t32[hit], t64[hit], d[hit])


def assert_t_close(t32, t64, d):
    """fp32 t against f64 t.  The bar is 1e-5 relative.  It is met by all but a
    few rays in ten thousand: those whose hit is closer than a few units, or
    that graze the surface, where the fp32 rounding of the coordinates themselves
    (ulp(555) = 6.1e-5, amplified by 1/cos of the incidence angle) exceeds 1e-5
    of the distance.  Those stay within 1e-3 relative and 1e-3 units."""
    rel = np.abs(t32 - t64) / np.maximum(np.abs(t64), 1e-30)
    assert (rel > 1e-5).mean() < 1e-3, (rel > 1e-5).mean()
    err = np.abs(t32 - t64) * np.linalg.norm(d, axis=1)
    loose = rel > 1e-5
    assert np.all((rel[loose] < 1e-3) & (err[loose] < 1e-3)), (rel[loose], err[loose])


def test_oracle_thread_split_modes_agree(orc, cornell):
    """Rendering rows across threads or samples across threads (main.rs:70-133)
    gives the same sums: each (pixel, sample) has its own Philox counter."""
    params = make_params(48, 48, 16, seed=7)
    a, a2 = orc.render(cornell.flat, cornell.camera.c, params, n_threads=3, split_samples=False)
    b, b2 = orc.render(cornell.flat, cornell.camera.c, params, n_threads=4, split_samples=True)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-12)
    assert np.allclose(a2, b2, rtol=1e-12, atol=1e-12)


def test_oracle_sample_ranges_add(orc, cornell):
    """Two sample ranges rendered separately sum to the whole (the property the
    multi-GPU sample-range split relies on)."""
    cam = cornell.camera.c
    whole, _ = orc.render(cornell.flat, cam, make_params(32, 32, 12, seed=9))
    lo, _ = orc.render(cornell.flat, cam, make_params(32, 32, 5, sample_begin=0, seed=9))
    hi, _ = orc.render(cornell.flat, cam, make_params(32, 32, 7, sample_begin=5, seed=9))
    assert np.allclose(whole, lo + hi, rtol=1e-12, atol=1e-12)


def test_oracle_tonemap(orc):
    # main.rs:136-155: sqrt, clamp high only, 255.99 scale, truncation; NaN -> 0
    sums = np.array([[[0.0, 4.0, 1.0], [0.25 * 4, 100.0, float("nan")]]])
    img = orc.tonemap(sums, 4)
    assert img.tolist() == [[[0, 255, 127], [127, 255, 0]]]


def test_restatement_reproduces_its_stored_vectors():
    """tests/golden/oracle_vectors.npz holds the f64 restatement's outputs on fixed
    inputs (made by tests/golden/make_oracle_vectors.py): the flattened Cornell box,
    2048 nearest hits, two small images.  Recomputed here they must come out the
    same, so a change to the restatement or the flattener cannot slip by.  (They pin
    the restatement against itself; nothing of the reference's exists to pin it.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_vectors", os.path.join(GOLDEN, "make_oracle_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.build()
    want = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    for k in ("nodes", "prims", "ray_o", "ray_d", "ray_prim"):
        assert np.array_equal(now[k], want[k]), k
    for k in ("ray_t", "ray_normal", "ray_p"):
        assert np.allclose(now[k], want[k], rtol=1e-12, atol=0.0, equal_nan=True), k
    # sin, cos and log come from the C library, whose last bit may differ from one CPU to another; one such bit can send a
    # path elsewhere, so a pixel in twenty may differ while the rest agree to rounding
    for k in ("cornell_sum", "cornell_sumsq", "smoke_sum", "smoke_sumsq"):
        close = np.isclose(now[k], want[k], rtol=1e-9, atol=0.0)
        assert close.mean() > 0.95, (k, close.mean())
        assert abs(now[k].sum() - want[k].sum()) < 0.05 * abs(want[k].sum()), k
