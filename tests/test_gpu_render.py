"""Parity of the HIP sample loop (rt_render through the C ABI) against the CPU
oracle.  Monte Carlo sums cannot be compared bit for bit with the reference
(its random numbers are OS seeded), so the bars are statistical, as
BASELINE.json's north_star states: per channel within 3 standard errors,
RMSE below a stated threshold, at equal sample counts."""
import numpy as np
import pytest

from raytracing_amd.distributed import shard_tile_mask, shard_width
from raytracing_amd import _lib, make_params

pytestmark = pytest.mark.gpu


def _z_scores(sum_a, sumsq_a, sum_b, sumsq_b, n):
    """Per pixel and channel: difference of the two means over the standard
    error of that difference, from each side's own sample variance."""
    ma, mb = sum_a / n, sum_b / n
    va = np.maximum(sumsq_a - n * ma * ma, 0.0) / (n - 1)  # sample variance of one sample
    vb = np.maximum(sumsq_b - n * mb * mb, 0.0) / (n - 1)
    se = np.sqrt(va / n + vb / n)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (ma - mb) / se
    return ma, mb, se, z


def test_render_same_stream_tracks_oracle(gpu, orc, cornell):
    """Same Philox seed on both sides: the device draws exactly the oracle's
    random numbers, so the only differences are fp32 rounding in shading and the
    hardware sin/cos.  Nearly every pixel sum then agrees to fp32 accuracy; the
    few that do not are paths that a last-bit difference sent to another surface."""
    gpu.upload_scene(cornell.flat)
    nx = ny = 64
    spp = 32
    params = make_params(nx, ny, spp, seed=0xC0FFEE)
    s, s2, stats = gpu.render(cornell.camera, params, want_sumsq=True, want_stats=True)
    os_, os2, ostats = orc.render(cornell.flat, cornell.camera.c, params, mode=orc.ORC_F32_FUSED, want_stats=True)
    assert stats.paths == nx * ny * spp == ostats.paths
    assert np.isfinite(s).all() and (s >= 0).all()
    close = np.isclose(s, os_, rtol=2e-4, atol=1e-4)
    assert close.mean() > 0.98, close.mean()
    # whole-image energy agrees tightly
    assert abs(s.sum() - os_.sum()) / os_.sum() < 2e-3
    # the device ends a path whose weight reached zero; the oracle follows it to the end
    assert stats.rays <= ostats.rays and stats.rays > 0.8 * ostats.rays
    assert stats.shade_events == stats.light_tests
    assert stats.kernel_ms > 0


def test_render_statistical_parity_independent_seeds(gpu, orc, cornell):
    """Independent seeds, equal spp, f64 oracle (the reference's arithmetic):
    per-channel differences within 3 standard errors for all but the share of
    pixels a normal distribution puts beyond 3 sigma (0.27 %; 1 % allowed because
    per-pixel variance is itself estimated from 256 heavy-tailed samples), no mean
    bias, and RMSE of the difference consistent with pure Monte Carlo noise."""
    gpu.upload_scene(cornell.flat)
    nx = ny = 64
    spp = 256
    s, s2 = gpu.render(cornell.camera, make_params(nx, ny, spp, seed=1111), want_sumsq=True)
    os_, os2 = orc.render(cornell.flat, cornell.camera.c, make_params(nx, ny, spp, seed=2222), mode=orc.ORC_F64)
    ma, mb, se, z = _z_scores(s.astype(np.float64), s2.astype(np.float64), os_, os2, spp)
    valid = se > 0
    assert valid.mean() > 0.7
    zz = z[valid]
    assert (np.abs(zz) > 3).mean() < 0.01, (np.abs(zz) > 3).mean()
    assert abs(zz.mean()) < 0.05, zz.mean()
    assert 0.9 < zz.std() < 1.1, zz.std()
    # RMSE threshold: the difference image's RMSE must not exceed 1.15x what the two noise levels predict
    rmse = np.sqrt(((ma - mb)[valid] ** 2).mean())
    predicted = np.sqrt((se[valid] ** 2).mean())
    assert rmse < 1.15 * predicted, (rmse, predicted)
    # pixels neither side ever lit (the black frame) are exactly zero on both
    assert np.all(ma[~valid] == mb[~valid])
    # image means agree to 1 %
    assert abs(ma.mean() - mb.mean()) / mb.mean() < 0.01


def test_render_sample_ranges_and_repeatability(gpu, cornell):
    """Sample-range sharding: two ranges add up to the whole (to fp32 summation
    order), and the same call twice gives the same bits."""
    gpu.upload_scene(cornell.flat)
    cam = cornell.camera
    whole = gpu.render(cam, make_params(72, 40, 24, seed=5))
    again = gpu.render(cam, make_params(72, 40, 24, seed=5))
    assert np.array_equal(whole, again)
    lo = gpu.render(cam, make_params(72, 40, 10, sample_begin=0, seed=5))
    hi = gpu.render(cam, make_params(72, 40, 14, sample_begin=10, seed=5))
    assert np.allclose(whole, lo + hi, rtol=1e-5, atol=1e-5)
    assert not np.array_equal(whole, gpu.render(cam, make_params(72, 40, 24, seed=6)))


def test_render_pixel_tile_shards_partition_the_image(gpu, cornell):
    """Pixel-tile sharding: each 8x8 tile is rendered by exactly one shard (the one
    include/rt_abi.h's numbering names), the others leave it zero, and the shards
    add up to the unsharded image."""
    gpu.upload_scene(cornell.flat)
    cam = cornell.camera
    nx, ny, spp = 200, 133, 8  # ragged: the last row of tiles is cut
    whole = gpu.render(cam, make_params(nx, ny, spp, seed=3))
    parts = [gpu.render(cam, make_params(nx, ny, spp, seed=3, shard_index=k, shard_count=3)) for k in range(3)]
    # each pixel comes from one shard; a shard's launch splits its samples into its own number of partial images, so the
    # sums agree to fp32 summation order rather than bit for bit
    assert np.allclose(whole, parts[0] + parts[1] + parts[2], rtol=1e-5, atol=1e-5)
    owner = np.zeros((ny, nx), dtype=int)
    for k in range(3):
        lit = parts[k].sum(axis=2) > 0
        owner[lit] += 1
        assert not lit[~shard_tile_mask(nx, ny, k, 3)].any()
    assert owner.max() == 1
    assert shard_width(nx, 3) == 25 and shard_width(nx, 5) == 26 and shard_width(1920, 8) == 241  # ceil(200/8) = 25 = 5 * 5
    # with enough samples for several partial images per shard; 5 shares a factor with the 25 tiles of a row
    whole = gpu.render(cam, make_params(nx, ny, 100, seed=4))
    for count in (2, 5, 13):
        parts = sum(gpu.render(cam, make_params(nx, ny, 100, seed=4, shard_index=k, shard_count=count)).astype(np.float64) for k in range(count))
        assert np.allclose(whole, parts, rtol=1e-5, atol=1e-5), count


def test_render_matches_oracle_tonemapped(gpu, orc, cornell):
    """End of the pipe (main.rs:136-155): device tonemap of device sums against
    the oracle's tonemap of the same sums, and against the oracle's own render on
    16x16 block means."""
    gpu.upload_scene(cornell.flat)
    spp = 64
    params = make_params(96, 96, spp, seed=31)
    s = gpu.render(cornell.camera, params)
    img = gpu.tonemap(s, spp)
    ref = orc.tonemap(s.astype(np.float64), spp)
    assert np.abs(img.astype(int) - ref.astype(int)).max() <= 1
    assert (img != ref).mean() < 0.01
    os_, _ = orc.render(cornell.flat, cornell.camera.c, make_params(96, 96, spp, seed=32), mode=orc.ORC_F64)
    oimg = orc.tonemap(os_, spp)
    blocks = lambda a: a.astype(np.float64).reshape(6, 16, 6, 16, 3).mean(axis=(1, 3))
    assert np.abs(blocks(img) - blocks(oimg)).max() < 3.0


def test_device_resident_tonemap_equals_the_host_entry_point(gpu, hipmem, cornell):
    """main.rs:136-155 on the buffers rt_render_device wrote: render, tonemap and
    one copy of the 8-bit image, with nothing allocated or staged by the library."""
    from raytracing_amd import make_params
    gpu.upload_scene(cornell.flat)
    p = make_params(80, 56, 40, seed=77)
    n = 80 * 56 * 3
    d_sum, d_u8 = hipmem.alloc(n * 4), hipmem.alloc(n)
    gpu.render_device(cornell.camera, p, d_sum, None, None)
    gpu.tonemap_device(d_sum, 80, 56, 40, d_u8, None)
    img = hipmem.to_host(d_u8, np.uint8, n).reshape(56, 80, 3)
    sums = hipmem.to_host(d_sum, np.float32, n).reshape(56, 80, 3)
    assert np.array_equal(sums, gpu.render(cornell.camera, p))
    assert np.array_equal(img, gpu.tonemap(sums, 40))
    assert img.max() == 255 and img.min() == 0


def test_upload_refuses_null_arrays_and_empty_light_rects(gpu, cornell):
    """A count above zero with a null array, or a light rect of no area (its pdf
    divides by the area, hitable.rs:249-251), is an error return, not a crash."""
    import ctypes as C
    from raytracing_amd import _lib
    from raytracing_amd._lib import RT_ERR_INVALID, RtError
    for field in ("lights", "textures", "objects"):
        bad = _lib.rt_scene()
        for name, _ in _lib.rt_scene._fields_:
            setattr(bad, name, getattr(cornell.flat, name))
        setattr(bad, field, C.cast(None, type(getattr(bad, field))))
        with pytest.raises(RtError) as e:
            gpu.upload_scene(bad)
        assert e.value.code == RT_ERR_INVALID
    bad = _lib.rt_scene()
    for name, _ in _lib.rt_scene._fields_:
        setattr(bad, name, getattr(cornell.flat, name))
    light = _lib.rt_light()
    light.f[0], light.f[1], light.f[2], light.f[3], light.f[4] = 213.0, 213.0, 227.0, 332.0, 554.0
    bad.lights = C.pointer(light)
    with pytest.raises(RtError) as e:
        gpu.upload_scene(bad)
    assert e.value.code == RT_ERR_INVALID
    gpu.upload_scene(cornell.flat)


def test_render_argument_errors(gpu, cornell):
    from raytracing_amd import RtError
    gpu.upload_scene(cornell.flat)
    for bad in (make_params(0, 8, 1), make_params(8, 8, 0), make_params(8, 8, 1, shard_index=2, shard_count=2),
                make_params(8, 8, 1, sky=7), make_params(8, 8, 1, t_min=-1.0), make_params(8, 8, 1, t_min=float("inf")),
                make_params(8, 8, 16, sample_begin=2**31 - 8)):
        with pytest.raises(RtError) as e:
            gpu.render(cornell.camera, bad)
        assert e.value.code == _lib.RT_ERR_INVALID


def test_render_degenerate_sizes_and_depths(gpu, orc, cornell):
    """One pixel, one sample, ragged tiles, max_depth 0 (camera rays only: no
    random direction is ever drawn, so the image must equal the oracle's to fp32
    rounding everywhere)."""
    gpu.upload_scene(cornell.flat)
    cam = cornell.camera
    for nx, ny, spp in ((1, 1, 1), (1, 1, 65), (7, 3, 5), (9, 17, 33)):
        p = make_params(nx, ny, spp, seed=17)
        s, stats = gpu.render(cam, p, want_stats=True)
        o, _ = orc.render(cornell.flat, cam.c, p, mode=orc.ORC_F32_FUSED)
        assert s.shape == (ny, nx, 3) and stats.paths == nx * ny * spp
        assert np.isfinite(s).all()
        assert np.isclose(s, o, rtol=2e-4, atol=1e-4).mean() > 0.9
    p = make_params(64, 64, 4, seed=9, max_depth=0)
    s, stats = gpu.render(cam, p, want_stats=True)
    o, _, ostats = orc.render(cornell.flat, cam.c, p, mode=orc.ORC_F32_FUSED, want_stats=True)
    assert stats.rays == stats.paths == ostats.rays  # nothing scatters
    assert stats.shade_events == 0
    assert np.isclose(s, o, rtol=1e-6, atol=0).mean() > 0.995  # a ray on the light's edge may fall either way
    assert s.max() > 0  # the light is in view


def test_render_sample_chunks_equal_separate_launches(gpu, cornell):
    """A launch that splits each tile's samples over several waves (chunks of at
    least 32 samples, the last one in halving parts, partial images added in a
    fixed order) gives what separate launches over the same sample ranges give."""
    gpu.upload_scene(cornell.flat)
    cam = cornell.camera
    nx, ny, spp = 40, 24, 256  # 15 tiles: the launch uses 7 chunks of 32 samples and the last 32 as 16 + 8 + 8
    whole = gpu.render(cam, make_params(nx, ny, spp, seed=12))
    parts = sum(gpu.render(cam, make_params(nx, ny, 32, sample_begin=32 * k, seed=12)).astype(np.float64) for k in range(8))
    assert np.allclose(whole, parts, rtol=1e-5, atol=1e-5)
    # spp that is not a multiple of the chunk size, and second moments through the same path
    s, s2 = gpu.render(cam, make_params(nx, ny, 100, seed=12), want_sumsq=True)
    a, a2 = gpu.render(cam, make_params(nx, ny, 60, seed=12), want_sumsq=True)
    b, b2 = gpu.render(cam, make_params(nx, ny, 40, sample_begin=60, seed=12), want_sumsq=True)
    assert np.allclose(s, a + b, rtol=1e-5, atol=1e-5)
    assert np.allclose(s2, a2 + b2, rtol=1e-5, atol=1e-4)


def test_render_every_sample_index_runs_exactly_once(gpu, cornell):
    """Sample counts around the sizes at which the last chunk is cut into parts:
    the launch gives the sum of one-sample launches over the same indices."""
    gpu.upload_scene(cornell.flat)
    cam = cornell.camera
    nx, ny = 24, 16
    ones = [gpu.render(cam, make_params(nx, ny, 1, sample_begin=k, seed=21)).astype(np.float64) for k in range(131)]
    for spp in (15, 16, 17, 31, 32, 40, 63, 64, 100, 131):
        whole = gpu.render(cam, make_params(nx, ny, spp, seed=21))
        assert np.allclose(whole, sum(ones[:spp]), rtol=1e-5, atol=1e-5), spp
    shifted = gpu.render(cam, make_params(nx, ny, 40, sample_begin=70, seed=21))
    assert np.allclose(shifted, sum(ones[70:110]), rtol=1e-5, atol=1e-5)


def test_render_without_light_shape_uses_cosine_pdf_alone(gpu, orc, cornell):
    """No light shape: color() samples CosinePDF only (pdf.rs:8-31)."""
    from raytracing_amd import Scene, scenes
    with Scene(3) as sc:
        cam, world = scenes.cornell_box_scene(1.0)
        flat = sc.flatten(world)
        assert flat.n_lights == 0
        gpu.upload_scene(flat)
        p = make_params(48, 48, 32, seed=21)
        s, stats = gpu.render(cam, p, want_stats=True)
        o, _ = orc.render(flat, cam.c, p, mode=orc.ORC_F32_FUSED)
        assert stats.light_tests == 0
        assert np.isclose(s, o, rtol=2e-4, atol=1e-4).mean() > 0.97
        assert abs(s.sum() - o.sum()) / o.sum() < 5e-3


def test_device_render_matches_reference_example_png(gpu, cornell, tmp_path):
    """The reference's own example render, examples/cornell_box.png (600x600),
    against the device path end to end: render 1000 samples per pixel, tonemap on
    the device, write and re-read the PNG, compare 20x20 block means with the
    fixture made from the reference's file.  The PNG's own sample count is not
    recorded, so the bar is in 8-bit levels on block means."""
    import os
    from conftest import GOLDEN
    from raytracing_amd.png import read_png, write_png
    blocks_ref = np.load(os.path.join(GOLDEN, "cornell_box_png_blocks.npy")).astype(np.float64)
    gpu.upload_scene(cornell.flat)
    spp = 1000
    s = gpu.render(cornell.camera, make_params(600, 600, spp, seed=0x5EED0001))
    img = gpu.tonemap(s, spp)
    path = str(tmp_path / "cornell_box.png")
    write_png(path, img)
    back = read_png(path)
    assert np.array_equal(back, img)
    blocks = back.astype(np.float64).reshape(30, 20, 30, 20, 3).mean(axis=(1, 3))
    diff = blocks - blocks_ref
    assert np.abs(diff).max() < 2.5, np.abs(diff).max()
    assert np.sqrt((diff ** 2).mean()) < 0.5, np.sqrt((diff ** 2).mean())
    assert abs(diff.mean()) < 0.3, diff.mean()
    assert back[:10].max() == 0 and back[:, :10].max() == 0  # the black frame


def test_device_render_matches_reference_example_png_at_fine_blocks(gpu, cornell):
    """The same file at 10x10 blocks against a 4096-sample device render: the PNG's
    sample count is estimated from its own noise and the tonemap's noise bias
    allowed for (conftest.reference_png_fine), so what is left in the difference is
    the PNG's own Monte Carlo noise (about 0.35 levels per block); and the colour
    ratios of the two painted walls."""
    from conftest import reference_png_fine
    gpu.upload_scene(cornell.flat)
    spp = 4096
    s, s2 = gpu.render(cornell.camera, make_params(600, 600, spp, seed=0x5EED0002), want_sumsq=True)
    n_hat, diff, ratios = reference_png_fine(s, s2, spp, gpu.tonemap(s, spp))
    print("estimated spp of the reference PNG %.0f; block diff max %.2f rms %.3f mean %.3f; ratios %r" % (
        n_hat, np.abs(diff).max(), np.sqrt((diff ** 2).mean()), diff.mean(), ratios))
    # measured: estimate 982 (the reference at quality 1000 renders 8 x 125 samples), max 1.45, rms 0.26, mean 0.04,
    # colour ratios equal to 2 parts in 10,000
    assert 800 < n_hat < 1250, n_hat
    assert np.abs(diff).max() < 2.0, np.abs(diff).max()
    assert np.sqrt((diff ** 2).mean()) < 0.35, np.sqrt((diff ** 2).mean())
    assert abs(diff.mean()) < 0.1, diff.mean()
    for key in ("green_wall_g_over_r", "red_wall_r_over_g"):
        png, ours = ratios[key]
        assert abs(ours / png - 1.0) < 0.003, (key, png, ours)


def test_render_large_image_low_spp(gpu, orc, cornell):
    """3840x2160 at 2 samples per pixel: 129,600 tiles, every index past 2^16,
    against the oracle on the same random stream."""
    gpu.upload_scene(cornell.flat)
    nx, ny, spp = 3840, 2160, 2
    p = make_params(nx, ny, spp, seed=4)
    s, stats = gpu.render(cornell.camera, p, want_stats=True)
    assert stats.paths == nx * ny * spp
    o, _ = orc.render(cornell.flat, cornell.camera.c, p, mode=orc.ORC_F32_FUSED, n_threads=8)
    assert np.isclose(s, o, rtol=2e-4, atol=1e-4).mean() > 0.98
    assert abs(s.sum() - o.sum()) / o.sum() < 2e-3
    # the corners and the centre are where an indexing slip would show
    for ys, xs in ((slice(0, 64), slice(0, 64)), (slice(ny - 64, ny), slice(nx - 64, nx)), (slice(1000, 1100), slice(1900, 2000))):
        assert np.isclose(s[ys, xs], o[ys, xs], rtol=2e-4, atol=1e-4).mean() > 0.97


def test_cpp_host_program_renders_the_same_image(gpu, tmp_path):
    """cornell_main (the reference's main() restated in C++ over the library)
    builds the scene with the C++ host classes, renders and writes a PNG; the Python
    binding, given the same seed, must produce the same bytes."""
    import os
    import subprocess
    from conftest import ROOT
    from raytracing_amd import Scene, scenes
    from raytracing_amd.png import read_png
    exe = os.path.join(ROOT, "raytracing_amd", "csrc", "cornell_main")
    assert os.path.exists(exe), "build it with make -C raytracing_amd/csrc"
    out = str(tmp_path / "cpp.png")
    seed = 24680
    run = subprocess.run([exe, "2", "16", out, str(seed)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    assert "200x200, 16 samples per pixel, 640000 paths" in run.stdout
    img_cpp = read_png(out)
    with Scene(seed) as sc:
        cam, world = scenes.cornell_box_scene(1.0)
        sc.light_xz_rect(*scenes.MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(world)
        gpu.upload_scene(flat)
        s = gpu.render(cam, make_params(200, 200, 16, seed=seed))
        img_py = gpu.tonemap(s, 16)
    assert img_cpp.shape == (200, 200, 3)
    assert np.array_equal(img_cpp, img_py)


def test_render_against_the_stored_images(gpu, cornell):
    """The device against tests/golden/oracle_vectors.npz: the f64 restatement's
    24x24 Cornell image and 16x16 cornell_smoke image at 32 samples (sums and second
    moments), compared with device renders of the same size from another seed by
    the standard-error test; no oracle code runs here."""
    import os
    from conftest import GOLDEN
    from raytracing_amd import Scene, scenes
    v = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    with Scene(0x5EED0001) as sc:
        cam, world = scenes.cornell_smoke_scene(1.0)
        sc.light_xz_rect(*scenes.SMOKE_LIGHT_SHAPE)
        smoke = sc.flatten(world)
        cases = [(cornell.flat, cornell.camera, 24, v["cornell_sum"], v["cornell_sumsq"]), (smoke, cam, 16, v["smoke_sum"], v["smoke_sumsq"])]
        for flat, camera, n, want, want2 in cases:
            gpu.upload_scene(flat)
            spp = 32
            s, s2 = gpu.render(camera, make_params(n, n, spp, seed=99), want_sumsq=True)
            ma, mb, se, z = _z_scores(s.astype(np.float64), s2.astype(np.float64), want, want2, spp)
            valid = se > 0
            zz = z[valid]
            assert valid.mean() > 0.6
            assert (np.abs(zz) > 3.5).mean() < 0.02, (np.abs(zz) > 3.5).mean()
            assert abs(zz.mean()) < 0.15, zz.mean()
            assert abs(ma.mean() - mb.mean()) / mb.mean() < 0.08, (ma.mean(), mb.mean())


def test_the_bench_launch_itself(gpu, orc):
    """Exactly the launch bench.py times (cornell_box_scene at 1024x1024, 1000 samples per pixel, seed 0x5EED0001, depth
    50): every path is traced once, the image is the same on a second launch, and the means of its 32x32-pixel blocks
    agree with the f64 oracle's (the reference's arithmetic) at 16 samples per pixel and another seed within 3 standard
    errors of the oracle's block means (the device's own error, at 1000 x 1024 samples per block, is a tenth of that)."""
    from conftest import CornellFixture
    c = CornellFixture(seed=0x5EED0001, ratio=1.0)
    gpu.upload_scene(c.flat)
    nx = ny = 1024
    p = make_params(nx, ny, 1000, seed=0x5EED0001)
    s, s2, st = gpu.render(c.camera, p, want_sumsq=True, want_stats=True)
    assert st.paths == nx * ny * 1000
    assert 2.3 * st.paths < st.rays < 2.7 * st.paths  # 2.489 rays per path
    assert np.isfinite(s).all() and (s >= 0).all()
    again = gpu.render(c.camera, p)
    assert np.array_equal(s, again)
    o_spp = 16
    o, o2 = orc.render(c.flat, c.camera.c, make_params(nx, ny, o_spp, seed=0xABCD), mode=orc.ORC_F64, n_threads=16)
    blk = lambda a: a.reshape(32, 32, 32, 32, 3).sum(axis=(1, 3))  # noqa: E731
    n_dev, n_orc = 1000.0 * 1024, float(o_spp) * 1024
    md, mo = blk(s.astype(np.float64)) / n_dev, blk(o) / n_orc
    vd = np.maximum(blk(s2.astype(np.float64)) / n_dev - md * md, 0.0)
    vo = np.maximum(blk(o2) / n_orc - mo * mo, 0.0)
    se = np.sqrt(vd / n_dev + vo / n_orc)
    valid = se > 0
    z = (md - mo)[valid] / se[valid]
    from conftest import log_achieved
    log_achieved("bench_launch_blocks", frac_over_3=(np.abs(z) > 3).mean(), mean_z=z.mean(), std_z=z.std(), blocks=valid.sum())
    assert valid.mean() > 0.8
    assert (np.abs(z) > 3).mean() < 0.02, (np.abs(z) > 3).mean()
    assert abs(z.mean()) < 0.15, z.mean()
    assert abs(md.mean() - mo.mean()) / mo.mean() < 0.005
