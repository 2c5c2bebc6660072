"""rt_render_device_noalloc (the entry point that allocates nothing: its work
counter and partial images live in a static device array) against
rt_render_device, the entry point it is a variant of, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gpu, hipmem, cam, params, want_sumsq=False):
    n = params.ny * params.nx * 3
    outs = []
    for fn in (gpu.render_device, gpu.render_device_noalloc):
        a = hipmem.alloc(n * 4)  # filled with 0xFF bytes (NaN): every float must be overwritten
        b = hipmem.alloc(n * 4) if want_sumsq else None
        fn(cam, params, a, b, None)
        outs.append((hipmem.to_host(a, np.float32, n), hipmem.to_host(b, np.float32, n) if want_sumsq else None))
    hipmem.free_all()
    return outs


def test_noalloc_matches_render_device_twice_in_a_row(gpu, hipmem, cornell):
    """Two calls in a row on one GPU, the second with a smaller image and fewer
    samples: each bit-identical to rt_render_device (same kernels, same chunking,
    fixed summation order), so nothing of the first call's partial images or
    counter may leak into the second."""
    from raytracing_amd import make_params
    gpu.upload_scene(cornell.flat)
    # 200x152 at 300 spp: 9 + 3 sample chunks (partial images in the scratch); then 96x64 at 70 spp: 2 + 2 chunks
    for nx, ny, spp, sumsq in ((200, 152, 300, True), (96, 64, 70, False)):
        p = make_params(nx, ny, spp, seed=0xA110C)
        (ref, ref2), (got, got2) = _both(gpu, hipmem, cornell.camera, p, want_sumsq=sumsq)
        assert np.isfinite(ref).all() and ref.sum() > 0
        assert np.array_equal(ref, got), "noalloc sums differ from rt_render_device at %dx%d@%d" % (nx, ny, spp)
        if sumsq:
            assert np.array_equal(ref2, got2)


def test_noalloc_single_chunk_and_shards(gpu, hipmem, cornell):
    """Fewer than 64 spp is one chunk (only the counter is used); a tile-sharded
    launch zeroes its partial images inside the scratch."""
    from raytracing_amd import make_params
    gpu.upload_scene(cornell.flat)
    for p in (make_params(130, 70, 9, seed=5), make_params(256, 192, 128, seed=6, shard_index=1, shard_count=3)):
        (ref, _), (got, _) = _both(gpu, hipmem, cornell.camera, p)
        assert np.isfinite(ref).all() and ref.sum() > 0
        assert np.array_equal(ref, got)


def test_noalloc_refuses_a_launch_beyond_its_scratch(gpu, hipmem, cornell):
    """1024x1024 at 1000 spp needs 15 partial images (189 MB): a clear error, and
    the context goes on working."""
    from raytracing_amd import make_params
    from raytracing_amd._lib import RT_ERR_UNSUPPORTED, RtError
    gpu.upload_scene(cornell.flat)
    n = 1024 * 1024 * 3
    a = hipmem.alloc(n * 4, fill_byte=0)
    with pytest.raises(RtError) as e:
        gpu.render_device_noalloc(cornell.camera, make_params(1024, 1024, 1000), a, None, None)
    assert e.value.code == RT_ERR_UNSUPPORTED and "scratch" in str(e.value)
    assert float(hipmem.to_host(a, np.float32, n).sum()) == 0.0
    hipmem.free_all()
    # at 100 spp the same image fits (5 partial images, 63 MB)
    (ref, _), (got, _) = _both(gpu, hipmem, cornell.camera, make_params(1024, 1024, 100))
    assert np.array_equal(ref, got)


def test_noalloc_on_a_tree_scene(gpu, hipmem):
    """A scene on the per-lane stack traversal: no partial images (its work items add
    into the output), and the deep end of the lanes' stacks lives in the scratch too.
    Such a scene's sums depend on scheduling in their last bits, hence the tolerance."""
    from raytracing_amd import Scene, make_params
    from test_gpu_materials import _sphere_field
    with Scene(77) as sc:
        cam, world = _sphere_field(300, 3, glass_and_metal=True)
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    for p in (make_params(160, 96, 200, seed=8), make_params(64, 40, 20, seed=9)):
        (ref, _), (got, _) = _both(gpu, hipmem, cam, p)
        assert np.isfinite(ref).all() and ref.sum() > 0
        assert np.allclose(ref, got, rtol=1e-5, atol=1e-4)


def test_noalloc_when_the_scratch_cannot_hold_the_stacks(gpu, hipmem):
    """Partial images (RT_OPT_FIXED_ORDER) of 1024x768 at 128 spp take 56.6 MB of the
    64 MiB scratch; with one stack entry wanted in LDS the 8192 waves' stack regions
    would need 4 MiB per further entry, so only two entries fit and the rest of each
    lane's stack has to stay in LDS: slower, same bits as rt_render_device."""
    from raytracing_amd import Scene, make_params
    from raytracing_amd._lib import RT_OPT_FIXED_ORDER, RT_OPT_STACK_LDS_ENTRIES
    from test_gpu_materials import _sphere_field
    with Scene(78) as sc:
        cam, world = _sphere_field(600, 4, glass_and_metal=True)
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    saved = gpu.get_option(RT_OPT_STACK_LDS_ENTRIES)
    gpu.set_option(RT_OPT_FIXED_ORDER, 1)
    gpu.set_option(RT_OPT_STACK_LDS_ENTRIES, 1)
    try:
        for p in (make_params(1024, 768, 128, seed=8), make_params(64, 40, 20, seed=9)):
            (ref, _), (got, _) = _both(gpu, hipmem, cam, p)
            assert np.isfinite(ref).all() and ref.sum() > 0
            assert np.array_equal(ref, got)
    finally:
        gpu.set_option(RT_OPT_FIXED_ORDER, 0)
        gpu.set_option(RT_OPT_STACK_LDS_ENTRIES, saved)
