"""The stack traversal keeps the first entries of a lane's stack in LDS and the rest
in a region of global memory: with the LDS part cut to one or two entries nearly
every push and pop goes through that region, and the image must not change (the
stack kernel's sums agree to the order of its float adds)."""
import numpy as np
import pytest

from raytracing_amd import Scene, make_params
from raytracing_amd import scenes
from raytracing_amd._lib import (RT_OPT_FIXED_ORDER, RT_OPT_HOLD_LANES, RT_OPT_READY_LANES, RT_OPT_STACK_LDS_ENTRIES,
                                 RT_OPT_WAVES_PER_CU)

pytestmark = pytest.mark.gpu


@pytest.fixture
def lds_cap(gpu):
    default = gpu.get_option(RT_OPT_STACK_LDS_ENTRIES)

    def set_cap(n):
        gpu.set_option(RT_OPT_STACK_LDS_ENTRIES, default if n is None else n)
    yield set_cap
    gpu.set_option(RT_OPT_STACK_LDS_ENTRIES, default)


def test_image_does_not_depend_on_where_the_stack_lives(gpu, orc, lds_cap):
    from test_gpu_materials import _sphere_field
    with Scene(31) as sc:
        cam, world = _sphere_field(400, 9, glass_and_metal=True)
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    p = make_params(96, 64, 48, seed=12)
    lds_cap(None)
    ref, st = gpu.render(cam, p, want_stats=True)
    assert st.node_tests > 4 * st.rays  # a real walk
    for cap in (1, 2, 5):
        lds_cap(cap)
        s, st2 = gpu.render(cam, p, want_stats=True)
        assert np.allclose(s, ref, rtol=1e-5, atol=1e-4), cap
        assert (st2.rays, st2.hits, st2.node_tests, st2.sphere_tests) == (st.rays, st.hits, st.node_tests, st.sphere_tests), cap
    # and against the oracle on the same stream
    lds_cap(1)
    s = gpu.render(cam, p)
    o = orc.render(flat, cam.c, p, mode=orc.ORC_F32_FUSED, n_threads=8, want_sumsq=False)
    assert np.isclose(s, o, rtol=5e-3, atol=5e-3).mean() > 0.8


def test_final_scene_with_the_stack_in_global_memory(gpu, lds_cap):
    """Instances, sub-trees with their frame markers, media: the final scene, whose
    worst-case stack is 30 entries, with one of them in LDS."""
    with Scene(0x5EED0000) as sc:
        cam, world = scenes.final_scene(1.5)
        sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    p = make_params(120, 80, 16, seed=4)
    lds_cap(None)
    ref = gpu.render(cam, p)
    lds_cap(1)
    s = gpu.render(cam, p)
    assert np.allclose(s, ref, rtol=1e-5, atol=1e-4)


def _field_150():
    from test_gpu_materials import _sphere_field
    with Scene(77) as sc:
        cam, world = _sphere_field(150, 5, glass_and_metal=True)
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        return cam, sc.flatten(world)


def test_fixed_order_option_gives_equal_bits(gpu):
    """RT_OPT_FIXED_ORDER: the stack kernel writes one partial image per work item,
    keeps one item open per wave and adds the partial images in a fixed order, so two
    renders of the same input agree bit for bit; without it the items meet in the
    output through float atomics and only the tolerance holds."""
    cam, flat = _field_150()
    gpu.upload_scene(flat)
    p = make_params(160, 96, 128, seed=5)
    free = gpu.render(cam, p)
    gpu.set_option(RT_OPT_FIXED_ORDER, 1)
    try:
        assert gpu.get_option(RT_OPT_FIXED_ORDER) == 1
        a, a2, st = gpu.render(cam, p, want_sumsq=True, want_stats=True)
        b, b2 = gpu.render(cam, p, want_sumsq=True)
        c = gpu.render(cam, p)
        assert st.node_tests > 4 * st.rays  # the stack traversal, not the scan
        assert np.array_equal(a, b) and np.array_equal(a2, b2) and np.array_equal(a, c)
    finally:
        gpu.set_option(RT_OPT_FIXED_ORDER, 0)
    assert np.allclose(free, a, rtol=1e-5, atol=1e-4)


def test_output_in_host_mapped_memory_is_written_with_plain_stores(gpu, hipmem):
    """The stack kernel's float atomics are only dependable on ordinary device
    allocations.  rt_render_device looks at what kind of memory it is given: a
    host-mapped output buffer gets the fixed-order path (partial images, plain stores),
    so its sums equal RT_OPT_FIXED_ORDER's in device memory bit for bit."""
    cam, flat = _field_150()
    gpu.upload_scene(flat)
    p = make_params(160, 96, 128, seed=5)
    n = 160 * 96 * 3
    gpu.set_option(RT_OPT_FIXED_ORDER, 1)
    try:
        want, want2 = gpu.render(cam, p, want_sumsq=True)
    finally:
        gpu.set_option(RT_OPT_FIXED_ORDER, 0)
    for both_mapped in (True, False):
        a = hipmem.alloc_host_mapped(n * 4)
        b = hipmem.alloc_host_mapped(n * 4) if both_mapped else hipmem.alloc(n * 4)
        gpu.render_device(cam, p, b, a, None)  # the sums in b, the sums of squares in the mapped buffer
        got = hipmem.to_host(b, np.float32, n).reshape(want.shape)
        got2 = hipmem.to_host(a, np.float32, n).reshape(want.shape)
        assert np.array_equal(got, want) and np.array_equal(got2, want2), both_mapped
    hipmem.free_all()


def test_tuning_options_are_clamped_and_cannot_hang_the_kernel(gpu):
    """A threshold of more than 64 lanes in 64 could never be met: the traversal phase
    would never hand the wave back to the shading code.  The options clamp to their
    ranges, and the phase also ends when no lane is left in a traversal."""
    cam, flat = _field_150()
    gpu.upload_scene(flat)
    p = make_params(64, 48, 16, seed=6)
    ref = gpu.render(cam, p)
    saved = {k: gpu.get_option(k) for k in (RT_OPT_READY_LANES, RT_OPT_HOLD_LANES, RT_OPT_STACK_LDS_ENTRIES, RT_OPT_WAVES_PER_CU)}
    try:
        for k, v, want in ((RT_OPT_READY_LANES, 65, 64), (RT_OPT_READY_LANES, 1 << 30, 64), (RT_OPT_HOLD_LANES, 0, 1),
                           (RT_OPT_HOLD_LANES, 999, 64), (RT_OPT_STACK_LDS_ENTRIES, -5, 1), (RT_OPT_STACK_LDS_ENTRIES, 1000, 64),
                           (RT_OPT_WAVES_PER_CU, 0, 1), (RT_OPT_WAVES_PER_CU, 64, 32)):
            gpu.set_option(k, v)
            assert gpu.get_option(k) == want
            s = gpu.render(cam, p)
            assert np.allclose(s, ref, rtol=1e-5, atol=1e-4), (k, v)
            gpu.set_option(k, saved[k])
    finally:
        for k, v in saved.items():
            gpu.set_option(k, v)
    with pytest.raises(Exception):
        gpu.set_option(99, 1)


def test_pool_kernel_traces_the_same_paths(gpu):
    """RT_OPT_POOL_SLOTS: the kernel that keeps a wave's paths in a pool and runs each
    phase of the loop on a full wave traces exactly the paths the default kernel
    traces: the same event counts, the same image to the order of fp32 additions
    (all materials, textures and media: the final scene; lens and sky: Book 1)."""
    from raytracing_amd._lib import RT_OPT_POOL_SLOTS, RT_SKY_GRADIENT
    cases = []
    with Scene(0x5EED0000) as sc:
        cam, world = scenes.final_scene(1.5)
        sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        cases.append((cam, sc.flatten(world), make_params(240, 160, 24, seed=4)))
    with Scene(3) as sc:
        cam, world = scenes.random_scene(1.5)
        cases.append((cam, sc.flatten(world), make_params(150, 100, 16, seed=8, sky=RT_SKY_GRADIENT)))
    cam, flat = _field_150()
    cases.append((cam, flat, make_params(96, 64, 32, seed=2)))
    for cam, flat, p in cases:
        gpu.upload_scene(flat)
        ref, ref2, st = gpu.render(cam, p, want_sumsq=True, want_stats=True)
        try:
            for slots in (64, 128, 200):
                gpu.set_option(RT_OPT_POOL_SLOTS, slots)
                s, s2, st2 = gpu.render(cam, p, want_sumsq=True, want_stats=True)
                assert st2.as_dict().items() - {("kernel_ms", st2.kernel_ms)} == st.as_dict().items() - {("kernel_ms", st.kernel_ms)}, slots
                assert np.allclose(s, ref, rtol=1e-5, atol=1e-4) and np.allclose(s2, ref2, rtol=1e-5, atol=1e-3), slots
        finally:
            gpu.set_option(RT_OPT_POOL_SLOTS, 0)
