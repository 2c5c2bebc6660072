"""The stack traversal keeps the first entries of a lane's stack in LDS and the rest
in a region of global memory: with the LDS part cut to one or two entries nearly
every push and pop goes through that region, and the image must not change (the
stack kernel's sums agree to the order of its float adds)."""
import os

import numpy as np
import pytest

from raytracing_amd import Scene, make_params
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture
def lds_cap():
    def set_cap(n):
        if n is None:
            os.environ.pop("RT_STACK_LDS_CAP", None)
        else:
            os.environ["RT_STACK_LDS_CAP"] = str(n)  # read by the library at every launch
    yield set_cap
    os.environ.pop("RT_STACK_LDS_CAP", None)


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
