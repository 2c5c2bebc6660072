"""rt_trace_closest_f64, the parity mode: the device computing the nearest hit in
the reference's own arithmetic (f64, the source's divisions, no fused
multiply-adds, bvh.rs's traversal with both children visited and the right one
winning ties) against the f64 CPU restatement, BIT FOR BIT: primitive, t, hit
point and normal.  Where the fp32 render path is compared with the reference's
arithmetic statistically, this path is compared exactly."""
import numpy as np
import pytest

from raytracing_amd import Scene
from raytracing_amd import scene as S
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def _check(gpu, orc, flat, o, d, min_hits=0.2):
    gpu.upload_scene(flat)
    o32 = np.ascontiguousarray(o, dtype=np.float32)
    d32 = np.ascontiguousarray(d, dtype=np.float32)
    prim, t, nrm, p = gpu.trace_closest_f64(o32.astype(np.float64), d32.astype(np.float64), want_normal=True, want_p=True)
    oprim, ot, onrm, op = orc.trace_closest(flat, o32, d32, mode=orc.ORC_F64, want_normal=True, want_p=True)
    assert np.array_equal(prim, oprim), "primitive ids differ on %d rays" % int((prim != oprim).sum())
    assert np.array_equal(t, ot), "t differs on %d rays" % int((t != ot).sum())
    assert np.array_equal(p, op) and np.array_equal(nrm, onrm)
    assert (prim >= 0).mean() > min_hits
    return prim, t


def test_cornell_box_f64_bit_exact(gpu, orc, cornell):
    from conftest import cornell_ray_batch
    o, d = cornell_ray_batch(cornell, 40000, 7)
    prim, t = _check(gpu, orc, cornell.flat, o, d, min_hits=0.8)
    # and the fp32 path finds the same surfaces as the reference's arithmetic (the batch aims a tenth of its rays at
    # edges and corners, where the two arithmetics may decide differently)
    p32, t32 = gpu.trace_closest(o, d)
    same = p32 == prim
    assert same.mean() > 0.995, same.mean()
    hit = same & (prim >= 0)
    assert (np.abs(t32[hit].astype(np.float64) - t[hit]) / np.maximum(t[hit], 1e-3) < 1e-5).mean() > 0.999


@pytest.mark.parametrize("seed,n_items", [(1, 3), (3, 15), (5, 40), (6, 120), (7, 400), (9, 70)])
def test_random_scenes_f64_bit_exact(gpu, orc, seed, n_items):
    """Every hitable and nesting the flattener takes (test_gpu_fuzz.py's scenes)."""
    from test_gpu_fuzz import _random_hitable
    rng = np.random.RandomState(1000 + seed)
    with Scene(seed) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        items = [_random_hitable(rng, mats) for _ in range(n_items)]
        flat = sc.flatten(S.HitableList.new(items))
    m = 20000
    o = rng.uniform(-14, 14, size=(m, 3)).astype(np.float32)
    d = (rng.uniform(-10, 10, size=(m, 3)).astype(np.float32) - o).astype(np.float32)
    d[: m // 10, rng.randint(3)] = 0.0  # some rays parallel to an axis plane
    _check(gpu, orc, flat, o, d, min_hits=0.01 if n_items < 10 else 0.05)


def test_final_scene_f64_bit_exact(gpu, orc):
    with Scene(0x5EED0000) as sc:
        cam, world = scenes.final_scene(1.5)
        flat = sc.flatten(world)
    rng = np.random.RandomState(3)
    n = 20000
    o = rng.uniform(-100, 600, size=(n, 3)).astype(np.float32)
    d = (rng.uniform(-100, 600, size=(n, 3)).astype(np.float32) - o).astype(np.float32)
    _check(gpu, orc, flat, o, d, min_hits=0.3)


def test_f64_argument_errors(gpu, cornell):
    from raytracing_amd._lib import RtError
    gpu.upload_scene(cornell.flat)
    with pytest.raises(ValueError):
        gpu.trace_closest_f64(np.zeros((4, 3)), np.zeros((5, 3)))
    prim, t = gpu.trace_closest_f64(np.zeros((0, 3)), np.zeros((0, 3)))
    assert prim.size == 0
    # an empty interval rejects everything, as the source's comparisons do
    prim, t = gpu.trace_closest_f64(np.array([[278.0, 278.0, -800.0]]), np.array([[0.0, 0.0, 1.0]]), t_min=5.0, t_max=1.0)
    assert prim[0] == -1
