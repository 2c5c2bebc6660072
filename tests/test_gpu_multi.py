"""rt_multi_*: one render over several ranks from one process (the thread fan-out
and image sum of main.rs:70-133 with GPUs for threads).  The GPU box has one GPU,
so ranks share it ("virtual ranks": repeated device ids, exchange by device
copies in rank order).  What RCCL can do on one device is driven with
RT_MULTI_FORCE_RCCL: dlopen, ncclCommInitAll, one in-place ncclReduce of the frame
onto the only rank and one empty send/receive group, i.e. the resolved symbols,
their signatures and the enum values; the exchange BETWEEN devices (several ranks
in a reduce, matched sends and receives, the staging offsets) has never run."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import MultiRenderer, RtError, Scene, make_params, scenes
from raytracing_amd._lib import RT_ERR_INVALID

pytestmark = pytest.mark.gpu


def test_one_rank_needs_no_rccl(gpu, cornell):
    m = MultiRenderer([0])
    try:
        assert not m.uses_rccl
        m.upload_scene(cornell.flat)
        p = make_params(64, 48, 8, seed=3)
        assert np.array_equal(m.render(cornell.camera, p), gpu_render(gpu, cornell, p))
    finally:
        m.close()


def gpu_render(gpu, cornell, p):
    gpu.upload_scene(cornell.flat)
    return gpu.render(cornell.camera, p)


def test_one_rank_through_rccl_equals_rt_render(gpu, cornell):
    """The RCCL calls on one rank: a reduce onto itself must leave the frame as it is."""
    gpu.upload_scene(cornell.flat)
    p = make_params(96, 80, 40, seed=3)
    want, want2 = gpu.render(cornell.camera, p, want_sumsq=True)
    m = MultiRenderer([0], force_rccl=True)
    try:
        assert m.uses_rccl
        m.upload_scene(cornell.flat)
        for mode in ("samples", "tiles"):
            got, got2, stats = m.render(cornell.camera, p, mode=mode, want_sumsq=True, want_stats=True)
            assert np.array_equal(got, want) and np.array_equal(got2, want2)
            assert stats[0].paths == 96 * 80 * 40 and stats[0].kernel_ms > 0
    finally:
        m.close()


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_virtual_ranks_reproduce_the_one_rank_image(gpu, cornell, n_ranks):
    """N ranks on one device: by samples the frames add up to the one-rank image up
    to fp32 summation order; by tiles the gathered image equals, bit for bit, the
    tile shards rendered one by one through rt_render and added."""
    gpu.upload_scene(cornell.flat)
    nx, ny, spp = 200, 133, 48  # 25 x 17 tiles, the last row cut; 8 ranks deal over a grid 25 wide, 5 ranks would need 26
    p = make_params(nx, ny, spp, seed=17)
    whole, whole2 = gpu.render(cornell.camera, p, want_sumsq=True)
    m = MultiRenderer([0] * n_ranks)
    try:
        assert not m.uses_rccl
        m.upload_scene(cornell.flat)
        s, s2, stats = m.render(cornell.camera, p, mode="samples", want_sumsq=True, want_stats=True)
        assert sum(st.paths for st in stats) == nx * ny * spp
        assert np.allclose(s, whole, rtol=2e-6, atol=1e-5) and np.allclose(s2, whole2, rtol=2e-6, atol=1e-4)
        t, t2 = m.render(cornell.camera, p, mode="tiles", want_sumsq=True)
        parts = np.zeros_like(whole)
        for k in range(n_ranks):
            parts += gpu.render(cornell.camera, make_params(nx, ny, spp, seed=17, shard_index=k, shard_count=n_ranks))
        assert np.array_equal(t, parts)
        assert np.allclose(t, whole, rtol=2e-6, atol=1e-5) and np.allclose(t2, whole2, rtol=2e-6, atol=1e-4)
    finally:
        m.close()


def test_more_ranks_than_blocks_and_a_tree_scene(gpu, orc):
    """16x8 pixels are two tiles: with five ranks three have nothing to render;
    100x60 is an ordinary small frame.  The scene is a sphere field on the stack traversal, whose per-pixel
    sums depend on scheduling in their last bits, hence the tolerance."""
    from test_gpu_materials import _sphere_field
    with Scene(61) as sc:
        cam, world = _sphere_field(120, 5, glass_and_metal=False)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    m = MultiRenderer([0] * 5)
    try:
        m.upload_scene(flat)
        for nx, ny in ((16, 8), (100, 60)):
            p = make_params(nx, ny, 32, seed=8)
            whole = gpu.render(cam, p)
            for mode in ("tiles", "samples"):
                s = m.render(cam, p, mode=mode)
                assert np.allclose(s, whole, rtol=1e-5, atol=1e-4), (mode, nx)
                assert s.sum() > 0
    finally:
        m.close()


def test_multi_argument_errors(gpu, cornell):
    with pytest.raises(RtError):
        MultiRenderer([0, 99])
    m = MultiRenderer([0, 0], no_rccl=True)
    try:
        with pytest.raises(RtError) as e:
            m.render(cornell.camera, make_params(8, 8, 4))
        assert "no scene" in str(e.value)
        m.upload_scene(cornell.flat)
        with pytest.raises(RtError) as e:
            m.render(cornell.camera, make_params(8, 8, 1), mode="samples")  # fewer samples than ranks
        assert e.value.code == RT_ERR_INVALID
        with pytest.raises(RtError):
            m.render(cornell.camera, make_params(8, 8, 4, shard_index=1, shard_count=2))
        assert m.render(cornell.camera, make_params(8, 8, 1), mode="tiles").sum() >= 0
    finally:
        m.close()


def test_cpp_host_program_over_three_ranks(gpu, tmp_path):
    """cornell_main with its samples spread over three ranks of GPU 0 (the
    reference spreads them over 8 threads) writes the one-rank image up to the
    last bit of a few pixels' sums."""
    from conftest import ROOT
    from raytracing_amd.png import read_png
    exe = os.path.join(ROOT, "raytracing_amd", "csrc", "cornell_main")
    one, three, tiles = (str(tmp_path / n) for n in ("one.png", "three.png", "tiles.png"))
    for out, extra in ((one, []), (three, ["-3"]), (tiles, ["-3", "tiles"])):
        run = subprocess.run([exe, "2", "24", out, "777"] + extra, capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, run.stderr
        assert "200x200, 24 samples per pixel, 960000 paths" in run.stdout
    a = read_png(one).astype(np.int32)
    for other in (three, tiles):
        b = read_png(other).astype(np.int32)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3


def test_per_rank_event_counts_are_complete(gpu, cornell):
    """Every rank's counters are zeroed on that rank's own stream, in order with its
    counting launch: eight ranks on one device, five renders in a row, and the paths
    the ranks report always add up to the whole image."""
    nx, ny, spp = 160, 120, 64
    p = make_params(nx, ny, spp, seed=3)
    m = MultiRenderer([0] * 8)
    try:
        m.upload_scene(cornell.flat)
        for _ in range(5):
            for mode in ("samples", "tiles"):
                _, stats = m.render(cornell.camera, p, mode=mode, want_stats=True)
                assert sum(st.paths for st in stats) == nx * ny * spp, mode
                assert all(st.rays >= st.paths for st in stats)
    finally:
        m.close()
