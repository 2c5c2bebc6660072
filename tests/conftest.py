import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def log_achieved(name, **values):
    """Appends what a parity test measured (fractions of equal pixels, z statistics)
    to gpurun_out/parity_achieved.jsonl, so that the bars in the tests can be set from
    what the kernels achieve.  Does nothing where that directory does not exist."""
    import inspect
    import json
    out = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(out):
        return
    caller = next((f.function for f in inspect.stack()[1:] if f.function.startswith("test_")), "?")
    with open(os.path.join(out, "parity_achieved.jsonl"), "a") as fh:
        fh.write(json.dumps(dict(test=caller, what=name, **{k: float(v) for k, v in values.items()})) + "\n")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): oracle/liboracle.so through ctypes."""
    from oracle import binding
    binding.lib()
    return binding


class CornellFixture(object):
    def __init__(self, seed=0x5EED0001, ratio=1.0):
        from raytracing_amd import Scene
        from raytracing_amd.scenes import MAIN_RS_LIGHT_SHAPE, cornell_box_scene
        self.scene = Scene(seed)
        with self.scene:
            self.camera, self.world = cornell_box_scene(ratio)
            self.scene.light_xz_rect(*MAIN_RS_LIGHT_SHAPE)
            self.flat = self.scene.flatten(self.world)


@pytest.fixture(scope="session")
def cornell():
    """cornell_box_scene(1.0) (scenes.rs:276-360) flattened, with the light
    shape of main.rs:177-178 registered for HitablePDF."""
    return CornellFixture()


@pytest.fixture(scope="session")
def gpu():
    """One Renderer on device 0.  Fails (does not skip) when the HIP library or
    the device is missing: GPU tests must not pass on a fallback."""
    from raytracing_amd import Renderer
    r = Renderer(0)
    yield r
    r.close()


class HipMemory(object):
    """Device buffers for tests of the device-pointer entry points, through the
    HIP runtime librt_amd.so itself is linked against (no torch: a second HIP
    runtime in the process does not see the device once the first has it)."""

    def __init__(self):
        import ctypes as C
        from raytracing_amd import _lib
        _lib.lib()  # the library first: the soname below then resolves to the runtime it loaded
        self.C = C
        self.hip = C.CDLL("libamdhip64.so.7")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.live = []
        self.live_host = []

    def alloc(self, nbytes, fill_byte=0xFF):
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), nbytes) == 0
        assert self.hip.hipMemset(p, fill_byte, nbytes) == 0
        self.live.append(p)
        return p.value

    def alloc_host_mapped(self, nbytes, fill_byte=0xFF):
        """Pinned host memory the device reaches through the same address (not an
        ordinary device allocation)."""
        p = self.C.c_void_p()
        self.hip.hipHostMalloc.argtypes = [self.C.POINTER(self.C.c_void_p), self.C.c_size_t, self.C.c_uint]
        assert self.hip.hipHostMalloc(self.C.byref(p), nbytes, 0) == 0
        self.C.memset(p.value, fill_byte, nbytes)
        self.live_host.append(p)
        return p.value

    def sync(self):
        assert self.hip.hipDeviceSynchronize() == 0

    def to_host(self, ptr, dtype, count):
        out = np.empty(count, dtype=dtype)
        self.sync()
        assert self.hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 4) == 0  # hipMemcpyDefault: device or mapped host memory
        return out

    def free_all(self):
        for p in self.live:
            self.hip.hipFree(p)
        self.live = []
        self.hip.hipHostFree.argtypes = [self.C.c_void_p]
        for p in self.live_host:
            self.hip.hipHostFree(p)
        self.live_host = []


@pytest.fixture()
def hipmem(gpu):
    m = HipMemory()
    yield m
    m.sync()
    m.free_all()


def cornell_ray_batch(cornell, n, seed):
    """Seeded rays for nearest-hit parity in the Cornell box: camera rays,
    rays from random interior points in random directions (what cosine bounces
    look like), axis-aligned rays and rays aimed at primitive edges/corners."""
    rng = np.random.RandomState(seed)
    cam = cornell.camera.c
    origin = np.array(list(cam.origin))
    llc = np.array(list(cam.lower_left_corner))
    hor = np.array(list(cam.horizontal))
    ver = np.array(list(cam.vertical))
    n_cam = n // 3
    n_int = n // 3
    n_axis = n // 12
    n_edge = n - n_cam - n_int - n_axis
    os_, ds_ = [], []
    # camera rays (camera.rs:53-62 with aperture 0)
    s = rng.rand(n_cam, 1)
    t = rng.rand(n_cam, 1)
    os_.append(np.repeat(origin[None, :], n_cam, axis=0))
    ds_.append(llc[None, :] + s * hor[None, :] + t * ver[None, :] - origin[None, :])
    # interior bounce-like rays
    os_.append(rng.rand(n_int, 3) * 555.0)
    d = rng.randn(n_int, 3)
    ds_.append(d * rng.uniform(0.1, 3.0, size=(n_int, 1)))
    # axis-aligned rays (zero components in the direction)
    o = rng.rand(n_axis, 3) * 555.0
    d = np.zeros((n_axis, 3))
    ax = rng.randint(0, 3, size=n_axis)
    d[np.arange(n_axis), ax] = rng.choice([-1.0, 1.0], size=n_axis) * rng.uniform(0.5, 2.0, size=n_axis)
    os_.append(o)
    ds_.append(d)
    # rays aimed from interior points at corners/edges of the boxes and walls (world-space AABB corners
    # of every node), jittered by a few ulps to straddle the edges
    flat = cornell.flat
    targets = []
    for i in range(flat.n_nodes):
        nd = flat.nodes[i]
        for cx in (nd.bmin[0], nd.bmax[0]):
            for cy in (nd.bmin[1], nd.bmax[1]):
                for cz in (nd.bmin[2], nd.bmax[2]):
                    targets.append((cx, cy, cz))
    targets = np.array(targets)
    o = rng.rand(n_edge, 3) * 500.0 + 25.0
    tgt = targets[rng.randint(0, len(targets), size=n_edge)] + rng.randn(n_edge, 3) * rng.choice([0.0, 1e-4, 1e-2, 1.0], size=(n_edge, 1))
    os_.append(o)
    ds_.append(tgt - o)
    o = np.concatenate(os_).astype(np.float32)
    d = np.concatenate(ds_).astype(np.float32)
    return o, d


def reference_png_fine(sum_rgb, sumsq_rgb, spp, img8):
    """Compares a 600x600 Cornell render with the reference's example PNG at 10x10
    pixel blocks (tests/golden/cornell_box_png_blocks10.npz: block means and pixel
    noise variance of examples/cornell_box.png).  sum/sumsq are the per-pixel
    sample sums and sums of squares at `spp` samples, img8 the tonemapped image.

    Returns (n_hat, diff, ratios): the PNG's sample count estimated from its noise
    (the per-sample variance of our render, carried through the sqrt tonemap,
    divided by the PNG's pixel noise variance, median over the smooth blocks);
    the block-mean differences in 8-bit levels after allowing for the tonemap's
    noise bias (E[sqrt(mean of N samples)] lies below sqrt(mean) by
    255.99 var / (8 N x^1.5), more at our spp than at the PNG's); and the wall
    colour ratios (PNG, ours) over the red-wall and green-wall blocks."""
    g = np.load(os.path.join(GOLDEN, "cornell_box_png_blocks10.npz"))
    mean10, noise10 = g["mean"].astype(np.float64), g["noise_var"].astype(np.float64)
    x = np.asarray(sum_rgb, dtype=np.float64) / spp
    var1 = np.maximum(np.asarray(sumsq_rgb, dtype=np.float64) / spp - x * x, 0.0)
    blk = lambda a: a.reshape(60, 10, 60, 10, 3).mean(axis=(1, 3))  # noqa: E731
    xs = np.clip(x, 1e-4, None)
    pred1 = blk(255.99 ** 2 / (4.0 * xs) * var1)  # pixel noise variance in 8-bit levels at ONE sample per pixel
    xm = blk(x)
    smooth = (xm > 0.05) & (xm < 0.8) & (noise10 > 0.3)
    n_hat = float(np.median(pred1[smooth] / np.maximum(noise10[smooth] - 1.0 / 12.0, 1e-3)))  # 1/12: the 8-bit rounding's own variance
    bias = lambda n: np.where(x < 1.0, 255.99 * var1 / (8.0 * n * xs ** 1.5), 0.0)  # noqa: E731
    corr = np.clip(bias(float(spp)) - bias(n_hat), 0.0, 5.0)
    ours = blk(np.asarray(img8, dtype=np.float64) + corr)
    diff = ours - mean10
    green = (mean10[..., 1] > 1.6 * mean10[..., 0]) & (mean10[..., 1] > 40)
    red = (mean10[..., 0] > 2.0 * mean10[..., 1]) & (mean10[..., 0] > 40)
    ratios = {
        "green_wall_g_over_r": (mean10[green, 1].sum() / mean10[green, 0].sum(), ours[green, 1].sum() / ours[green, 0].sum()),
        "red_wall_r_over_g": (mean10[red, 0].sum() / mean10[red, 1].sum(), ours[red, 0].sum() / ours[red, 1].sum()),
        "blocks": (int(green.sum()), int(red.sum())),
    }
    return n_hat, diff, ratios
