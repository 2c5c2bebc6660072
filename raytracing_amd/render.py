"""The sample loop of src/main.rs:67-133 behind the C ABI.

`Renderer` wraps one rt_ctx (one GPU).  There is no CPU implementation here:
every method ends in a call into librt_amd.so and raises RtError when the
library or the device is not usable.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, make_params


class Renderer(object):
    def __init__(self, device_id=0):
        self._ctx = C.c_void_p()
        check(_lib.lib().rt_create(device_id, C.byref(self._ctx)))
        self.device_id = device_id

    def close(self):
        if self._ctx:
            _lib.lib().rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, flat_scene):
        """rt_upload_scene; the wall time of its two parts (host build of the
        device's tree views, device copies) is kept in `upload_ms`."""
        ms = (C.c_double * 2)()
        check(_lib.lib().rt_upload_scene_timed(self._ctx, C.byref(flat_scene), ms))
        self.upload_ms = (float(ms[0]), float(ms[1]))

    def set_option(self, option, value):
        """rt_set_option: a context option (RT_OPT_* of rt_abi.h); out-of-range
        values are clamped by the library."""
        check(_lib.lib().rt_set_option(self._ctx, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int(0)
        check(_lib.lib().rt_get_option(self._ctx, int(option), C.byref(v)))
        return int(v.value)

    def render(self, camera, params, want_sumsq=False, want_stats=False):
        """Blocking render to host arrays: per-pixel sums, shape (ny, nx, 3),
        top row first; optionally sums of squares and the event counters."""
        shape = (params.ny, params.nx, 3)
        s = np.zeros(shape, dtype=np.float32)
        s2 = np.zeros(shape, dtype=np.float32) if want_sumsq else None
        stats = _lib.rt_stats() if want_stats else None
        check(_lib.lib().rt_render(self._ctx, C.byref(camera.c), C.byref(params), s.ctypes.data,
                                   s2.ctypes.data if want_sumsq else None,
                                   C.byref(stats) if want_stats else None))
        out = [s]
        if want_sumsq:
            out.append(s2)
        if want_stats:
            out.append(stats)
        return tuple(out) if len(out) > 1 else s

    def render_device(self, camera, params, d_sum_ptr, d_sumsq_ptr=None, stream=None):
        """Enqueues the render on `stream` (a hipStream_t as int) writing into
        device memory the caller owns (for example a torch tensor's data_ptr())."""
        check(_lib.lib().rt_render_device(self._ctx, C.byref(camera.c), C.byref(params), C.c_void_p(d_sum_ptr),
                                          C.c_void_p(d_sumsq_ptr) if d_sumsq_ptr else None,
                                          C.c_void_p(stream) if stream else None))

    def render_device_noalloc(self, camera, params, d_sum_ptr, d_sumsq_ptr=None, stream=None):
        """render_device through the entry point that allocates nothing (its
        scratch is a static device array; launches whose partial images exceed
        it raise RtError with RT_ERR_UNSUPPORTED)."""
        check(_lib.lib().rt_render_device_noalloc(self._ctx, C.byref(camera.c), C.byref(params), C.c_void_p(d_sum_ptr),
                                                  C.c_void_p(d_sumsq_ptr) if d_sumsq_ptr else None,
                                                  C.c_void_p(stream) if stream else None))

    def tonemap_device(self, d_sum_ptr, nx, ny, spp_total, d_rgb8_ptr, stream=None):
        """main.rs:136-155 on device-resident buffers, enqueued on `stream`."""
        check(_lib.lib().rt_tonemap_device(self._ctx, C.c_void_p(d_sum_ptr), nx, ny, spp_total, C.c_void_p(d_rgb8_ptr),
                                           C.c_void_p(stream) if stream else None))

    def trace_closest(self, o, d, t_min=0.001, t_max=float("inf"), want_normal=False, want_p=False):
        """world.hit(r, t_min, t_max) of main.rs:166 for a batch of rays."""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        if o.ndim != 2 or o.shape[1] != 3 or o.shape != d.shape:
            raise ValueError("o and d must both have shape (n, 3)")
        n = o.shape[0]
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        normal = np.zeros((n, 3), dtype=np.float32) if want_normal else None
        p = np.zeros((n, 3), dtype=np.float32) if want_p else None
        check(_lib.lib().rt_trace_closest(self._ctx, o.ctypes.data, d.ctypes.data, n, t_min, t_max, prim.ctypes.data,
                                          t.ctypes.data, normal.ctypes.data if want_normal else None,
                                          p.ctypes.data if want_p else None))
        out = [prim, t]
        if want_normal:
            out.append(normal)
        if want_p:
            out.append(p)
        return tuple(out)

    def trace_closest_f64(self, o, d, t_min=0.001, t_max=float("inf"), want_normal=False, want_p=False):
        """The same query in the reference's own arithmetic (f64, the source's
        divisions and traversal order): rt_trace_closest_f64, the parity mode."""
        o = np.ascontiguousarray(o, dtype=np.float64)
        d = np.ascontiguousarray(d, dtype=np.float64)
        if o.ndim != 2 or o.shape[1] != 3 or o.shape != d.shape:
            raise ValueError("o and d must both have shape (n, 3)")
        n = o.shape[0]
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        normal = np.zeros((n, 3), dtype=np.float64) if want_normal else None
        p = np.zeros((n, 3), dtype=np.float64) if want_p else None
        check(_lib.lib().rt_trace_closest_f64(self._ctx, o.ctypes.data, d.ctypes.data, n, t_min, t_max, prim.ctypes.data,
                                              t.ctypes.data, normal.ctypes.data if want_normal else None,
                                              p.ctypes.data if want_p else None))
        out = [prim, t]
        if want_normal:
            out.append(normal)
        if want_p:
            out.append(p)
        return tuple(out)

    def tonemap(self, sum_rgb, spp_total):
        """main.rs:136-155 on the device: u8 RGB, shape (ny, nx, 3)."""
        a = np.ascontiguousarray(sum_rgb, dtype=np.float32)
        ny, nx = a.shape[0], a.shape[1]
        out = np.zeros((ny, nx, 3), dtype=np.uint8)
        check(_lib.lib().rt_tonemap(self._ctx, a.ctypes.data, nx, ny, spp_total, out.ctypes.data))
        return out

    def build_lbvh(self, flat_scene):
        """The scene with its world tree rebuilt on the device as a linear BVH
        (rt_build_lbvh): a copy of `flat_scene` whose node array belongs to the
        returned object; the other arrays are shared with `flat_scene`."""
        nodes = (_lib.rt_node * flat_scene.n_nodes)()
        check(_lib.lib().rt_build_lbvh(self._ctx, C.byref(flat_scene), nodes))
        out = _lib.rt_scene()
        for name, _ in _lib.rt_scene._fields_:
            setattr(out, name, getattr(flat_scene, name))
        out.nodes = C.cast(nodes, C.POINTER(_lib.rt_node))
        out._keep = (nodes, flat_scene)  # the arrays must outlive the struct that points at them
        return out

    def selftest_reciprocal(self):
        """Runs the kernels' reciprocal sequence against the correctly rounded
        division for all 2^32 float bit patterns on the device.  Returns
        (mismatch count, up to 16 example bit patterns)."""
        import ctypes as C
        count = C.c_uint64(0)
        bad = (C.c_uint32 * 16)()
        check(_lib.lib().rt_selftest_reciprocal(self._ctx, C.byref(count), bad))
        return int(count.value), [int(b) for b in bad]


class MultiRenderer(object):
    """One render over several GPUs from this process (rt_multi_*): the thread
    fan-out and the image sum of main.rs:70-133 with GPUs for threads.  A device id
    may repeat (ranks that share a GPU)."""

    def __init__(self, device_ids, no_rccl=False, force_rccl=False):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._m = C.c_void_p()
        self.n = len(device_ids)
        flags = (_lib.RT_MULTI_NO_RCCL if no_rccl else 0) | (_lib.RT_MULTI_FORCE_RCCL if force_rccl else 0)
        check(_lib.lib().rt_multi_create(ids, len(device_ids), flags, C.byref(self._m)))

    @property
    def uses_rccl(self):
        return bool(_lib.lib().rt_multi_uses_rccl(self._m))

    def close(self):
        if self._m:
            _lib.lib().rt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, flat_scene):
        check(_lib.lib().rt_multi_upload_scene(self._m, C.byref(flat_scene)))

    def render(self, camera, params, mode="samples", want_sumsq=False, want_stats=False):
        """Sums (ny, nx, 3), top row first; optionally sums of squares and the
        per-rank event counters."""
        shape = (params.ny, params.nx, 3)
        s = np.zeros(shape, dtype=np.float32)
        s2 = np.zeros(shape, dtype=np.float32) if want_sumsq else None
        stats = (_lib.rt_stats * self.n)() if want_stats else None
        shard = {"samples": _lib.RT_SHARD_SAMPLES, "tiles": _lib.RT_SHARD_TILES}[mode]
        check(_lib.lib().rt_multi_render(self._m, C.byref(camera.c), C.byref(params), shard, s.ctypes.data,
                                         s2.ctypes.data if want_sumsq else None, stats))
        out = [s]
        if want_sumsq:
            out.append(s2)
        if want_stats:
            out.append(list(stats))
        return tuple(out) if len(out) > 1 else s


def render_image(scene, camera, world, nx, ny, ns, seed=0x5EED0001, light_shape=None, device_id=0, png_path=None):
    """main(): flatten `world`, render ns samples per pixel, return the 8-bit
    image (main.rs:65-155) and, with `png_path`, write it as the reference does
    (main.rs:157-160).  `light_shape` = (x0, x1, z0, z1, k) is the XZRect color()
    samples (main.rs:177-178); None renders with CosinePDF alone."""
    if light_shape is not None:
        scene.light_xz_rect(*light_shape)
    flat = scene.flatten(world)
    r = Renderer(device_id)
    try:
        r.upload_scene(flat)
        sums = r.render(camera, make_params(nx, ny, ns, seed=seed))
        img = r.tonemap(sums, ns)
        if png_path is not None:
            from .png import write_png
            write_png(png_path, img)
        return img
    finally:
        r.close()
