"""ctypes binding of librt_amd.so (include/rt_abi.h, include/rt_scene_api.h).

The library is the product: there is no Python or CPU fallback for any of its
entry points.  If it has not been built, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB lets a tuning session load another build of the same library.
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "csrc", "librt_amd.so"))

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_DEVICE = -2
RT_ERR_UNSUPPORTED = -3
RT_ERR_NO_SCENE = -4

RT_PRIM_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT = 0, 1, 2, 3
RT_PRIM_TYPE_MASK = 0xFF
RT_PRIM_FLIP = 0x100
RT_OBJ_LIST, RT_OBJ_BVH, RT_OBJ_MEDIUM, RT_OBJ_MEDIUM_BVH = 0, 1, 2, 3
RT_XFORM_ROTATE_Y, RT_XFORM_TRANSLATE = 1, 2
RT_MAT_LAMBERTIAN, RT_MAT_DIFFUSE_LIGHT, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_ISOTROPIC = 0, 1, 2, 3, 4
RT_TEX_CONSTANT, RT_TEX_CHECKER, RT_TEX_NOISE = 0, 1, 2
RT_SKY_BLACK, RT_SKY_GRADIENT = 0, 1
RT_SHARD_SAMPLES, RT_SHARD_TILES = 0, 1
RT_LIGHT_XZ_RECT, RT_LIGHT_XY_RECT, RT_LIGHT_YZ_RECT, RT_LIGHT_SPHERE = 0, 1, 2, 3
RT_MULTI_NO_RCCL = 1
RT_MULTI_FORCE_RCCL = 2
(RT_OPT_FIXED_ORDER, RT_OPT_READY_LANES, RT_OPT_HOLD_LANES, RT_OPT_STACK_LDS_ENTRIES, RT_OPT_WAVES_PER_CU,
 RT_OPT_ITEM_ROUNDS, RT_OPT_MIN_CHUNK_SPP, RT_OPT_LAST_CHUNK_SPP, RT_OPT_POOL_SLOTS, RT_OPT_SHADE_BATCH,
 RT_OPT_REFILL_LANES, RT_OPT_TEX_BATCH, RT_OPT_HUNGRY_LANES, RT_OPT_KEEP_TREE) = range(1, 15)


class rt_node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("left", C.c_int32), ("bmax", C.c_float * 3), ("right", C.c_int32)]


class rt_prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("f", C.c_float * 6)]


class rt_object(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("first", C.c_int32), ("count", C.c_int32), ("xform", C.c_int32),
        ("sin_theta", C.c_float), ("cos_theta", C.c_float), ("offset", C.c_float * 3),
        ("density", C.c_float), ("material", C.c_int32), ("medium_index", C.c_int32),
    ]


class rt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_float * 3), ("fuzz", C.c_float),
                ("ref_idx", C.c_float), ("pad", C.c_int32)]


class rt_texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("odd", C.c_int32), ("even", C.c_int32),
                ("scale", C.c_float), ("perlin", C.c_int32)]


class rt_perlin(C.Structure):
    _fields_ = [("ranvec", (C.c_float * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class rt_light(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad", C.c_int32), ("f", C.c_float * 6)]


class rt_scene(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(rt_node)), ("n_nodes", C.c_int32),
        ("prims", C.POINTER(rt_prim)), ("n_prims", C.c_int32),
        ("objects", C.POINTER(rt_object)), ("n_objects", C.c_int32),
        ("materials", C.POINTER(rt_material)), ("n_materials", C.c_int32),
        ("textures", C.POINTER(rt_texture)), ("n_textures", C.c_int32),
        ("perlins", C.POINTER(rt_perlin)), ("n_perlins", C.c_int32),
        ("lights", C.POINTER(rt_light)), ("n_lights", C.c_int32),
    ]


class rt_camera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in
                ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + [("lens_radius", C.c_double)]


class rt_render_params(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("spp", C.c_int32), ("sample_begin", C.c_int32),
        ("max_depth", C.c_int32), ("sky", C.c_int32), ("t_min", C.c_float),
        ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("seed", C.c_uint64),
    ]


class rt_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "rays", "hits", "node_tests", "rect_tests", "sphere_tests", "instance_ops", "shade_events",
                 "light_tests")] + [("kernel_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def make_params(nx, ny, spp, sample_begin=0, max_depth=50, sky=RT_SKY_BLACK, t_min=0.001, shard_index=0,
                shard_count=1, seed=0x5EED0001):
    """rt_render_params with the reference's constants as defaults
    (max depth 50 main.rs:173, t_min 0.001 main.rs:166, black sky main.rs:32)."""
    return rt_render_params(nx, ny, spp, sample_begin, max_depth, sky, t_min, shard_index, shard_count, seed)


_lib = None


def lib():
    """Loads librt_amd.so once; raises if it is missing (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    d3 = C.POINTER(C.c_double)
    vp = C.c_void_p

    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("rt_last_error", C.c_char_p)
    sig("rt_camera_init", C.c_int, C.POINTER(rt_camera), d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_double)
    sig("rt_create", C.c_int, C.c_int, C.POINTER(vp))
    sig("rt_destroy", C.c_int, vp)
    sig("rt_set_option", C.c_int, vp, C.c_int, C.c_int)
    sig("rt_get_option", C.c_int, vp, C.c_int, C.POINTER(C.c_int))
    sig("rt_upload_scene", C.c_int, vp, C.POINTER(rt_scene))
    sig("rt_upload_scene_timed", C.c_int, vp, C.POINTER(rt_scene), C.POINTER(C.c_double))
    sig("rt_render", C.c_int, vp, C.POINTER(rt_camera), C.POINTER(rt_render_params), vp, vp, C.POINTER(rt_stats))
    sig("rt_render_device", C.c_int, vp, C.POINTER(rt_camera), C.POINTER(rt_render_params), vp, vp, vp)
    sig("rt_render_device_noalloc", C.c_int, vp, C.POINTER(rt_camera), C.POINTER(rt_render_params), vp, vp, vp)
    sig("rt_tonemap_device", C.c_int, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp)
    sig("rt_multi_create", C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint, C.POINTER(vp))
    sig("rt_multi_destroy", C.c_int, vp)
    sig("rt_multi_uses_rccl", C.c_int, vp)
    sig("rt_multi_upload_scene", C.c_int, vp, C.POINTER(rt_scene))
    sig("rt_multi_render", C.c_int, vp, C.POINTER(rt_camera), C.POINTER(rt_render_params), C.c_int, vp, vp, vp)
    sig("rt_trace_closest", C.c_int, vp, vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp, vp, vp)
    sig("rt_trace_closest_f64", C.c_int, vp, vp, vp, C.c_size_t, C.c_double, C.c_double, vp, vp, vp, vp)
    sig("rt_tonemap", C.c_int, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp)
    sig("rt_selftest_reciprocal", C.c_int, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32))
    sig("rt_build_lbvh", C.c_int, vp, C.POINTER(rt_scene), vp)

    sig("rtb_create", C.c_int, C.c_uint64, C.POINTER(vp))
    sig("rtb_destroy", C.c_int, vp)
    sig("rtb_texture_constant", C.c_int, vp, C.c_double, C.c_double, C.c_double)
    sig("rtb_texture_checker", C.c_int, vp, C.c_int, C.c_int)
    sig("rtb_texture_noise", C.c_int, vp, C.c_double)
    sig("rtb_lambertian", C.c_int, vp, C.c_int)
    sig("rtb_metal", C.c_int, vp, C.c_double, C.c_double, C.c_double, C.c_double)
    sig("rtb_dielectric", C.c_int, vp, C.c_double)
    sig("rtb_diffuse_light", C.c_int, vp, C.c_int)
    sig("rtb_isotropic", C.c_int, vp, C.c_int)
    sig("rtb_sphere", C.c_int, vp, d3, C.c_double, C.c_int)
    for name in ("rtb_xy_rect", "rtb_xz_rect", "rtb_yz_rect"):
        sig(name, C.c_int, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int)
    sig("rtb_flip_normals", C.c_int, vp, C.c_int)
    sig("rtb_cuboid", C.c_int, vp, d3, d3, C.c_int)
    sig("rtb_translate", C.c_int, vp, C.c_int, d3)
    sig("rtb_rotate_y", C.c_int, vp, C.c_int, C.c_double)
    sig("rtb_constant_medium", C.c_int, vp, C.c_int, C.c_double, C.c_int)
    sig("rtb_hitable_list", C.c_int, vp, C.POINTER(C.c_int), C.c_int)
    sig("rtb_bvh_node", C.c_int, vp, C.POINTER(C.c_int), C.c_int)
    sig("rtb_light_xz_rect", C.c_int, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)
    sig("rtb_light_from_world", C.c_int, vp, C.c_int)
    sig("rtb_light_add", C.c_int, vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)
    sig("rtb_lights_from_world", C.c_int, vp, C.c_int)
    sig("rtb_flatten", C.c_int, vp, C.c_int, C.POINTER(rt_scene))
    _lib = L
    return L


class RtError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "rt error %d: %s" % (code, text))
        self.code = code


def check(rc):
    """Raises RtError for a negative return code; passes handles through."""
    if rc < 0:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))
    return rc


def vec3_arg(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))
