"""Scene construction with the reference's names and argument order.

Each `X.new(...)` below stands for the Rust constructor `X::new(...)` of the
reference (file:line in the docstrings) and forwards to the C handle API of
librt_amd.so (include/rt_scene_api.h).  Constructors register in the innermost
active `Scene` (a context manager), which plays the role of the Rust program's
heap: it owns the textures, materials and hitables until it is closed.
"""
import ctypes as C

from . import _lib
from ._lib import check, vec3_arg


class Vec3(object):
    """vec3.rs:6-11 (only what scene descriptions need)."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __add__(self, o):
        return Vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __repr__(self):
        return "vec3(%r, %r, %r)" % (self.x, self.y, self.z)


def vec3(x, y, z):
    """vec3.rs:13-15"""
    return Vec3(x, y, z)


def scalar(f):
    """vec3.rs:16-18"""
    return Vec3(f, f, f)


_stack = []


def current_scene():
    if not _stack:
        raise RuntimeError("no active Scene: use `with Scene(seed) as sc:` around scene construction")
    return _stack[-1]


class Scene(object):
    """Owns one rt_builder.  `seed` replaces rand::thread_rng() for the
    construction-time randomness (bvh.rs:107 split axes, perlin.rs:67-79)."""

    def __init__(self, seed=0x5EED0000):
        self._b = C.c_void_p()
        check(_lib.lib().rtb_create(C.c_uint64(seed), C.byref(self._b)))
        self._flat = None

    def __enter__(self):
        _stack.append(self)
        return self

    def __exit__(self, *exc):
        _stack.pop()
        return False

    def close(self):
        if self._b:
            _lib.lib().rtb_destroy(self._b)
            self._b = C.c_void_p()
            self._flat = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._b

    def light_xz_rect(self, x0, x1, z0, z1, k):
        """The XZRect that color() gives to HitablePDF (main.rs:177-178)."""
        return check(_lib.lib().rtb_light_xz_rect(self._b, x0, x1, z0, z1, k))

    def light_add(self, kind, a, b, c, d, e=0.0):
        """Appends a shape to the list HitablePDF samples (beyond the reference,
        whose color() samples one XZRect): kind is "xz_rect", "xy_rect", "yz_rect"
        (a0, a1, b0, b1, k in the constructor's order) or "sphere" (cx, cy, cz, r).
        Returns the number of shapes registered."""
        types = {"xz_rect": _lib.RT_LIGHT_XZ_RECT, "xy_rect": _lib.RT_LIGHT_XY_RECT, "yz_rect": _lib.RT_LIGHT_YZ_RECT,
                 "sphere": _lib.RT_LIGHT_SPHERE}
        return check(_lib.lib().rtb_light_add(self._b, types[kind], a, b, c, d, e))

    def lights_from_world(self, world):
        """Registers every emissive rect (XY, XZ, YZ, bare or flipped) and sphere of
        `world` as a light shape, replacing any set before.  Returns the count."""
        if world.scene is not self:
            raise ValueError("world belongs to another Scene")
        return check(_lib.lib().rtb_lights_from_world(self._b, world.handle))

    def light_from_world(self, world):
        """Takes the light shape from the scene: the first emissive XZRect of
        `world` (bare or under FlipNormals).  Returns how many there are."""
        if world.scene is not self:
            raise ValueError("world belongs to another Scene")
        return check(_lib.lib().rtb_light_from_world(self._b, world.handle))

    def flatten(self, world):
        """BvhNode over world.list (main.rs:85-90), flattened to an rt_scene whose
        buffers live as long as this Scene does (or until the next flatten)."""
        if world.scene is not self:
            raise ValueError("world belongs to another Scene")
        flat = _lib.rt_scene()
        check(_lib.lib().rtb_flatten(self._b, world.handle, C.byref(flat)))
        self._flat = flat
        flat._keep = self  # the struct points into this Scene's buffers: keep it alive as long as the struct
        return flat


class _Handle(object):
    __slots__ = ("scene", "handle")

    def __init__(self, scene, handle):
        self.scene, self.handle = scene, handle


class Texture(_Handle):
    """texture.rs:5-17"""

    @staticmethod
    def new_constant(color):
        """texture.rs:36-38"""
        sc = current_scene()
        return Texture(sc, check(_lib.lib().rtb_texture_constant(sc.handle, color[0], color[1], color[2])))

    @staticmethod
    def new_checker(odd, even):
        """texture.rs:40-45"""
        sc = current_scene()
        return Texture(sc, check(_lib.lib().rtb_texture_checker(sc.handle, odd.handle, even.handle)))

    @staticmethod
    def new_noise(scale):
        """texture.rs:47-52"""
        sc = current_scene()
        return Texture(sc, check(_lib.lib().rtb_texture_noise(sc.handle, scale)))


class Material(_Handle):
    """material.rs:11-17"""


class Lambertian(object):
    @staticmethod
    def new(texture):
        """material.rs:57-59"""
        sc = current_scene()
        return Material(sc, check(_lib.lib().rtb_lambertian(sc.handle, texture.handle)))

    @staticmethod
    def new_from_color(albedo):
        """material.rs:61-65"""
        return Lambertian.new(Texture.new_constant(albedo))


class Metal(object):
    @staticmethod
    def new(albedo, fuzz):
        """material.rs:93-98 (commented out in the reference)"""
        sc = current_scene()
        return Material(sc, check(_lib.lib().rtb_metal(sc.handle, albedo[0], albedo[1], albedo[2], fuzz)))


class Dielectric(object):
    @staticmethod
    def new(ref_idx):
        """material.rs:121-123 (commented out in the reference)"""
        sc = current_scene()
        return Material(sc, check(_lib.lib().rtb_dielectric(sc.handle, ref_idx)))


class DiffuseLight(object):
    @staticmethod
    def new(emit):
        """material.rs:189-191"""
        sc = current_scene()
        return Material(sc, check(_lib.lib().rtb_diffuse_light(sc.handle, emit.handle)))


class Isotropic(object):
    @staticmethod
    def new(albedo):
        """material.rs:213-215 (commented out in the reference)"""
        sc = current_scene()
        return Material(sc, check(_lib.lib().rtb_isotropic(sc.handle, albedo.handle)))


class Hitable(_Handle):
    """hitable.rs:19-30"""


class Sphere(object):
    @staticmethod
    def new(center, radius, material):
        """hitable.rs:40-46"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_sphere(sc.handle, vec3_arg(center), radius, material.handle)))


class XYRect(object):
    @staticmethod
    def new(x0, x1, y0, y1, k, material):
        """hitable.rs:151-160"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_xy_rect(sc.handle, x0, x1, y0, y1, k, material.handle)))


class XZRect(object):
    @staticmethod
    def new(x0, x1, z0, z1, k, material):
        """hitable.rs:204-213"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_xz_rect(sc.handle, x0, x1, z0, z1, k, material.handle)))


class YZRect(object):
    @staticmethod
    def new(y0, y1, z0, z1, k, material):
        """hitable.rs:278-287"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_yz_rect(sc.handle, y0, y1, z0, z1, k, material.handle)))


class FlipNormals(object):
    """hitable.rs:329-339: the three constructors differ only in the rect type
    they accept; the handle API checks that the argument is a rect."""

    @staticmethod
    def _new(rect):
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_flip_normals(sc.handle, rect.handle)))

    new_xy = _new
    new_xz = _new
    new_yz = _new


class Cuboid(object):
    @staticmethod
    def new(p0, p1, material):
        """hitable.rs:380-423"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_cuboid(sc.handle, vec3_arg(p0), vec3_arg(p1), material.handle)))


class Translate(object):
    @staticmethod
    def new(hitable, offset):
        """hitable.rs:433-435"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_translate(sc.handle, hitable.handle, vec3_arg(offset))))


class RotateY(object):
    @staticmethod
    def new(hitable, angle):
        """hitable.rs:465-505 (angle in degrees)"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_rotate_y(sc.handle, hitable.handle, angle)))


class ConstantMedium(object):
    @staticmethod
    def new(boundary, density, a):
        """hitable.rs:551-557 (commented out in the reference)"""
        sc = current_scene()
        return Hitable(sc, check(_lib.lib().rtb_constant_medium(sc.handle, boundary.handle, density, a.handle)))


def _handles(items):
    arr = (C.c_int * len(items))(*[h.handle for h in items])
    return arr, len(items)


class HitableList(Hitable):
    __slots__ = ("list",)

    @staticmethod
    def new(items):
        """hitable.rs:100-102"""
        sc = current_scene()
        items = list(items)
        arr, n = _handles(items)
        out = HitableList(sc, check(_lib.lib().rtb_hitable_list(sc.handle, arr, n)))
        out.list = items
        return out


class BvhNode(object):
    @staticmethod
    def new(items, time0=0.0, time1=0.0):
        """bvh.rs:106-146 (time0/time1 are unused there as well)"""
        sc = current_scene()
        arr, n = _handles(list(items))
        return Hitable(sc, check(_lib.lib().rtb_bvh_node(sc.handle, arr, n)))


class Camera(object):
    """camera.rs:5-14"""

    def __init__(self, c_camera):
        self.c = c_camera

    @staticmethod
    def new(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
        """camera.rs:17-51"""
        cam = _lib.rt_camera()
        check(_lib.lib().rt_camera_init(C.byref(cam), vec3_arg(lookfrom), vec3_arg(lookat), vec3_arg(vup),
                                        vfov, aspect, aperture, focus_dist))
        return Camera(cam)
