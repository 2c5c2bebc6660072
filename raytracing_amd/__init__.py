"""MI355X-native replacement for the per-pixel path-tracing sample loop of
jesseyoungmann/raytracing (src/main.rs:96-116 calling color(), main.rs:165-209).

Layout:
  csrc/rt_device.hip  gfx950 kernels + the rt_* C ABI (include/rt_abi.h)
  csrc/rt_scene.cpp   host scene classes, BVH build, flattener, rtb_* C handles
  scene.py, scenes.py the reference's constructor names and scene builders
  render.py           Renderer: one rt_ctx per GPU
"""
from ._lib import RtError, make_params  # noqa: F401
from .render import MultiRenderer, Renderer, render_image  # noqa: F401
from .scene import Scene  # noqa: F401
