#!/usr/bin/env python3
"""A/B of the two kernels for scenes on the stack traversal (render_stack_kernel: RT_OPT_POOL_SLOTS 0; render_pool_kernel:
the path pool in LDS), in one process on one device: same image (to the order of fp32 additions), same event counts,
production-kernel time by HIP events, for a list of option settings.

  python tools/pool_ab.py final 1920 1080 64 "pool=0" "pool=128" "pool=128,hold=24"
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracing_amd import Renderer, Scene, make_params, _lib  # noqa: E402
from bench_scenes import build  # noqa: E402

NAMES = {"pool": _lib.RT_OPT_POOL_SLOTS, "shade": _lib.RT_OPT_SHADE_BATCH, "refill": _lib.RT_OPT_REFILL_LANES, "tex": _lib.RT_OPT_TEX_BATCH,
         "hungry": _lib.RT_OPT_HUNGRY_LANES, "hold": _lib.RT_OPT_HOLD_LANES, "ready": _lib.RT_OPT_READY_LANES, "lds": _lib.RT_OPT_STACK_LDS_ENTRIES,
         "waves": _lib.RT_OPT_WAVES_PER_CU, "fixed": _lib.RT_OPT_FIXED_ORDER, "rounds": _lib.RT_OPT_ITEM_ROUNDS, "minspp": _lib.RT_OPT_MIN_CHUNK_SPP,
         "lastspp": _lib.RT_OPT_LAST_CHUNK_SPP, "keep": _lib.RT_OPT_KEEP_TREE}


def main():
    scene, nx, ny, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    settings = sys.argv[5:] or ["pool=0", "pool=128"]

    class A(object):
        spheres = 1 << 20
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    sc = Scene(0x5EED0000)
    with sc:
        cam, world, light, sky = build(scene, float(nx) / ny, A)
        if light is not None:
            sc.light_xz_rect(*light)
        flat = sc.flatten(world)
    r = Renderer(0)
    r.upload_scene(flat)
    defaults = {}
    for k, v in NAMES.items():
        try:
            defaults[k] = r.get_option(v)
        except Exception:  # (an older build of the library that does not have this option)
            pass
    p = make_params(nx, ny, spp, sky=sky)
    buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    ref_img, ref_counts = None, None
    check = os.environ.get("AB_CHECK", "1") == "1"
    for s in settings:
        for k, v in defaults.items():
            r.set_option(NAMES[k], v)
        for kv in s.split(","):
            k, v = kv.split("=")
            r.set_option(NAMES[k], int(v))
        times = []
        for k in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            if k > 0:
                times.append(e0.elapsed_time(e1))
        line = "%-44s %9.3f ms (min of 3; %s)" % (s, min(times), " ".join("%.2f" % t for t in times))
        if check:
            img = buf.cpu().numpy().copy()
            _, st = r.render(cam, p, want_stats=True)
            counts = (st.paths, st.rays, st.hits, st.node_tests, st.rect_tests, st.sphere_tests, st.instance_ops, st.shade_events, st.light_tests)
            if ref_img is None:
                ref_img, ref_counts = img, counts
                line += "  [reference] sum %.6e" % float(img.astype(np.float64).sum())
            else:
                bad = int((~np.isclose(img, ref_img, rtol=1e-5, atol=1e-4)).sum())
                line += "  image: %d of %d values beyond fp32 order; counts %s" % (bad, img.size, "equal" if counts == ref_counts else "DIFFER %r vs %r" % (counts, ref_counts))
        print(line, flush=True)
    r.close()


if __name__ == "__main__":
    main()
