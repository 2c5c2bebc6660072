#!/usr/bin/env python3
"""Builds variants/timeline.so: the library with render_kernel recording, per work
item, its start and end time (100 MHz counter) and the hardware unit it ran on.
With RT_TIMELINE_OUT=<file> in the environment every render writes those records
(3 x uint64 per item) to the file; tools/timeline_summary.py reads them.  A tuning
aid only: the patch is applied to a temporary copy of rt_device.hip."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracing_amd", "csrc")


def patch(s):
    def rep(old, new):
        nonlocal s
        assert old in s, old
        s = s.replace(old, new, 1)
    rep("  int n_chunks, chunk_spp;", "  unsigned long long* timeline;\n  int n_chunks, chunk_spp;")
    rep("  const int chunk = item / n_tiles;", "  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();\n  const int chunk = item / n_tiles;")
    rep("  // ---- write the tile (main.rs:112-114; sums, not means) ----", """  if (A.timeline && lane == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    A.timeline[3 * (size_t)item] = tl_start;
    A.timeline[3 * (size_t)item + 1] = __builtin_amdgcn_s_memrealtime();
    A.timeline[3 * (size_t)item + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
  // ---- write the tile (main.rs:112-114; sums, not means) ----""")
    rep("  const DevCamera dc = to_dev_camera(cam);\n  const int n_items = n_tiles * A.n_chunks;", """  const DevCamera dc = to_dev_camera(cam);
  const int n_items = n_tiles * A.n_chunks;
  A.timeline = nullptr;
  static unsigned long long* tl_buf = nullptr;
  static size_t tl_items = 0;
  if (std::getenv("RT_TIMELINE_OUT")) {
    if ((size_t)n_items > tl_items) {
      if (tl_buf) (void)hipFree(tl_buf);
      HIP_TRY(hipMalloc((void**)&tl_buf, (size_t)n_items * 3 * 8));
      tl_items = (size_t)n_items;
    }
    A.timeline = tl_buf;
  }""")
    rep("""  HIP_TRY(hipGetLastError());
  return RT_OK;
}

}  // namespace

int rt::context_device""", """  HIP_TRY(hipGetLastError());
  if (A.timeline) {
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)n_items * 3);
    HIP_TRY(hipMemcpy(h.data(), tl_buf, h.size() * 8, hipMemcpyDeviceToHost));
    FILE* f = std::fopen(std::getenv("RT_TIMELINE_OUT"), "wb");
    if (f) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
  }
  return RT_OK;
}

}  // namespace

int rt::context_device""")
    return s


def main():
    src = patch(open(os.path.join(SRC, "rt_device.hip")).read())
    tmp = os.path.join(SRC, "tmp_timeline.hip")
    open(tmp, "w").write(src)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    try:
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function",
                               "-shared", "-o", os.path.join(ROOT, "variants", "timeline.so"), tmp, os.path.join(SRC, "rt_lbvh.hip"),
                               os.path.join(SRC, "rt_scene.cpp")] + sys.argv[1:], cwd=SRC)
    finally:
        os.remove(tmp)


if __name__ == "__main__":
    main()
