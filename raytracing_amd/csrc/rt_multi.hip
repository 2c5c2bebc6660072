// rt_multi.hip -- one render over several GPUs of a node from ONE process: the
// thread fan-out and the 8-way image sum of main.rs:70-133, with GPUs for threads.
//
// A multi-context holds one rt_ctx, one stream and one frame buffer per rank.  A
// rank is a device id; the same id may appear more than once ("virtual ranks" that
// share a GPU: how the split is tested on a one-GPU machine).
//   RT_SHARD_SAMPLES: rank g renders samples [g*spp/N, (g+1)*spp/N) of every pixel
//                     (what main.rs:74-78 does with threads); the frames are SUMMED
//                     onto rank 0 (main.rs:128-133).
//   RT_SHARD_TILES:   rank g renders the 8x8-pixel tiles t with t % N == g (numbered as
//                     rt_abi.h describes for rt_render_params.shard_index) at the
//                     full sample count; each rank packs its own tiles into a
//                     contiguous buffer and rank 0 GATHERS them: nx*ny*12/N bytes per
//                     rank on the wire and no arithmetic.
// The exchange is RCCL (ncclReduce / grouped ncclSend+ncclRecv, one communicator per
// rank from ncclCommInitAll, each on its rank's stream) when all device ids are
// distinct; with repeated ids (or RT_MULTI_NO_RCCL) the frames are copied to rank
// 0's device and added or unpacked there in rank order.  librccl.so is opened with
// dlopen when the first multi-context of more than one distinct device is created:
// single-GPU users of the library never load it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_internal.hpp"

namespace {

int hip_fail(hipError_t e, const char* what) { return rt::set_error(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

// ---- the few RCCL entry points used, resolved at run time (rccl.h's own types are plain: an opaque
// communicator pointer, enums passed as int) ----
typedef void* comm_t;
struct Rccl {
  void* handle;
  int (*CommInitAll)(comm_t*, int, const int*);
  int (*CommDestroy)(comm_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  int (*Reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
};
const int kNcclFloat = 7;  // ncclFloat32 in rccl.h's ncclDataType_t
const int kNcclSum = 0;    // ncclSum in ncclRedOp_t

// Why the library could not be used, kept from the failing dlopen/dlsym itself: dlerror() hands its message out once
// and any later library call may replace it.
std::string& rccl_load_error() {
  static std::string text;
  return text;
}
Rccl* load_rccl() {
  static Rccl r;
  static bool tried = false, ok = false;
  if (tried) return ok ? &r : nullptr;
  tried = true;
  r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!r.handle) {
    const char* why = dlerror();
    rccl_load_error() = why ? why : "dlopen failed";
    return nullptr;
  }
#define RT_SYM(field, name)                                        \
  *(void**)(&r.field) = dlsym(r.handle, name);                     \
  if (!r.field) {                                                  \
    const char* why = dlerror();                                   \
    rccl_load_error() = why ? why : "missing symbol " name;        \
    return nullptr;                                                \
  }
  RT_SYM(CommInitAll, "ncclCommInitAll")
  RT_SYM(CommDestroy, "ncclCommDestroy")
  RT_SYM(GroupStart, "ncclGroupStart")
  RT_SYM(GroupEnd, "ncclGroupEnd")
  RT_SYM(Reduce, "ncclReduce")
  RT_SYM(Send, "ncclSend")
  RT_SYM(Recv, "ncclRecv")
  RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
  ok = true;
  return &r;
}

// ---- kernels of the exchange ----
// dst += src, float4 at a time (n4 float4s), then the tail.
__global__ void __launch_bounds__(256) add_frames_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = ((float4*)dst)[i];
    const float4 b = ((const float4*)src)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    ((float4*)dst)[i] = a;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// The k-th tile of shard `index` of `count` is tile number index + k*count of a grid `width` tiles wide (rt_shard_width).
// Packed layout: tile k at packed[k * 8*8*3], rows of 8 pixels; tiles and pixels beyond the frame's edge unused.
// PACK copies frame -> packed, !PACK copies packed -> frame.  One thread per float of a tile, one tile per 192 threads.
template <bool PACK>
__global__ void __launch_bounds__(192) move_blocks_kernel(float* __restrict__ frame, float* __restrict__ packed, int nx, int ny, int index,
                                                         int count, int width, int n_mine) {
  const int k = blockIdx.x;
  if (k >= n_mine) return;
  const int t = index + k * count;
  const int x0 = (t % width) * 8, y0 = (t / width) * 8;
  const int c = threadIdx.x % 24, row = threadIdx.x / 24;  // float within the tile's row (8 pixels x 3), row of the tile
  const int y = y0 + row;
  if (y >= ny || x0 + c / 3 >= nx) return;
  float* f = frame + ((size_t)y * nx + x0) * 3 + c;
  float* p = packed + (size_t)k * 192 + threadIdx.x;
  if (PACK) *p = *f;
  else *f = *p;
}

}  // namespace

struct rt_multi {
  int n;
  std::vector<int> device;
  std::vector<rt_ctx*> ctx;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> done, t0, t1;
  std::vector<float*> frame, frame2;    // per rank: sums (and sums of squares), nx*ny*3 floats, grown on demand
  std::vector<float*> packed, packed2;  // per rank: its blocks, contiguous (tile mode)
  std::vector<size_t> frame_floats, frame2_floats, packed_floats, packed2_floats;
  float* stage;  // on rank 0's device: one other rank's frame or packed blocks at a time (copy path), or all of them (RCCL gather)
  size_t stage_floats;
  bool use_rccl;
  Rccl* rccl;
  std::vector<comm_t> comm;
  std::string rccl_note;  // why RCCL is not in use although it was wanted
};

namespace {

int grow(int device, float** buf, size_t* have, size_t need) {
  if (need <= *have) return RT_OK;
  HIP_TRY(hipSetDevice(device));
  if (*buf) HIP_TRY(hipFree(*buf));
  *buf = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc((void**)buf, need * sizeof(float)));
  *have = need;
  return RT_OK;
}

int nccl_fail(const Rccl* r, int code, const char* what) {
  return rt::set_error(RT_ERR_DEVICE, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "RCCL error"));
}
#define NCCL_TRY(m, expr)                                      \
  do {                                                         \
    const int c_ = (expr);                                     \
    if (c_ != 0) return nccl_fail((m)->rccl, c_, #expr);       \
  } while (0)

// how many tile numbers shard `index` of `count` has (the launch's own count, rt_device.hip)
int blocks_of(int nx, int ny, int index, int count) {
  const long long numbers = (long long)rt_shard_width(nx, count) * ((ny + 7) / 8);
  return index < numbers ? (int)((numbers - index + count - 1) / count) : 0;
}

// Sums (samples) or gathers (tiles) the ranks' buffers `buf` (frames) onto rank 0's frame.
int exchange(rt_multi* m, const rt_render_params* p, int shard_mode, std::vector<float*>& frame, std::vector<float*>& packed,
             std::vector<size_t>& packed_floats) {
  const size_t n = (size_t)p->nx * p->ny * 3;
  const int N = m->n;
  // One rank has nothing to exchange.  With RT_MULTI_FORCE_RCCL it still goes through the calls (a reduce onto itself, an
  // empty send/receive group), so that the resolved entry points, their signatures and the enum values run on one GPU.
  if (N == 1 && !m->use_rccl) return RT_OK;
  const size_t block_floats = 8 * 8 * 3;
  if (shard_mode == RT_SHARD_TILES) {
    // every other rank packs its blocks
    for (int r = 1; r < N; ++r) {
      const int mine = blocks_of(p->nx, p->ny, r, N);
      int rc = grow(m->device[r], &packed[r], &packed_floats[r], (size_t)(mine > 0 ? mine : 1) * block_floats);
      if (rc != RT_OK) return rc;
      if (mine == 0) continue;
      HIP_TRY(hipSetDevice(m->device[r]));
      hipLaunchKernelGGL(move_blocks_kernel<true>, dim3((unsigned)mine), dim3(192), 0, m->stream[r], frame[r], packed[r], p->nx, p->ny, r, N, rt_shard_width(p->nx, N), mine);
      HIP_TRY(hipGetLastError());
    }
  }
  if (m->use_rccl) {
    if (shard_mode == RT_SHARD_SAMPLES) {
      NCCL_TRY(m, m->rccl->GroupStart());
      for (int r = 0; r < N; ++r) {
        HIP_TRY(hipSetDevice(m->device[r]));
        NCCL_TRY(m, m->rccl->Reduce(frame[r], frame[r], n, kNcclFloat, kNcclSum, 0, m->comm[r], m->stream[r]));
      }
      NCCL_TRY(m, m->rccl->GroupEnd());
      return RT_OK;
    }
    // tiles: rank 0 receives every other rank's packed blocks into its staging buffer, then unpacks them
    size_t total = 0;
    std::vector<size_t> at((size_t)N, 0);
    for (int r = 1; r < N; ++r) {
      at[(size_t)r] = total;
      total += (size_t)blocks_of(p->nx, p->ny, r, N) * block_floats;
    }
    int rc = grow(m->device[0], &m->stage, &m->stage_floats, total ? total : 1);
    if (rc != RT_OK) return rc;
    NCCL_TRY(m, m->rccl->GroupStart());
    for (int r = 1; r < N; ++r) {
      const size_t cnt = (size_t)blocks_of(p->nx, p->ny, r, N) * block_floats;
      if (cnt == 0) continue;
      HIP_TRY(hipSetDevice(m->device[r]));
      NCCL_TRY(m, m->rccl->Send(packed[r], cnt, kNcclFloat, 0, m->comm[r], m->stream[r]));
      HIP_TRY(hipSetDevice(m->device[0]));
      NCCL_TRY(m, m->rccl->Recv(m->stage + at[(size_t)r], cnt, kNcclFloat, r, m->comm[0], m->stream[0]));
    }
    NCCL_TRY(m, m->rccl->GroupEnd());
    HIP_TRY(hipSetDevice(m->device[0]));
    for (int r = 1; r < N; ++r) {
      const int mine = blocks_of(p->nx, p->ny, r, N);
      if (mine == 0) continue;
      hipLaunchKernelGGL(move_blocks_kernel<false>, dim3((unsigned)mine), dim3(192), 0, m->stream[0], frame[0], m->stage + at[(size_t)r], p->nx, p->ny, r, N, rt_shard_width(p->nx, N), mine);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
  }
  // Copy path: rank 0's stream takes the other ranks one after the other, in rank order (so the sum's order is fixed):
  // wait for the rank's work, copy its buffer to rank 0's device, add or unpack.
  for (int r = 1; r < N; ++r) {
    const int mine = shard_mode == RT_SHARD_TILES ? blocks_of(p->nx, p->ny, r, N) : 0;
    const size_t cnt = shard_mode == RT_SHARD_TILES ? (size_t)mine * block_floats : n;
    HIP_TRY(hipSetDevice(m->device[r]));
    HIP_TRY(hipEventRecord(m->done[r], m->stream[r]));
    if (cnt == 0) continue;
    int rc = grow(m->device[0], &m->stage, &m->stage_floats, cnt);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(m->device[0]));
    HIP_TRY(hipStreamWaitEvent(m->stream[0], m->done[r], 0));
    const float* src = shard_mode == RT_SHARD_TILES ? packed[r] : frame[r];
    HIP_TRY(hipMemcpyAsync(m->stage, src, cnt * sizeof(float), hipMemcpyDeviceToDevice, m->stream[0]));
    if (shard_mode == RT_SHARD_TILES) {
      hipLaunchKernelGGL(move_blocks_kernel<false>, dim3((unsigned)mine), dim3(192), 0, m->stream[0], frame[0], m->stage, p->nx, p->ny, r, N, rt_shard_width(p->nx, N), mine);
    } else {
      unsigned blocks = (unsigned)((n / 4 + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      if (blocks == 0) blocks = 1;
      hipLaunchKernelGGL(add_frames_kernel, dim3(blocks), dim3(256), 0, m->stream[0], frame[0], m->stage, n);
    }
    HIP_TRY(hipGetLastError());
  }
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_multi_destroy(rt_multi* m) {
  if (!m) return RT_OK;
  for (int r = 0; r < m->n; ++r) {
    if (!m->ctx[(size_t)r]) continue;  // never created (rt_multi_create failed at this rank): its device id may not even exist
    (void)hipSetDevice(m->device[(size_t)r]);
    if ((size_t)r < m->comm.size() && m->comm[(size_t)r]) (void)m->rccl->CommDestroy(m->comm[(size_t)r]);
    float* bufs[] = {m->frame[(size_t)r], m->frame2[(size_t)r], m->packed[(size_t)r], m->packed2[(size_t)r]};
    for (float* b : bufs)
      if (b) (void)hipFree(b);
    if (m->stream[(size_t)r]) (void)hipStreamDestroy(m->stream[(size_t)r]);
    if (m->done[(size_t)r]) (void)hipEventDestroy(m->done[(size_t)r]);
    if (m->t0[(size_t)r]) (void)hipEventDestroy(m->t0[(size_t)r]);
    if (m->t1[(size_t)r]) (void)hipEventDestroy(m->t1[(size_t)r]);
    if (m->ctx[(size_t)r]) (void)rt_destroy(m->ctx[(size_t)r]);
  }
  if (m->stage) {
    (void)hipSetDevice(m->device[0]);
    (void)hipFree(m->stage);
  }
  (void)hipGetLastError();  // nothing of the teardown is left for a later call to trip over
  delete m;
  return RT_OK;
}

int rt_multi_create(const int* device_ids, int n_devices, unsigned flags, rt_multi** out) {
  if (!out || !device_ids || n_devices < 1 || n_devices > 64) return rt::set_error(RT_ERR_INVALID, "rt_multi_create: bad argument");
  if (flags & ~(unsigned)(RT_MULTI_NO_RCCL | RT_MULTI_FORCE_RCCL)) return rt::set_error(RT_ERR_INVALID, "rt_multi_create: unknown flag");
  rt_multi* m = new rt_multi();
  const size_t N = (size_t)n_devices;
  m->n = n_devices;
  m->device.assign(device_ids, device_ids + n_devices);
  m->ctx.assign(N, nullptr);
  m->stream.assign(N, nullptr);
  m->done.assign(N, nullptr); m->t0.assign(N, nullptr); m->t1.assign(N, nullptr);
  m->frame.assign(N, nullptr); m->frame2.assign(N, nullptr); m->packed.assign(N, nullptr); m->packed2.assign(N, nullptr);
  m->frame_floats.assign(N, 0); m->frame2_floats.assign(N, 0); m->packed_floats.assign(N, 0); m->packed2_floats.assign(N, 0);
  m->stage = nullptr;
  m->stage_floats = 0;
  m->use_rccl = false;
  m->rccl = nullptr;
  for (int r = 0; r < n_devices; ++r) {
    int rc = rt_create(device_ids[r], &m->ctx[(size_t)r]);
    hipError_t e = hipSuccess;
    if (rc == RT_OK) e = hipStreamCreateWithFlags(&m->stream[(size_t)r], hipStreamNonBlocking);
    if (rc == RT_OK && e == hipSuccess) e = hipEventCreateWithFlags(&m->done[(size_t)r], hipEventDisableTiming);
    if (rc == RT_OK && e == hipSuccess) e = hipEventCreate(&m->t0[(size_t)r]);
    if (rc == RT_OK && e == hipSuccess) e = hipEventCreate(&m->t1[(size_t)r]);
    if (rc == RT_OK && e != hipSuccess) rc = hip_fail(e, "rt_multi_create");
    if (rc != RT_OK) {
      rt_multi_destroy(m);
      return rc;
    }
  }
  bool distinct = true;
  for (int a = 0; a < n_devices; ++a)
    for (int b = a + 1; b < n_devices; ++b)
      if (device_ids[a] == device_ids[b]) distinct = false;
  // RCCL only where there is something to exchange between distinct devices (a one-rank context never loads it, unless
  // RT_MULTI_FORCE_RCCL asks for the calls to be exercised).  When the library cannot be loaded or initialised the
  // context falls back to the exchange by device copies (which every test covers) and rt_multi_uses_rccl() says so;
  // with RT_MULTI_FORCE_RCCL that is an error instead.
  const bool force = (flags & RT_MULTI_FORCE_RCCL) != 0;
  if (distinct && !(flags & RT_MULTI_NO_RCCL) && (n_devices > 1 || force)) {
    std::string why;
    m->rccl = load_rccl();
    if (!m->rccl) {
      why = "cannot load librccl.so (" + rccl_load_error() + ")";
    } else {
      m->comm.assign(N, nullptr);
      const int c = m->rccl->CommInitAll(m->comm.data(), n_devices, device_ids);
      if (c != 0) {
        why = std::string("ncclCommInitAll: ") + (m->rccl->GetErrorString ? m->rccl->GetErrorString(c) : "RCCL error");
        m->comm.clear();
      } else {
        m->use_rccl = true;
      }
    }
    if (!m->use_rccl) {
      m->rccl_note = why;
      if (force) {
        rt_multi_destroy(m);
        return rt::set_error(RT_ERR_DEVICE, "rt_multi_create: " + why);
      }
      std::fprintf(stderr, "rt_multi_create: %s; exchanging by device copies instead\n", why.c_str());
    }
  }
  *out = m;
  return RT_OK;
}

int rt_multi_uses_rccl(const rt_multi* m) { return m && m->use_rccl ? 1 : 0; }

int rt_multi_upload_scene(rt_multi* m, const rt_scene* scene) {
  if (!m) return rt::set_error(RT_ERR_INVALID, "rt_multi_upload_scene: null context");
  for (int r = 0; r < m->n; ++r) {
    const int rc = rt_upload_scene(m->ctx[(size_t)r], scene);  // the scene (at most ~100 MB) is replicated on every rank
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_render_params* params, int shard_mode, float* out_sum, float* out_sumsq,
                    rt_stats* stats_per_rank) {
  if (!m || !cam || !params || !out_sum) return rt::set_error(RT_ERR_INVALID, "rt_multi_render: null argument");
  if (shard_mode != RT_SHARD_SAMPLES && shard_mode != RT_SHARD_TILES) return rt::set_error(RT_ERR_INVALID, "rt_multi_render: unknown shard mode");
  if (params->shard_count != 1 || params->shard_index != 0) return rt::set_error(RT_ERR_INVALID, "rt_multi_render: params must describe the whole image (shard_count 1)");
  if (params->nx <= 0 || params->ny <= 0) return rt::set_error(RT_ERR_INVALID, "image size out of range");
  const int N = m->n;
  if (shard_mode == RT_SHARD_SAMPLES && params->spp < N) return rt::set_error(RT_ERR_INVALID, "rt_multi_render: fewer samples per pixel than ranks");
  const size_t n = (size_t)params->nx * params->ny * 3;
  // every rank's share, enqueued on its own stream before anything is waited for
  for (int r = 0; r < N; ++r) {
    int rc = grow(m->device[(size_t)r], &m->frame[(size_t)r], &m->frame_floats[(size_t)r], n);
    if (rc == RT_OK && out_sumsq) rc = grow(m->device[(size_t)r], &m->frame2[(size_t)r], &m->frame2_floats[(size_t)r], n);
    if (rc != RT_OK) return rc;
    rt_render_params p = *params;
    if (shard_mode == RT_SHARD_SAMPLES) {
      const int begin = (int)((long long)params->spp * r / N), end = (int)((long long)params->spp * (r + 1) / N);
      p.sample_begin = params->sample_begin + begin;
      p.spp = end - begin;
    } else {
      p.shard_index = r;
      p.shard_count = N;
    }
    if (stats_per_rank && (rc = rt::reset_counters(m->ctx[(size_t)r], (void*)m->stream[(size_t)r])) != RT_OK) return rc;
    HIP_TRY(hipSetDevice(m->device[(size_t)r]));
    HIP_TRY(hipEventRecord(m->t0[(size_t)r], m->stream[(size_t)r]));
    rc = rt::enqueue_render(m->ctx[(size_t)r], cam, &p, m->frame[(size_t)r], out_sumsq ? m->frame2[(size_t)r] : nullptr, stats_per_rank != nullptr,
                            m->stream[(size_t)r]);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(m->device[(size_t)r]));
    HIP_TRY(hipEventRecord(m->t1[(size_t)r], m->stream[(size_t)r]));
  }
  int rc = exchange(m, params, shard_mode, m->frame, m->packed, m->packed_floats);
  if (rc == RT_OK && out_sumsq) rc = exchange(m, params, shard_mode, m->frame2, m->packed2, m->packed2_floats);
  if (rc != RT_OK) return rc;
  for (int r = N - 1; r >= 0; --r) {
    HIP_TRY(hipSetDevice(m->device[(size_t)r]));
    HIP_TRY(hipStreamSynchronize(m->stream[(size_t)r]));
  }
  HIP_TRY(hipMemcpy(out_sum, m->frame[0], n * sizeof(float), hipMemcpyDeviceToHost));
  if (out_sumsq) HIP_TRY(hipMemcpy(out_sumsq, m->frame2[0], n * sizeof(float), hipMemcpyDeviceToHost));
  if (stats_per_rank) {
    for (int r = 0; r < N; ++r) {
      if ((rc = rt::read_counters(m->ctx[(size_t)r], &stats_per_rank[r])) != RT_OK) return rc;
      float ms = 0.0f;
      HIP_TRY(hipEventElapsedTime(&ms, m->t0[(size_t)r], m->t1[(size_t)r]));
      stats_per_rank[r].kernel_ms = ms;
    }
  }
  return RT_OK;
}

}  // extern "C"
