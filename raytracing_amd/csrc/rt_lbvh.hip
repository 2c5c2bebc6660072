// rt_lbvh.hip -- the world tree rebuilt on the device.
//
// The reference builds its BVH on the host by a random-axis median split
// (src/bvh.rs:105-147), which every thread repeats (src/main.rs:85-90).  That
// tree is the contract for parity and it is a poor one for large scenes (about a
// hundred box tests per ray over 2^20 spheres).  rt_build_lbvh replaces the world
// tree's interior nodes by a linear BVH over the same leaves: Morton order of the
// leaf-box centres, the hierarchy of Karras (HPG 2012) from the sorted keys, boxes
// fitted bottom-up.  Leaves, their boxes and the sub-trees inside objects stay as
// they are, so the result is again an rt_scene node array that rt_upload_scene
// (and the CPU oracle) take.  Nearest hits are those of the reference's tree
// except for exact ties, which depend on leaf order.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_internal.hpp"

namespace {

struct LeafBox {
  float bmin[3];
  int node;  // index of the leaf in the caller's node array
  float bmax[3];
  int pad;
};

struct Interior {
  float bmin[3];
  int left;  // >= 0: interior k; < 0: ~(sorted leaf position)
  float bmax[3];
  int right;
};

// 10 bits per axis, interleaved (x highest).
__device__ __forceinline__ unsigned spread10(unsigned v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// key = Morton code of the box centre << 32 | leaf number: unique, so the
// hierarchy below needs no special case for equal codes.
__global__ void morton_kernel(const LeafBox* __restrict__ leaves, int n, float3 lo, float3 inv_extent, unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const LeafBox b = leaves[i];
  const float cx = (0.5f * (b.bmin[0] + b.bmax[0]) - lo.x) * inv_extent.x;
  const float cy = (0.5f * (b.bmin[1] + b.bmax[1]) - lo.y) * inv_extent.y;
  const float cz = (0.5f * (b.bmin[2] + b.bmax[2]) - lo.z) * inv_extent.z;
  const unsigned x = (unsigned)fminf(fmaxf(cx * 1024.0f, 0.0f), 1023.0f);
  const unsigned y = (unsigned)fminf(fmaxf(cy * 1024.0f, 0.0f), 1023.0f);
  const unsigned z = (unsigned)fminf(fmaxf(cz * 1024.0f, 0.0f), 1023.0f);
  const unsigned code = (spread10(x) << 2) | (spread10(y) << 1) | spread10(z);
  keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

// Length of the common prefix of keys i and j, -1 when j is out of range.
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012, section 3: interior node i covers a range of sorted leaves that
// starts or ends at i; its split is where the highest differing bit changes.
__global__ void hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, Interior* __restrict__ nodes, int* __restrict__ parent_of_interior,
                                 int* __restrict__ parent_of_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = delta(keys, n, i, i + 1) > delta(keys, n, i, i - 1) ? 1 : -1;
  const int d_min = delta(keys, n, i, i - d);
  int l_max = 2;
  while (delta(keys, n, i, i + l_max * d) > d_min) l_max *= 2;
  int l = 0;
  for (int t = l_max / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > d_min) l += t;
  const int j = i + l * d;
  const int d_node = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > d_node) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int first = i < j ? i : j, last = i < j ? j : i;
  const int left = first == gamma ? ~gamma : gamma;
  const int right = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  nodes[i].left = left;
  nodes[i].right = right;
  if (left >= 0) parent_of_interior[left] = i; else parent_of_leaf[~left] = i;
  if (right >= 0) parent_of_interior[right] = i; else parent_of_leaf[~right] = i;
  if (i == 0) parent_of_interior[0] = -1;
}

// One thread per leaf climbs towards the root; the second thread to reach a node
// finds both children's boxes written and carries on, the first one stops.
__global__ void fit_kernel(const LeafBox* __restrict__ leaves, const unsigned long long* __restrict__ keys, int n, Interior* __restrict__ nodes,
                           const int* __restrict__ parent_of_interior, const int* __restrict__ parent_of_leaf, unsigned* __restrict__ arrivals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int node = parent_of_leaf[i];
  while (node >= 0) {
    __threadfence();
    if (atomicAdd(&arrivals[node], 1u) == 0u) return;
    __threadfence();  // the other child's box was written before its thread's fence; see it
    float lo[3], hi[3];
    for (int side = 0; side < 2; ++side) {
      const int c = side == 0 ? nodes[node].left : nodes[node].right;
      float clo[3], chi[3];
      if (c < 0) {
        const LeafBox b = leaves[(unsigned)keys[~c]];
        for (int k = 0; k < 3; ++k) { clo[k] = b.bmin[k]; chi[k] = b.bmax[k]; }
      } else {
        // written by the thread that arrived first at `node` through this child, before its fence
        const volatile Interior* ci = nodes + c;
        for (int k = 0; k < 3; ++k) { clo[k] = ci->bmin[k]; chi[k] = ci->bmax[k]; }
      }
      for (int k = 0; k < 3; ++k) {
        lo[k] = side == 0 ? clo[k] : fminf(lo[k], clo[k]);
        hi[k] = side == 0 ? chi[k] : fmaxf(hi[k], chi[k]);
      }
    }
    for (int k = 0; k < 3; ++k) { nodes[node].bmin[k] = lo[k]; nodes[node].bmax[k] = hi[k]; }
    node = parent_of_interior[node];
  }
}

int hip_fail(hipError_t e, const char* what) {
  return rt::set_error(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

struct DeviceBuffers {
  void* p[8];
  int n;
  DeviceBuffers() : n(0) {}
  ~DeviceBuffers() {
    for (int i = 0; i < n; ++i)
      if (p[i]) (void)hipFree(p[i]);
  }
  hipError_t alloc(void** out, size_t bytes) {
    *out = nullptr;
    const hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
};

// World leaves and interior nodes in the caller's array: everything reachable from
// node 0 without entering an object's sub-tree.
int collect_world(const rt_scene* s, int idx, int depth, std::vector<int>* leaves, std::vector<int>* interior) {
  if (idx < 0 || idx >= s->n_nodes) return rt::set_error(RT_ERR_INVALID, "node index out of range");
  if (depth > 64) return rt::set_error(RT_ERR_INVALID, "node tree deeper than 64 levels (cycle?)");
  const rt_node& n = s->nodes[idx];
  if (n.left < 0) {
    leaves->push_back(idx);
    return RT_OK;
  }
  if (n.right < 0) return rt::set_error(RT_ERR_INVALID, "interior node with a negative right child");
  interior->push_back(idx);
  int rc = collect_world(s, n.left, depth + 1, leaves, interior);
  if (rc != RT_OK) return rc;
  return collect_world(s, n.right, depth + 1, leaves, interior);
}

}  // namespace

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

extern "C" int rt_build_lbvh(rt_ctx* ctx, const rt_scene* s, rt_node* nodes_out) {
  if (!ctx || !s || !nodes_out || !s->nodes || s->n_nodes <= 0) return rt::set_error(RT_ERR_INVALID, "rt_build_lbvh: null argument");
  std::vector<int> leaf_nodes, interior_nodes;
  int rc = collect_world(s, 0, 0, &leaf_nodes, &interior_nodes);
  if (rc != RT_OK) return rc;
  std::memcpy(nodes_out, s->nodes, (size_t)s->n_nodes * sizeof(rt_node));
  const int n = (int)leaf_nodes.size();
  if (n < 2) return RT_OK;  // the root is the only leaf
  if ((int)interior_nodes.size() != n - 1) return rt::set_error(RT_ERR_INVALID, "rt_build_lbvh: the world tree is not a full binary tree");
  // interior node k of the new tree takes the k-th smallest index the old interior nodes had: the root stays node 0
  std::sort(interior_nodes.begin(), interior_nodes.end());
  if (interior_nodes[0] != 0) return rt::set_error(RT_ERR_INVALID, "rt_build_lbvh: node 0 is not the root");

  std::vector<LeafBox> boxes((size_t)n);
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = 0; i < n; ++i) {
    const rt_node& ln = s->nodes[leaf_nodes[(size_t)i]];
    LeafBox& b = boxes[(size_t)i];
    for (int k = 0; k < 3; ++k) {
      b.bmin[k] = ln.bmin[k];
      b.bmax[k] = ln.bmax[k];
      const float c = 0.5f * (ln.bmin[k] + ln.bmax[k]);
      if (c < lo[k]) lo[k] = c;
      if (c > hi[k]) hi[k] = c;
    }
    b.node = leaf_nodes[(size_t)i];
    b.pad = 0;
  }
  float3 lo3 = make_float3(lo[0], lo[1], lo[2]);
  float3 inv3 = make_float3(hi[0] > lo[0] ? 1.0f / (hi[0] - lo[0]) : 0.0f, hi[1] > lo[1] ? 1.0f / (hi[1] - lo[1]) : 0.0f,
                            hi[2] > lo[2] ? 1.0f / (hi[2] - lo[2]) : 0.0f);

  HIP_TRY(hipSetDevice(rt::context_device(ctx)));
  DeviceBuffers dev;
  LeafBox* d_leaves = nullptr;
  unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
  Interior* d_nodes = nullptr;
  int *d_parent_int = nullptr, *d_parent_leaf = nullptr;
  unsigned* d_arrivals = nullptr;
  void* d_temp = nullptr;
  HIP_TRY(dev.alloc((void**)&d_leaves, (size_t)n * sizeof(LeafBox)));
  HIP_TRY(dev.alloc((void**)&d_keys, (size_t)n * sizeof(unsigned long long)));
  HIP_TRY(dev.alloc((void**)&d_sorted, (size_t)n * sizeof(unsigned long long)));
  HIP_TRY(dev.alloc((void**)&d_nodes, (size_t)(n - 1) * sizeof(Interior)));
  HIP_TRY(dev.alloc((void**)&d_parent_int, (size_t)(n - 1) * sizeof(int)));
  HIP_TRY(dev.alloc((void**)&d_parent_leaf, (size_t)n * sizeof(int)));
  HIP_TRY(dev.alloc((void**)&d_arrivals, (size_t)(n - 1) * sizeof(unsigned)));
  HIP_TRY(hipMemcpy(d_leaves, boxes.data(), (size_t)n * sizeof(LeafBox), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d_arrivals, 0, (size_t)(n - 1) * sizeof(unsigned)));
  HIP_TRY(hipMemset(d_nodes, 0, (size_t)(n - 1) * sizeof(Interior)));

  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(256), 0, 0, d_leaves, n, lo3, inv3, d_keys);
  HIP_TRY(hipGetLastError());
  size_t temp_bytes = 0;
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_sorted, n));
  HIP_TRY(dev.alloc(&d_temp, temp_bytes));
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_sorted, n));
  hipLaunchKernelGGL(hierarchy_kernel, dim3(blocks), dim3(256), 0, 0, d_sorted, n, d_nodes, d_parent_int, d_parent_leaf);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(fit_kernel, dim3(blocks), dim3(256), 0, 0, d_leaves, d_sorted, n, d_nodes, d_parent_int, d_parent_leaf, d_arrivals);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());

  std::vector<Interior> built((size_t)(n - 1));
  std::vector<unsigned long long> sorted((size_t)n);
  HIP_TRY(hipMemcpy(built.data(), d_nodes, (size_t)(n - 1) * sizeof(Interior), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sorted.data(), d_sorted, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int k = 0; k < n - 1; ++k) {
    const Interior& b = built[(size_t)k];
    rt_node& out = nodes_out[interior_nodes[(size_t)k]];
    for (int c = 0; c < 3; ++c) { out.bmin[c] = b.bmin[c]; out.bmax[c] = b.bmax[c]; }
    out.left = b.left >= 0 ? interior_nodes[(size_t)b.left] : leaf_nodes[(size_t)(unsigned)sorted[(size_t)~b.left]];
    out.right = b.right >= 0 ? interior_nodes[(size_t)b.right] : leaf_nodes[(size_t)(unsigned)sorted[(size_t)~b.right]];
  }
  return RT_OK;
}
