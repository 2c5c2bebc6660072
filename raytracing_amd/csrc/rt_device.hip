// rt_device.hip -- gfx950 kernels and the rt_* entry points of include/rt_abi.h.
//
// Kernels:
//   render_kernel       : the whole sample loop of main.rs:96-116 with color()
//                         (main.rs:165-209) unrolled into an iterative path loop; scenes of a
//                         handful of leaves and scenes on the skip-link scan.
//   render_stack_kernel : the same loop for scenes on the per-lane stack traversal: a lane's
//                         walk through the tree goes on across the shading code (see there).
//   trace_kernel        : nearest hit for a batch of rays (the world.hit of main.rs:166).
//   trace64_kernel      : the same in the reference's own arithmetic (f64, the source's
//                         divisions and traversal order): the parity mode.
//   reduce_slabs_kernel, tonemap_kernel, selftest_rcp_kernel : small helpers.
//
// Execution model of render_kernel.  One single-wave workgroup owns a work item:
// an 8x8 pixel tile and a chunk of its samples, i.e. a pool of 64*chunk_spp
// (pixel, sample) paths.  Every trip of the path loop each lane advances its
// path by one vertex; lanes whose path ended are found with a ballot, ranked with
// mbcnt and refilled from the head of the pool, so the wave stays full until the
// pool is empty.  Radiance goes to per-tile accumulators in LDS (ds_add_f32) and
// is written to HBM once, at the end.  Control flow is wave-uniform; per-lane
// conditions are selects.
//
// Traversal, chosen per scene by rt_upload_scene, one kernel build each:
//   TRAV_FLAT  - a handful of leaves (the Cornell box): the wave walks all of them
//                with a wave-uniform index, leaf data arriving through scalar loads
//                into SGPRs; the leaves' box tests are deferred to the winner.
//   TRAV_SCAN  - the same walk over a depth-first array with skip links.
//   TRAV_STACK - larger trees: every lane follows its own ray through a four-wide tree that the
//                upload builds over the given leaves, with a stack in LDS (and, for the rare
//                deep walk, in global memory).
// In the first two, leaves are met in the reference's left-to-right order, so
// BvhNode::hit's "right wins ties" rule (bvh.rs:212-217) becomes "a later hit
// replaces the best unless best < t"; the third keeps a depth-first rank per leaf.
//
// Arithmetic is fp32.  This file is compiled with -ffp-contract=off; every
// fused multiply-add is written as fmaf() so that the CPU oracle's fp32 "fused"
// mode can repeat the intersection arithmetic bit for bit.  Square roots in the
// intersection code are the correctly rounded ones, and so is the reciprocal
// (exact_rcp below).  Shading uses the hardware's approximate instructions.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_internal.hpp"

namespace {

// ---------------------------------------------------------------------------
// Device-side scene
// ---------------------------------------------------------------------------
enum {
  K_INTERIOR = 0,  // box only; children follow
  K_RECT_XY = 1,   // leaf holding one rect: f = a0,a1,b0,b1,k ; a = prim index
  K_RECT_XZ = 2,
  K_RECT_YZ = 3,
  K_SPHERE = 4,    // leaf holding one sphere: f = cx,cy,cz,r ; a = prim index
  K_LIST = 5,      // leaf holding prims[a, a+b) as a HitableList in the node's own frame
  K_OBJ_LIST = 6,  // Translate(RotateY(list)): f = sin,cos,ox,oy,oz ; xform ; prims[a, a+b)
  K_OBJ_BVH = 7,   // Translate(RotateY(BvhNode)): same f/xform; the sub-tree's nodes follow up to skip
  K_MEDIUM = 8     // ConstantMedium over Translate(RotateY(list)): as K_OBJ_LIST; the last extension slot holds
                   // density, medium ordinal, DevPrim index
};

// One node of the depth-first ("scan") layout, 64 bytes = one s_load_dwordx16.
// Children of an interior node follow it immediately; `skip` is the index of the
// first node after its subtree.  Leaves carry what their test needs inline.
struct ScanNode {
  float bmin[3];
  int skip;
  float bmax[3];
  int kind;
  float f[6];
  int a;
  int b;  // prim count; for K_OBJ_*: count | xform << 24
};

// Everything the path loop needs to know about the primitive it hit, gathered
// with one 64-byte read: geometry, the transform of the object it lives in, and
// its material with a constant texture already resolved to a colour.
#define DP_TYPE_MASK 0x7   // RT_PRIM_* or DP_TYPE_MEDIUM
#define DP_TYPE_MEDIUM 4   // a ConstantMedium's scattering point: p = ray(t), normal (1,0,0) (hitable.rs:586-593)
#define DP_FLIP 0x8
#define DP_XFORM_SHIFT 4   // 2 bits: RT_XFORM_*
#define DP_MAT_SHIFT 6     // 3 bits: RT_MAT_*
#define DP_TEXTURED 0x200  // the colour must come from Texture::value (checker / noise)
#define DP_CHAINED 0x400   // the primitive lies inside more than one object: see DevScene::prim_obj
#define DP_INDEX_SHIFT 12  // material index
struct DevPrim {
  int flags;
  float f[5];
  float color[3];  // albedo / emission when the texture is constant; Metal albedo
  float sin_t, cos_t;
  float off[3];
  float fuzz, ref_idx;
};

// Scene buffers are written by rt_upload_scene and only read by kernels, so
// wave-uniform reads go through the constant address space with a vector type:
// that is what makes the compiler emit one scalar load (s_load_dwordx16 into
// SGPRs) per record.
#define RT_CONSTANT_AS __attribute__((address_space(4)))
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

struct DevScene {
  const ScanNode* scan;
  const rt_prim* prims;
  const DevPrim* dprims;
  const rt_material* materials;
  const rt_texture* textures;
  const rt_perlin* perlins;
  int n_scan, n_prims, n_dprims, n_lights;
  int mat_mask;  // bit RT_MAT_* set when a primitive or medium uses that material
  int textured;  // some material needs Texture::value (checker / noise)
  int has_spheres;
  int flat;  // 1: the scan array holds leaves only (every skip is the next index)
  const float4* leaf_box;  // per DevPrim: bmin, bmax of the scan leaf that holds it (two float4)
  // Stack traversal (scenes with more than a handful of leaves): the breadth-first
  // nodes with packed leaf information, and the objects their leaves refer to.  In
  // that mode `scan` holds only the ConstantMedium nodes, walked after the tree.
  const struct BNode* bnodes;
  const struct BNode4* bnodes4;
  int root_ref;            // the world tree's root as a reference
  const int* sub_root_ref;  // per object, two values: the reference of an RT_OBJ_BVH's sub-tree root, and the node whose box is the root's own
  int root_box;             // the same node for the world's tree
  const int* box6_index;   // per primitive: >= 0 where a list of six rects in Cuboid::new's order starts, the index of its record
  const float4* box6_recs; // 5 float4 per record: the 18 floats of box6_test, padded
  const int* prim_rank;  // per primitive: depth-first rank of its leaf
  const rt_object* objects;
  // Objects inside objects (RotateY(Translate(x)), an instanced sub-tree that holds instances ...): a primitive under
  // more than one of them carries DP_CHAINED, and its hit is carried back to the world through the objects
  // chain_objs[obj_chain[2 * o] .. + obj_chain[2 * o + 1]) (outermost first) of its innermost object o = prim_obj[prim].
  int has_chains;
  const int* prim_obj;
  const int* obj_chain;
  const int* chain_objs;
  // ConstantMedia whose boundary is a sub-tree (RT_OBJ_MEDIUM_BVH): walked after the tree like the others, by their own code
  const struct MediumBvh* media_bvh;
  int n_media_bvh;
  const struct ScanNode* media_sph;  // see media_sphere_pass; n_media_sph > 0 only when EVERY medium of the scan array has such a record
  int n_media_sph;
  const rt_node* ref_nodes;  // the node array as it was passed in (trace64_kernel walks it the reference's way)
  int n_ref_nodes;
  int stack_mode;
  int no_prune;     // some geometry lies outside the boxes above it (a sphere of negative radius in a list): boxes are never skipped by distance
  int stack_depth;  // entries per lane of the traversal stack that the deepest walk can need
  int stack_lds;    // of which this many live in LDS
  rt_light light;          // the first shape (the only one in the reference's own case: one XZRect)
  const rt_light* lights;  // all of them
  int light_general;       // several shapes, or a shape other than an XZRect: the general HitablePDF code
};

static_assert(sizeof(rt_prim) == 32 && sizeof(ScanNode) == 64, "record sizes the byte-offset loads assume");

// A ConstantMedium whose boundary is a sub-tree: the box of the world-tree leaf that holds it (bvh.rs:198 tests it
// before the medium is asked), the object, and what ConstantMedium::hit needs.
struct MediumBvh {
  float bmin[3], bmax[3];
  int object;    // index into objects (its transform; its sub-tree's box node is objects[object].first)
  float density;
  int ordinal;   // selects the random-number stream (rt_object.medium_index)
  int dprim;     // the DevPrim of its scattering point
  int pad[2];
};
static_assert(sizeof(MediumBvh) == 48, "three 16-byte scalar loads");

// The breadth-first node array as the stack traversal keeps it, 32 bytes per node:
// the box (read for the root and for the root of an object's sub-tree) and, for a
// leaf, left = ~ref (negative) and right = primitive count, 0 meaning objects[ref].
struct BNode {
  float bmin[3];
  int left;
  float bmax[3];
  int right;
};
#define BN_INDEX_MASK 0x0fffffff
// What the stack traversal fetches at an interior node, 128 bytes (one cache line): the boxes of up to four
// descendants and where they are.  The binary tree of bvh.rs is collapsed on upload: a node's entry holds up to four
// of its descendants (its two children, the larger interior ones opened into their own children while there is
// room).  Leaving out the box test of a node in between changes no result: a box lies inside its parent's and the slab arithmetic is monotonic
// in the bounds, so a ray that passes a node's Aabb::hit passes every ancestor's.  Half as many dependent
// fetches per ray, each deciding four box tests.  A reference is an index into this array for an interior
// descendant, the descendant's index in the breadth-first array with BN2_LEAF set for a leaf, and for a leaf of
// one primitive the primitive's index; empty slots hold BN4_EMPTY and a box no ray passes.
struct BNode4 {
  float lox[4], loy[4], loz[4];
  float hix[4], hiy[4], hiz[4];
  int ref[4];
  float top[4];  // y_top of each descendant (see slab_core)
};
static_assert(sizeof(BNode4) == 128, "one cache line");
#define BN2_LEAF 0x40000000
#define BN2_PRIM 0x20000000  // with BN2_LEAF: a single-primitive leaf, the low bits are the primitive's index
#define BN2_BOX6 0x10000000  // with BN2_LEAF: a list of six rects in Cuboid::new's order, the low bits are its record's index
#define BN4_EMPTY (-2)
#define RT_MAX_TREE_DEPTH 64  // rt_upload_scene refuses node trees deeper than this
#define RT_MAX_NEST 8         // and objects nested deeper than this (an object in a sub-tree of an object ...)
#define STACK_DEPTH 64  // deepest tree the stack traversal takes; the launch sizes the LDS stack to the scene's own depth
#define STACK_RESTORE 0xffffffffu

struct DevCamera {
  float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
  float lens_radius;
};

struct F3 {
  float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 sel3(bool c, F3 a, F3 b) { return f3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
// z*z' + (y*y' + x*x') with single-rounding multiply-adds (the oracle's fused dot).
__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

struct RayF {
  F3 o, d, inv;
};

// The correctly rounded reciprocal, 1.0f / x, in five instructions instead of the
// dozen of the general division: the hardware estimate (1 ulp) and one Newton
// step with single-rounding multiply-adds.  rt_selftest_reciprocal compares it
// with the division for every one of the 2^32 bit patterns on the device it runs
// on; the inputs outside RCP_LO..RCP_HI in magnitude (and zero, infinity, NaN),
// where the step's residual is not representable, take the division itself.
#define RT_RCP_LO 1.0e-30f
#define RT_RCP_HI 1.0e+30f
__device__ __forceinline__ float rcp_newton(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = fmaf(-x, r, 1.0f);
  return fmaf(e, r, r);
}
__device__ __forceinline__ bool rcp_in_range(float x) {
  const float a = fabsf(x);
  return (a >= RT_RCP_LO) & (a <= RT_RCP_HI);
}
__device__ __forceinline__ float exact_rcp(float x) {
  float r = rcp_newton(x);
  // wave-uniform: the division is only compiled into a branch no ordinary ray takes
  if (__builtin_expect(__ballot(!rcp_in_range(x)) != 0ull, 0)) r = rcp_in_range(x) ? r : 1.0f / x;
  return r;
}
__device__ __forceinline__ RayF make_ray(F3 o, F3 d) {
  RayF r;
  r.o = o;
  r.d = d;
  r.inv = f3(rcp_newton(d.x), rcp_newton(d.y), rcp_newton(d.z));
  // one range test for the three components
  const float lo = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)), hi = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  if (__builtin_expect(__ballot(!((lo >= RT_RCP_LO) & (hi <= RT_RCP_HI))) != 0ull, 0)) {
    r.inv.x = rcp_in_range(d.x) ? r.inv.x : 1.0f / d.x;
    r.inv.y = rcp_in_range(d.y) ? r.inv.y : 1.0f / d.y;
    r.inv.z = rcp_in_range(d.z) ? r.inv.z : 1.0f / d.z;
  }
  return r;
}

// Translate::hit (hitable.rs:440) then RotateY::hit (hitable.rs:510-518): the ray
// in the object's frame.  d.y, and so inv.y, is unchanged.  WANT_INV = false
// skips the two reciprocals (the caller only needs o and d).  `xform` may differ
// from lane to lane, so the two steps are selects, not branches.
template <bool WANT_INV, bool UNIFORM = false>
__device__ __forceinline__ RayF to_object_frame(const RayF& w, int xform, float sin_t, float cos_t, float ox, float oy, float oz) {
  RayF r = w;
  const bool tr = (xform & RT_XFORM_TRANSLATE) != 0, ro = (xform & RT_XFORM_ROTATE_Y) != 0;
  if (UNIFORM) {
    // the whole wave is at the same object: branch instead of selecting
    if (tr) r.o = f3(w.o.x - ox, w.o.y - oy, w.o.z - oz);
    if (ro) {
      const F3 o = r.o, d = r.d;
      r.o.x = fmaf(cos_t, o.x, -(sin_t * o.z));
      r.o.z = fmaf(sin_t, o.x, cos_t * o.z);
      r.d.x = fmaf(cos_t, d.x, -(sin_t * d.z));
      r.d.z = fmaf(sin_t, d.x, cos_t * d.z);
      if (WANT_INV) {
        r.inv.x = exact_rcp(r.d.x);
        r.inv.z = exact_rcp(r.d.z);
      }
    }
    return r;
  }
  r.o = sel3(tr, f3(w.o.x - ox, w.o.y - oy, w.o.z - oz), w.o);
  const F3 o = r.o, d = r.d;
  r.o.x = ro ? fmaf(cos_t, o.x, -(sin_t * o.z)) : o.x;
  r.o.z = ro ? fmaf(sin_t, o.x, cos_t * o.z) : o.z;
  r.d.x = ro ? fmaf(cos_t, d.x, -(sin_t * d.z)) : d.x;
  r.d.z = ro ? fmaf(sin_t, d.x, cos_t * d.z) : d.z;
  if (WANT_INV) {
    // d is unchanged without a rotation, so 1/d is the same value either way.
    r.inv.x = exact_rcp(r.d.x);
    r.inv.z = exact_rcp(r.d.z);
  }
  return r;
}

__device__ __forceinline__ ScanNode load_node(const ScanNode* base, int i) {
  // a 32-bit unsigned byte offset goes straight into the scalar load's offset operand (no 64-bit address arithmetic)
  const u32x16 v = *(const RT_CONSTANT_AS u32x16*)((const RT_CONSTANT_AS char*)base + (unsigned)i * 64u);
  ScanNode n;
  __builtin_memcpy(&n, &v, sizeof(n));
  return n;
}
__device__ __forceinline__ rt_prim load_prim_uniform(const rt_prim* base, int i) {
  const u32x8 v = *(const RT_CONSTANT_AS u32x8*)((const RT_CONSTANT_AS char*)base + (unsigned)i * 32u);
  rt_prim q;
  __builtin_memcpy(&q, &v, sizeof(q));
  return q;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and the stream layout of rt_abi.h
// ---------------------------------------------------------------------------
struct U4 {
  unsigned x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const unsigned long long p0 = (unsigned long long)M0 * c0;
    const unsigned long long p1 = (unsigned long long)M1 * c2;
    // one three-input XOR each (v_bitop3_b32 with truth table 0x96; the compiler leaves a ^ b ^ c as two instructions)
    const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1, k0, 0x96);
    const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3, k1, 0x96);
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
    k0 += W0;
    k1 += W1;
  }
  U4 r;
  r.x = c0; r.y = c1; r.z = c2; r.w = c3;
  return r;
}
// The same with the twenty round keys read from LDS (rk[2 * round], rk[2 * round + 1]; every lane reads the same
// addresses) instead of made by eighteen scalar additions per call: five LDS reads.
__device__ __forceinline__ U4 philox4x32_10_keys(unsigned c0, unsigned c1, unsigned c2, unsigned c3, const unsigned* rk) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint4 kk[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) kk[i] = ((const uint4*)rk)[i];
  unsigned k[20];
  __builtin_memcpy(k, kk, sizeof(k));
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const unsigned long long p0 = (unsigned long long)M0 * c0;
    const unsigned long long p1 = (unsigned long long)M1 * c2;
    const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1, k[2 * round], 0x96);
    const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3, k[2 * round + 1], 0x96);
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
  }
  U4 r;
  r.x = c0; r.y = c1; r.z = c2; r.w = c3;
  return r;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// What a ConstantMedium needs to draw its random number inside the traversal.
struct PathRng {
  unsigned pixel, sample, depth, key0, key1;
};

struct Counters {
  unsigned rays, hits, node_tests, rect_tests, sphere_tests, instance_ops, shade_events, light_tests, paths;
};

struct Hit {
  float t;
  int prim;  // index into prims, -1 = none
};

// The control flow below is wave-uniform wherever possible: per-lane conditions
// become selects ("h & on"), not branches, because a wave of incoherent rays
// executes both sides anyway and exec-mask bookkeeping only adds scalar work.
// Lanes that take no part (dead paths, rays past the end of a batch) run the same
// arithmetic on stale registers; they never touch memory and their results are
// masked out.

// Aabb::hit (bvh.rs:46-61) with the reference's comparisons: a NaN slab value
// leaves the interval unchanged (v_max/v_min return the other operand), a
// negative reciprocal swaps the pair, an empty or zero-width interval misses.
// The three per-axis exits of the source collapse into one test because the
// interval only ever shrinks.  `best_t` adds the pruning test: a box entered
// beyond the best hit so far cannot hold the winner.
struct Signs {
  bool x, y, z;  // reciprocal direction negative, per axis
};
__device__ __forceinline__ Signs ray_signs(const RayF& r) {
  Signs s;
  s.x = r.inv.x < 0.0f;
  s.y = r.inv.y < 0.0f;
  s.z = r.inv.z < 0.0f;
  return s;
}
// `y_top` (>= y1) is how high the geometry under the box really reaches: a Cuboid's
// top face lies 0.01 above the cuboid's own box (hitable.rs:398-399), so a ray coming
// down can hit it before it enters the box.  The box test itself uses the box, as the
// reference does; the pruning test uses the entry distance into the box raised to
// y_top, which no hit under this node can undercut.
__device__ __forceinline__ bool slab_core(float x0, float y0, float z0, float x1, float y1, float z1, float y_top, const RayF& r,
                                          const Signs& sg, float t_min, float t_max, float best_t, float& t_enter) {
  const float t_min0 = t_min;
  float t0 = (x0 - r.o.x) * r.inv.x, t1 = (x1 - r.o.x) * r.inv.x;
  const float nx = sg.x ? t1 : t0;
  t_min = fmaxf(nx, t_min);
  t_max = fminf(sg.x ? t0 : t1, t_max);
  t0 = (y0 - r.o.y) * r.inv.y; t1 = (y1 - r.o.y) * r.inv.y;
  const float ny = sg.y ? t1 : t0;
  t_min = fmaxf(ny, t_min);
  t_max = fminf(sg.y ? t0 : t1, t_max);
  t0 = (z0 - r.o.z) * r.inv.z; t1 = (z1 - r.o.z) * r.inv.z;
  const float nz = sg.z ? t1 : t0;
  t_min = fmaxf(nz, t_min);
  t_max = fminf(sg.z ? t0 : t1, t_max);
  const float ny_top = sg.y ? (y_top - r.o.y) * r.inv.y : ny;
  const float t_prune = fmaxf(fmaxf(nx, nz), fmaxf(ny_top, t_min0));
  t_enter = t_prune;  // the distance below which nothing under this box can be hit (not negative: t_min0 is not)
  // Not in the reference, which works in f64: a rect's box is 0.0002 thick, and from an origin a few thousand
  // units away (bound - origin) rounds to the same float for both faces, so the entry and exit distances are
  // equal and bvh.rs:58's `<=` rejects a box the ray goes through.  The exit distance is widened by two units
  // in the last place, which never rejects a box the exact test accepts.
  return !(t_max * 1.00000024f <= t_min) & !(t_prune > best_t);
}
// Without pruning (best_t = +inf) y_top plays no part.
__device__ __forceinline__ bool slab_box(float x0, float y0, float z0, float x1, float y1, float z1, const RayF& r, const Signs& sg,
                                         float t_min, float t_max) {
  float te;
  return slab_core(x0, y0, z0, x1, y1, z1, y1, r, sg, t_min, t_max, 3.402823466e+38f, te);
}
// The same with pruning against best_t, also handing back the pruning distance (where the ray enters the box raised to y_top).
__device__ __forceinline__ bool slab_box_enter(float4 lo, float4 hi, float y_top, const RayF& r, const Signs& sg, float t_min, float t_max,
                                               float best_t, float& t_enter) {
  return slab_core(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, y_top, r, sg, t_min, t_max, best_t, t_enter);
}
// A scan node keeps its y_top in f[5].
__device__ __forceinline__ bool slab_test(const ScanNode& n, const RayF& r, const Signs& sg, float t_min, float t_max, float best_t) {
  float te;
  return slab_core(n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2], n.f[5], r, sg, t_min, t_max, best_t, te);
}

// XYRect/XZRect/YZRect::hit (hitable.rs:163-173 and twins).  C is the fixed
// axis, A and B the in-plane axes in constructor order; f = a0,a1,b0,b1,k.  The
// source's rejects are kept as written so a NaN passes them exactly as it does
// there.  t = (k - o[C]) * (1/d[C]) replaces the source's division.
template <int C, int A, int B>
__device__ __forceinline__ bool rect_test(const float* f, const RayF& r, float t0, float t1, float& t_out) {
  const float oc = C == 0 ? r.o.x : (C == 1 ? r.o.y : r.o.z);
  const float ic = C == 0 ? r.inv.x : (C == 1 ? r.inv.y : r.inv.z);
  const float oa = A == 0 ? r.o.x : (A == 1 ? r.o.y : r.o.z);
  const float da = A == 0 ? r.d.x : (A == 1 ? r.d.y : r.d.z);
  const float ob = B == 0 ? r.o.x : (B == 1 ? r.o.y : r.o.z);
  const float db = B == 0 ? r.d.x : (B == 1 ? r.d.y : r.d.z);
  const float t = (f[4] - oc) * ic;
  const float a = fmaf(t, da, oa);
  const float b = fmaf(t, db, ob);
  t_out = t;
  // x lies in [lo, hi] exactly when the median of the three is x (the bounds are
  // ordered: rt_upload_scene refuses a rect with a0 > a1, the callers keep
  // t0 <= t1).  A NaN x makes the median one of the bounds and the comparison
  // unordered, which counts as inside, as the source's two rejects do.
  const float mt = __builtin_amdgcn_fmed3f(t, t0, t1), ma = __builtin_amdgcn_fmed3f(a, f[0], f[1]), mb = __builtin_amdgcn_fmed3f(b, f[2], f[3]);
  return !((mt < t) | (mt > t)) & !((ma < a) | (ma > a)) & !((mb < b) | (mb > b));
}

// Sphere::hit (hitable.rs:50-82): half-b quadratic, nearer root first, strict
// bounds; f = cx,cy,cz,r.  Not in the reference, which works in f64: b*b - a*c
// cancels badly in fp32 when the origin is far from a small sphere (2000 units
// from a sphere of radius 2.4: grazing hits come and go, t is off by up to 3e-4
// relative), so the discriminant is taken from the part of oc perpendicular to
// the ray, l = oc - (b/a) d:  b*b - a*c = a (r*r - l.l), and the roots are
// t = -b/a -+ sqrt((r*r - l.l) / a); every term is of the sphere's own size.
// With it the 2^20-sphere scene gives the f64 restatement's sphere on every test
// ray and t within 1e-5 relative on 99.98 % of them.  A negative h makes the
// square root, and with it every comparison, fail on its own; h == 0 is excluded
// as discriminant == 0 is in the source.
// h (the discriminant's sign) and the two roots; the roots are NaN when h < 0.
__device__ __forceinline__ void sphere_roots(const float* f, const RayF& r, float& h, float& t_near, float& t_far) {
  const F3 oc = f3(r.o.x - f[0], r.o.y - f[1], r.o.z - f[2]);
  const float a = dot3(r.d, r.d);
  const float b = dot3(oc, r.d);
  const float inv_a = exact_rcp(a);
  const float k = b * inv_a;
  const F3 l = f3(fmaf(-k, r.d.x, oc.x), fmaf(-k, r.d.y, oc.y), fmaf(-k, r.d.z, oc.z));
  h = fmaf(f[3], f[3], -dot3(l, l));
  const float sq = sqrtf(h * inv_a);
  t_near = -k - sq;
  t_far = -k + sq;
}
__device__ __forceinline__ bool sphere_test(const float* f, const RayF& r, float t_min, float t_max, float& t_out) {
  float h, t_near, t_far;
  sphere_roots(f, r, h, t_near, t_far);
  const bool near_ok = (t_near < t_max) & (t_near > t_min);
  const bool far_ok = (t_far < t_max) & (t_far > t_min);
  t_out = near_ok ? t_near : t_far;
  return (h > 0.0f) & (near_ok | far_ok);
}

// HitableList::hit (hitable.rs:105-117) over prims[first, first+count) with a
// wave-uniform loop, so the primitive records are scalar loads.  The list's own
// closest_so_far starts at t_max; its winner then meets the running best under
// BvhNode::hit's rule (bvh.rs:212-217): the earlier hit survives only if it is
// strictly nearer.
// The list's winner as a candidate for the caller's running best: `take` says whether it replaces it.
template <bool STATS>
__device__ __forceinline__ void list_cand(const rt_prim* __restrict__ prims, int first, int count, bool lane_on, const RayF& r,
                                          float t_min, float t_max, float best_t, float& cand_t, int& cand_prim, bool& take, Counters& cnt) {
  float loc_t = t_max;
  int loc_prim = -1;
  for (int qi = first; qi < first + count; ++qi) {
    const int qu = __builtin_amdgcn_readfirstlane(qi);
    const rt_prim q = load_prim_uniform(prims, qu);
    const int type = q.type & RT_PRIM_TYPE_MASK;
    float t = 0.0f;
    bool h;
    if (type == RT_PRIM_XY_RECT) h = rect_test<2, 0, 1>(q.f, r, t_min, loc_t, t);
    else if (type == RT_PRIM_XZ_RECT) h = rect_test<1, 0, 2>(q.f, r, t_min, loc_t, t);
    else if (type == RT_PRIM_YZ_RECT) h = rect_test<0, 1, 2>(q.f, r, t_min, loc_t, t);
    else h = sphere_test(q.f, r, t_min, loc_t, t);
    h &= lane_on;
    if (STATS) {
      if (type == RT_PRIM_SPHERE) cnt.sphere_tests += lane_on;
      else cnt.rect_tests += lane_on;
    }
    loc_t = h ? t : loc_t;
    loc_prim = h ? qu : loc_prim;
  }
  take = (loc_prim >= 0) & !(best_t < loc_t);
  cand_t = loc_t;
  cand_prim = loc_prim;
}
template <bool STATS>
__device__ __forceinline__ void list_test(const rt_prim* __restrict__ prims, int first, int count, bool lane_on, const RayF& r,
                                          float t_min, float t_max, Hit& best, Counters& cnt) {
  float ct;
  int cp;
  bool take;
  list_cand<STATS>(prims, first, count, lane_on, r, t_min, t_max, best.t, ct, cp, take, cnt);
  best.t = take ? ct : best.t;
  best.prim = take ? cp : best.prim;
}

// A HitableList of exactly six rects in the order Cuboid::new builds them
// (hitable.rs:381-416: XY, XY, XZ, XZ, YZ, YZ) whose pairs share their in-plane
// bounds: the same sequential semantics as list_test, from an 18-float record
// (per pair a0,a1,b0,b1,k_first,k_second) stored in the two scan slots that
// follow the node, fetched by two scalar loads, with no per-primitive dispatch.
#define LIST_IS_BOX6 0x800000
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// `k` is the pair's position in the list, a compile-time constant, so the selects
// that track the winner need no register for it.
template <int C, int A, int B>
__device__ __forceinline__ void box_pair(const float* g, int k, bool lane_on, const RayF& r, float t_min, float& loc_t, int& loc_k) {
  float f[5] = {g[0], g[1], g[2], g[3], g[4]};
  float t;
  bool h = rect_test<C, A, B>(f, r, t_min, loc_t, t) & lane_on;
  loc_t = h ? t : loc_t;
  loc_k = h ? k : loc_k;
  f[4] = g[5];
  h = rect_test<C, A, B>(f, r, t_min, loc_t, t) & lane_on;
  loc_t = h ? t : loc_t;
  loc_k = h ? k + 1 : loc_k;
}
template <bool STATS>
__device__ __forceinline__ void box6_cand(const ScanNode* __restrict__ scan, int node, int first, bool lane_on, const RayF& r, float t_min,
                                          float t_max, float best_t, float& cand_t, int& cand_prim, bool& take, Counters& cnt) {
  const u32x16 v0 = *(const RT_CONSTANT_AS u32x16*)((const RT_CONSTANT_AS char*)scan + (unsigned)node * 64u + 64u);
  const u32x2 v1 = *(const RT_CONSTANT_AS u32x2*)((const RT_CONSTANT_AS char*)scan + (unsigned)node * 64u + 128u);
  float g[18];
  __builtin_memcpy(&g[0], &v0, 64);
  __builtin_memcpy(&g[16], &v1, 8);
  float loc_t = t_max;
  int loc_k = -1;
  box_pair<2, 0, 1>(&g[0], 0, lane_on, r, t_min, loc_t, loc_k);
  box_pair<1, 0, 2>(&g[6], 2, lane_on, r, t_min, loc_t, loc_k);
  box_pair<0, 1, 2>(&g[12], 4, lane_on, r, t_min, loc_t, loc_k);
  if (STATS) cnt.rect_tests += lane_on ? 6u : 0u;
  take = (loc_k >= 0) & !(best_t < loc_t);
  cand_t = loc_t;
  cand_prim = first + loc_k;
}
template <bool STATS>
__device__ __forceinline__ void box6_test(const ScanNode* __restrict__ scan, int node, int first, bool lane_on, const RayF& r, float t_min,
                                          float t_max, Hit& best, Counters& cnt) {
  float ct;
  int cp;
  bool take;
  box6_cand<STATS>(scan, node, first, lane_on, r, t_min, t_max, best.t, ct, cp, take, cnt);
  best.t = take ? ct : best.t;
  best.prim = take ? cp : best.prim;
}

// The two boundary searches of ConstantMedium::hit (hitable.rs:562-566) over a Cuboid's six rects in one pass: a
// rect's t and whether its hit point lies inside the rect do not depend on the interval searched, so they are computed
// once per rect, and each search then walks the six (t, inside) pairs the way HitableList::hit walks the rects (the
// upper bound shrinks to every hit accepted, a later rect wins an equal t).  Same arithmetic, same comparisons and same
// order as two box6_test calls; 96 instructions instead of 150.  `next_of` maps the first search's t to the second's
// lower bound.
template <int C, int A, int B>
__device__ __forceinline__ void rect_parts(const float* f, float k, const RayF& r, float& t, bool& inside) {
  const float oc = C == 0 ? r.o.x : (C == 1 ? r.o.y : r.o.z);
  const float ic = C == 0 ? r.inv.x : (C == 1 ? r.inv.y : r.inv.z);
  const float oa = A == 0 ? r.o.x : (A == 1 ? r.o.y : r.o.z);
  const float da = A == 0 ? r.d.x : (A == 1 ? r.d.y : r.d.z);
  const float ob = B == 0 ? r.o.x : (B == 1 ? r.o.y : r.o.z);
  const float db = B == 0 ? r.d.x : (B == 1 ? r.d.y : r.d.z);
  t = (k - oc) * ic;
  const float a = fmaf(t, da, oa);
  const float b = fmaf(t, db, ob);
  const float ma = __builtin_amdgcn_fmed3f(a, f[0], f[1]), mb = __builtin_amdgcn_fmed3f(b, f[2], f[3]);
  inside = !((ma < a) | (ma > a)) & !((mb < b) | (mb > b));
}
__device__ __forceinline__ void six_search(const float* t, const bool* inside, bool lane_on, float t0, float& loc_t, int& loc_k) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float mt = __builtin_amdgcn_fmed3f(t[i], t0, loc_t);
    const bool h = inside[i] & !((mt < t[i]) | (mt > t[i])) & lane_on;
    loc_t = h ? t[i] : loc_t;
    loc_k = h ? i : loc_k;
  }
}
template <bool STATS>
__device__ __forceinline__ void box6_two_searches(const ScanNode* __restrict__ scan, int node, int first, bool lane_on, const RayF& r, Hit& b1,
                                                  Hit& b2, Counters& cnt) {
  const u32x16 v0 = *(const RT_CONSTANT_AS u32x16*)((const RT_CONSTANT_AS char*)scan + (unsigned)node * 64u + 64u);
  const u32x2 v1 = *(const RT_CONSTANT_AS u32x2*)((const RT_CONSTANT_AS char*)scan + (unsigned)node * 64u + 128u);
  float g[18];
  __builtin_memcpy(&g[0], &v0, 64);
  __builtin_memcpy(&g[16], &v1, 8);
  float t[6];
  bool inside[6];
  rect_parts<2, 0, 1>(&g[0], g[4], r, t[0], inside[0]);
  rect_parts<2, 0, 1>(&g[0], g[5], r, t[1], inside[1]);
  rect_parts<1, 0, 2>(&g[6], g[10], r, t[2], inside[2]);
  rect_parts<1, 0, 2>(&g[6], g[11], r, t[3], inside[3]);
  rect_parts<0, 1, 2>(&g[12], g[16], r, t[4], inside[4]);
  rect_parts<0, 1, 2>(&g[12], g[17], r, t[5], inside[5]);
  if (STATS) cnt.rect_tests += lane_on ? 12u : 0u;  // (the events of the two searches)
  const float big = 3.402823466e+38f;
  float loc_t = big;
  int loc_k = -1;
  six_search(t, inside, lane_on, -big, loc_t, loc_k);
  bool take = (loc_k >= 0) & !(b1.t < loc_t);
  b1.t = take ? loc_t : b1.t;
  b1.prim = take ? first + loc_k : b1.prim;
  // hitable.rs:566 searches on from rec1.t + 0.0001; in fp32 at least from the next float above rec1.t (see scan_range)
  float t_next = b1.t + 0.0001f;
  {
    const int bits = __float_as_int(b1.t);
    const float up = __int_as_float(bits >= 0 ? bits + 1 : (bits == (int)0x80000000 ? 1 : bits - 1));
    t_next = t_next > b1.t ? t_next : up;
  }
  loc_t = big;
  loc_k = -1;
  six_search(t, inside, lane_on, t_next, loc_t, loc_k);
  take = (loc_k >= 0) & !(b2.t < loc_t);
  b2.t = take ? loc_t : b2.t;
  b2.prim = take ? first + loc_k : b2.prim;
}

__device__ __forceinline__ int wave_min_i32(int v) {
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}

// The two boundary searches of ConstantMedium::hit (hitable.rs:562-566) when the boundary is one sphere `f` (centre, radius): they are
// the two roots of one quadratic, so it is solved once.  Each search takes the nearer root inside its interval, else the
// farther one, exactly as Sphere::hit does.
template <bool STATS>
__device__ __forceinline__ void medium_sphere_searches(const float* f, int prim, bool on, const RayF& q, Hit& b1, Hit& b2, Counters& cnt) {
  float h, t_near, t_far;
  sphere_roots(f, q, h, t_near, t_far);
  if (STATS) cnt.sphere_tests += on ? 2u : 0u;
  const float big = 3.402823466e+38f;
  const bool n1 = (t_near < big) & (t_near > -big), f1 = (t_far < big) & (t_far > -big);
  const bool v1 = on & (h > 0.0f) & (n1 | f1);
  b1.t = v1 ? (n1 ? t_near : t_far) : b1.t;
  b1.prim = v1 ? prim : -1;
  float t_next = b1.t + 0.0001f;
  {
    const int bits = __float_as_int(b1.t);
    const float up = __int_as_float(bits >= 0 ? bits + 1 : (bits == (int)0x80000000 ? 1 : bits - 1));
    t_next = t_next > b1.t ? t_next : up;
  }
  const bool n2 = (t_near < big) & (t_near > t_next), f2 = (t_far < big) & (t_far > t_next);
  const bool v2 = on & (h > 0.0f) & (n2 | f2);
  b2.t = v2 ? (n2 ? t_near : t_far) : b2.t;
  b2.prim = v2 ? prim : -1;
}

// The rest of ConstantMedium::hit (hitable.rs:567-596) once the two boundary hits are known: the clamps, the draw, the hit.
// A hit replaces `best` unless best is nearer.
template <bool STATS>
__device__ __forceinline__ void medium_draw(bool on, const Hit& b1, const Hit& b2, const RayF& r, float t_min, float t_max, float density, int ordinal,
                                            int dprim, const PathRng& rng, Hit& best, Counters& cnt) {
  float t1 = b1.t, t2 = b2.t;
  t1 = t1 < t_min ? t_min : t1;
  t2 = t2 > t_max ? t_max : t2;
  bool ok = on & (b1.prim >= 0) & (b2.prim >= 0) & !(t1 >= t2);
  if (STATS) cnt.instance_ops += on;
  if (__ballot(ok) != 0ull) {  // the draw is only made where some ray crosses the medium
    t1 = t1 < 0.0f ? 0.0f : t1;
    const float dlen = sqrtf(dot3(r.d, r.d));
    const float inside = (t2 - t1) * dlen;
    const U4 mw = philox4x32_10(rng.pixel, rng.sample, rng.depth, 4096u + (unsigned)ordinal, rng.key0, rng.key1);
    const float hit_distance = -(1.0f / density) * __logf(u01(mw.x));
    const float t = t1 + hit_distance / dlen;
    ok &= (hit_distance < inside) & !(best.t < t);
    best.t = ok ? t : best.t;
    best.prim = ok ? dprim : best.prim;
  }
}

// ConstantMedium::hit, hitable.rs:560-598 (commented out in the reference), for the K_MEDIUM node `n` at scan[i] and the
// lanes in `on` (some lane is): the two boundary searches, then the draw.
template <bool STATS>
__device__ __forceinline__ void medium_node(const DevScene& S, int i, const ScanNode& n, bool on, const RayF& r, float t_min, float t_max,
                                            const PathRng& rng, Hit& best, Counters& cnt) {
  const RayF q = to_object_frame<true, true>(r, n.b >> 24, n.f[0], n.f[1], n.f[2], n.f[3], n.f[4]);
  const bool box6 = (n.b & LIST_IS_BOX6) != 0;
  const ScanNode ext = load_node(S.scan, n.skip - 1);  // density, ordinal, DevPrim index
  Hit b1, b2;
  b1.t = 3.402823466e+38f; b1.prim = -1;
  b2.t = 3.402823466e+38f; b2.prim = -1;
  const int n_boundary = n.b & 0x7fffff;
  const rt_prim bq = load_prim_uniform(S.prims, n.a);
  if (!box6 && n_boundary == 1 && (bq.type & RT_PRIM_TYPE_MASK) == RT_PRIM_SPHERE) {
    medium_sphere_searches<STATS>(bq.f, n.a, on, q, b1, b2, cnt);
  } else if (box6) {
    box6_two_searches<STATS>(S.scan, i, n.a, on, q, b1, b2, cnt);
  } else {
    list_test<STATS>(S.prims, n.a, n_boundary, on, q, -3.402823466e+38f, 3.402823466e+38f, b1, cnt);
    // hitable.rs:566 searches on from rec1.t + 0.0001.  Not in the reference, which works in f64: beyond t = 2048 that
    // sum is rec1.t itself in fp32 and a rect (whose test accepts t == t_min) would be found again, so the search
    // starts at the next float above rec1.t at least.
    float t_next = b1.t + 0.0001f;
    {
      const int bits = __float_as_int(b1.t);
      const float up = __int_as_float(bits >= 0 ? bits + 1 : (bits == (int)0x80000000 ? 1 : bits - 1));
      t_next = t_next > b1.t ? t_next : up;
    }
    list_test<STATS>(S.prims, n.a, n_boundary, on, q, t_next, 3.402823466e+38f, b2, cnt);
  }
  medium_draw<STATS>(on, b1, b2, r, t_min, t_max, ext.f[0], ext.a, ext.b, rng, best, cnt);
}

// The media of a scene on the stack traversal when each of them is bounded by one sphere in the world frame (the Book-2
// final scene's mist and smoke): everything ConstantMedium::hit needs of a medium in ONE 64-byte record, so a medium costs the
// wave one scalar load where the scan array's three records (node, scattering record, boundary primitive) cost it two
// round trips one after the other.  Record = a ScanNode whose box and y_top are the leaf's, f[0..3] the sphere, f[4] the
// density, skip the medium's ordinal, a the boundary primitive, b the DevPrim of the scattering point.
// Same tests in the same order as scan_range over the scan array of such a scene.
template <bool STATS>
__device__ __forceinline__ void media_sphere_pass(const DevScene& S, bool act, const RayF& r, float t_min, const PathRng& rng, Hit& best, Counters& cnt) {
  const Signs sg = ray_signs(r);
  for (int m = 0; m < S.n_media_sph; ++m) {
    const ScanNode n = load_node(S.media_sph, m);
    const bool on = act & slab_test(n, r, sg, t_min, INFINITY, 3.402823466e+38f);
    if (STATS) cnt.node_tests += act;
    if (__ballot(on) == 0ull) continue;
    Hit b1, b2;
    b1.t = 3.402823466e+38f; b1.prim = -1;
    b2.t = 3.402823466e+38f; b2.prim = -1;
    medium_sphere_searches<STATS>(n.f, n.a, on, r, b1, b2, cnt);
    medium_draw<STATS>(on, b1, b2, r, t_min, INFINITY, n.f[4], n.skip, n.b, rng, best, cnt);
  }
}

// The deferred scan of a flat array (the Cornell box, with MEDIA also its variants with smoke; rt_trace_closest): every slot
// it tests is a single rect, a single sphere, a list, a list under Translate/RotateY or a ConstantMedium, so the loop needs
// none of scan_range's bookkeeping for skipping lanes and sub-trees.  Scalar instructions take the same issue slots as vector ones (tools/probe_valu.sh), and
// the general loop spends about 25 of them per leaf on its dispatch.  Same tests in the same order as scan_range.
template <bool STATS, bool MEDIA = false>
__device__ __forceinline__ void scan_flat_lean(const DevScene& S, int end, bool active, const RayF& r, float t_min, float t_max, Hit& best,
                                               Counters& cnt, const PathRng* rng = nullptr) {
  const bool on_t = STATS ? active : true;  // the caller masks the result with `active` (trace_scan's `got`)
  if (end <= 0) return;
#ifdef RT_PROBE_CULL
  const Signs sg_cull = ray_signs(r);
#endif
  int i = 0;
  do {
    const ScanNode n = load_node(S.scan, i);
    const int kind = n.kind;
    float cand_t = 0.0f;
    int cand_prim = -1;
    bool take = false;
#ifdef RT_PROBE_CULL
    // Diagnostic build: a list's tests are skipped when no lane of the wave passes the list's box (raised to y_top, so
    // that a Cuboid's lid is inside it).  In the counting build sphere_tests counts the lanes such a skip served.
    if (kind == K_LIST || kind == K_OBJ_LIST) {
      const bool in_box = active & slab_test(n, r, sg_cull, t_min, t_max, 3.402823466e+38f);
      if (__ballot(in_box) == 0ull) {
        if (STATS) cnt.sphere_tests += active;
        i = n.skip;
        continue;
      }
    }
#endif
    if (kind <= K_RECT_YZ) {
      // A single rect: its own t_max test against the running best is the tie rule.
      if (kind == K_RECT_XY) take = rect_test<2, 0, 1>(n.f, r, t_min, best.t, cand_t);
      else if (kind == K_RECT_XZ) take = rect_test<1, 0, 2>(n.f, r, t_min, best.t, cand_t);
      else take = rect_test<0, 1, 2>(n.f, r, t_min, best.t, cand_t);
      take &= on_t;
      if (STATS) cnt.rect_tests += active;
      cand_prim = n.a;
    } else if (kind == K_SPHERE) {
      take = sphere_test(n.f, r, t_min, t_max, cand_t) & on_t;
      if (STATS) cnt.sphere_tests += active;
      take &= !(best.t < cand_t);
      cand_prim = n.a;
    } else if (kind == K_LIST) {
      if (n.b & LIST_IS_BOX6) box6_cand<STATS>(S.scan, i, n.a, on_t, r, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
      else list_cand<STATS>(S.prims, n.a, n.b & 0x7fffff, on_t, r, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
    } else if (kind == K_OBJ_LIST) {  // (what is left is a ConstantMedium: skipped in a query that leaves media out, rt_trace_closest)
      const RayF q = to_object_frame<true, true>(r, n.b >> 24, n.f[0], n.f[1], n.f[2], n.f[3], n.f[4]);
      if (STATS) cnt.instance_ops += active;
      if (n.b & LIST_IS_BOX6) box6_cand<STATS>(S.scan, i, n.a, on_t, q, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
      else list_cand<STATS>(S.prims, n.a, n.b & 0x7fffff, on_t, q, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
    } else if (MEDIA && kind == K_MEDIUM) {
      if (__ballot(active) != 0ull) medium_node<STATS>(S, i, n, active, r, t_min, t_max, *rng, best, cnt);
    }
    best.t = take ? cand_t : best.t;
    best.prim = take ? cand_prim : best.prim;
    i = n.skip;
  } while (i < end);
}

// Walks scan[begin, end) for the lanes with active == true, all 64 lanes of the
// wave taking part in the control flow.  LEVEL 0 is the world frame; a K_OBJ_BVH
// node re-enters at LEVEL 1 with the ray moved into the object's frame.  FLAT
// means every node is a leaf, so no lane ever skips ahead.  DEFER (flat arrays
// only) leaves the leaves' box tests out: trace_scan applies the winner's box
// test afterwards.
template <bool STATS, bool FLAT, bool MEDIA, int LEVEL, bool DEFER = false>
__device__ __forceinline__ void scan_range(const DevScene& S, int begin, int end, bool active, const RayF& r, float t_min, float t_max,
                                           const PathRng& rng, Hit& best, Counters& cnt) {
  const Signs sg = ray_signs(r);
  int lane_next = active ? begin : 0x7fffffff;
  int i = begin;
  while (i < end) {
    bool act = active;
    if (!FLAT) {
      act = lane_next <= i;
      if (__ballot(act) == 0ull) {
        i = __builtin_amdgcn_readfirstlane(min(wave_min_i32(lane_next), end));
        continue;
      }
    }
    // readfirstlane pins the index to an SGPR so the node arrives by scalar load.
    i = __builtin_amdgcn_readfirstlane(i);
    const ScanNode n = load_node(S.scan, i);
    // Hierarchical arrays also skip a box entered beyond the best hit so far.  A flat
    // array does not: its leaves are few, and a Cuboid's top face lies 0.01 above the
    // cuboid's own box (hitable.rs:398-399), where that shortcut is not exact.
    const bool on = DEFER ? act : (act & slab_test(n, r, sg, t_min, t_max, (FLAT || S.no_prune) ? 3.402823466e+38f : best.t));
    if (STATS && !DEFER) cnt.node_tests += act;
    // Without box tests every lane of the scan has the same `on`, the caller's `active`, and the caller masks the result
    // with it again (trace_scan's `got`): the primitive tests then run unmasked, which saves a mask operation per test.
    const bool on_t = (DEFER && !STATS) ? true : on;
    if (!FLAT) lane_next = (act & !on) ? n.skip : lane_next;
    const int kind = n.kind;
    // Every kind of leaf hands its winner to one update of the running best at the end of the trip (`take`: it replaces
    // the best so far), so the best hit lives in one pair of registers around the whole loop.
    float cand_t = 0.0f;
    int cand_prim = -1;
    bool take = false;
    if (kind >= K_RECT_XY && kind <= K_RECT_YZ) {
      // A single rect: its own t_max test against the running best is the tie rule.
      float t;
      bool h;
      if (kind == K_RECT_XY) h = rect_test<2, 0, 1>(n.f, r, t_min, best.t, t);
      else if (kind == K_RECT_XZ) h = rect_test<1, 0, 2>(n.f, r, t_min, best.t, t);
      else h = rect_test<0, 1, 2>(n.f, r, t_min, best.t, t);
      h &= on_t;
      if (STATS) cnt.rect_tests += on;
      take = h;
      cand_t = t;
      cand_prim = n.a;
    } else if (kind == K_SPHERE) {
      float t;
      bool h = sphere_test(n.f, r, t_min, t_max, t) & on_t;
      if (STATS) cnt.sphere_tests += on;
      h &= !(best.t < t);
      take = h;
      cand_t = t;
      cand_prim = n.a;
    } else if (kind == K_LIST) {
      if (__ballot(on) != 0ull) {
        if (n.b & LIST_IS_BOX6) box6_cand<STATS>(S.scan, i, n.a, on_t, r, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
        else list_cand<STATS>(S.prims, n.a, n.b & 0x7fffff, on_t, r, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
      }
    } else if (kind == K_OBJ_LIST || kind == K_OBJ_BVH) {
      if (LEVEL == 0) {
        if (__ballot(on) != 0ull) {
          const RayF q = to_object_frame<true, true>(r, n.b >> 24, n.f[0], n.f[1], n.f[2], n.f[3], n.f[4]);
          if (STATS) cnt.instance_ops += on;
          if (kind == K_OBJ_LIST) {
            if (n.b & LIST_IS_BOX6) box6_cand<STATS>(S.scan, i, n.a, on_t, q, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
            else list_cand<STATS>(S.prims, n.a, n.b & 0x7fffff, on_t, q, t_min, t_max, best.t, cand_t, cand_prim, take, cnt);
          }
          else if (!FLAT) scan_range<STATS, false, false, 1>(S, i + 1, n.skip, on, q, t_min, t_max, rng, best, cnt);
        }
        if (kind == K_OBJ_BVH) {
          i = n.skip;
          continue;
        }
      }
    }
    else if (MEDIA && kind == K_MEDIUM && LEVEL == 0) {
      if (__ballot(on) != 0ull) medium_node<STATS>(S, i, n, on, r, t_min, t_max, rng, best, cnt);
    }
    best.t = take ? cand_t : best.t;
    best.prim = take ? cand_prim : best.prim;
    // A leaf's skip is the next node (past any extension slots); an interior node's first child follows it.
    i = (kind == K_INTERIOR) ? i + 1 : n.skip;
  }
}

// HitableList::hit (hitable.rs:105-117) over prims[first, first+count) as one lane of the stack traversal meets
// it: per-lane loads, a per-lane loop.  An exact tie with the best hit so far goes to the leaf that comes later in
// depth-first order (see trace_stack).
template <bool STATS>
__device__ __forceinline__ void prim_list_lane(const DevScene& S, int first, int count, const RayF& r, float t_min, float t_max, Hit& best,
                                               Counters& cnt) {
  float loc_t = t_max;
  int loc_prim = -1;
  for (int qi = first; qi < first + count; ++qi) {
    const rt_prim q = *(const rt_prim*)((const char*)S.prims + ((unsigned)qi << 5));
    const int type = q.type & RT_PRIM_TYPE_MASK;
    float t = 0.0f;
    bool h;
    if (type == RT_PRIM_SPHERE) {
      h = sphere_test(q.f, r, t_min, loc_t, t);
      if (STATS) cnt.sphere_tests++;
    } else {
      if (type == RT_PRIM_XY_RECT) h = rect_test<2, 0, 1>(q.f, r, t_min, loc_t, t);
      else if (type == RT_PRIM_XZ_RECT) h = rect_test<1, 0, 2>(q.f, r, t_min, loc_t, t);
      else h = rect_test<0, 1, 2>(q.f, r, t_min, loc_t, t);
      if (STATS) cnt.rect_tests++;
    }
    if (h) {
      loc_t = t;
      loc_prim = qi;
    }
  }
  if (loc_prim < 0) return;
  bool take = loc_t < best.t;
  // an exact tie goes to the leaf that comes later in depth-first order; the ranks are only read then
  if (!take && !(best.t < loc_t)) take = best.prim < 0 || S.prim_rank[loc_prim] > S.prim_rank[best.prim];
  if (take) {
    best.t = loc_t;
    best.prim = loc_prim;
  }
}

// A list of six rects in Cuboid::new's order, as one lane meets it in the stack traversal: the 18-float record of
// box6_test read per lane (five 16-byte loads in flight together) and the same three pair tests, no loop and no dispatch.
template <bool STATS>
__device__ __forceinline__ void box6_lane(const DevScene& S, int rec, int first, const RayF& r, float t_min, float t_max, Hit& best, Counters& cnt) {
  const float4* src = (const float4*)((const char*)S.box6_recs + (unsigned)rec * 80u);
  const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
  if (first < 0) first = __float_as_int(v4.z);  // (a lane that came by the record alone: the record names its list)
  const float g[18] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y};
  float loc_t = t_max;
  int loc_k = -1;
  box_pair<2, 0, 1>(&g[0], 0, true, r, t_min, loc_t, loc_k);
  box_pair<1, 0, 2>(&g[6], 2, true, r, t_min, loc_t, loc_k);
  box_pair<0, 1, 2>(&g[12], 4, true, r, t_min, loc_t, loc_k);
  if (STATS) cnt.rect_tests += 6u;
  if (loc_k < 0) return;
  const int loc_prim = first + loc_k;
  bool take = loc_t < best.t;
  if (!take && !(best.t < loc_t)) take = best.prim < 0 || S.prim_rank[loc_prim] > S.prim_rank[best.prim];
  if (take) {
    best.t = loc_t;
    best.prim = loc_prim;
  }
}
template <bool STATS>
__device__ __forceinline__ void leaf_list_lane(const DevScene& S, int first, int count, const RayF& r, float t_min, float t_max, Hit& best, Counters& cnt) {
  int rec = -1;
  if (count == 6) rec = S.box6_index[first];
  if (rec >= 0) box6_lane<STATS>(S, rec, first, r, t_min, t_max, best, cnt);
  else prim_list_lane<STATS>(S, first, count, r, t_min, t_max, best, cnt);
}

// The lane's stack: a column of LDS (consecutive entries 64 dwords apart).  A walk's worst case (three waiting
// descendants per level) is several times what walks really use, and the LDS a wave holds decides how many waves a
// CU runs, which is what hides the node fetches: so the render kernel keeps only the first n_lds entries in LDS and
// the deeper ones in a column of global memory of the wave's own (`ovf`, same layout), touched by the rare walk that
// gets that deep.  The trace kernel keeps the whole stack in LDS (n_lds = the scene's depth, ovf unused).
#define RT_STACK_MAX 64  // deepest walk the stack traversal takes (LDS + overflow)
#ifndef RT_STACK_LDS_RENDER
#define RT_STACK_LDS_RENDER 10  // entries (of two dwords) the render kernel keeps in LDS
#endif
// (The two parts are pointers into their own address spaces, so that an access is a ds_ or a global_ instruction under
// a branch: with plain pointers the compiler selects between the two addresses and issues one FLAT access, which for
// the LDS part takes the global path's latency and waits on both memory counters.)
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(1))) unsigned glb_u32;
template <bool OVF>
struct LaneStackT {
  static constexpr bool kOvf = OVF;
  lds_u32* lds;  // this lane's column: entry `at` is the pair lds[at * 128] (reference), lds[at * 128 + 64] (distance key)
  int n_lds;     // entries in LDS
  glb_u32* ovf;  // this lane's column of the wave's overflow region, same layout (OVF only)
  float prune_scale;  // 1, or +inf for a scene walked without skipping boxes by distance (DevScene::no_prune): the best hit so far times this is what a box's entry distance is compared with
  __device__ __forceinline__ void put(int at, unsigned ref, unsigned key) {
    if (!OVF || at < n_lds) { lds[at * 128] = ref; lds[at * 128 + 64] = key; }
    else { ovf[(at - n_lds) * 128] = ref; ovf[(at - n_lds) * 128 + 64] = key; }
  }
  __device__ __forceinline__ unsigned get(int at) const {
    if (!OVF || at < n_lds) return lds[at * 128];
    return ovf[(at - n_lds) * 128];
  }
  __device__ __forceinline__ unsigned get_key(int at) const {
    if (!OVF || at < n_lds) return lds[at * 128 + 64];
    return ovf[(at - n_lds) * 128 + 64];
  }
};

// The state of one lane's walk through the four-wide tree.
struct Trav {
  RayF r;    // the ray in the frame of the sub-tree the lane is in
  int sp;    // entries on the lane's stack
  int cur;   // the reference to look at next, or TRAV_POP
  Hit best;
};
#define TRAV_POP (-1)
// True while `cur` is a list or instance leaf: the lane waits for the wave's leaf code.
__device__ __forceinline__ bool trav_holds_leaf(int cur) { return (cur != TRAV_POP) & ((cur & BN2_LEAF) != 0); }

// The root's own box (bvh.rs:198), then the root as a reference.  `ref` is the root's reference, `box` its node.
template <bool STATS>
__device__ __forceinline__ int trav_enter(const DevScene& S, int ref, int box, const RayF& r, float t_min, float t_max, Counters& cnt) {
  const float4 lo = *(const float4*)&S.bnodes[box];
  const float4 hi = *((const float4*)&S.bnodes[box] + 1);
  float te;
  if (STATS) cnt.node_tests++;
  // nothing to prune against yet at the world root; inside an object the root gets the box test only, as the first node of a BvhNode::hit
  const bool h = slab_box_enter(lo, hi, hi.y, r, ray_signs(r), t_min, t_max, 3.402823466e+38f, te);
  return h ? ref : TRAV_POP;
}

// One step of a lane that is not holding a leaf: pop if there is nothing to look at, then fetch a four-wide node,
// test its boxes and go to the nearest descendant the ray enters, pushing the others farthest first.  Returns false
// when the walk has ended.  (Leaves, single primitives included, are left to trav_leaf: the wave runs that code for
// many lanes at once, which was measured faster than testing a primitive where a lane meets it.)
// CHECKED: the build also serves scenes that must be walked without skipping boxes by distance (stk.prune_scale = +inf):
// only the most general render build and the nearest-hit query pay for the multiplication.
template <bool STATS, bool CHECKED, class STK>
__device__ __forceinline__ bool trav_step_impl(const DevScene& S, STK& stk, Trav& T, float t_min, float t_max, Counters& cnt) {
  if (T.cur == TRAV_POP) {
    // Entries wait on the stack with their pruning distance (the sort key of the step that pushed them: the distance
    // rounded down by its two lowest bits): what has been found since may lie nearer than an entry's box begins,
    // and then the entry goes without a fetch.  (An exact tie cannot hide in it: every hit under the box is farther.)
    if (T.sp == 0) return false;
    --T.sp;
    const unsigned x = stk.get(T.sp);
    const unsigned key = stk.get_key(T.sp);
    if (x == STACK_RESTORE) {
      // back in the world frame: the world ray waits under the marker (trav_leaf put it there), its reciprocals are made again
      const F3 wo = f3(__uint_as_float(stk.get(T.sp - 3)), __uint_as_float(stk.get(T.sp - 2)), __uint_as_float(stk.get(T.sp - 1)));
      const F3 wd = f3(__uint_as_float(stk.get_key(T.sp - 3)), __uint_as_float(stk.get_key(T.sp - 2)), __uint_as_float(stk.get_key(T.sp - 1)));
      T.sp -= 3;
      T.r = make_ray(wo, wd);
      return true;
    }
    // (no loop: a lane whose entry is pruned pops the next one in its next trip; looping here, or looking at two
    // entries at once, was measured slower)
    if (__uint_as_float(key & ~3u) > (CHECKED ? T.best.t * stk.prune_scale : T.best.t)) return true;
    T.cur = (int)x;
  }
  const int cur = T.cur;
  if (cur & BN2_LEAF) return true;  // (a leaf popped from the stack: the lane now holds it)
  // base + 32-bit byte offset: the loads take the array's address from scalar registers and one shifted index per lane
  const float4* rec = (const float4*)((const char*)S.bnodes4 + ((unsigned)cur << 7));
  const float4 lox = rec[0], loy = rec[1], loz = rec[2], hix = rec[3], hiy = rec[4], hiz = rec[5], refs_f = rec[6], top = rec[7];
  const Signs sg = ray_signs(T.r);
  float t0, t1, t2, t3;
  // (Skipping a box entered beyond the best hit so far takes for granted that what lies under a box lies inside it.  A
  // sphere of negative radius has an inverted box of its own, hitable.rs:86-91, so the boxes above a list that holds
  // one do not contain it: scenes with such a sphere are walked without that shortcut, as the reference walks every scene.)
  const float prune_t = CHECKED ? T.best.t * stk.prune_scale : T.best.t;
  const bool h0 = slab_core(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, top.x, T.r, sg, t_min, t_max, prune_t, t0);
  const bool h1 = slab_core(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, top.y, T.r, sg, t_min, t_max, prune_t, t1);
  const bool h2 = slab_core(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, top.z, T.r, sg, t_min, t_max, prune_t, t2);
  const bool h3 = slab_core(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, top.w, T.r, sg, t_min, t_max, prune_t, t3);
  const int r0 = __float_as_int(refs_f.x), r1 = __float_as_int(refs_f.y), r2 = __float_as_int(refs_f.z), r3 = __float_as_int(refs_f.w);
  if (STATS) cnt.node_tests += 2u + (r2 != BN4_EMPTY) + (r3 != BN4_EMPTY);  // (empty slots are the last ones)
  // Order by entry distance: the distance (not negative, so its bits order like an integer) with its two lowest bits
  // replaced by the slot number is one sortable key per descendant; a miss sorts last.  Which of two equally near
  // boxes comes first does not matter: the winner is decided by t and, on an exact tie, by leaf rank.
  unsigned k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xffffffffu;
  unsigned k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xffffffffu;
  unsigned k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xffffffffu;
  unsigned k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xffffffffu;
  unsigned lo_, hi_;
  lo_ = min(k0, k1); hi_ = max(k0, k1); k0 = lo_; k1 = hi_;
  lo_ = min(k2, k3); hi_ = max(k2, k3); k2 = lo_; k3 = hi_;
  lo_ = min(k0, k2); hi_ = max(k0, k2); k0 = lo_; k2 = hi_;
  lo_ = min(k1, k3); hi_ = max(k1, k3); k1 = lo_; k3 = hi_;
  lo_ = min(k1, k2); hi_ = max(k1, k2); k1 = lo_; k2 = hi_;
  const int n_hit = (int)h0 + (int)h1 + (int)h2 + (int)h3;
#define RT_REF_OF(k) (((k) & 2u) ? (((k) & 1u) ? r3 : r2) : (((k) & 1u) ? r1 : r0))
  if (n_hit >= 4) { stk.put(T.sp, (unsigned)RT_REF_OF(k3), k3); ++T.sp; }
  if (n_hit >= 3) { stk.put(T.sp, (unsigned)RT_REF_OF(k2), k2); ++T.sp; }
  if (n_hit >= 2) { stk.put(T.sp, (unsigned)RT_REF_OF(k1), k1); ++T.sp; }
  T.cur = n_hit >= 1 ? RT_REF_OF(k0) : TRAV_POP;
#undef RT_REF_OF
  return true;
}

// A stack with a global part decides at every access which part an entry lives in, a branch around each of a step's
// five accesses.  Walks seldom leave the LDS part, so the wave first asks whether any of its stepping lanes could (a
// step pops one entry or pushes three at most): if none can, the step runs on a view of the stack that has no global part.
template <bool STATS, bool CHECKED = true, class STK>
__device__ __forceinline__ bool trav_step(const DevScene& S, STK& stk, Trav& T, float t_min, float t_max, Counters& cnt) {
  if (STK::kOvf) {
    if (__ballot(T.sp + 3 > stk.n_lds) == 0ull) {
      LaneStackT<false> in_lds;
      in_lds.lds = stk.lds;
      in_lds.n_lds = stk.n_lds;
      in_lds.prune_scale = stk.prune_scale;
      in_lds.ovf = nullptr;
      return trav_step_impl<STATS, CHECKED>(S, in_lds, T, t_min, t_max, cnt);
    }
  }
  return trav_step_impl<STATS, CHECKED>(S, stk, T, t_min, t_max, cnt);
}

// The leaf a lane holds (its box has been passed): a primitive list in the current frame, or an object:
// Translate/RotateY over a list or over a BvhNode, whose sub-tree the lane enters in the object's frame with a
// marker on the stack that switches it back.  ConstantMedium leaves are left to the caller (they draw random numbers).
template <bool STATS, class STK>
__device__ __forceinline__ void trav_leaf(const DevScene& S, STK& stk, Trav& T, float t_min, float t_max, Counters& cnt) {
  const int node = T.cur & BN_INDEX_MASK;
  const bool single = (T.cur & BN2_PRIM) != 0;
  const bool by_record = (T.cur & BN2_BOX6) != 0;
  T.cur = TRAV_POP;
  if (by_record) {
    box6_lane<STATS>(S, node, -1, T.r, t_min, t_max, T.best, cnt);
    return;
  }
  int ref = node, count = 1;
  if (!single) {
    ref = ~S.bnodes[node].left;
    count = S.bnodes[node].right;
  }
  if (count > 0) {
    if (single) prim_list_lane<STATS>(S, ref, 1, T.r, t_min, t_max, T.best, cnt);
    else leaf_list_lane<STATS>(S, ref, count, T.r, t_min, t_max, T.best, cnt);
    return;
  }
  const rt_object ob = S.objects[ref];
  if (ob.kind >= RT_OBJ_MEDIUM) return;
  // (T.r is the ray in the frame the lane is in: the world's, or an enclosing object's)
  const RayF q = to_object_frame<true>(T.r, ob.xform, ob.sin_theta, ob.cos_theta, ob.offset[0], ob.offset[1], ob.offset[2]);
  if (STATS) cnt.instance_ops++;
  if (ob.kind == RT_OBJ_LIST) {
    leaf_list_lane<STATS>(S, ob.first, ob.count, q, t_min, t_max, T.best, cnt);
  } else {
    // the lane keeps one ray, the one of the frame it is in: the ray of the frame it leaves goes on the stack (three
    // entries: a component of the origin where a reference would be, of the direction where its key would be) under the marker
    stk.put(T.sp, __float_as_uint(T.r.o.x), __float_as_uint(T.r.d.x));
    stk.put(T.sp + 1, __float_as_uint(T.r.o.y), __float_as_uint(T.r.d.y));
    stk.put(T.sp + 2, __float_as_uint(T.r.o.z), __float_as_uint(T.r.d.z));
    stk.put(T.sp + 3, STACK_RESTORE, 0u);
    T.sp += 4;
    T.r = q;
    T.cur = trav_enter<STATS>(S, S.sub_root_ref[2 * ref], S.sub_root_ref[2 * ref + 1], T.r, t_min, t_max, cnt);
  }
}

// Per-lane nearest hit over the four-wide tree with a stack in LDS (one column per lane; consecutive entries
// are 64 dwords apart).  Used for scenes too large for the wave-uniform scan: every lane follows its own ray,
// nearer boxes first, skipping boxes entered beyond its best hit.  The reference visits both children and lets the
// right one win an exact tie (bvh.rs:204-217); visiting in another order gives the same winner when a tie goes to
// the leaf that comes later in depth-first order, so every primitive has the rank of its leaf in prim_rank[],
// read only on a tie.
// The walk of one tree for one lane, from its root's box test on: `root_ref` the root's reference, `box_node` the node
// that holds the root's box.
template <bool STATS, class STK>
__device__ __forceinline__ void trace_tree(const DevScene& S, STK& stk, int root_ref, int box_node, const RayF& w, float t_min, float t_max, Hit& best,
                                           Counters& cnt) {
  Trav T;
  T.r = w;
  T.sp = 0;
  T.best = best;
  T.cur = trav_enter<STATS>(S, root_ref, box_node, w, t_min, t_max, cnt);
  for (;;) {
    // down until this lane holds a list or instance leaf (lanes that have one wait for the rest of the wave,
    // so the leaf code runs once for many lanes instead of once per step)
    bool more = true;
    while (more && !trav_holds_leaf(T.cur)) more = trav_step<STATS>(S, stk, T, t_min, t_max, cnt);
    if (!more) break;
    trav_leaf<STATS>(S, stk, T, t_min, t_max, cnt);
  }
  best = T.best;
}
template <bool STATS>
__device__ __forceinline__ void trace_stack(const DevScene& S, unsigned* __restrict__ stk_lds, const RayF& w, float t_min, float t_max, Hit& best,
                                            Counters& cnt) {
  LaneStackT<false> stk;
  stk.lds = (lds_u32*)stk_lds;
  stk.n_lds = S.stack_lds;
  stk.prune_scale = S.no_prune ? INFINITY : 1.0f;
  stk.ovf = nullptr;
  trace_tree<STATS>(S, stk, S.root_ref, S.root_box, w, t_min, t_max, best, cnt);
}

// ConstantMedium::hit (hitable.rs:560-598, commented out in the reference) for the media whose boundary is a sub-tree:
// the two boundary searches are walks of that tree by the lane's own stack (free at this point: the lane's walk of the
// world has ended), the rest is scan_range's K_MEDIUM code.  All 64 lanes take part; `act` marks the ones with a ray.
template <bool STATS, class STK>
__device__ __forceinline__ void media_bvh_pass(const DevScene& S, STK& stk, bool act, const RayF& r, float t_min, float t_max, const PathRng& rng,
                                               Hit& best, Counters& cnt) {
  const float big = 3.402823466e+38f;
  for (int m = 0; m < S.n_media_bvh; ++m) {
    const MediumBvh mb = S.media_bvh[m];
    const rt_object ob = S.objects[mb.object];
    if (STATS) cnt.node_tests += act;
    const bool on = act & slab_box(mb.bmin[0], mb.bmin[1], mb.bmin[2], mb.bmax[0], mb.bmax[1], mb.bmax[2], r, ray_signs(r), t_min, t_max);
    if (__ballot(on) == 0ull) continue;
    const RayF q = to_object_frame<true, true>(r, ob.xform, ob.sin_theta, ob.cos_theta, ob.offset[0], ob.offset[1], ob.offset[2]);
    if (STATS) cnt.instance_ops += on;
    Hit b1, b2;
    b1.t = big; b1.prim = -1;
    b2.t = big; b2.prim = -1;
    const int root = S.sub_root_ref[2 * mb.object], root_box = S.sub_root_ref[2 * mb.object + 1];
    if (on) trace_tree<STATS>(S, stk, root, root_box, q, -big, big, b1, cnt);
    float t_next = b1.t + 0.0001f;
    {
      const int bits = __float_as_int(b1.t);
      const float up = __int_as_float(bits >= 0 ? bits + 1 : (bits == (int)0x80000000 ? 1 : bits - 1));
      t_next = t_next > b1.t ? t_next : up;
    }
    if (on & (b1.prim >= 0)) trace_tree<STATS>(S, stk, root, root_box, q, t_next, big, b2, cnt);
    float t1 = b1.t, t2 = b2.t;
    t1 = t1 < t_min ? t_min : t1;
    t2 = t2 > t_max ? t_max : t2;
    bool ok = on & (b1.prim >= 0) & (b2.prim >= 0) & !(t1 >= t2);
    if (__ballot(ok) != 0ull) {
      t1 = t1 < 0.0f ? 0.0f : t1;
      const float dlen = sqrtf(dot3(r.d, r.d));
      const float inside = (t2 - t1) * dlen;
      const U4 mw = philox4x32_10(rng.pixel, rng.sample, rng.depth, 4096u + (unsigned)mb.ordinal, rng.key0, rng.key1);
      const float hit_distance = -(1.0f / mb.density) * __logf(u01(mw.x));
      const float t = t1 + hit_distance / dlen;
      ok &= (hit_distance < inside) & !(best.t < t);
      best.t = ok ? t : best.t;
      best.prim = ok ? mb.dprim : best.prim;
    }
  }
}

// Nearest hit of BvhNode::hit (bvh.rs:197-228).  Must be called by all 64 lanes.
// TRAV is chosen by the host per scene: each variant is its own kernel so the
// register allocator only sees the code that scene can reach.
enum { TRAV_SCAN = 0, TRAV_FLAT = 1, TRAV_STACK = 2 };
template <bool STATS, int TRAV, bool MEDIA>
__device__ __forceinline__ Hit trace_scan(const DevScene& S, unsigned* stk, const float4* leaf_box, bool active, const RayF& w, float t_min,
                                          float t_max, const PathRng& rng, Counters& cnt) {
  Hit best;
  best.t = t_max;
  best.prim = -1;
  if (TRAV == TRAV_STACK) {
    if (active) trace_stack<STATS>(S, stk, w, t_min, t_max, best, cnt);
    // what is left in the scan array in this mode: the ConstantMedium nodes
    if (MEDIA && S.n_scan > 0) scan_range<STATS, true, true, 0>(S, 0, S.n_scan, active, w, t_min, t_max, rng, best, cnt);
  } else if (TRAV == TRAV_FLAT) {
    // A flat array of leaves: the nearest hit is the candidate with the smallest t,
    // the later leaf winning a tie, among those whose leaf box the ray also passes
    // (bvh.rs:198-203).  The box test almost never rejects a candidate whose own
    // test accepted it (only rounding at a box face can), so take the winner over
    // all candidates first and box-test it alone: if it passes it is also the
    // winner among the box-tested candidates, each of which is one of the
    // candidates it beat.  If it fails for any lane, the wave redoes the scan
    // with every box test in place.
    scan_flat_lean<STATS, MEDIA>(S, S.n_scan, active, w, t_min, t_max, best, cnt, &rng);
    const bool got = active & (best.prim >= 0);
    const float4* lb = leaf_box + 2 * (got ? best.prim : 0);
    const float4 b0 = lb[0], b1 = lb[1];
    if (STATS) cnt.node_tests += got;
    const bool ok = slab_box(b0.x, b0.y, b0.z, b1.x, b1.y, b1.z, w, ray_signs(w), t_min, t_max);
    if (__builtin_expect(__ballot(got & !ok) != 0ull, 0)) {
      best.t = t_max;
      best.prim = -1;
      scan_range<STATS, true, MEDIA, 0, false>(S, 0, S.n_scan, active, w, t_min, t_max, rng, best, cnt);
    }
  } else {
    scan_range<STATS, TRAV == TRAV_FLAT, MEDIA, 0>(S, 0, S.n_scan, active, w, t_min, t_max, rng, best, cnt);
  }
  return best;
}

struct Surface {
  F3 p, normal;
};

// The HitRecord fields color() needs (hitable.rs:175-182, 62-67), carried back
// to the world frame by RotateY::hit (hitable.rs:521-528) and Translate::hit
// (hitable.rs:442).  `q` is the hit primitive's DevPrim.
// A primitive inside more than one object (DP_CHAINED; `prim` is its index) goes through its objects one by one,
// outermost first on the way in and innermost first on the way back, as the nested hit() calls of the source do.
// CHAINS: the code for such primitives is only compiled into the builds that scenes with them are given (the render
// kernels' most general build, and the trace kernel), so that the other builds keep their registers.
template <bool CHAINS>
__device__ __forceinline__ Surface finalize_hit(const DevScene& S, const DevPrim& q, int prim, const RayF& w, float t, bool has_spheres) {
  Surface s;
  const int xform = (q.flags >> DP_XFORM_SHIFT) & 3;
  RayF r = to_object_frame<false>(w, xform, q.sin_t, q.cos_t, q.off[0], q.off[1], q.off[2]);
  const bool chained = CHAINS && S.has_chains != 0 && (q.flags & DP_CHAINED) != 0;
  int ch_first = 0, ch_len = 0;
  if (CHAINS && S.has_chains != 0 && __ballot(chained) != 0ull) {  // (wave-uniform: scenes without such nestings never come here)
    if (chained) {
      const int ob = S.prim_obj[prim];
      ch_first = S.obj_chain[2 * ob];
      ch_len = S.obj_chain[2 * ob + 1];
      r = w;
      for (int i = 0; i < ch_len; ++i) {
        const rt_object o = S.objects[S.chain_objs[ch_first + i]];
        r = to_object_frame<false>(r, o.xform, o.sin_theta, o.cos_theta, o.offset[0], o.offset[1], o.offset[2]);
      }
    }
  }
  F3 p = f3(fmaf(t, r.d.x, r.o.x), fmaf(t, r.d.y, r.o.y), fmaf(t, r.d.z, r.o.z));
  const int type = q.flags & DP_TYPE_MASK;
  const float sgn = (q.flags & DP_FLIP) ? -1.0f : 1.0f;
  F3 nrm = f3(type == RT_PRIM_YZ_RECT ? sgn : 0.0f, type == RT_PRIM_XZ_RECT ? sgn : 0.0f, type == RT_PRIM_XY_RECT ? sgn : 0.0f);
  if (has_spheres) {  // wave-uniform: a property of the scene
    // Not in the reference, which works in f64: in fp32 ray(t) lies off the sphere by the rounding of t * d
    // (1e-4 at a few thousand units), enough for the next ray to hit the same sphere again from inside
    // (measured: +14 % radiance on a scene of small spheres at coordinates up to 1000).  The point is put
    // back at distance |radius| from the centre; the normal is the reference's (p - c) / r of that point.
    const bool sph = type == RT_PRIM_SPHERE;
    const F3 dd = f3(p.x - q.f[0], p.y - q.f[1], p.z - q.f[2]);
    const float scale = fabsf(q.f[3]) * exact_rcp(sqrtf(dot3(dd, dd)));
    p = sel3(sph, f3(fmaf(dd.x, scale, q.f[0]), fmaf(dd.y, scale, q.f[1]), fmaf(dd.z, scale, q.f[2])), p);
    const float inv_r = exact_rcp(q.f[3]);
    nrm = sel3(sph, f3((p.x - q.f[0]) * inv_r, (p.y - q.f[1]) * inv_r, (p.z - q.f[2]) * inv_r), nrm);
  }
  nrm = sel3(type == DP_TYPE_MEDIUM, f3(1.0f, 0.0f, 0.0f), nrm);
  const bool ro = (xform & RT_XFORM_ROTATE_Y) != 0, tr = (xform & RT_XFORM_TRANSLATE) != 0;
  const F3 pl = p, nl = nrm;
  p.x = ro ? fmaf(q.cos_t, pl.x, q.sin_t * pl.z) : pl.x;
  p.z = ro ? fmaf(-q.sin_t, pl.x, q.cos_t * pl.z) : pl.z;
  nrm.x = ro ? fmaf(q.cos_t, nl.x, q.sin_t * nl.z) : nl.x;
  nrm.z = ro ? fmaf(-q.sin_t, nl.x, q.cos_t * nl.z) : nl.z;
  p = sel3(tr, f3(p.x + q.off[0], p.y + q.off[1], p.z + q.off[2]), p);
  if (CHAINS && S.has_chains != 0 && __ballot(chained) != 0ull) {
    if (chained) {
      for (int i = ch_len - 1; i >= 0; --i) {
        const rt_object o = S.objects[S.chain_objs[ch_first + i]];
        if (o.xform & RT_XFORM_ROTATE_Y) {
          const F3 pl2 = p, nl2 = nrm;
          p.x = fmaf(o.cos_theta, pl2.x, o.sin_theta * pl2.z);
          p.z = fmaf(-o.sin_theta, pl2.x, o.cos_theta * pl2.z);
          nrm.x = fmaf(o.cos_theta, nl2.x, o.sin_theta * nl2.z);
          nrm.z = fmaf(-o.sin_theta, nl2.x, o.cos_theta * nl2.z);
        }
        if (o.xform & RT_XFORM_TRANSLATE) p = f3(p.x + o.offset[0], p.y + o.offset[1], p.z + o.offset[2]);
      }
    }
  }
  s.p = p;
  s.normal = nrm;
  return s;
}

// ---------------------------------------------------------------------------
// trace_kernel
// ---------------------------------------------------------------------------
template <int TRAV>
__global__ void __launch_bounds__(256) trace_kernel(DevScene S, const float* __restrict__ o, const float* __restrict__ d, size_t n,
                                                    float t_min, float t_max, int* __restrict__ prim_id, float* __restrict__ t_out,
                                                    float* __restrict__ normal, float* __restrict__ p_out) {
  extern __shared__ unsigned trace_stack_lds[];  // TRAV_STACK: stack_lds entries (two dwords) of 64 lanes for each of the 4 waves
  unsigned* stk = trace_stack_lds + (TRAV == TRAV_STACK ? (threadIdx.x >> 6) * S.stack_lds * 128 : 0) + (threadIdx.x & 63);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // Whole waves stay in the loop together: the traversal uses wave-wide operations.
  const size_t n_round = (n + 63) / 64 * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_round; idx += stride) {
    const bool active = idx < n;
    const size_t at = active ? idx : 0;
    const RayF w = make_ray(f3(o[3 * at], o[3 * at + 1], o[3 * at + 2]), f3(d[3 * at], d[3 * at + 1], d[3 * at + 2]));
    Counters cnt;
    PathRng rng = {0u, 0u, 0u, 0u, 0u};  // a deterministic query takes no random numbers: media are skipped
    const Hit h = trace_scan<false, TRAV, false>(S, stk, S.leaf_box, active, w, t_min, t_max, rng, cnt);
    if (active) {
      prim_id[idx] = h.prim;
      t_out[idx] = h.prim >= 0 ? h.t : 0.0f;
      if (normal || p_out) {
        Surface s;
        s.p = f3(0, 0, 0);
        s.normal = f3(0, 0, 0);
        if (h.prim >= 0) s = finalize_hit<true>(S, S.dprims[h.prim], h.prim, w, h.t, S.has_spheres != 0);
        if (normal) { normal[3 * idx] = s.normal.x; normal[3 * idx + 1] = s.normal.y; normal[3 * idx + 2] = s.normal.z; }
        if (p_out) { p_out[3 * idx] = s.p.x; p_out[3 * idx + 1] = s.p.y; p_out[3 * idx + 2] = s.p.z; }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// trace64_kernel: the nearest hit in the REFERENCE'S OWN ARITHMETIC (parity mode)
// ---------------------------------------------------------------------------
// f64 throughout, a division where the source divides, no fused multiply-adds, and the source's traversal: every node
// of the tree as it was passed in gets its Aabb::hit (bvh.rs:198), both children are visited with the same t_max and the
// right one wins an exact tie (bvh.rs:204-217).  None of the render path's fp32 steps (outward-widened slabs, the
// perpendicular-offset discriminant, the re-projected sphere point) applies here.  It is a check, not a fast path: one
// ray per lane, a private stack, no pruning.
struct D3 {
  double x, y, z;
};
__device__ __forceinline__ double d3_get(const D3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
__device__ __forceinline__ double d3_dot(const D3& a, const D3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:176-178
struct Hit64 {
  double t;
  int prim;
  D3 p, n;
};

// Aabb::hit, bvh.rs:46-61
__device__ __forceinline__ bool aabb_hit64(const rt_node& n, const D3& o, const D3& d, double t_min, double t_max) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double inv_d = 1.0 / d3_get(d, a);
    double t0 = ((double)n.bmin[a] - d3_get(o, a)) * inv_d;
    double t1 = ((double)n.bmax[a] - d3_get(o, a)) * inv_d;
    if (inv_d < 0.0) { const double tmp = t0; t0 = t1; t1 = tmp; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    if (t_max <= t_min) return false;
  }
  return true;
}

// XYRect/XZRect/YZRect::hit (hitable.rs:163-183 and twins) and Sphere::hit (hitable.rs:49-84), FlipNormals applied
__device__ __forceinline__ bool prim_hit64(const rt_prim& q, int idx, const D3& o, const D3& d, double t0, double t1, Hit64& rec) {
  const int type = q.type & RT_PRIM_TYPE_MASK;
  if (type == RT_PRIM_SPHERE) {
    const D3 c = {(double)q.f[0], (double)q.f[1], (double)q.f[2]};
    const double radius = (double)q.f[3];
    const D3 oc = {o.x - c.x, o.y - c.y, o.z - c.z};
    const double a = d3_dot(d, d), b = d3_dot(oc, d), cc = d3_dot(oc, oc) - radius * radius;
    const double disc = b * b - a * cc;
    if (!(disc > 0.0)) return false;
    const double sq = sqrt(disc);
    for (int root = 0; root < 2; ++root) {
      const double temp = root == 0 ? (-b - sq) / a : (-b + sq) / a;
      if (temp < t1 && temp > t0) {
        rec.t = temp;
        rec.p = D3{o.x + d.x * temp, o.y + d.y * temp, o.z + d.z * temp};
        rec.n = D3{(rec.p.x - c.x) / radius, (rec.p.y - c.y) / radius, (rec.p.z - c.z) / radius};
        rec.prim = idx;
        return true;
      }
    }
    return false;
  }
  int c, ca, cb;
  if (type == RT_PRIM_XY_RECT) { c = 2; ca = 0; cb = 1; }
  else if (type == RT_PRIM_XZ_RECT) { c = 1; ca = 0; cb = 2; }
  else { c = 0; ca = 1; cb = 2; }
  const double a0 = q.f[0], a1 = q.f[1], b0 = q.f[2], b1 = q.f[3], k = q.f[4];
  const double t = (k - d3_get(o, c)) / d3_get(d, c);
  if (t < t0 || t > t1) return false;
  const double a = d3_get(o, ca) + t * d3_get(d, ca);
  const double b = d3_get(o, cb) + t * d3_get(d, cb);
  if (a < a0 || a > a1 || b < b0 || b > b1) return false;
  rec.t = t;
  rec.prim = idx;
  rec.p = D3{o.x + d.x * t, o.y + d.y * t, o.z + d.z * t};
  const double s = (q.type & RT_PRIM_FLIP) ? -1.0 : 1.0;
  // (the flipped zero components are -0.0 in the reference too: `-normal` negates every component, hitable.rs:341-353)
  rec.n = D3{c == 0 ? s : s * 0.0, c == 1 ? s : s * 0.0, c == 2 ? s : s * 0.0};
  return true;
}

// HitableList::hit, hitable.rs:105-117
__device__ __forceinline__ bool list_hit64(const DevScene& S, int first, int count, const D3& o, const D3& d, double t_min, double t_max, Hit64& out) {
  bool any = false;
  double closest = t_max;
  for (int i = first; i < first + count; ++i) {
    Hit64 rec;
    if (prim_hit64(S.prims[i], i, o, d, t_min, closest, rec)) {
      closest = rec.t;
      out = rec;
      any = true;
    }
  }
  return any;
}

// BvhNode::hit over the tree under `root` (bvh.rs:197-228), iteratively: the leaves are met left to right, every one is
// tested against the same (t_min, t_max), and a leaf's hit replaces the one held unless that one is strictly nearer,
// which is the source's `if left.t < right.t { left } else { right }` folded over the leaves.  LEVEL: how many objects
// the tree lies inside (0: the world tree); an object's sub-tree is walked by the next level's copy of the function,
// up to RT_MAX_NEST (rt_upload_scene refuses deeper nestings), so the recursion is unrolled at compile time.
template <int LEVEL>
__device__ bool tree_hit64(const DevScene& S, int root, const D3& o, const D3& d, double t_min, double t_max, Hit64& best);
template <>
__device__ bool tree_hit64<RT_MAX_NEST + 1>(const DevScene&, int, const D3&, const D3&, double, double, Hit64&) { return false; }

template <int LEVEL>
__device__ __forceinline__ bool object_hit64(const DevScene& S, const rt_object& ob, const D3& o, const D3& d, double t_min, double t_max, Hit64& rec) {
  // Translate::hit around RotateY::hit, hitable.rs:438-446, 508-532
  D3 lo = o, ld = d;
  const double cos_t = (double)ob.cos_theta, sin_t = (double)ob.sin_theta;
  const D3 off = {(double)ob.offset[0], (double)ob.offset[1], (double)ob.offset[2]};
  if (ob.xform & RT_XFORM_TRANSLATE) lo = D3{lo.x - off.x, lo.y - off.y, lo.z - off.z};
  if (ob.xform & RT_XFORM_ROTATE_Y) {
    const D3 po = lo, pd = ld;
    lo.x = cos_t * po.x - sin_t * po.z;
    lo.z = sin_t * po.x + cos_t * po.z;
    ld.x = cos_t * pd.x - sin_t * pd.z;
    ld.z = sin_t * pd.x + cos_t * pd.z;
  }
  bool hit;
  if (ob.kind == RT_OBJ_BVH) hit = tree_hit64<LEVEL + 1>(S, ob.first, lo, ld, t_min, t_max, rec);
  else hit = list_hit64(S, ob.first, ob.count, lo, ld, t_min, t_max, rec);
  if (!hit) return false;
  if (ob.xform & RT_XFORM_ROTATE_Y) {
    const D3 p = rec.p, n = rec.n;
    rec.p.x = cos_t * p.x + sin_t * p.z;
    rec.p.z = -sin_t * p.x + cos_t * p.z;
    rec.n.x = cos_t * n.x + sin_t * n.z;
    rec.n.z = -sin_t * n.x + cos_t * n.z;
  }
  if (ob.xform & RT_XFORM_TRANSLATE) rec.p = D3{rec.p.x + off.x, rec.p.y + off.y, rec.p.z + off.z};
  return true;
}

template <int LEVEL>
__device__ bool tree_hit64(const DevScene& S, int root, const D3& o, const D3& d, double t_min, double t_max, Hit64& best) {
  int stack[RT_MAX_TREE_DEPTH + 2];
  int sp = 0;
  stack[sp++] = root;
  bool any = false;
  while (sp > 0) {
    const int idx = stack[--sp];
    const rt_node n = S.ref_nodes[idx];
    if (!aabb_hit64(n, o, d, t_min, t_max)) continue;
    if (n.left >= 0) {
      stack[sp++] = n.right;  // (popped after the left one: leaves are met left to right)
      stack[sp++] = n.left;
      continue;
    }
    Hit64 rec;
    bool hit;
    const int ref = ~n.left;
    if (n.right > 0) {
      hit = list_hit64(S, ref, n.right, o, d, t_min, t_max, rec);
    } else {
      const rt_object ob = S.objects[ref];
      if (ob.kind >= RT_OBJ_MEDIUM) continue;  // a deterministic query skips media, as rt_trace_closest does
      if (LEVEL < RT_MAX_NEST) hit = object_hit64<LEVEL>(S, ob, o, d, t_min, t_max, rec);
      else continue;
    }
    if (hit && (!any || !(best.t < rec.t))) {
      best = rec;
      any = true;
    }
  }
  return any;
}

__global__ void __launch_bounds__(64) trace64_kernel(DevScene S, const double* __restrict__ o, const double* __restrict__ d, size_t n, double t_min,
                                                     double t_max, int* __restrict__ prim_id, double* __restrict__ t_out,
                                                     double* __restrict__ normal, double* __restrict__ p_out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const D3 ro = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, rd = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    Hit64 best;
    best.t = 0.0;
    best.prim = -1;
    best.p = D3{0.0, 0.0, 0.0};
    best.n = D3{0.0, 0.0, 0.0};
    Hit64 rec;
    if (S.n_ref_nodes > 0 && tree_hit64<0>(S, 0, ro, rd, t_min, t_max, rec)) best = rec;
    prim_id[i] = best.prim;
    t_out[i] = best.t;
    if (normal) { normal[3 * i] = best.n.x; normal[3 * i + 1] = best.n.y; normal[3 * i + 2] = best.n.z; }
    if (p_out) { p_out[3 * i] = best.p.x; p_out[3 * i + 1] = best.p.y; p_out[3 * i + 2] = best.p.z; }
  }
}

// Shading-side helpers.  Nothing from here on is compared bit for bit with the
// CPU (Monte Carlo sums are checked statistically), so the hardware's 1-ulp
// reciprocal, reciprocal square root and square root are used directly.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {  // vec3.rs:180-186
  return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ F3 unit3(F3 a) { return a * fast_rsq(dot3(a, a)); }  // vec3.rs:188-190

// sin(x) through v_sin_f32, which takes turns and needs its argument reduced
// first.  At |x| of a few thousand the result is good to about 1e-3, which moves
// a checker boundary by less than a thousandth of a cell.
__device__ __forceinline__ float fast_sin(float x) {
  const float turns = x * 0.15915494309189533577f;
  return __builtin_amdgcn_sinf(turns - floorf(turns));
}

// `f64 as usize` of perlin.rs:34-36 (saturating, negatives and NaN to 0), reduced
// mod 256 the way the wrapping `(i + di) & 255` of perlin.rs:43-45 sees it.
__device__ __forceinline__ unsigned perlin_cell(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 18446744073709551616.0f) return 255u;  // usize::MAX
  if (f >= 4294967296.0f) return 0u;               // a multiple of 512 at this magnitude
  return (unsigned)f & 255u;
}

// Perlin::noise (perlin.rs:29-50) with perlin_interpolate (perlin.rs:102-124).
__device__ __forceinline__ float perlin_noise(const rt_perlin* __restrict__ pn_generic, F3 p) {
  // (the tables are in global memory; said so, the reads are global_ loads and not FLAT ones, which the out-of-line
  // perlin_turb would otherwise get because it cannot see where its pointer comes from)
  const __attribute__((address_space(1))) rt_perlin* pn = (const __attribute__((address_space(1))) rt_perlin*)pn_generic;
  const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
  const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
  const unsigned i = perlin_cell(fx), j = perlin_cell(fy), k = perlin_cell(fz);
  const float uu = u * u * (3.0f - 2.0f * u);
  const float vv = v * v * (3.0f - 2.0f * v);
  const float ww = w * w * (3.0f - 2.0f * w);
  float accum = 0.0f;
  // the outer loop stays a loop: eight corners' table reads in flight at once cost more registers than the function has
#pragma unroll 1
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        const int idx = pn->perm_x[(i + di) & 255u] ^ pn->perm_y[(j + dj) & 255u] ^ pn->perm_z[(k + dk) & 255u];
        const __attribute__((address_space(1))) float* c = pn->ranvec[idx & 255];
        const float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
        accum += wi * wj * wk * (c[0] * (u - (float)di) + c[1] * (v - (float)dj) + c[2] * (w - (float)dk));
      }
  return accum;
}

// Perlin::turb (perlin.rs:52-63).
__device__ __noinline__ float perlin_turb(const rt_perlin* __restrict__ pn, F3 p, int depth) {
  float accum = 0.0f, weight = 1.0f;
  for (int o = 0; o < depth; ++o) {
    accum += weight * perlin_noise(pn, p);
    weight *= 0.5f;
    p = p * 2.0f;
  }
  return fabsf(accum);
}

// Texture::value (texture.rs:22-35).  A checker of checkers is followed down
// (the source recurses); eight levels are more than any scene here nests.
__device__ __forceinline__ F3 texture_value(const rt_texture* __restrict__ textures, const rt_perlin* __restrict__ perlins, int tex, F3 p) {
  for (int level = 0; level < 8; ++level) {
    const rt_texture t = textures[tex];
    if (t.type == RT_TEX_CHECKER) {
      const float sines = fast_sin(10.0f * p.x) * fast_sin(10.0f * p.y) * fast_sin(10.0f * p.z);
      tex = sines < 0.0f ? t.odd : t.even;
      continue;
    }
    if (t.type == RT_TEX_NOISE) {
      const float g = 0.5f * (1.0f + fast_sin(t.scale * p.z + 10.0f * perlin_turb(perlins + t.perlin, p, 7)));
      return f3(g, g, g);
    }
    return f3(t.color[0], t.color[1], t.color[2]);
  }
  return f3(0, 0, 0);
}

// Perlin::turb(p, 7) for the lanes in `need`, computed by the WHOLE wave one point at a time: 7 octaves x 8 lattice
// corners are 56 independent terms, so lane 8*o + c takes corner c (di, dj, dk = the bits of c, perlin_noise's loop
// order) of octave o and the point costs two rounds of table reads in all, where one lane walking the octaves alone
// makes twenty-eight rounds one after the other while the rest of the wave waits for it (textured vertices are rare:
// two or three lanes of a wave at a time).  The terms are added in perlin_noise's and perlin_turb's own order (corner
// sums on lane 8*o, then the octaves), so the value equals perlin_turb's bit for bit.  Must be called by all 64 lanes.
__device__ __forceinline__ float perlin_turb_wave(const rt_perlin* __restrict__ perlins, bool need, int perlin_index, F3 p) {
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
  const int o = lane >> 3;  // (lanes 56..63 would be an eighth octave: computed, never used)
  const int di = (lane >> 2) & 1, dj = (lane >> 1) & 1, dk = lane & 1;
  const float scale = __int_as_float((127 + o) << 23);  // 2^o: perlin_turb doubles p once per octave
  float result = 0.0f;
  unsigned long long m = __ballot(need);
  while (m != 0ull) {
    const int src = (int)__builtin_ctzll(m);
    m &= m - 1ull;
    const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.x), src));
    const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.y), src));
    const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p.z), src));
    const int pi = __builtin_amdgcn_readlane(perlin_index, src);
    const __attribute__((address_space(1))) rt_perlin* pn = (const __attribute__((address_space(1))) rt_perlin*)(perlins + pi);
    const float qx = px * scale, qy = py * scale, qz = pz * scale;
    const float fx = floorf(qx), fy = floorf(qy), fz = floorf(qz);
    const float u = qx - fx, v = qy - fy, w = qz - fz;
    const unsigned i = perlin_cell(fx), j = perlin_cell(fy), k = perlin_cell(fz);
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    const int idx = pn->perm_x[(i + di) & 255u] ^ pn->perm_y[(j + dj) & 255u] ^ pn->perm_z[(k + dk) & 255u];
    const __attribute__((address_space(1))) float* c = pn->ranvec[idx & 255];
    const float wi = di ? uu : 1.0f - uu, wj = dj ? vv : 1.0f - vv, wk = dk ? ww : 1.0f - ww;
    const float term = wi * wj * wk * (c[0] * (u - (float)di) + c[1] * (v - (float)dj) + c[2] * (w - (float)dk));
    // perlin_noise: accum = 0; accum += term(corner 0), ... term(corner 7), on the first lane of each octave's eight
    float noise = 0.0f + term;
#pragma unroll
    for (int cnr = 1; cnr < 8; ++cnr) noise += __shfl(term, (lane & ~7) + cnr, 64);
    // perlin_turb: accum += weight * noise, the weight halving per octave
    float accum = 0.0f, weight = 1.0f;
#pragma unroll
    for (int oct = 0; oct < 7; ++oct) {
      accum += weight * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(noise), 8 * oct));
      weight *= 0.5f;
    }
    result = lane == src ? fabsf(accum) : result;
  }
  return result;
}

// Texture::value for the lanes in `need` (the others get an unspecified colour).  Must be called by all 64 lanes: the
// checker chain is followed per lane, the turbulence of the noise textures is made by the whole wave when few lanes
// need it (perlin_turb_wave) and by each lane for itself when many do.
#ifndef RT_TURB_WAVE_MAX
#define RT_TURB_WAVE_MAX 12  // lanes up to which the wave makes their turbulence one lane after the other (beyond: every lane its own)
#endif
__device__ __forceinline__ F3 texture_value_wave(const rt_texture* __restrict__ textures, const rt_perlin* __restrict__ perlins, bool need, int tex, F3 p) {
  F3 col = f3(0, 0, 0);
  bool open = need, noise = false;
  float scale = 0.0f;
  int perlin_index = 0;
  for (int level = 0; level < 8 && __ballot(open) != 0ull; ++level) {
    if (open) {
      const rt_texture t = textures[tex];
      if (t.type == RT_TEX_CHECKER) {
        const float sines = fast_sin(10.0f * p.x) * fast_sin(10.0f * p.y) * fast_sin(10.0f * p.z);
        tex = sines < 0.0f ? t.odd : t.even;
      } else {
        open = false;
        noise = t.type == RT_TEX_NOISE;
        scale = t.scale;
        perlin_index = t.perlin;
        col = f3(t.color[0], t.color[1], t.color[2]);
      }
    }
  }
  const unsigned long long m_noise = __ballot(noise);
  if (m_noise != 0ull) {
    float turb = 0.0f;
    if (__popcll(m_noise) <= RT_TURB_WAVE_MAX) {
      turb = perlin_turb_wave(perlins, noise, perlin_index, p);
    } else if (noise) {
      turb = perlin_turb(perlins + perlin_index, p, 7);
    }
    const float g = 0.5f * (1.0f + fast_sin(scale * p.z + 10.0f * turb));
    col = sel3(noise, f3(g, g, g), col);
  }
  return col;
}

// reflect / refract / schlick of material.rs:162-180 (commented out in the reference).
__device__ __forceinline__ F3 reflect3(F3 v, F3 n) {
  const float k = 2.0f * dot3(v, n);
  return f3(v.x - k * n.x, v.y - k * n.y, v.z - k * n.z);
}

#define RT_INV_PI_F 0.31830988618379067154f

// Hitable::random / Hitable::pdf_value of a light shape other than the reference's single XZRect (which shade_vertex
// has inline): XY and YZ rects with XZRect's formulas (hitable.rs:245-264) on their own axes, spheres with the
// pdf_value/random of "Ray Tracing: The Rest of Your Life" (a cone towards the sphere).  The reference has no code
// for these; the oracle restates the same formulas.
__device__ __forceinline__ F3 light_random_general(int type, const float* f, F3 o, float xi1, float xi2) {
  // rect: a point of the rect minus o
  const float a = fmaf(xi1, f[1] - f[0], f[0]), b = fmaf(xi2, f[3] - f[2], f[2]), k = f[4];
  const F3 pt = type == RT_LIGHT_XZ_RECT ? f3(a, k, b) : (type == RT_LIGHT_XY_RECT ? f3(a, b, k) : f3(k, a, b));
  const F3 rect_dir = f3(pt.x - o.x, pt.y - o.y, pt.z - o.z);
  // sphere: a direction inside the cone it subtends
  const F3 dc = f3(f[0] - o.x, f[1] - o.y, f[2] - o.z);
  const float d2 = dot3(dc, dc);
  const F3 w = dc * fast_rsq(d2);
  const bool big_x = fabsf(w.x) > 0.9f;
  const F3 ax = big_x ? f3(0.0f, 1.0f, 0.0f) : f3(1.0f, 0.0f, 0.0f);
  const F3 v = unit3(cross3(w, ax));
  const F3 u = cross3(w, v);
  const float z = 1.0f + xi2 * (fast_sqrt(fmaxf(0.0f, 1.0f - f[3] * f[3] * fast_rcp(d2))) - 1.0f);
  const float sxy = fast_sqrt(fmaxf(0.0f, 1.0f - z * z));
  const float cx = __builtin_amdgcn_cosf(xi1) * sxy, cy = __builtin_amdgcn_sinf(xi1) * sxy;
  const F3 sph_dir = f3(fmaf(z, w.x, fmaf(cy, v.x, cx * u.x)), fmaf(z, w.y, fmaf(cy, v.y, cx * u.y)), fmaf(z, w.z, fmaf(cy, v.z, cx * u.z)));
  return sel3(type == RT_LIGHT_SPHERE, sph_dir, rect_dir);
}
// `type` and `f` are wave-uniform here (the caller loops over the shapes).
__device__ __forceinline__ float light_pdf_general(int type, const float* f, F3 o, F3 v, float dlen2, float inv_len) {
  if (type == RT_LIGHT_SPHERE) {
    const F3 oc = f3(o.x - f[0], o.y - f[1], o.z - f[2]);
    const float inv_a = fast_rcp(dlen2);
    const float k = dot3(oc, v) * inv_a;
    const F3 l = f3(fmaf(-k, v.x, oc.x), fmaf(-k, v.y, oc.y), fmaf(-k, v.z, oc.z));
    const float h = fmaf(f[3], f[3], -dot3(l, l));
    const float sq = fast_sqrt(h * inv_a);
    const float t_near = -k - sq, t_far = -k + sq;
    const bool hit = (h > 0.0f) & (((t_near < 3.402823466e+38f) & (t_near > 0.001f)) | ((t_far < 3.402823466e+38f) & (t_far > 0.001f)));
    const float cos_max = fast_sqrt(fmaxf(0.0f, 1.0f - f[3] * f[3] * fast_rcp(dot3(oc, oc))));
    const float solid = 6.28318530717958647692f * (1.0f - cos_max);
    return hit ? fast_rcp(solid) : 0.0f;
  }
  const float oc = type == RT_LIGHT_XZ_RECT ? o.y : (type == RT_LIGHT_XY_RECT ? o.z : o.x);
  const float vc = type == RT_LIGHT_XZ_RECT ? v.y : (type == RT_LIGHT_XY_RECT ? v.z : v.x);
  const float oa = type == RT_LIGHT_YZ_RECT ? o.y : o.x, va = type == RT_LIGHT_YZ_RECT ? v.y : v.x;
  const float ob = type == RT_LIGHT_XY_RECT ? o.y : o.z, vb = type == RT_LIGHT_XY_RECT ? v.y : v.z;
  const float t = (f[4] - oc) * fast_rcp(vc);
  const float la = fmaf(t, va, oa), lb = fmaf(t, vb, ob);
  const bool miss = (t < 0.001f) | (t > 3.402823466e+38f) | (la < f[0]) | (la > f[1]) | (lb < f[2]) | (lb > f[3]);
  const float area = (f[1] - f[0]) * (f[3] - f[2]);
  const float cosine = fabsf(vc * inv_len);
  return miss ? 0.0f : (t * t * dlen2) * fast_rcp(cosine * area);
}

// What one path vertex comes to: color() of main.rs:165-209 for the hit `h` of `ray`.
struct Shaded {
  F3 radiance;  // what this vertex adds to the sample (weight x emitted or sky), before de_nan
  F3 dir;       // direction of the ray that continues the path (not normalised, as in the source)
  F3 thr_next;  // the path's weight after this vertex
  F3 p;         // the hit point
  bool scattered, textured, emits;
  int mat_index;
};
// Every lane runs it; dead lanes and misses work on primitive 0 with their results masked by the caller.  `w` is the
// Philox call of this vertex.
// The 64-byte record of the primitive a lane hit (primitive 0 for a lane without a hit), as four 16-byte loads.  A caller
// with other work to do before it shades (the media, the vertex's Philox call) issues these first.
struct PrimRec {
  float4 q0, q1, q2, q3;
};
__device__ __forceinline__ PrimRec load_prim_rec(const float4* dprim_tab, bool alive, int prim) {
  const float4* src = (const float4*)((const char*)dprim_tab + ((unsigned)((alive & (prim >= 0)) ? prim : 0) << 6));
  PrimRec r;
  r.q0 = src[0]; r.q1 = src[1]; r.q2 = src[2]; r.q3 = src[3];
  return r;
}
template <bool STATS, int GENERAL>
__device__ __forceinline__ Shaded shade_vertex(const DevScene& S, const PrimRec& rec, bool alive, const Hit& h, const RayF& ray, F3 thr,
                                               int depth, const U4& w, int a_max_depth, int a_sky, unsigned pixel_id, unsigned sample,
                                               unsigned key0, unsigned key1, Counters& cnt) {
  Shaded out;
  // ---- color(), main.rs:165-209, one vertex; every lane runs it, dead lanes and
  // misses on primitive 0 with their results masked ----
  const bool hit = alive & (h.prim >= 0);
  DevPrim q;
  {
    const float4 q0 = rec.q0, q1 = rec.q1, q2 = rec.q2, q3 = rec.q3;
    __builtin_memcpy((char*)&q, &q0, 16);
    __builtin_memcpy((char*)&q + 16, &q1, 16);
    __builtin_memcpy((char*)&q + 32, &q2, 16);
    __builtin_memcpy((char*)&q + 48, &q3, 16);
  }
  const Surface sf = finalize_hit<GENERAL == 3>(S, q, hit ? h.prim : 0, ray, h.t, S.has_spheres != 0);
  const int mat_type = (q.flags >> DP_MAT_SHIFT) & 7;
  F3 color = f3(q.color[0], q.color[1], q.color[2]);
  if (STATS) cnt.hits += hit;
  // A checker or noise texture's colour comes at the top of the next trip (see there); until then it counts as white.
  const bool textured = GENERAL >= 2 && S.textured && hit && (q.flags & DP_TEXTURED);
  if (GENERAL >= 2) color = sel3(textured, f3(1.0f, 1.0f, 1.0f), color);

  // material.rs:197-203: a light emits towards the side its normal points to and never scatters.
  const bool is_light = mat_type == RT_MAT_DIFFUSE_LIGHT;
  const bool emits = hit & is_light & (dot3(sf.normal, ray.d) < 0.0f);
  // main.rs:203-208
  F3 sky = f3(0, 0, 0);
  if (a_sky == RT_SKY_GRADIENT) {
    const F3 ud = unit3(ray.d);
    const float t = 0.5f * (ud.y + 1.0f);
    sky = f3((1.0f - t) + t * 0.5f, (1.0f - t) + t * 0.7f, (1.0f - t) + t * 1.0f);
  }
  const F3 le = sel3(emits, (GENERAL >= 2) ? sel3(textured, f3(0, 0, 0), color) : color, sel3(hit, f3(0, 0, 0), sky));
  out.radiance = f3(thr.x * le.x, thr.y * le.y, thr.z * le.z);  // what this vertex adds to the sample, before de_nan

  // Lambertian under MixturePDF(HitablePDF(light), CosinePDF(normal)), main.rs:173-197.
  const bool scatters = hit & (mat_type == RT_MAT_LAMBERTIAN) & (depth < a_max_depth);
  if (STATS) cnt.shade_events += scatters;
  const float xi1 = u01(w.z), xi2 = u01(w.w);
  // onb.rs:10-21.  The normal is already of unit length (to rounding), and the cross
  // product with an axis vector is written out: w x (0,1,0) = (-w.z, 0, w.x),
  // w x (1,0,0) = (0, w.z, -w.y).
  const F3 ww = sf.normal;
  const bool big_x = fabsf(ww.x) > 0.9f;
  const float other = big_x ? ww.x : ww.y;
  const float inv_v = fast_rsq(fmaf(ww.z, ww.z, other * other));
  const F3 vv = f3(big_x ? -ww.z * inv_v : 0.0f, big_x ? 0.0f : ww.z * inv_v, big_x ? ww.x * inv_v : -ww.y * inv_v);
  const F3 uu = cross3(ww, vv);
  const bool pick_light = (S.n_lights > 0) & ((w.y >> 31) != 0u);  // MixturePDF's coin, pdf.rs:70
  // XZRect::random, hitable.rs:256-264
  F3 dir_light = f3(fmaf(xi1, S.light.f[1] - S.light.f[0], S.light.f[0]) - sf.p.x, S.light.f[4] - sf.p.y,
                    fmaf(xi2, S.light.f[3] - S.light.f[2], S.light.f[2]) - sf.p.z);
  if (GENERAL == 3 && S.light_general) {
    // HitablePDF over a list of shapes: random() picks one uniformly; counter (pixel, sample, depth, 8192), word 0
    rt_light lt = S.light;
    if (S.n_lights > 1) {
      const U4 lw = philox4x32_10(pixel_id, sample, (unsigned)depth, 8192u, key0, key1);
      const unsigned pick = (unsigned)(((unsigned long long)lw.x * (unsigned long long)(unsigned)S.n_lights) >> 32);
      const float4* src = (const float4*)((const char*)S.lights + (pick << 5));
      const float4 l0 = src[0], l1 = src[1];
      __builtin_memcpy((char*)&lt, &l0, 16);
      __builtin_memcpy((char*)&lt + 16, &l1, 16);
    }
    dir_light = light_random_general(lt.type, lt.f, sf.p, xi1, xi2);
  }
  // random_cosine_direction, main.rs:239-249 (x and y keep the source's factor 2);
  // v_sin/v_cos take their argument in turns, so phi = 2*pi*r1 is r1 turns.
  const float cz = fast_sqrt(1.0f - xi2);
  const float r2s = 2.0f * fast_sqrt(xi2);
  const float cx = __builtin_amdgcn_cosf(xi1) * r2s;
  const float cy = __builtin_amdgcn_sinf(xi1) * r2s;
  const F3 dir_cos = f3(fmaf(cz, ww.x, fmaf(cy, vv.x, cx * uu.x)), fmaf(cz, ww.y, fmaf(cy, vv.y, cx * uu.y)), fmaf(cz, ww.z, fmaf(cy, vv.z, cx * uu.z)));
  F3 dir = sel3(pick_light, dir_light, dir_cos);
  const float dlen2 = dot3(dir, dir);
  const float inv_len = fast_rsq(dlen2);
  // pdf.rs:19-26
  const float cos_w = dot3(dir, ww) * inv_len;
  float pdf_val = cos_w > 0.0f ? cos_w * RT_INV_PI_F : 0.0f;
  if (S.n_lights > 0) {
    // XZRect::pdf_value, hitable.rs:245-254 (its hit test uses 0.001 and f64::MAX)
    if (STATS) cnt.light_tests += scatters;
    const float t = (S.light.f[4] - sf.p.y) * fast_rcp(dir.y);
    const float lx = fmaf(t, dir.x, sf.p.x), lz = fmaf(t, dir.z, sf.p.z);
    const bool miss = (t < 0.001f) | (t > 3.402823466e+38f) | (lx < S.light.f[0]) | (lx > S.light.f[1]) | (lz < S.light.f[2]) | (lz > S.light.f[3]);
    const float area = (S.light.f[1] - S.light.f[0]) * (S.light.f[3] - S.light.f[2]);
    const float cosine = fabsf(dir.y * inv_len);
    float lp = miss ? 0.0f : (t * t * dlen2) * fast_rcp(cosine * area);
    if (GENERAL == 3 && S.light_general) {
      // pdf_value() of the list: the mean of the shapes' (a wave-uniform loop: the records arrive by scalar load)
      lp = 0.0f;
      for (int k = 0; k < S.n_lights; ++k) {
        const u32x8 v8 = *(const RT_CONSTANT_AS u32x8*)((const RT_CONSTANT_AS char*)S.lights + (unsigned)k * 32u);
        rt_light lk;
        __builtin_memcpy(&lk, &v8, sizeof(lk));
        lp += light_pdf_general(lk.type, lk.f, sf.p, dir, dlen2, inv_len);
        if (STATS && k > 0) cnt.light_tests += scatters;
      }
      lp *= fast_rcp((float)S.n_lights);
    }
    pdf_val = 0.5f * lp + 0.5f * pdf_val;
  }
  // material.rs:76-82
  const float cos_n = fmaxf(dot3(sf.normal, dir) * inv_len, 0.0f);
  const float f = (cos_n * RT_INV_PI_F) * fast_rcp(pdf_val);
  F3 thr_next = f3(thr.x * color.x * f, thr.y * color.y * f, thr.z * color.z * f);
  bool scattered = scatters;
  if (GENERAL && (S.mat_mask & ((1 << RT_MAT_METAL) | (1 << RT_MAT_DIELECTRIC) | (1 << RT_MAT_ISOTROPIC)))) {
    // Materials of the reference's older scatter() -> (attenuation, ray) form
    // (material.rs:100-112, 125-159, 217-221, commented out there): the path goes on
    // along the returned ray weighted by the attenuation, with no pdf involved.
    const bool deep = depth < a_max_depth;
    const bool is_metal = hit & deep & (mat_type == RT_MAT_METAL);
    const bool is_diel = hit & deep & (mat_type == RT_MAT_DIELECTRIC);
    const bool is_iso = hit & deep & (mat_type == RT_MAT_ISOTROPIC);
    if (__ballot(is_metal | is_diel | is_iso) != 0ull) {
      // random_in_unit_sphere (main.rs:211-221): attempt a draws counter (pixel, sample, depth, 256+a)
      F3 rs = f3(0, 0, 0);
      bool looking = is_metal | is_iso;
      unsigned attempt = 0;
      while (__ballot(looking) != 0ull) {
        const U4 sw = philox4x32_10(pixel_id, sample, (unsigned)depth, 256u + attempt, key0, key1);
        const F3 c = f3(2.0f * u01(sw.x) - 1.0f, 2.0f * u01(sw.y) - 1.0f, 2.0f * u01(sw.z) - 1.0f);
        const bool ok = looking & (dot3(c, c) < 1.0f);
        rs = sel3(ok, c, rs);
        looking &= !ok;
        ++attempt;
      }
      const F3 n = sf.normal;
      // Metal::scatter
      const F3 refl_unit = reflect3(unit3(ray.d), n);
      const F3 m_dir = f3(refl_unit.x + q.fuzz * rs.x, refl_unit.y + q.fuzz * rs.y, refl_unit.z + q.fuzz * rs.z);
      const bool m_ok = dot3(m_dir, n) > 0.0f;
      // Dielectric::scatter (reflects the un-normalised incoming direction, material.rs:126)
      const F3 reflected = reflect3(ray.d, n);
      const float dn = dot3(ray.d, n);
      const float d_len = fast_sqrt(dot3(ray.d, ray.d));
      const bool outside = !(dn > 0.0f);
      const F3 on_ = outside ? n : f3(-n.x, -n.y, -n.z);
      const float ni_over_nt = outside ? fast_rcp(q.ref_idx) : q.ref_idx;
      const float cosine = outside ? -dn * fast_rcp(d_len) : q.ref_idx * dn * fast_rcp(d_len);
      const F3 uvd = unit3(ray.d);
      const float dt = dot3(uvd, on_);
      const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
      const float sq = fast_sqrt(disc);
      const F3 refracted = f3(ni_over_nt * (uvd.x - on_.x * dt) - on_.x * sq, ni_over_nt * (uvd.y - on_.y * dt) - on_.y * sq,
                              ni_over_nt * (uvd.z - on_.z * dt) - on_.z * sq);
      float r0 = (1.0f - q.ref_idx) * fast_rcp(1.0f + q.ref_idx);
      r0 = r0 * r0;
      const float om = 1.0f - cosine;
      const float reflect_prob = disc > 0.0f ? r0 + (1.0f - r0) * (om * om) * (om * om) * om : 1.0f;
      const float xi = u01(w.y << 1);
      const F3 d_dir = xi < reflect_prob ? reflected : refracted;
      // Isotropic::scatter
      const F3 s_dir = is_metal ? m_dir : (is_diel ? d_dir : rs);
      const F3 atten = is_diel ? f3(1.0f, 1.0f, 1.0f) : color;
      const bool s_any = is_metal | is_diel | is_iso;
      const bool s_ok = (is_metal & m_ok) | is_diel | is_iso;
      dir = sel3(s_any, s_dir, dir);
      thr_next = sel3(s_any, f3(thr.x * atten.x, thr.y * atten.y, thr.z * atten.z), thr_next);
      scattered = s_any ? s_ok : scattered;
    }
  }
  out.dir = dir;
  out.thr_next = thr_next;
  out.p = sf.p;
  out.scattered = scattered;
  out.textured = textured;
  out.emits = emits;
  out.mat_index = q.flags >> DP_INDEX_SHIFT;
  return out;
}

struct RenderArgs {
  int nx, ny, spp, sample_begin, max_depth, sky;
  float t_min, inv_nx, inv_ny;
  int shard_index, shard_count, shard_width;
  int n_tiles;  // tiles this launch works on: all of them, or this shard's tile numbers (some of them beyond the image's edge)
  unsigned key0, key1;
  int tiles_x, tiles_y;
  int n_chunks, chunk_spp;  // the samples of a tile are split into n_chunks work items of chunk_spp samples,
  int split_parts;          // the last of which is cut into this many parts, each half the one before
  int ready_lanes;          // render_stack_kernel: finished rays in a wave (in 64 live ones) that end a traversal phase
  int hold_lanes;           // and lanes holding a leaf (in 64 traversing ones) that make the wave run the leaf code
  int accumulate;           // render_stack_kernel: items ADD their tile sums to the (zeroed) output instead of writing partial images
  int max_slots;            // render_stack_kernel: work items a wave may have open at once (1: a pixel's sum does not depend on scheduling)
  int pool_slots;           // render_pool_kernel: path slots per wave (0: the launch uses render_stack_kernel)
  void* pool_mem;           //   pool_slots 64-byte records for every wave of the grid
  int refill_lanes;         //   finished walks in a wave that make it hand them over and take new rays
  int shade_batch;          //   paths waiting to be shaded (or started) that make the wave run the shading code
  int tex_batch;            //   paths waiting for a Texture::value that make the wave run the texture code
  int hungry_lanes;         //   idle lanes, with the ray queue empty, that make the wave run a batch of any size
  int stack_lds;            // render_stack_kernel: entries of a lane's stack kept in LDS; deeper ones go to stack_ovf
  unsigned* stack_ovf;      // (scene's stack depth - stack_lds) x 128 dwords for every wave of the grid
};

// Waves per SIMD the register allocator must leave room for (launch_bounds' second argument).
#ifndef RT_RENDER_WAVES_PER_SIMD
#define RT_RENDER_WAVES_PER_SIMD 7
#endif


// ---------------------------------------------------------------------------
// render_kernel
// ---------------------------------------------------------------------------
// GENERAL = 0 is the build for scenes made of Lambertian and DiffuseLight
// surfaces with constant textures and no media (the Cornell box); 1 adds Metal,
// Dielectric, Isotropic and ConstantMedium; 2 adds checker and noise textures; 3
// adds light shapes other than one XZRect (lists of shapes, XY/YZ rects, spheres).
template <bool STATS, bool SUMSQ, int TRAV, int GENERAL>
#ifndef RT_FLAT_G1_WAVES
#define RT_FLAT_G1_WAVES 5  // (the allocator takes 79 registers by itself, six waves per SIMD; bounding it to 72 for seven was measured: see the experiments)
#endif
__global__ void __launch_bounds__(64, (TRAV == TRAV_STACK && GENERAL < 2) ? 6 : ((TRAV == TRAV_FLAT && GENERAL == 1) ? RT_FLAT_G1_WAVES : ((GENERAL != 0 || TRAV == TRAV_STACK) ? 5 : RT_RENDER_WAVES_PER_SIMD))) render_kernel(DevScene S, DevCamera cam, RenderArgs A, float* __restrict__ out_sum,
                                                     float* __restrict__ out_sumsq, unsigned long long* __restrict__ stats, unsigned* __restrict__ queue) {
  // Single-wave workgroups: every wave is on its own, none waits for the slowest
  // wave of a larger group.
  __shared__ float acc[1][6][64];
  // A flat scene's primitive tables live in LDS (4 + 2 float4 per primitive, the
  // launch sizes the array): a path vertex then costs no global memory access.
  extern __shared__ float4 prim_lds[];
  // The same launch-sized array is the traversal stack of a large scene: stack_depth entries of 64 lanes.
  unsigned* stk = (unsigned*)prim_lds + (threadIdx.x & 63);
  const int lane = threadIdx.x & 63;
  // The camera (19 floats, used once per path) lives in LDS rather than in scalar
  // registers for the life of the kernel; every lane reads the same address.
  __shared__ float cam_lds[20];
  __shared__ __align__(16) unsigned rk_lds[20];  // Philox's round keys
  if (lane < 10) {
    rk_lds[2 * lane] = A.key0 + (unsigned)lane * 0x9E3779B9u;
    rk_lds[2 * lane + 1] = A.key1 + (unsigned)lane * 0xBB67AE85u;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cam_lds[k] = cam.origin[k];
      cam_lds[3 + k] = cam.llc[k];
      cam_lds[6 + k] = cam.horizontal[k];
      cam_lds[9 + k] = cam.vertical[k];
      cam_lds[12 + k] = cam.u[k];
      cam_lds[15 + k] = cam.v[k];
    }
    cam_lds[18] = cam.lens_radius;
  }
  const bool has_lens = cam.lens_radius != 0.0f;
  const float4* dprim_tab = (const float4*)S.dprims;
  const float4* box_tab = S.leaf_box;
  if (TRAV == TRAV_FLAT) {
    for (int k = lane; k < S.n_dprims * 4; k += 64) prim_lds[k] = dprim_tab[k];
    for (int k = lane; k < S.n_dprims * 2; k += 64) prim_lds[S.n_dprims * 4 + k] = box_tab[k];
    dprim_tab = prim_lds;
    box_tab = prim_lds + S.n_dprims * 4;
    __syncthreads();
  }
  const int wib = 0;
  Counters cnt;
  memset(&cnt, 0, sizeof(cnt));
  // One work item = (tile, sample chunk), run by one wave from start to end.  With
  // more than one chunk each item writes its partial sums to its own slab;
  // reduce_slabs_kernel adds the slabs in a fixed order, so the result does not
  // depend on scheduling.  The launch has one wave per hardware slot and the waves
  // take items from a counter until none are left: workgroups are handed to the
  // shader engines in equal numbers although the engines do not have equal numbers
  // of CUs, so a launch of one workgroup per item leaves the larger engines waiting.
  const int n_tiles = A.n_tiles;
  const int n_items = n_tiles * A.n_chunks;
  // The kernel's arguments arrive as sixteen-register loads, and a value used from one
  // keeps all sixteen alive (and spilled and reloaded together) for the life of the
  // loop.  The ones the path loop reads are copied to registers of their own here.
  float a_t_min = A.t_min;
  int a_max_depth = A.max_depth, a_sky = A.sky;
  asm volatile("" : "+s"(a_t_min), "+s"(a_max_depth), "+s"(a_sky));
  for (;;) {
  int item = 0;
  if (lane == 0) item = (int)atomicAdd(queue, 1u);
  item = __builtin_amdgcn_readfirstlane(item);
  if (item >= n_items) break;
  const int chunk = item / n_tiles;
  const int tile = item - chunk * n_tiles;
  int tx = tile % A.tiles_x, ty = tile / A.tiles_x;
  bool mine = true;
  if (A.shard_count > 1) {
    // A shard's launch runs over its own tiles only: the numbers shard_index + k * shard_count of a grid shard_width
    // tiles wide (rt_shard_width), columns beyond the image's edge empty.  The host has zeroed the output, so other
    // shards' pixels stay 0.
    const int t = A.shard_index + tile * A.shard_count;
    tx = t % A.shard_width;
    ty = t / A.shard_width;
    mine = (tx < A.tiles_x) & (ty < A.tiles_y);
  }
  const size_t slab = (size_t)chunk * A.nx * A.ny * 3;
  float* __restrict__ o_sum = out_sum + slab;
  float* __restrict__ o_sumsq = SUMSQ ? out_sumsq + slab : nullptr;
#pragma unroll
  for (int c = 0; c < (SUMSQ ? 6 : 3); ++c) acc[wib][c][lane] = 0.0f;

  // Items are handed out chunk by chunk, so the launch ends on the last chunk's items:
  // that chunk comes in halving parts, which keeps the waves' finishing times close.
  const int n_whole = A.n_chunks - A.split_parts;
  int s_first = chunk * A.chunk_spp, s_len = A.chunk_spp;
  if (chunk >= n_whole) {
    const int k = chunk - n_whole;
    const int last = A.spp - n_whole * A.chunk_spp;
    const int o0 = last - (last >> k);
    const int o1 = k + 1 < A.split_parts ? last - (last >> (k + 1)) : last;
    s_first = n_whole * A.chunk_spp + o0;
    s_len = o1 - o0;
  }
  const int s_count = mine ? s_len : 0;
  const int pool_size = 64 * s_count;
  int pool_next = 0;

  // Per-lane path state.  A lane is either on a path (alive) or waiting for an
  // item; `fresh` marks a path that has not traced its camera ray yet.
  bool alive = false, fresh = false;
  F3 ro = f3(0, 0, 0), rd = f3(0, 0, 1);  // ray to trace next (camera ray or scattered ray)
  F3 thr = f3(0, 0, 0);                   // running product of albedo * scattering_pdf / pdf
  int depth = 0;
  int pix = 0;            // pixel within the tile
  unsigned pixel_id = 0;  // j*nx + i
  unsigned sample = 0;
  int pend = -1;          // GENERAL >= 2: a colour still to come from Texture::value, material index << 1 | ended-on-a-light

  for (;;) {
    if (GENERAL >= 2 && S.textured) {
      // ---- Texture::value of the previous vertex (texture.rs:22-35).  It is evaluated here, where only the path
      // state is live, instead of in the middle of the shading code: Perlin turbulence is a call, and everything live
      // across a call has to sit in the registers the callee preserves.  The vertex used colour (1,1,1); a path that
      // went on gets its weight multiplied now, one that ended on a textured light adds its sample now. ----
      if (__ballot(pend >= 0) != 0ull) {
        const F3 c = texture_value_wave(S.textures, S.perlins, pend >= 0, pend >= 0 ? S.materials[pend >> 1].texture : 0, ro);
        if (pend >= 0) {
          const F3 v = f3(thr.x * c.x, thr.y * c.y, thr.z * c.z);
          if (pend & 1) {
            const float r = v.x != v.x ? 0.0f : v.x, g = v.y != v.y ? 0.0f : v.y, b = v.z != v.z ? 0.0f : v.z;  // de_nan
            if (r != 0.0f || g != 0.0f || b != 0.0f) {
              atomicAdd(&acc[wib][0][pix], r);
              atomicAdd(&acc[wib][1][pix], g);
              atomicAdd(&acc[wib][2][pix], b);
              if (SUMSQ) {
                atomicAdd(&acc[wib][3][pix], r * r);
                atomicAdd(&acc[wib][4][pix], g * g);
                atomicAdd(&acc[wib][5][pix], b * b);
              }
            }
          } else {
            thr = v;
          }
        }
        pend = -1;
      }
    }
    // ---- hand (pixel, sample) items to the lanes without a path (main.rs:96-101):
    // ballot the idle lanes, rank them, take consecutive items from the pool ----
    const unsigned long long need = __ballot(!alive);
    if (need != 0ull && pool_next < pool_size) {
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
      const int k = pool_next + rank;
      pool_next += __popcll(need);
      const bool take = !alive & (k < pool_size);
      const int kk = take ? k : 0;
      const int c_pix = kk & 63;
      const int px = tx * 8 + (c_pix & 7);
      const int row = ty * 8 + (c_pix >> 3);  // output row, top first
      const bool start = take & (px < A.nx) & (row < A.ny);
      pix = start ? c_pix : pix;
      pixel_id = start ? (unsigned)((A.ny - 1 - row) * A.nx + px) : pixel_id;
      sample = start ? (unsigned)(A.sample_begin + s_first + (kk >> 6)) : sample;
      depth = start ? 0 : depth;
      thr = sel3(start, f3(1.0f, 1.0f, 1.0f), thr);
      fresh |= start;
      alive |= start;
      if (STATS) cnt.paths += start;
    }
    if (__ballot(alive) == 0ull) {
      if (pool_next >= pool_size) break;
      continue;
    }

    // The empty asm makes the key opaque to the optimiser once per iteration.  Without
    // it the ten round keys (key + round * constant) are computed before the loop and
    // held in twenty scalar registers for its whole life, which this kernel does not
    // have to spare; with it they are nine scalar additions per call.
    unsigned key0 = A.key0, key1 = A.key1;
    asm volatile("" : "+s"(key0), "+s"(key1));
    // ---- the one Philox call of this vertex: counter (pixel, sample, depth, 0) ----
    // (the build for Lambertian scenes only: in the others the twenty registers the keys need for a moment cost spills)
    U4 w;
    if (GENERAL == 0) {
      int rk0 = 0;
      asm volatile("" : "+v"(rk0));  // (an index the optimiser cannot see through keeps the reads of the keys inside the loop)
      w = philox4x32_10_keys(pixel_id, sample, (unsigned)depth, 0u, rk_lds + rk0);
    } else {
      w = philox4x32_10(pixel_id, sample, (unsigned)depth, 0u, key0, key1);
    }

    // ---- camera ray of the fresh lanes (main.rs:102-105, camera.rs:53-62) ----
    if (__ballot(fresh) != 0ull) {
      const int px = tx * 8 + (pix & 7);
      const int j = A.ny - 1 - (ty * 8 + (pix >> 3));
      const float u = ((float)px + (float)(w.x >> 16) * (1.0f / 65536.0f)) * A.inv_nx;
      const float v = ((float)j + (float)(w.x & 0xffffu) * (1.0f / 65536.0f)) * A.inv_ny;
      // an index the optimiser cannot see through keeps these reads inside the loop
      int c0 = 0;
      asm volatile("" : "+v"(c0));
      const float* cl = cam_lds + c0;
      float rdx = 0.0f, rdy = 0.0f;
      F3 offset = f3(0.0f, 0.0f, 0.0f);
      if (has_lens) {
        // random_in_unit_disk, camera.rs:65-75: attempt a draws counter (pixel, sample, 0, 1+a)
        const float lens_radius = cl[18];
        unsigned attempt = 0;
        bool looking = fresh;
        while (__ballot(looking) != 0ull) {
          const U4 lw = philox4x32_10(pixel_id, sample, 0u, 1u + attempt, key0, key1);
          const float dx = 2.0f * u01(lw.x) - 1.0f, dy = 2.0f * u01(lw.y) - 1.0f;
          const bool ok = looking & (fmaf(dy, dy, dx * dx) < 1.0f);
          rdx = ok ? lens_radius * dx : rdx;
          rdy = ok ? lens_radius * dy : rdy;
          looking &= !ok;
          ++attempt;
        }
        offset = f3(fmaf(cl[15], rdy, cl[12] * rdx), fmaf(cl[16], rdy, cl[13] * rdx), fmaf(cl[17], rdy, cl[14] * rdx));
      }
      const F3 org = f3(cl[0] + offset.x, cl[1] + offset.y, cl[2] + offset.z);
      const F3 dir = f3(fmaf(v, cl[9], fmaf(u, cl[6], cl[3])) - cl[0] - offset.x,
                        fmaf(v, cl[10], fmaf(u, cl[7], cl[4])) - cl[1] - offset.y,
                        fmaf(v, cl[11], fmaf(u, cl[8], cl[5])) - cl[2] - offset.z);
      ro = sel3(fresh, org, ro);
      rd = sel3(fresh, dir, rd);
      fresh = false;
    }
    const RayF ray = make_ray(ro, rd);
#ifdef RT_PROBE
    // Diagnostic build (-DRT_PROBE=k, tools/probe_valu.sh): 50 more instructions of one kind in every trip of the path
    // loop, results unused (depth comes back unchanged).  How much the launch slows down per added instruction says
    // what bounds the kernel: a unit with slack absorbs them.  No probe exists in the normal build.
#if RT_PROBE == 1
    asm volatile(".rept 50\n s_nop 0\n .endr\n");
#elif RT_PROBE == 2
    { int tmp_ = depth; asm volatile(".rept 50\n v_xor_b32 %0, 1, %0\n .endr\n" : "+v"(tmp_)); depth = tmp_; }
#elif RT_PROBE == 3
    { int tmp_ = depth; asm volatile(".rept 50\n v_cmp_lt_i32 vcc, 1, %0\n .endr\n" : "+v"(tmp_) : : "vcc"); depth = tmp_; }
#elif RT_PROBE == 4
    { int tmp_ = depth; asm volatile("v_cmp_lt_i32 vcc, 1000, %0\n .rept 50\n v_cndmask_b32 %0, %0, %0, vcc\n .endr\n" : "+v"(tmp_) : : "vcc"); depth = tmp_; }
#elif RT_PROBE == 5
    { float tmp_ = __int_as_float(depth); asm volatile(".rept 50\n v_med3_f32 %0, %0, %0, %0\n .endr\n" : "+v"(tmp_)); depth = __float_as_int(tmp_); }
#elif RT_PROBE == 6
    { int tmp_ = depth; asm volatile(".rept 50\n s_and_b64 vcc, vcc, exec\n .endr\n" : "+v"(tmp_) : : "vcc"); depth = tmp_; }
#endif
#endif

    // ---- world.hit(r, 0.001, INFINITY), main.rs:166 ----
    if (STATS) cnt.rays += alive;
    const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
    const Hit h = trace_scan<STATS, TRAV, GENERAL != 0>(S, stk, box_tab, alive, ray, a_t_min, INFINITY, rng, cnt);

    // ---- color(), main.rs:165-209, one vertex ----
    const Shaded sh = shade_vertex<STATS, GENERAL>(S, load_prim_rec(dprim_tab, alive, h.prim), alive, h, ray, thr, depth, w, a_max_depth, a_sky, pixel_id, sample, key0, key1, cnt);
    const F3 radiance = sh.radiance, dir = sh.dir, thr_next = sh.thr_next;
    const bool scattered = sh.scattered, textured = sh.textured, emits = sh.emits;
    // A zero or NaN weight fixes the sample's value (0 after de_nan) whatever the
    // rest of the path does, so the path ends there.
    const bool go_on = scattered & ((thr_next.x > 0.0f) | (thr_next.y > 0.0f) | (thr_next.z > 0.0f));
    const bool done = alive & !go_on;
    if (done) {
      // de_nan (main.rs:252-264) then col += (main.rs:108)
      const float r = radiance.x != radiance.x ? 0.0f : radiance.x;
      const float g = radiance.y != radiance.y ? 0.0f : radiance.y;
      const float b = radiance.z != radiance.z ? 0.0f : radiance.z;
      if (r != 0.0f || g != 0.0f || b != 0.0f) {
        atomicAdd(&acc[wib][0][pix], r);
        atomicAdd(&acc[wib][1][pix], g);
        atomicAdd(&acc[wib][2][pix], b);
        if (SUMSQ) {
          atomicAdd(&acc[wib][3][pix], r * r);
          atomicAdd(&acc[wib][4][pix], g * g);
          atomicAdd(&acc[wib][5][pix], b * b);
        }
      }
    }
    // A lane whose path ended gets a new origin and weight with its next path, so only
    // the direction is kept clean (it feeds the reciprocals' range test meanwhile).
    ro = sh.p;
    rd = sel3(go_on, dir, rd);
    if (GENERAL >= 2) {
      const bool lit = textured & emits;  // its sample is thr * colour: the weight is kept for the top of the next trip
      pend = (textured & (go_on | emits)) ? (sh.mat_index << 1) | (lit ? 1 : 0) : -1;
      thr = sel3(lit, thr, thr_next);
    } else {
      thr = thr_next;
    }
    depth += go_on ? 1 : 0;
    alive = go_on;
  }

  // ---- write the tile (main.rs:112-114; sums, not means) ----
  {
    const int px = tx * 8 + (lane & 7);
    const int row = ty * 8 + (lane >> 3);
    if (px < A.nx && row < A.ny) {
      const size_t at = ((size_t)row * A.nx + px) * 3;
      o_sum[at] = acc[wib][0][lane];
      o_sum[at + 1] = acc[wib][1][lane];
      o_sum[at + 2] = acc[wib][2][lane];
      if (SUMSQ) {
        o_sumsq[at] = acc[wib][3][lane];
        o_sumsq[at + 1] = acc[wib][4][lane];
        o_sumsq[at + 2] = acc[wib][5][lane];
      }
    }
  }
  }  // next item
  if (STATS) {
    const unsigned v[9] = {cnt.paths, cnt.rays, cnt.hits, cnt.node_tests, cnt.rect_tests, cnt.sphere_tests, cnt.instance_ops, cnt.shade_events, cnt.light_tests};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0) atomicAdd(&stats[k], x);
    }
  }
}

// ---------------------------------------------------------------------------
// render_stack_kernel: the sample loop for scenes whose tree needs the per-lane
// stack traversal.  Same work items, same pool of (pixel, sample) paths per wave,
// same accumulators as render_kernel; what differs is that a lane's traversal
// state (node, stack, best hit, ray in the current frame) lives across the shading
// code.  The wave alternates between two phases:
//   shading   - the lanes that are NOT in a traversal: finish the hit (media), shade
//               the vertex, end or continue the path, take a new path from the pool,
//               make the camera ray, start the next traversal;
//   traversal - every lane with a ray steps through its tree until enough lanes
//               have finished (RT_READY_LANES of them, or half of the live ones) to
//               make a shading phase worth running.
// So a lane whose ray ended early gets its next ray while its neighbours are still
// inside their trees, instead of idling until the slowest ray of the wave is done.
// ---------------------------------------------------------------------------
#ifndef RT_READY_LANES
#define RT_READY_LANES 48
#endif
#ifndef RT_SLOTS
// (scenes with media or the older materials have paths of depth 50 beside paths of depth 3: a third open item keeps
// the lanes fed while two items wait for their last paths: final scene -5 %; the Lambertian build gains nothing from it)
#define RT_SLOTS (GENERAL >= 1 ? 3 : 2)
#endif
#ifndef RT_STACK_LAST_SPP
#define RT_STACK_LAST_SPP 4  // render_stack_kernel: the last sample chunk of a tile is halved down to parts of this many samples
#endif
// Waves per SIMD the register allocator leaves room for.  The LDS a wave holds lets a CU run 20 waves, five per SIMD,
// whatever the registers allow: the Lambertian build fits 80 registers without spills; the build with all materials
// needs 96 for that (with 80 it spilled 56 bytes per lane: Book-1 scene 18.8 -> 17.7 ms with 96).
#ifndef RT_STACK_WAVES
#define RT_STACK_WAVES 6
#endif
#ifndef RT_STACK_WAVES_G2
#define RT_STACK_WAVES_G2 5
#endif
// Diagnostic build (-DRT_PHASE_TIMING): shader-clock stamps around the parts of the loop, summed per wave into
// stats[16..31] by the counting build and printed by rt_render.  No stamp exists in the normal build.
#ifdef RT_PHASE_TIMING
#define RT_STAMP(k) do { if (STATS) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[k] += now_ - t_last; t_last = now_; } } while (0)
#define RT_COUNT(k, v) do { if (STATS) ph[k] += (v); } while (0)
#else
#define RT_STAMP(k) do { } while (0)
#define RT_COUNT(k, v) do { } while (0)
#endif
template <bool STATS, bool SUMSQ, int GENERAL>
__global__ void __launch_bounds__(64, GENERAL < 1 ? RT_STACK_WAVES : (GENERAL < 2 ? 5 : RT_STACK_WAVES_G2)) render_stack_kernel(DevScene S, DevCamera cam, RenderArgs A, float* __restrict__ out_sum,
                                                     float* __restrict__ out_sumsq, unsigned long long* __restrict__ stats, unsigned* __restrict__ queue) {
  // Up to RT_SLOTS work items are open per wave: the one lanes take new paths from (slot f) and earlier ones whose
  // pools are empty and whose last paths are still on their way.  Paths reach depth 50 in scenes with media, and a
  // wave that waited for its item's last path before opening the next one would run at a few lanes for most of its
  // life.  A lane carries its item's slot with its pixel (slot * 64 + pixel in the tile) and adds its samples to
  // that slot's accumulators; a slot's tile is written when its last path has ended.
  __shared__ float acc[SUMSQ ? 6 : 3][RT_SLOTS * 64];
  __shared__ int slot_info[RT_SLOTS][4];  // tile x, tile y, chunk
  extern __shared__ float4 prim_lds[];  // the LDS part of the traversal stack: A.stack_lds entries of 64 lanes
  LaneStackT<true> stk;
  stk.lds = (lds_u32*)((unsigned*)prim_lds + (threadIdx.x & 63));
  stk.n_lds = A.stack_lds;
  {
    unsigned ps = S.no_prune ? 0x7f800000u : 0x3f800000u;  // +inf or 1.0f
    asm volatile("" : "+s"(ps));  // (a register of its own: see the copies of the kernel's arguments below)
    stk.prune_scale = __uint_as_float(ps);
  }
  stk.ovf = (glb_u32*)(A.stack_ovf + (size_t)blockIdx.x * (size_t)(S.stack_depth - A.stack_lds) * 128 + (threadIdx.x & 63));
  const int lane = threadIdx.x & 63;
  __shared__ float cam_lds[20];
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cam_lds[k] = cam.origin[k];
      cam_lds[3 + k] = cam.llc[k];
      cam_lds[6 + k] = cam.horizontal[k];
      cam_lds[9 + k] = cam.vertical[k];
      cam_lds[12 + k] = cam.u[k];
      cam_lds[15 + k] = cam.v[k];
    }
    cam_lds[18] = cam.lens_radius;
  }
  const bool has_lens = cam.lens_radius != 0.0f;
  const float4* dprim_tab = (const float4*)S.dprims;
  Counters cnt;
  memset(&cnt, 0, sizeof(cnt));
  const int n_tiles = A.n_tiles;
  const int n_items = n_tiles * A.n_chunks;
  float a_t_min = A.t_min;
  int a_max_depth = A.max_depth, a_sky = A.sky;
  int a_ready = A.ready_lanes, a_hold = A.hold_lanes;
  int slots_cap = A.max_slots < RT_SLOTS ? A.max_slots : RT_SLOTS;
  asm volatile("" : "+s"(a_t_min), "+s"(a_max_depth), "+s"(a_sky), "+s"(a_ready), "+s"(a_hold), "+s"(slots_cap));
#ifdef RT_PHASE_TIMING
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  // Wave-uniform state of the open items.
  unsigned open_mask = 0;      // slots that hold an item
  int f = 0;                   // slot that new paths come from
  bool fill_open = false;      // slot f's pool is not empty yet
  bool queue_empty = false;
  int tx = 0, ty = 0;          // tile of the item in slot f
  int s_first = 0, pool_size = 0, pool_next = 0;  // its first sample and its pool

  // Per-lane path state.
  bool alive = false;  // on a path
  bool trav = false;   // in a traversal (implies alive)
  F3 thr = f3(0, 0, 0);
  int depth = 0, pix = 0;  // pix: slot * 64 + pixel within the tile
  unsigned pixel_id = 0, sample = 0;
  int pend = -1;
  // Per-lane traversal state.  A lane keeps ONE ray, T.r: the world ray of the segment it shades or traverses, or,
  // while it is inside an object's sub-tree, that ray in the object's frame (the world ray then waits on the lane's
  // stack, see trav_leaf).  A walk that has ended is back in the world frame.
  Trav T;
  T.r = make_ray(f3(0, 0, 0), f3(0, 0, 1));
  F3& ro = T.r.o;  // (the names the shading code uses)
  F3& rd = T.r.d;
#define RT_WORLD_RAY() (T.r)
  T.sp = 0;
  T.cur = TRAV_POP;
  T.best.t = INFINITY;
  T.best.prim = -1;

  for (;;) {
    // ======================= shading phase: the lanes that are not in a traversal =======================
    const bool ready = !trav;
    if (GENERAL >= 2 && S.textured) {
      // Texture::value of the vertex before (see render_kernel): evaluated where little else is live.
      if (__ballot(ready & (pend >= 0)) != 0ull) {
        const bool tex_now = ready & (pend >= 0);
        const F3 c = texture_value_wave(S.textures, S.perlins, tex_now, tex_now ? S.materials[pend >> 1].texture : 0, ro);
        if (tex_now) {
          const F3 v = f3(thr.x * c.x, thr.y * c.y, thr.z * c.z);
          if (pend & 1) {
            const float cr = v.x != v.x ? 0.0f : v.x, cg = v.y != v.y ? 0.0f : v.y, cb = v.z != v.z ? 0.0f : v.z;  // de_nan
            if (cr != 0.0f || cg != 0.0f || cb != 0.0f) {
              atomicAdd(&acc[0][pix], cr);
              atomicAdd(&acc[1][pix], cg);
              atomicAdd(&acc[2][pix], cb);
              if (SUMSQ) {
                atomicAdd(&acc[3][pix], cr * cr);
                atomicAdd(&acc[4][pix], cg * cg);
                atomicAdd(&acc[5][pix], cb * cb);
              }
            }
          } else {
            thr = v;
          }
          pend = -1;
        }
      }
    }
    RT_STAMP(0);  // texture
    unsigned key0 = A.key0, key1 = A.key1;
    asm volatile("" : "+s"(key0), "+s"(key1));
    const bool act = ready & alive;  // lanes whose traversal has ended: shade that vertex
    RT_COUNT(8, 1); RT_COUNT(9, __popcll(__ballot(act)));
    if (__ballot(act) != 0ull) {
      // The record of the primitive the walk found is asked for now and used after the media and the vertex's Philox call,
      // which hide its latency; the few lanes whose hit a medium replaces ask again.
      // (The builds with textures are at their register limit: they ask where they always did, just before shading;
      // asking early cost the final scene 1.2 % and gave the Book-1 scene 1.4 %.)
      PrimRec hit_rec;
      if (GENERAL < 2) hit_rec = load_prim_rec(dprim_tab, act, T.best.prim);
      const int walk_prim = T.best.prim;
      // what is left of world.hit: the ConstantMedium nodes (they draw random numbers)
      if (GENERAL != 0 && S.n_media_sph > 0) {
        const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
        media_sphere_pass<STATS>(S, act, RT_WORLD_RAY(), a_t_min, rng, T.best, cnt);
      } else if (GENERAL != 0 && S.n_scan > 0) {
        const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
        scan_range<STATS, true, true, 0>(S, 0, S.n_scan, act, RT_WORLD_RAY(), a_t_min, INFINITY, rng, T.best, cnt);
      }
      if (GENERAL == 3 && S.n_media_bvh > 0) {
        const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
        media_bvh_pass<STATS>(S, stk, act, RT_WORLD_RAY(), a_t_min, INFINITY, rng, T.best, cnt);
      }
      RT_STAMP(1);  // media
      const U4 w = philox4x32_10(pixel_id, sample, (unsigned)depth, 0u, key0, key1);
      if (GENERAL >= 2) hit_rec = load_prim_rec(dprim_tab, act, T.best.prim);
      if (GENERAL == 1) {
        const bool moved = act & (T.best.prim != walk_prim);
        if (__ballot(moved) != 0ull) {
          const PrimRec again = load_prim_rec(dprim_tab, act, T.best.prim);
          if (moved) hit_rec = again;
        }
      }
      const Shaded sh = shade_vertex<STATS, GENERAL>(S, hit_rec, act, T.best, RT_WORLD_RAY(), thr, depth, w, a_max_depth, a_sky, pixel_id, sample, key0, key1, cnt);
      RT_STAMP(2);  // philox + shade
      const bool go_on = act & sh.scattered & ((sh.thr_next.x > 0.0f) | (sh.thr_next.y > 0.0f) | (sh.thr_next.z > 0.0f));
      const bool done = act & !go_on;
      if (done) {
        // de_nan (main.rs:252-264) then col += (main.rs:108)
        const float cr = sh.radiance.x != sh.radiance.x ? 0.0f : sh.radiance.x;
        const float cg = sh.radiance.y != sh.radiance.y ? 0.0f : sh.radiance.y;
        const float cb = sh.radiance.z != sh.radiance.z ? 0.0f : sh.radiance.z;
        if (cr != 0.0f || cg != 0.0f || cb != 0.0f) {
          atomicAdd(&acc[0][pix], cr);
          atomicAdd(&acc[1][pix], cg);
          atomicAdd(&acc[2][pix], cb);
          if (SUMSQ) {
            atomicAdd(&acc[3][pix], cr * cr);
            atomicAdd(&acc[4][pix], cg * cg);
            atomicAdd(&acc[5][pix], cb * cb);
          }
        }
      }
      ro = sel3(act, sh.p, ro);
      rd = sel3(go_on, sh.dir, rd);
      if (GENERAL >= 2) {
        const bool lit = sh.textured & sh.emits;
        pend = act ? ((sh.textured & (go_on | sh.emits)) ? (sh.mat_index << 1) | (lit ? 1 : 0) : -1) : pend;
        thr = sel3(act, sel3(lit, thr, sh.thr_next), thr);
      } else {
        thr = sel3(act, sh.thr_next, thr);
      }
      depth += go_on ? 1 : 0;
      alive = act ? go_on : alive;
    }
    RT_STAMP(3);  // deposit + state update
    // ---- work items, and (pixel, sample) paths for the free lanes (main.rs:96-101).  Twice when the item the paths
    // come from runs dry with lanes still free: the next item is opened at once, so no lane waits a traversal phase. ----
    bool fresh = false, finished = false;
    int fresh_px = 0, fresh_j = 0;
    for (int pass = 0; pass < 2; ++pass) {
      // ======================= work items (wave-uniform) =======================
      {
        const bool busy = alive | (GENERAL >= 2 && pend >= 0);
        if (fill_open && pool_next >= pool_size) fill_open = false;  // slot f goes on as a draining slot
        unsigned draining = open_mask & ~(fill_open ? (1u << f) : 0u);
        while (draining != 0u) {
          const int d = __builtin_ctz(draining);
          draining &= draining - 1u;
          if (__ballot(busy & ((pix >> 6) == d)) != 0ull) continue;
          // the last path of this item has ended: write its tile (main.rs:112-114; sums, not means)
          const int dtx = slot_info[d][0], dty = slot_info[d][1], chunk = slot_info[d][2];
          const size_t slab = A.accumulate ? 0 : (size_t)chunk * A.nx * A.ny * 3;
          const int px = dtx * 8 + (lane & 7);
          const int row = dty * 8 + (lane >> 3);
          if (px < A.nx && row < A.ny) {
            const size_t at = slab + ((size_t)row * A.nx + px) * 3;
            if (A.accumulate) {
              // the items of a tile (its sample chunks) meet in the output: hardware float adds, no partial images
  #pragma unroll
              for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&out_sum[at + c], acc[c][d * 64 + lane]);
              if (SUMSQ) {
  #pragma unroll
                for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&out_sumsq[at + c], acc[3 + c][d * 64 + lane]);
              }
            } else {
              out_sum[at] = acc[0][d * 64 + lane];
              out_sum[at + 1] = acc[1][d * 64 + lane];
              out_sum[at + 2] = acc[2][d * 64 + lane];
              if (SUMSQ) {
                out_sumsq[at] = acc[3][d * 64 + lane];
                out_sumsq[at + 1] = acc[4][d * 64 + lane];
                out_sumsq[at + 2] = acc[5][d * 64 + lane];
              }
            }
          }
          open_mask &= ~(1u << d);
        }
        if (!fill_open && !queue_empty && __popc(open_mask) < slots_cap) {
          int item = 0;
          if (lane == 0) item = (int)atomicAdd(queue, 1u);
          item = __builtin_amdgcn_readfirstlane(item);
          if (item >= n_items) {
            queue_empty = true;
          } else {
            f = __builtin_ctz(~open_mask);
            const int chunk = item / n_tiles;
            const int tile = item - chunk * n_tiles;
            tx = tile % A.tiles_x;
            ty = tile / A.tiles_x;
            bool mine = true;
            if (A.shard_count > 1) {
              const int t = A.shard_index + tile * A.shard_count;
              tx = t % A.shard_width;
              ty = t / A.shard_width;
              mine = (tx < A.tiles_x) & (ty < A.tiles_y);
            }
  #pragma unroll
            for (int c = 0; c < (SUMSQ ? 6 : 3); ++c) acc[c][f * 64 + lane] = 0.0f;
            if (lane == 0) {
              slot_info[f][0] = tx;
              slot_info[f][1] = ty;
              slot_info[f][2] = chunk;
            }
            // the last chunk comes in halving parts (see render_kernel)
            const int n_whole = A.n_chunks - A.split_parts;
            int s_len = A.chunk_spp;
            s_first = chunk * A.chunk_spp;
            if (chunk >= n_whole) {
              const int k = chunk - n_whole;
              const int last = A.spp - n_whole * A.chunk_spp;
              const int o0 = last - (last >> k);
              const int o1 = k + 1 < A.split_parts ? last - (last >> (k + 1)) : last;
              s_first = n_whole * A.chunk_spp + o0;
              s_len = o1 - o0;
            }
            pool_size = mine ? 64 * s_len : 0;
            pool_next = 0;
            open_mask |= 1u << f;
            fill_open = true;
          }
        }
        if (open_mask == 0u && queue_empty) {
          finished = true;
          break;
        }
      }
      // A path that ended on a textured light keeps its lane until the colour has been added (top of the next shading phase).
      const bool free_lane = ready & !alive & !(GENERAL >= 2 && pend >= 0);
      const unsigned long long need = __ballot(free_lane);
      if (need == 0ull) break;
      if (fill_open && pool_next < pool_size) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
        const int k = pool_next + rank;
        pool_next += __popcll(need);
        const bool take = free_lane & (k < pool_size);
        const int kk = take ? k : 0;
        const int c_pix = kk & 63;
        const int px = tx * 8 + (c_pix & 7);
        const int row = ty * 8 + (c_pix >> 3);
        const bool start = take & (px < A.nx) & (row < A.ny);
        pix = start ? (f * 64 + c_pix) : pix;
        pixel_id = start ? (unsigned)((A.ny - 1 - row) * A.nx + px) : pixel_id;
        sample = start ? (unsigned)(A.sample_begin + s_first + (kk >> 6)) : sample;
        depth = start ? 0 : depth;
        thr = sel3(start, f3(1.0f, 1.0f, 1.0f), thr);
        fresh |= start;
        fresh_px = start ? px : fresh_px;
        fresh_j = start ? A.ny - 1 - row : fresh_j;
        alive |= start;
        if (STATS) cnt.paths += start;
      }
      if (pool_next < pool_size || queue_empty) break;  // lanes are only left free by an item that ran dry
    }
    if (finished) break;
    RT_STAMP(4);  // work items, refill
    if (__ballot(alive) == 0ull) continue;  // nothing to trace: back to the items (a pool at the image's edge, a pending colour)
    // ---- camera ray of the fresh lanes (main.rs:102-105, camera.rs:53-62); word 0 of the vertex's Philox call ----
    if (__ballot(fresh) != 0ull) {
      const U4 w = philox4x32_10(pixel_id, sample, 0u, 0u, key0, key1);
      const int px = fresh_px, j = fresh_j;  // (not from tx, ty: a lane of this phase may belong to the item before)
      const float u = ((float)px + (float)(w.x >> 16) * (1.0f / 65536.0f)) * A.inv_nx;
      const float v = ((float)j + (float)(w.x & 0xffffu) * (1.0f / 65536.0f)) * A.inv_ny;
      int c0 = 0;
      asm volatile("" : "+v"(c0));
      const float* cl = cam_lds + c0;
      float rdx = 0.0f, rdy = 0.0f;
      F3 offset = f3(0.0f, 0.0f, 0.0f);
      if (has_lens) {
        const float lens_radius = cl[18];
        unsigned attempt = 0;
        bool looking = fresh;
        while (__ballot(looking) != 0ull) {
          const U4 lw = philox4x32_10(pixel_id, sample, 0u, 1u + attempt, key0, key1);
          const float dx = 2.0f * u01(lw.x) - 1.0f, dy = 2.0f * u01(lw.y) - 1.0f;
          const bool ok = looking & (fmaf(dy, dy, dx * dx) < 1.0f);
          rdx = ok ? lens_radius * dx : rdx;
          rdy = ok ? lens_radius * dy : rdy;
          looking &= !ok;
          ++attempt;
        }
        offset = f3(fmaf(cl[15], rdy, cl[12] * rdx), fmaf(cl[16], rdy, cl[13] * rdx), fmaf(cl[17], rdy, cl[14] * rdx));
      }
      const F3 org = f3(cl[0] + offset.x, cl[1] + offset.y, cl[2] + offset.z);
      const F3 dir = f3(fmaf(v, cl[9], fmaf(u, cl[6], cl[3])) - cl[0] - offset.x,
                        fmaf(v, cl[10], fmaf(u, cl[7], cl[4])) - cl[1] - offset.y,
                        fmaf(v, cl[11], fmaf(u, cl[8], cl[5])) - cl[2] - offset.z);
      ro = sel3(fresh, org, ro);
      rd = sel3(fresh, dir, rd);
    }
    RT_STAMP(4);  // camera
    // ---- world.hit(r, 0.001, INFINITY), main.rs:166: the lanes with a new segment start at the root ----
    {
      const bool go = ready & alive;
      const RayF nr = make_ray(ro, rd);
      if (STATS) cnt.rays += go;
      Counters c2;  // (the root test of lanes that do not start a ray is not an event)
      memset(&c2, 0, sizeof(c2));
      const int first = trav_enter<STATS>(S, S.root_ref, S.root_box, nr, a_t_min, INFINITY, c2);
      if (STATS) cnt.node_tests += go;
      T.r.inv = sel3(go, nr.inv, T.r.inv);
      T.cur = go ? first : T.cur;
      T.sp = go ? 0 : T.sp;
      T.best.t = go ? INFINITY : T.best.t;
      T.best.prim = go ? -1 : T.best.prim;
      trav |= go;
    }
    RT_STAMP(5);  // traversal start
    // ======================= traversal phase =======================
    for (;;) {
      RT_COUNT(10, 1); RT_COUNT(11, __popcll(__ballot(trav)));
      // one step for every lane that has one to make; a lane holding a list or instance leaf waits for the leaf code below
      if (trav & !trav_holds_leaf(T.cur)) trav = trav_step<STATS, GENERAL == 3>(S, stk, T, a_t_min, INFINITY, cnt);
      RT_STAMP(6);  // step
      const bool hold = trav & trav_holds_leaf(T.cur);
      const unsigned long long m_trav = __ballot(trav), m_hold = __ballot(hold);
      // the leaf code runs once for all lanes that hold a leaf, when enough of the traversing lanes do
      if (m_hold != 0ull && __popcll(m_hold) * 64 >= a_hold * __popcll(m_trav)) {
        RT_COUNT(12, 1); RT_COUNT(13, __popcll(m_hold)); RT_COUNT(14, __popcll(__ballot(hold & ((T.cur & BN2_PRIM) != 0))));
        if (hold) trav_leaf<STATS>(S, stk, T, a_t_min, INFINITY, cnt);
      }
      RT_STAMP(7);  // leaf code
      // back to the shading phase when enough of the live lanes have a finished ray (RT_READY_LANES in 64)
      const int n_alive = __popcll(__ballot(alive));
      const int n_ready = n_alive - __popcll(__ballot(trav));
      // (the second test ends the phase when no lane is left in a traversal, whatever the threshold is)
      if (n_ready * 64 >= a_ready * n_alive || n_ready == n_alive) break;
    }
  }

  if (STATS) {
    const unsigned v[9] = {cnt.paths, cnt.rays, cnt.hits, cnt.node_tests, cnt.rect_tests, cnt.sphere_tests, cnt.instance_ops, cnt.shade_events, cnt.light_tests};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0) atomicAdd(&stats[k], x);
    }
#ifdef RT_PHASE_TIMING
    if (lane == 0)
      for (int k = 0; k < 16; ++k) atomicAdd(&stats[16 + k], ph[k]);
#endif
  }
}

// ---------------------------------------------------------------------------
// render_pool_kernel: the sample loop for scenes on the per-lane stack traversal,
// with the wave's paths kept in a pool of records and handed between the phases
// of the loop through queues, so that every phase runs on a full wave.
//
// render_stack_kernel keeps a path in one lane from its camera ray to its end, so
// the lanes of a wave are in different phases at any moment (walking the tree,
// holding a leaf, waiting to be shaded) and each phase's code runs for a fraction
// of the wave.  Here a wave owns P path slots (more than its 64 lanes), each a
// 64-byte record (ray, hit, weight, pixel, sample, depth) in a region of global
// memory that only this wave touches, and four queues of slot numbers in LDS:
//   ray queue      paths with a ray to trace;
//   shade queue    paths whose walk has ended (hit found or missed);
//   texture queue  paths waiting for a Texture::value (checker / noise);
//   free queue     slots without a path.
// A lane is only a place where a walk runs: it takes a ray from the ray queue,
// walks the tree (trav_step / trav_leaf; the walk's stack and best hit live in the
// lane as before), writes the hit into the record, puts the slot on the shade
// queue and takes the next ray.  When the shade queue holds a wave's worth of paths
// the wave runs the shading code (media, Philox, color(), a new camera path for
// each one that ended, the root's box test of the next ray) once for 64 of them,
// and they go to the ray queue; the walks the lanes were in wait in their
// registers meanwhile.  Texture::value runs the same way for a batch of textured
// vertices.  The queues and records are the wave's own: no atomics, no barriers,
// and what a wave computes does not depend on what other waves do.
// Work items, tile accumulators and the way a tile's sums reach the output are
// render_stack_kernel's.
// ---------------------------------------------------------------------------
struct __align__(16) PoolRec {
  float ox, oy, oz, t;   // ray origin; after the walk: distance of the hit
  float dx, dy, dz;      // ray direction
  int ref;               // waiting for a lane: the reference the walk starts at (the root's box test is made by the
                         // shading code); after the walk: the primitive hit, -1 for none
  float tr, tg, tb;      // the path's weight
  unsigned depth_pix;    // depth << 8 | item slot * 64 + pixel within the tile
  unsigned pixel_id, sample;
  int pend;              // a colour still to come from Texture::value: material index << 1 | ended-on-a-light, or -1
  int pad;
};
static_assert(sizeof(PoolRec) == 64, "four 16-byte loads");
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f32x4 glb_f4;  // (global_ loads and stores, not flat_ ones: see LaneStackT)
typedef __attribute__((address_space(1))) float glb_f32;
typedef __attribute__((address_space(1))) char glb_char;
#define RT_POOL_LDS_DWORDS(P) (4 * (P))  // the four queues
template <bool STATS, bool SUMSQ, int GENERAL>
__global__ void __launch_bounds__(64, GENERAL < 2 ? 5 : 4) render_pool_kernel(DevScene S, DevCamera cam, RenderArgs A, float* __restrict__ out_sum,
                                                            float* __restrict__ out_sumsq, unsigned long long* __restrict__ stats, unsigned* __restrict__ queue) {
  __shared__ float acc[SUMSQ ? 6 : 3][RT_SLOTS * 64];
  __shared__ int slot_info[RT_SLOTS][4];  // tile x, tile y, chunk
  __shared__ float cam_lds[20];
  extern __shared__ float4 prim_lds[];  // A.stack_lds stack entries of 64 lanes, then the four queues
  const int lane = threadIdx.x & 63;
  LaneStackT<true> stk;
  stk.lds = (lds_u32*)((unsigned*)prim_lds + lane);
  stk.n_lds = A.stack_lds;
  {
    unsigned ps = S.no_prune ? 0x7f800000u : 0x3f800000u;  // +inf or 1.0f
    asm volatile("" : "+s"(ps));  // (a register of its own: see the copies of the kernel's arguments below)
    stk.prune_scale = __uint_as_float(ps);
  }
  stk.ovf = (glb_u32*)(A.stack_ovf + (size_t)blockIdx.x * (size_t)(S.stack_depth - A.stack_lds) * 128 + lane);
  const int P = A.pool_slots;
  glb_char* const recs = (glb_char*)A.pool_mem + (size_t)blockIdx.x * (size_t)P * sizeof(PoolRec);  // this wave's records
  lds_u32* const ray_q = (lds_u32*)((unsigned*)prim_lds + A.stack_lds * 128);
  lds_u32* const shade_q = ray_q + P;
  lds_u32* const tex_q = shade_q + P;
  lds_u32* const free_q = tex_q + P;
  for (int k = lane; k < P; k += 64) free_q[k] = (unsigned)k;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cam_lds[k] = cam.origin[k];
      cam_lds[3 + k] = cam.llc[k];
      cam_lds[6 + k] = cam.horizontal[k];
      cam_lds[9 + k] = cam.vertical[k];
      cam_lds[12 + k] = cam.u[k];
      cam_lds[15 + k] = cam.v[k];
    }
    cam_lds[18] = cam.lens_radius;
  }
  const bool has_lens = cam.lens_radius != 0.0f;
  const float4* dprim_tab = (const float4*)S.dprims;
  Counters cnt;
  memset(&cnt, 0, sizeof(cnt));
  const int n_tiles = A.n_tiles;
  const int n_items = n_tiles * A.n_chunks;
  float a_t_min = A.t_min;
  int a_max_depth = A.max_depth, a_sky = A.sky;
  int a_hold = A.hold_lanes, a_refill = A.refill_lanes, a_shade = A.shade_batch, a_tex = A.tex_batch, a_hungry = A.hungry_lanes;
  int slots_cap = A.max_slots < RT_SLOTS ? A.max_slots : RT_SLOTS;
  asm volatile("" : "+s"(a_t_min), "+s"(a_max_depth), "+s"(a_sky), "+s"(a_hold), "+s"(a_refill), "+s"(a_shade), "+s"(a_tex), "+s"(a_hungry), "+s"(slots_cap));
  int idle_trips = 0;
#ifdef RT_PHASE_TIMING
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  // The queues: rings of P entries each; heads and counts are the same for all lanes (scalar registers).
  int ray_head = 0, ray_n = 0, shade_head = 0, shade_n = 0, tex_head = 0, tex_n = 0, free_head = 0, free_n = P;
  // appends the slots of the lanes with `mine` to a ring (a slot is in one queue at a time, so a ring never overflows)
#define RT_RING_PUSH(q, head, n, mine, id)                                                                          \
  do {                                                                                                              \
    const unsigned long long m_ = __ballot(mine);                                                                   \
    if (m_ != 0ull) {                                                                                               \
      const int rank_ = __builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0)); \
      if (mine) {                                                                                                   \
        int at_ = (head) + (n) + rank_;                                                                             \
        at_ -= at_ >= P ? P : 0;                                                                                    \
        q[at_] = (unsigned)(id);                                                                                    \
      }                                                                                                             \
      (n) += __popcll(m_);                                                                                          \
    }                                                                                                               \
  } while (0)
  // Wave-uniform state of the open work items (see render_stack_kernel); live[d]: paths of slot d's item still on their way.
  unsigned open_mask = 0;
  int f = 0;
  bool fill_open = false, queue_empty = false, items_dirty = true;
  int tx = 0, ty = 0;
  int s_first = 0, pool_size = 0, pool_next = 0;
  int live[RT_SLOTS];
#pragma unroll
  for (int d = 0; d < RT_SLOTS; ++d) live[d] = 0;

  // Per-lane state: the walk this lane runs, and the slot of the path it belongs to (-1: none).
  int slot = -1;
  bool trav = false;
  Trav T;
  T.r = make_ray(f3(0, 0, 0), f3(1, 1, 1));
  T.sp = 0;
  T.cur = TRAV_POP;
  T.best.t = INFINITY;
  T.best.prim = -1;

  for (;;) {
    // ======================= work items (wave-uniform), after a pass that may have ended or started paths =======================
    if (items_dirty) {
      items_dirty = false;
      if (fill_open && pool_next >= pool_size) fill_open = false;  // slot f goes on as a draining slot
      unsigned draining = open_mask & ~(fill_open ? (1u << f) : 0u);
      while (draining != 0u) {
        const int d = __builtin_ctz(draining);
        draining &= draining - 1u;
        int live_d = 0;
#pragma unroll
        for (int e = 0; e < RT_SLOTS; ++e) live_d = e == d ? live[e] : live_d;
        if (live_d != 0) continue;
        // the last path of this item has ended: write its tile (main.rs:112-114; sums, not means)
        const int dtx = slot_info[d][0], dty = slot_info[d][1], chunk = slot_info[d][2];
        const size_t slab = A.accumulate ? 0 : (size_t)chunk * A.nx * A.ny * 3;
        const int px = dtx * 8 + (lane & 7);
        const int row = dty * 8 + (lane >> 3);
        if (px < A.nx && row < A.ny) {
          const size_t at = slab + ((size_t)row * A.nx + px) * 3;
          if (A.accumulate) {
#pragma unroll
            for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&out_sum[at + c], acc[c][d * 64 + lane]);
            if (SUMSQ) {
#pragma unroll
              for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&out_sumsq[at + c], acc[3 + c][d * 64 + lane]);
            }
          } else {
            out_sum[at] = acc[0][d * 64 + lane];
            out_sum[at + 1] = acc[1][d * 64 + lane];
            out_sum[at + 2] = acc[2][d * 64 + lane];
            if (SUMSQ) {
              out_sumsq[at] = acc[3][d * 64 + lane];
              out_sumsq[at + 1] = acc[4][d * 64 + lane];
              out_sumsq[at + 2] = acc[5][d * 64 + lane];
            }
          }
        }
        open_mask &= ~(1u << d);
      }
      if (!fill_open && !queue_empty && __popc(open_mask) < slots_cap) {
        int item = 0;
        if (lane == 0) item = (int)atomicAdd(queue, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) {
          queue_empty = true;
        } else {
          f = __builtin_ctz(~open_mask);
          const int chunk = item / n_tiles;
          const int tile = item - chunk * n_tiles;
          tx = tile % A.tiles_x;
          ty = tile / A.tiles_x;
          bool mine = true;
          if (A.shard_count > 1) {
            const int t = A.shard_index + tile * A.shard_count;
            tx = t % A.shard_width;
            ty = t / A.shard_width;
            mine = (tx < A.tiles_x) & (ty < A.tiles_y);
          }
#pragma unroll
          for (int c = 0; c < (SUMSQ ? 6 : 3); ++c) acc[c][f * 64 + lane] = 0.0f;
          if (lane == 0) {
            slot_info[f][0] = tx;
            slot_info[f][1] = ty;
            slot_info[f][2] = chunk;
          }
          const int n_whole = A.n_chunks - A.split_parts;
          int s_len = A.chunk_spp;
          s_first = chunk * A.chunk_spp;
          if (chunk >= n_whole) {
            const int k = chunk - n_whole;
            const int last = A.spp - n_whole * A.chunk_spp;
            const int o0 = last - (last >> k);
            const int o1 = k + 1 < A.split_parts ? last - (last >> (k + 1)) : last;
            s_first = n_whole * A.chunk_spp + o0;
            s_len = o1 - o0;
          }
          pool_size = mine ? 64 * s_len : 0;
          pool_next = 0;
          open_mask |= 1u << f;
          fill_open = true;
          items_dirty = pool_size == 0;  // (an item beyond the image's edge: closed again on the next trip)
#pragma unroll
          for (int e = 0; e < RT_SLOTS; ++e) live[e] = e == f ? 0 : live[e];
        }
      }
      RT_STAMP(4);  // work items
    }

    const int n_trav = __popcll(__ballot(trav));
    // The ray queue is empty and lanes stand idle: the fuller of the two waiting batches runs now, whatever its size.
    const bool hungry = ray_n == 0 && n_trav <= 64 - a_hungry;
    const int spawn_room = fill_open ? pool_size - pool_next : 0;
    const int spawnable = free_n < spawn_room ? free_n : spawn_room;
    const int shadable = shade_n + spawnable;
    unsigned key0 = A.key0, key1 = A.key1;
    asm volatile("" : "+s"(key0), "+s"(key1));

    if (shadable >= a_shade || (hungry && shadable > 0 && shadable >= tex_n)) {
      // ======================= shading: color() for up to 64 paths of the shade queue; new paths for ended ones and free slots =======================
      idle_trips = 0;
      items_dirty = true;
      const int k_rec = shade_n < 64 ? shade_n : 64;
      const bool act = lane < k_rec;
      int my = -1;
      {
        int at = shade_head + (act ? lane : 0);
        at -= at >= P ? P : 0;
        const int id = (int)shade_q[at];
        my = act ? id : -1;
        shade_head += k_rec;
        shade_head -= shade_head >= P ? P : 0;
        shade_n -= k_rec;
      }
      int n_grab = 64 - k_rec;
      n_grab = n_grab < spawnable ? n_grab : spawnable;
      const bool grab = (lane >= k_rec) & (lane < k_rec + n_grab);
      {
        int at = free_head + (grab ? lane - k_rec : 0);
        at -= at >= P ? P : 0;
        const int id = (int)free_q[at];
        my = grab ? id : my;
        free_head += n_grab;
        free_head -= free_head >= P ? P : 0;
        free_n -= n_grab;
      }
      RT_COUNT(8, 1); RT_COUNT(9, k_rec);
      glb_f4* const rec = (glb_f4*)(recs + ((unsigned)(my >= 0 ? my : 0) << 6));
      const f32x4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
      F3 ro = f3(q0.x, q0.y, q0.z);
      F3 rd = sel3(act, f3(q1.x, q1.y, q1.z), f3(1.0f, 1.0f, 1.0f));
      Hit h;
      h.t = q0.w;
      h.prim = act ? __float_as_int(q1.w) : -1;
      F3 thr = f3(q2.x, q2.y, q2.z);
      const unsigned dp = act ? __float_as_uint(q2.w) : 0u;
      unsigned pixel_id = __float_as_uint(q3.x), sample = __float_as_uint(q3.y);
      int depth = (int)(dp >> 8), pix = (int)(dp & 255u);
      const RayF ray = make_ray(ro, rd);
      // what is left of world.hit: the ConstantMedium nodes (they draw random numbers)
      if (GENERAL != 0 && S.n_media_sph > 0) {
        const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
        media_sphere_pass<STATS>(S, act, ray, a_t_min, rng, h, cnt);
      } else if (GENERAL != 0 && S.n_scan > 0) {
        const PathRng rng = {pixel_id, sample, (unsigned)depth, key0, key1};
        scan_range<STATS, true, true, 0>(S, 0, S.n_scan, act, ray, a_t_min, INFINITY, rng, h, cnt);
      }
      RT_STAMP(1);  // media
      const U4 w = philox4x32_10(pixel_id, sample, (unsigned)depth, 0u, key0, key1);
      const Shaded sh = shade_vertex<STATS, GENERAL>(S, load_prim_rec(dprim_tab, act, h.prim), act, h, ray, thr, depth, w, a_max_depth, a_sky, pixel_id, sample, key0, key1, cnt);
      RT_STAMP(2);  // philox + shade
      const bool go_on = act & sh.scattered & ((sh.thr_next.x > 0.0f) | (sh.thr_next.y > 0.0f) | (sh.thr_next.z > 0.0f));
      const bool done = act & !go_on;
      if (done) {
        // de_nan (main.rs:252-264) then col += (main.rs:108)
        const float cr = sh.radiance.x != sh.radiance.x ? 0.0f : sh.radiance.x;
        const float cg = sh.radiance.y != sh.radiance.y ? 0.0f : sh.radiance.y;
        const float cb = sh.radiance.z != sh.radiance.z ? 0.0f : sh.radiance.z;
        if (cr != 0.0f || cg != 0.0f || cb != 0.0f) {
          atomicAdd(&acc[0][pix], cr);
          atomicAdd(&acc[1][pix], cg);
          atomicAdd(&acc[2][pix], cb);
          if (SUMSQ) {
            atomicAdd(&acc[3][pix], cr * cr);
            atomicAdd(&acc[4][pix], cg * cg);
            atomicAdd(&acc[5][pix], cb * cb);
          }
        }
      }
      // A checker or noise texture's colour comes from the texture pass (the vertex used white): the path waits in the
      // texture queue, with the hit point where its next ray starts, and a path that ended on a textured light keeps
      // its weight and its slot until its sample has been added there.
      int pend = -1;
      if (GENERAL >= 2) {
        const bool lit = sh.textured & sh.emits;
        pend = (act & sh.textured & (go_on | sh.emits)) ? ((sh.mat_index << 1) | (lit ? 1 : 0)) : -1;
        thr = sel3(act, sel3(lit, thr, sh.thr_next), thr);
      } else {
        thr = sel3(act, sh.thr_next, thr);
      }
      ro = sel3(act, sh.p, ro);
      rd = sel3(go_on, sh.dir, rd);
      depth += go_on ? 1 : 0;
      const bool to_tex = GENERAL >= 2 && pend >= 0;
      const bool ended = done & !to_tex;
#pragma unroll
      for (int e = 0; e < RT_SLOTS; ++e) live[e] -= __popcll(__ballot(ended & ((pix >> 6) == e)));
      RT_STAMP(3);  // deposit + state update
      // ---- (pixel, sample) paths for the slots whose path ended and for the free slots taken above (main.rs:96-101) ----
      const bool cand = ended | grab;
      const unsigned long long need = __ballot(cand);
      bool start = false;
      int fresh_px = 0, fresh_j = 0;
      if (need != 0ull && fill_open && pool_next < pool_size) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0));
        const int k = pool_next + rank;
        pool_next += __popcll(need);
        const bool take = cand & (k < pool_size);
        const int kk = take ? k : 0;
        const int c_pix = kk & 63;
        const int px = tx * 8 + (c_pix & 7);
        const int row = ty * 8 + (c_pix >> 3);
        start = take & (px < A.nx) & (row < A.ny);
        pix = start ? (f * 64 + c_pix) : pix;
        pixel_id = start ? (unsigned)((A.ny - 1 - row) * A.nx + px) : pixel_id;
        sample = start ? (unsigned)(A.sample_begin + s_first + (kk >> 6)) : sample;
        depth = start ? 0 : depth;
        thr = sel3(start, f3(1.0f, 1.0f, 1.0f), thr);
        fresh_px = px;
        fresh_j = A.ny - 1 - row;
        if (STATS) cnt.paths += start;
#pragma unroll
        for (int e = 0; e < RT_SLOTS; ++e) live[e] += e == f ? __popcll(__ballot(start)) : 0;
      }
      // ---- camera ray of the new paths (main.rs:102-105, camera.rs:53-62); word 0 of the vertex's Philox call ----
      if (__ballot(start) != 0ull) {
        const U4 cw = philox4x32_10(pixel_id, sample, 0u, 0u, key0, key1);
        const float u = ((float)fresh_px + (float)(cw.x >> 16) * (1.0f / 65536.0f)) * A.inv_nx;
        const float v = ((float)fresh_j + (float)(cw.x & 0xffffu) * (1.0f / 65536.0f)) * A.inv_ny;
        int c0 = 0;
        asm volatile("" : "+v"(c0));
        const float* cl = cam_lds + c0;
        float rdx = 0.0f, rdy = 0.0f;
        F3 offset = f3(0.0f, 0.0f, 0.0f);
        if (has_lens) {
          const float lens_radius = cl[18];
          unsigned attempt = 0;
          bool looking = start;
          while (__ballot(looking) != 0ull) {
            const U4 lw = philox4x32_10(pixel_id, sample, 0u, 1u + attempt, key0, key1);
            const float dx = 2.0f * u01(lw.x) - 1.0f, dy = 2.0f * u01(lw.y) - 1.0f;
            const bool ok = looking & (fmaf(dy, dy, dx * dx) < 1.0f);
            rdx = ok ? lens_radius * dx : rdx;
            rdy = ok ? lens_radius * dy : rdy;
            looking &= !ok;
            ++attempt;
          }
          offset = f3(fmaf(cl[15], rdy, cl[12] * rdx), fmaf(cl[16], rdy, cl[13] * rdx), fmaf(cl[17], rdy, cl[14] * rdx));
        }
        const F3 org = f3(cl[0] + offset.x, cl[1] + offset.y, cl[2] + offset.z);
        const F3 dir = f3(fmaf(v, cl[9], fmaf(u, cl[6], cl[3])) - cl[0] - offset.x,
                          fmaf(v, cl[10], fmaf(u, cl[7], cl[4])) - cl[1] - offset.y,
                          fmaf(v, cl[11], fmaf(u, cl[8], cl[5])) - cl[2] - offset.z);
        ro = sel3(start, org, ro);
        rd = sel3(start, dir, rd);
      }
      // ---- the root's own box test of the next ray (bvh.rs:198), made here for the whole batch: a lane that takes
      // the ray starts at the reference it gives; then the records go back and the slots to their next queue ----
      const bool to_ray = (go_on & !to_tex) | start;
      const bool keeps = to_ray | to_tex;
      rd = sel3(keeps, rd, f3(1.0f, 1.0f, 1.0f));
      Counters c2;  // (counted where a lane takes the ray)
      memset(&c2, 0, sizeof(c2));
      const int first = trav_enter<STATS>(S, S.root_ref, S.root_box, make_ray(ro, rd), a_t_min, INFINITY, c2);
      if (keeps) {
        rec[0] = (f32x4){ro.x, ro.y, ro.z, 0.0f};
        rec[1] = (f32x4){rd.x, rd.y, rd.z, __int_as_float(first)};
        rec[2] = (f32x4){thr.x, thr.y, thr.z, __uint_as_float(((unsigned)depth << 8) | (unsigned)pix)};
        rec[3] = (f32x4){__uint_as_float(pixel_id), __uint_as_float(sample), __int_as_float(pend), 0.0f};
      }
      RT_RING_PUSH(ray_q, ray_head, ray_n, to_ray, my);
      if (GENERAL >= 2) RT_RING_PUSH(tex_q, tex_head, tex_n, to_tex, my);
      const bool to_free = (my >= 0) & !keeps;
      RT_RING_PUSH(free_q, free_head, free_n, to_free, my);
      // (the records are read again by other lanes of this wave: the stores are waited for here, once per batch)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RT_STAMP(5);  // new paths, camera, root test, write-back
      continue;
    }

    if (GENERAL >= 2 && (tex_n >= a_tex || (hungry && tex_n > 0))) {
      // ======================= Texture::value (texture.rs:22-35) for up to 64 paths of the texture queue =======================
      idle_trips = 0;
      items_dirty = true;
      const int k_rec = tex_n < 64 ? tex_n : 64;
      const bool act = lane < k_rec;
      int at = tex_head + (act ? lane : 0);
      at -= at >= P ? P : 0;
      const int my = (int)tex_q[at];
      tex_head += k_rec;
      tex_head -= tex_head >= P ? P : 0;
      tex_n -= k_rec;
      glb_f4* const rec = (glb_f4*)(recs + ((unsigned)(act ? my : 0) << 6));
      const f32x4 q0 = rec[0], q2 = rec[2], q3 = rec[3];
      const int pend = act ? __float_as_int(q3.z) : 0;
      const int pix = (int)(__float_as_uint(q2.w) & 255u);
      const bool lit = act & ((pend & 1) != 0);
      const F3 c = texture_value_wave(S.textures, S.perlins, act, act ? S.materials[pend >> 1].texture : 0, f3(q0.x, q0.y, q0.z));
      if (act) {
        const F3 v = f3(q2.x * c.x, q2.y * c.y, q2.z * c.z);
        if (lit) {
          const float cr = v.x != v.x ? 0.0f : v.x, cg = v.y != v.y ? 0.0f : v.y, cb = v.z != v.z ? 0.0f : v.z;  // de_nan
          if (cr != 0.0f || cg != 0.0f || cb != 0.0f) {
            atomicAdd(&acc[0][pix], cr);
            atomicAdd(&acc[1][pix], cg);
            atomicAdd(&acc[2][pix], cb);
            if (SUMSQ) {
              atomicAdd(&acc[3][pix], cr * cr);
              atomicAdd(&acc[4][pix], cg * cg);
              atomicAdd(&acc[5][pix], cb * cb);
            }
          }
        } else {
          rec[2] = (f32x4){v.x, v.y, v.z, q2.w};
        }
      }
#pragma unroll
      for (int e = 0; e < RT_SLOTS; ++e) live[e] -= __popcll(__ballot(lit & ((pix >> 6) == e)));
      RT_RING_PUSH(ray_q, ray_head, ray_n, act & !lit, my);
      RT_RING_PUSH(free_q, free_head, free_n, lit, my);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RT_STAMP(0);  // texture
      continue;
    }

    if (n_trav == 0 && ray_n == 0 && __ballot(slot >= 0) == 0ull) {
      // Nothing to walk, to shade or to colour, and no path to start: every path of the open items has ended.  The
      // item code at the top of the loop writes their tiles and opens the next item, or there is none left.
      if (open_mask == 0u && queue_empty) break;
      // (Every wave must leave this loop: a wave that finds nothing to do this many trips in a row, which the
      // bookkeeping above rules out, gives up rather than spin.)
      if (++idle_trips > 8) break;
      items_dirty = true;
      continue;
    }
    idle_trips = 0;

    // ======================= traversal: hand-overs, steps and leaf rounds until a batch is due =======================
    for (;;) {
      // ---- finished walks are handed over and idle lanes take new rays: when enough walks have ended to make it
      // worth a pass, when nothing else is walking, or when the ray queue is empty and lanes stand idle ----
      const bool fin = !trav & (slot >= 0);
      const unsigned long long m_fin = __ballot(fin);
      const int n_trav0 = __popcll(__ballot(trav));
      if (m_fin != 0ull && (__popcll(m_fin) >= a_refill || n_trav0 == 0 || (ray_n == 0 && 64 - n_trav0 >= a_hungry))) {
        if (fin) {
          glb_f32* const rec = (glb_f32*)(recs + ((unsigned)slot << 6));
          rec[3] = T.best.t;
          rec[7] = __int_as_float(T.best.prim);
        }
        RT_RING_PUSH(shade_q, shade_head, shade_n, fin, slot);
        slot = fin ? -1 : slot;
      }
      const bool want = !trav & (slot < 0);
      const unsigned long long m_want = __ballot(want);
      int n_take = __popcll(m_want);
      n_take = n_take < ray_n ? n_take : ray_n;
      if (n_take > 0) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_want, 0));
        const bool go = want & (rank < n_take);
        if (go) {
          int at = ray_head + rank;
          at -= at >= P ? P : 0;
          const int id = (int)ray_q[at];
          const glb_f4* const rec = (const glb_f4*)(recs + ((unsigned)id << 6));
          const f32x4 q0 = rec[0], q1 = rec[1];
          // ---- world.hit(r, 0.001, INFINITY), main.rs:166: the walk starts where the root's box test sent it ----
          T.r = make_ray(f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z));
          T.cur = __float_as_int(q1.w);
          T.sp = 0;
          T.best.t = INFINITY;
          T.best.prim = -1;
          slot = id;
          trav = true;
          if (STATS) { cnt.rays++; cnt.node_tests++; }
        }
        ray_head += n_take;
        ray_head -= ray_head >= P ? P : 0;
        ray_n -= n_take;
      }
      RT_STAMP(5);  // hand-over
      const unsigned long long m_trav1 = __ballot(trav);
      if (m_trav1 == 0ull) break;
      // a batch is due: the shade queue holds a wave's worth, or the ray queue is empty and lanes stand idle
      if (shade_n + spawnable >= a_shade) break;
      if (ray_n == 0 && __popcll(m_trav1) <= 64 - a_hungry && (shade_n + spawnable > 0 || tex_n > 0)) break;
      RT_COUNT(10, 1); RT_COUNT(11, __popcll(m_trav1));
      // one step for every lane that has one to make; a lane holding a list or instance leaf waits for the leaf code below
      if (trav & !trav_holds_leaf(T.cur)) trav = trav_step<STATS, GENERAL == 3>(S, stk, T, a_t_min, INFINITY, cnt);
      RT_STAMP(6);  // step
      const bool hold = trav & trav_holds_leaf(T.cur);
      const unsigned long long m_trav = __ballot(trav), m_hold = __ballot(hold);
      // the leaf code runs once for all lanes that hold a leaf, when enough of the traversing lanes do
      if (m_hold != 0ull && __popcll(m_hold) * 64 >= a_hold * __popcll(m_trav)) {
        RT_COUNT(12, 1); RT_COUNT(13, __popcll(m_hold)); RT_COUNT(14, __popcll(__ballot(hold & ((T.cur & BN2_PRIM) != 0))));
        if (hold) trav_leaf<STATS>(S, stk, T, a_t_min, INFINITY, cnt);
      }
      RT_STAMP(7);  // leaf code
    }
  }
#undef RT_RING_PUSH

  if (STATS) {
    const unsigned v[9] = {cnt.paths, cnt.rays, cnt.hits, cnt.node_tests, cnt.rect_tests, cnt.sphere_tests, cnt.instance_ops, cnt.shade_events, cnt.light_tests};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      unsigned long long x = v[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (lane == 0) atomicAdd(&stats[k], x);
    }
#ifdef RT_PHASE_TIMING
    if (lane == 0)
      for (int k = 0; k < 16; ++k) atomicAdd(&stats[16 + k], ph[k]);
#endif
  }
}

// ---------------------------------------------------------------------------
// selftest_rcp_kernel: exact_rcp against the correctly rounded division for all
// 2^32 inputs.  counts[0] = mismatching inputs, counts[1] = in-range inputs
// whose Newton step alone (no fallback) mismatches, bad[0..15] = examples.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) selftest_rcp_kernel(unsigned long long* __restrict__ counts, unsigned* __restrict__ bad) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned mism = 0, mism_newton = 0;
  for (unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; v < (1ull << 32); v += stride) {
    const float x = __uint_as_float((unsigned)v);
    const float want = 1.0f / x;
    const float got = exact_rcp(x);
    const bool same = (__float_as_uint(want) == __float_as_uint(got)) | ((want != want) & (got != got));
    const float nw = rcp_newton(x);
    const bool same_n = (__float_as_uint(want) == __float_as_uint(nw)) | ((want != want) & (nw != nw));
    if (!same) {
      const unsigned slot = mism < 1 ? atomicAdd(&bad[16], 1u) : 16u;
      if (slot < 16u) bad[slot] = (unsigned)v;
      ++mism;
    }
    if (!same_n && rcp_in_range(x)) ++mism_newton;
  }
  if (mism) atomicAdd(&counts[0], (unsigned long long)mism);
  if (mism_newton) atomicAdd(&counts[1], (unsigned long long)mism_newton);
}

// ---------------------------------------------------------------------------
// reduce_slabs_kernel: out[i] = slab0[i] + slab1[i] + ... in that order.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float4* __restrict__ slabs, size_t n4, int n_slabs, float4* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = slabs[i];
    for (int c = 1; c < n_slabs; ++c) {
      const float4 b = slabs[(size_t)c * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    out[i] = a;
  }
}

// ---------------------------------------------------------------------------
// tonemap_kernel (main.rs:136-155)
// ---------------------------------------------------------------------------
__global__ void tonemap_kernel(const float* __restrict__ sum, size_t n, float spp, unsigned char* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float c = sqrtf(sum[i] / spp);
  if (c > 1.0f) c = 1.0f;
  const float x = 255.99f * c;
  // `as u8`: saturating, NaN -> 0
  unsigned char b = 0;
  if (x > 0.0f) b = x >= 255.0f ? 255 : (unsigned char)x;
  out[i] = b;
}

}  // namespace

// ---------------------------------------------------------------------------
// Scratch of rt_render_device_noalloc: the work-item counter (first 256 bytes)
// and the partial-sum slabs of a chunked render.  Every GPU the library runs on
// holds its own copy of this array, at its own address, until the process ends.
// In host code the name is only a handle for hipGetSymbolAddress: rt_create asks
// for the array's address on its device once and everything else uses that pointer.
// ---------------------------------------------------------------------------
#define RT_NOALLOC_SCRATCH_BYTES ((size_t)64 << 20)
#define RT_NOALLOC_SLAB_OFFSET ((size_t)256)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

// ---------------------------------------------------------------------------
// Host side of the ABI
// ---------------------------------------------------------------------------
#define RT_OPT_COUNT 15  // RT_OPT_* of rt_abi.h are 1..14
struct rt_ctx {
  int device;
  bool has_scene;
  int general;  // build of the render kernel the scene needs: 0 Lambertian and lights only, 1 every material and media, 2 also textures, 3 also general light shapes
  DevScene dev;
  void* d_scan;
  void* d_prims;
  void* d_ref_nodes;
  void* d_dprims;
  void* d_leaf_box;
  void* d_bnodes;
  void* d_bnodes4;
  void* d_sub_root_ref;
  void* d_box6_index;
  void* d_box6_recs;
  void* d_prim_rank;
  void* d_objects;
  void* d_prim_obj;
  void* d_obj_chain;
  void* d_chain_objs;
  void* d_media_bvh;
  void* d_media_sph;
  void* d_materials;
  void* d_textures;
  void* d_perlins;
  void* d_lights;
  unsigned long long* d_stats;
  unsigned* d_queue;    // the work-item counter of render_kernel
  unsigned* d_stack_ovf;  // render_stack_kernel: the deep ends of the lanes' stacks (grown on demand)
  size_t stack_ovf_bytes;
  void* d_pool;           // render_pool_kernel: the waves' path records (grown on demand)
  size_t pool_bytes;
  int n_cus;
  float* d_slabs;       // partial-sum slabs of chunked renders, grown on demand
  size_t slab_floats;
  unsigned char* scratch_ptr;  // this device's copy of `scratch` (hipGetSymbolAddress in rt_create)
  void* d_stage[6];     // device staging of the host-buffer entry points (rt_render, rt_trace_closest, rt_tonemap), grown on demand
  size_t stage_bytes[6];
  hipEvent_t ev0, ev1;
  hipEvent_t ev_launch;      // recorded after the last enqueued render: a launch on another stream waits for it,
  hipStream_t last_stream;   // because all launches of a context share its work counter and slabs
  bool launched;
  bool verbose;              // RT_VERBOSE was set when the context was made: rt_upload_scene reports what it built and how long each part took
  int opt[RT_OPT_COUNT];     // context options (rt_set_option), indexed by RT_OPT_*; defaults from the environment, read once in rt_create
  double upload_ms[2];       // wall time of the last rt_upload_scene: host build, device copies
};

namespace {

// Range and default of every context option, and the environment variable that can preset it (read once, in rt_create).
struct OptSpec {
  int lo, hi, def;
  const char* env;
};
const OptSpec kOptSpec[RT_OPT_COUNT] = {
    {0, 0, 0, nullptr},
    {0, 1, 0, "RT_NO_ACCUMULATE"},        // RT_OPT_FIXED_ORDER
    {0, 64, 0, "RT_READY_LANES"},         // RT_OPT_READY_LANES (0: per scene)
    {1, 64, 16, "RT_HOLD_LANES"},         // RT_OPT_HOLD_LANES
    {1, 64, RT_STACK_LDS_RENDER, "RT_STACK_LDS_CAP"},  // RT_OPT_STACK_LDS_ENTRIES
    {1, 32, 32, "RT_WAVES_PER_CU"},       // RT_OPT_WAVES_PER_CU
    {1, 1000, 30, "RT_WANT_ROUNDS"},      // RT_OPT_ITEM_ROUNDS
    {0, 65536, 0, "RT_MIN_CHUNK_SPP"},    // RT_OPT_MIN_CHUNK_SPP (0: per launch)
    {1, 65536, RT_STACK_LAST_SPP, "RT_STACK_LAST_SPP"},  // RT_OPT_LAST_CHUNK_SPP
    {0, 512, 0, "RT_POOL_SLOTS"},         // RT_OPT_POOL_SLOTS (0: render_stack_kernel; otherwise at least 64)
    {1, 64, 64, "RT_SHADE_BATCH"},        // RT_OPT_SHADE_BATCH
    {1, 64, 16, "RT_REFILL_LANES"},       // RT_OPT_REFILL_LANES
    {1, 64, 24, "RT_TEX_BATCH"},          // RT_OPT_TEX_BATCH
    {1, 64, 8, "RT_HUNGRY_LANES"},        // RT_OPT_HUNGRY_LANES
    {0, 1, 0, "RT_KEEP_TREE"},            // RT_OPT_KEEP_TREE
};
int clamp_opt(int option, int value) {
  const OptSpec& o = kOptSpec[option];
  if (option == RT_OPT_POOL_SLOTS && value > 0 && value < 64) value = 64;  // (a pool holds a wave's worth of paths at least)
  return value < o.lo ? o.lo : (value > o.hi ? o.hi : value);
}

int hip_fail(hipError_t e, const char* what) {
  return rt::set_error(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

void free_scene(rt_ctx* ctx) {
  void** ptrs[] = {&ctx->d_scan, &ctx->d_prims, &ctx->d_ref_nodes, &ctx->d_dprims, &ctx->d_leaf_box, &ctx->d_bnodes, &ctx->d_bnodes4, &ctx->d_sub_root_ref, &ctx->d_box6_index, &ctx->d_box6_recs, &ctx->d_prim_rank, &ctx->d_objects, &ctx->d_prim_obj, &ctx->d_obj_chain, &ctx->d_chain_objs, &ctx->d_media_bvh, &ctx->d_media_sph, &ctx->d_materials, &ctx->d_textures, &ctx->d_perlins, &ctx->d_lights};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) {
    if (*ptrs[i]) (void)hipFree(*ptrs[i]);
    *ptrs[i] = nullptr;
  }
  ctx->has_scene = false;
}

// Checks every index the kernels follow, so a malformed scene is an error
// return instead of an out-of-bounds access on the device.
int validate_scene(const rt_scene* s) {
  if (!s) return rt::set_error(RT_ERR_INVALID, "scene is null");
  if (s->n_nodes <= 0 || !s->nodes) return rt::set_error(RT_ERR_INVALID, "scene has no nodes");
  if (s->n_prims <= 0 || !s->prims) return rt::set_error(RT_ERR_INVALID, "scene has no primitives");
  if (s->n_materials <= 0 || !s->materials) return rt::set_error(RT_ERR_INVALID, "scene has no materials");
  if (s->n_textures < 0 || s->n_objects < 0 || s->n_perlins < 0) return rt::set_error(RT_ERR_INVALID, "negative count");
  if (s->n_materials >= (1 << 19) || s->n_prims >= (1 << 23)) return rt::set_error(RT_ERR_UNSUPPORTED, "too many materials or primitives");
  if ((s->n_textures > 0 && !s->textures) || (s->n_objects > 0 && !s->objects) || (s->n_perlins > 0 && !s->perlins) || (s->n_lights > 0 && !s->lights))
    return rt::set_error(RT_ERR_INVALID, "a count is positive but its array pointer is null");
  if (s->n_lights < 0 || s->n_lights > RT_MAX_LIGHTS) return rt::set_error(RT_ERR_UNSUPPORTED, "n_lights must be 0 .. RT_MAX_LIGHTS");
  for (int i = 0; i < s->n_lights; ++i) {
    const rt_light& l = s->lights[i];
    if (l.type == RT_LIGHT_SPHERE) {
      if (!(l.f[3] > 0.0f)) return rt::set_error(RT_ERR_INVALID, "light sphere must have a positive radius");
    } else if (l.type == RT_LIGHT_XZ_RECT || l.type == RT_LIGHT_XY_RECT || l.type == RT_LIGHT_YZ_RECT) {
      // XZRect::pdf_value divides by the rect's area (hitable.rs:249-251)
      if (!(l.f[0] < l.f[1]) || !(l.f[2] < l.f[3])) return rt::set_error(RT_ERR_INVALID, "light rect must have a0 < a1 and b0 < b1");
    } else {
      return rt::set_error(RT_ERR_INVALID, "unknown light type");
    }
  }
  for (int i = 0; i < s->n_textures; ++i) {
    const rt_texture& t = s->textures[i];
    if (t.type == RT_TEX_CHECKER) {
      // children are created before their parent (texture.rs:40-45 takes them by value), which also rules out cycles
      if (t.odd < 0 || t.odd >= i || t.even < 0 || t.even >= i) return rt::set_error(RT_ERR_INVALID, "checker texture children must be earlier textures");
    } else if (t.type == RT_TEX_NOISE) {
      if (t.perlin < 0 || t.perlin >= s->n_perlins) return rt::set_error(RT_ERR_INVALID, "noise texture perlin index out of range");
    } else if (t.type != RT_TEX_CONSTANT) {
      return rt::set_error(RT_ERR_INVALID, "unknown texture type");
    }
  }
  for (int i = 0; i < s->n_perlins; ++i)
    for (int k = 0; k < 256; ++k) {
      const rt_perlin& pn = s->perlins[i];
      if ((pn.perm_x[k] | pn.perm_y[k] | pn.perm_z[k]) & ~255) return rt::set_error(RT_ERR_INVALID, "perlin permutation entry outside 0..255");
    }
  for (int i = 0; i < s->n_materials; ++i) {
    const rt_material& m = s->materials[i];
    if (m.type < RT_MAT_LAMBERTIAN || m.type > RT_MAT_ISOTROPIC) return rt::set_error(RT_ERR_INVALID, "unknown material type");
    const bool textured = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_DIFFUSE_LIGHT || m.type == RT_MAT_ISOTROPIC;
    if (textured && (m.texture < 0 || m.texture >= s->n_textures)) return rt::set_error(RT_ERR_INVALID, "material texture index out of range");
  }
  for (int i = 0; i < s->n_prims; ++i) {
    const rt_prim& p = s->prims[i];
    const int type = p.type & RT_PRIM_TYPE_MASK;
    if (type < RT_PRIM_SPHERE || type > RT_PRIM_YZ_RECT) return rt::set_error(RT_ERR_INVALID, "unknown primitive type");
    if ((p.type & ~(RT_PRIM_TYPE_MASK | RT_PRIM_FLIP)) != 0) return rt::set_error(RT_ERR_INVALID, "unknown primitive flag");
    if (p.material < 0 || p.material >= s->n_materials) return rt::set_error(RT_ERR_INVALID, "primitive material index out of range");
    // the kernels' range test takes ordered bounds; a rect built the other way round can never be hit anyway
    if (type != RT_PRIM_SPHERE && (p.f[0] > p.f[1] || p.f[2] > p.f[3]))
      return rt::set_error(RT_ERR_UNSUPPORTED, "rect with a lower bound above its upper bound");
  }
  for (int i = 0; i < s->n_objects; ++i) {
    const rt_object& o = s->objects[i];
    if (o.kind == RT_OBJ_LIST || o.kind == RT_OBJ_MEDIUM) {
      if (o.first < 0 || o.count <= 0 || o.first + o.count > s->n_prims) return rt::set_error(RT_ERR_INVALID, "object primitive range out of bounds");
      if (o.kind == RT_OBJ_MEDIUM) {
        if (o.material < 0 || o.material >= s->n_materials) return rt::set_error(RT_ERR_INVALID, "medium material index out of range");
        if (o.medium_index < 0 || o.medium_index >= s->n_objects) return rt::set_error(RT_ERR_INVALID, "medium ordinal out of range");
        if (!(o.density > 0.0f)) return rt::set_error(RT_ERR_INVALID, "medium density must be positive");
      }
    } else if (o.kind == RT_OBJ_BVH || o.kind == RT_OBJ_MEDIUM_BVH) {
      if (o.first <= 0 || o.first >= s->n_nodes) return rt::set_error(RT_ERR_INVALID, "object sub-tree root out of range");
      if (o.kind == RT_OBJ_MEDIUM_BVH) {
        if (o.material < 0 || o.material >= s->n_materials) return rt::set_error(RT_ERR_INVALID, "medium material index out of range");
        if (o.medium_index < 0 || o.medium_index >= s->n_objects) return rt::set_error(RT_ERR_INVALID, "medium ordinal out of range");
        if (!(o.density > 0.0f)) return rt::set_error(RT_ERR_INVALID, "medium density must be positive");
      }
    } else {
      return rt::set_error(RT_ERR_INVALID, "unknown object kind");
    }
  }
  return RT_OK;
}

// Leaves under each node of the breadth-first array, with the structural checks
// (indices in range, no cycles) done on the way.
int count_leaves(const rt_scene* s, int idx, int depth, std::vector<int>* memo) {
  if (idx < 0 || idx >= s->n_nodes) return rt::set_error(RT_ERR_INVALID, "node index out of range");
  if (depth > 64) return rt::set_error(RT_ERR_INVALID, "node tree deeper than 64 levels (cycle?)");
  if ((*memo)[idx] > 0) return rt::set_error(RT_ERR_INVALID, "node array is not a tree (a node has two parents)");
  const rt_node& n = s->nodes[idx];
  int leaves = 1;
  if (n.left >= 0) {
    if (n.right < 0) return rt::set_error(RT_ERR_INVALID, "interior node with a negative right child");
    const int a = count_leaves(s, n.left, depth + 1, memo);
    if (a < 0) return a;
    const int b = count_leaves(s, n.right, depth + 1, memo);
    if (b < 0) return b;
    leaves = a + b;
  }
  (*memo)[idx] = leaves;
  return leaves;
}

// Which object every primitive and every object lies in, with the structural checks of every tree of the scene (the
// world's and every object's sub-tree: indices in range, no node or object reached twice, bounded depth and nesting)
// done on the way, so that everything after it can follow the indices without looking.
struct FrameWalk {
  const rt_scene* s;
  std::vector<int>* leaves;     // count_leaves' memo, shared by all trees (a node reached from two places is an error)
  std::vector<int> prim_obj;    // per primitive: the innermost object it lies in, -1 for none
  std::vector<int> obj_parent;  // per object: the object it lies in, -1 for the world
  std::vector<char> obj_seen;
  bool nested;                  // an object inside an object, or a medium bounded by a sub-tree: only the stack traversal takes those
  int walk_tree(int root, int parent_obj, int nest) {
    if (nest > RT_MAX_NEST) return rt::set_error(RT_ERR_UNSUPPORTED, "objects nested more than 8 deep");
    const int rc = count_leaves(s, root, 0, leaves);
    if (rc < 0) return rc;
    return visit(root, parent_obj, nest);
  }
  int visit(int idx, int parent_obj, int nest) {
    const rt_node& n = s->nodes[idx];
    if (n.left >= 0) {
      const int rc = visit(n.left, parent_obj, nest);
      if (rc != RT_OK) return rc;
      return visit(n.right, parent_obj, nest);
    }
    const int ref = ~n.left;
    if (n.right > 0) {
      if (ref < 0 || ref > s->n_prims - n.right) return rt::set_error(RT_ERR_INVALID, "leaf primitive range out of bounds");
      for (int k = ref; k < ref + n.right; ++k) prim_obj[(size_t)k] = parent_obj;
      return RT_OK;
    }
    if (n.right < 0) return rt::set_error(RT_ERR_INVALID, "leaf with negative count");
    if (ref < 0 || ref >= s->n_objects) return rt::set_error(RT_ERR_INVALID, "leaf object index out of range");
    if (obj_seen[(size_t)ref]) return rt::set_error(RT_ERR_INVALID, "an object is the leaf of two nodes");
    obj_seen[(size_t)ref] = 1;
    obj_parent[(size_t)ref] = parent_obj;
    const rt_object& ob = s->objects[ref];
    if (parent_obj >= 0) nested = true;
    if (ob.kind >= RT_OBJ_MEDIUM && parent_obj >= 0)
      return rt::set_error(RT_ERR_UNSUPPORTED, "a ConstantMedium inside another object is not supported: media must be leaves of the world tree");
    if (ob.kind == RT_OBJ_LIST || ob.kind == RT_OBJ_MEDIUM) {
      for (int k = ob.first; k < ob.first + ob.count; ++k) prim_obj[(size_t)k] = ref;
      return RT_OK;
    }
    if (ob.kind == RT_OBJ_MEDIUM_BVH) nested = true;
    return walk_tree(ob.first, ref, nest + 1);
  }
};

// Sub-trees with at most this many leaves are stored as their leaves alone.  A
// leaf's box lies inside every ancestor's box and the slab arithmetic is
// monotonic in the box bounds, so a ray that passes the leaf's Aabb::hit passes
// every ancestor's: dropping the interior tests changes no result, and for a
// handful of leaves a wave of 64 incoherent rays would have visited them all anyway.
const int kTreeletLeaves = 16;
// A scene is run as "flat" when its scan array has no interior node and it has at
// most this many primitives (96 bytes of LDS each in the render kernel).
const int kFlatMaxPrims = 64;

struct ScanBuilder {
  const rt_scene* s;
  std::vector<int> leaves;
  std::vector<ScanNode> out;
  std::vector<int> prim_obj;
  std::vector<float> leaf_box;  // 8 floats per primitive: the box of the scan leaf that holds it
  std::vector<float> y_top;     // per node, see compute_y_top
  std::vector<int> media;  // object index of each ConstantMedium, in scan order
  bool any_interior;
  bool incomplete;  // the array leaves something out that only the stack traversal takes (objects inside objects, media bounded by a sub-tree)
  void set_leaf_box(int prim, const rt_node& n) {
    float* b = &leaf_box[(size_t)prim * 8];
    for (int c = 0; c < 3; ++c) { b[c] = n.bmin[c]; b[4 + c] = n.bmax[c]; }
  }
};

// True when prims[first, first+count) is six rects in Cuboid::new's order.
int box6_flag(const rt_scene* s, int first, int count) {
  static const int order[6] = {RT_PRIM_XY_RECT, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT, RT_PRIM_YZ_RECT};
  if (count != 6) return 0;
  for (int k = 0; k < 6; ++k)
    if ((s->prims[first + k].type & RT_PRIM_TYPE_MASK) != order[k]) return 0;
  for (int k = 0; k < 6; k += 2)
    for (int c = 0; c < 4; ++c)
      if (s->prims[first + k].f[c] != s->prims[first + k + 1].f[c]) return 0;
  return LIST_IS_BOX6;
}

// The two extension slots of a six-rect box: per pair a0,a1,b0,b1,k_first,k_second.
void push_box6(const rt_scene* s, int first, std::vector<ScanNode>* out) {
  float g[32];
  std::memset(g, 0, sizeof(g));
  for (int pair = 0; pair < 3; ++pair) {
    const rt_prim& q0 = s->prims[first + 2 * pair];
    const rt_prim& q1 = s->prims[first + 2 * pair + 1];
    for (int c = 0; c < 4; ++c) g[6 * pair + c] = q0.f[c];
    g[6 * pair + 4] = q0.f[4];
    g[6 * pair + 5] = q1.f[4];
  }
  ScanNode ext[2];
  std::memcpy(ext, g, sizeof(ext));
  out->push_back(ext[0]);
  out->push_back(ext[1]);
}

// The smallest float not below v, moved up two more steps to cover the rounding of
// the device's own fp32 arithmetic on the way to a hit point.
float round_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = std::nextafter(f, INFINITY);
  return std::nextafter(std::nextafter(f, INFINITY), INFINITY);
}

// `work` on a thread of its own if one can be had; where the system refuses a thread (std::system_error) the work is done
// here and now, and the returned thread is not joinable.  The build's threads only ever make it faster.
template <class F> std::thread try_thread(F work) {
  try {
    return std::thread(work);
  } catch (const std::system_error&) {
    work();
    return std::thread();
  }
}

// How high the primitives of prims[first, first+count) reach, in their own frame.
double geometry_top(const rt_scene* s, int first, int count) {
  double top = -INFINITY;
  for (int k = first; k < first + count; ++k) {
    const rt_prim& q = s->prims[k];
    const int type = q.type & RT_PRIM_TYPE_MASK;
    const double y = type == RT_PRIM_SPHERE ? (double)q.f[1] + std::fabs((double)q.f[3])
                   : type == RT_PRIM_XY_RECT ? (double)q.f[3] : (type == RT_PRIM_XZ_RECT ? (double)q.f[4] : (double)q.f[1]);
    if (y > top) top = y;
  }
  return top;
}

// y_top of every node (see slab_core): the node's box top, raised where the
// geometry under it reaches higher, which in the reference's constructors is the
// top face of a Cuboid (hitable.rs:398-399).  Nodes of an object's sub-tree get
// theirs in the object's frame; RotateY and Translate leave heights as they are
// up to the offset.
float compute_y_top(const rt_scene* s, int idx, int depth, std::vector<float>* y_top) {
  if (idx < 0 || idx >= s->n_nodes || depth > 130) return -INFINITY;
  const rt_node& n = s->nodes[idx];
  float top = n.bmax[1];
  if (n.left >= 0) {
    if (depth < 4 && s->n_nodes >= (1 << 16)) {
      // (the tree has been checked by now: no node is reached twice, so the two halves write different elements)
      float top_left = -INFINITY;
      std::thread left_half = try_thread([&]() { top_left = compute_y_top(s, n.left, depth + 1, y_top); });
      top = std::max(top, compute_y_top(s, n.right, depth + 1, y_top));
      if (left_half.joinable()) left_half.join();
      top = std::max(top, top_left);
    } else {
      top = std::max(top, compute_y_top(s, n.left, depth + 1, y_top));
      top = std::max(top, compute_y_top(s, n.right, depth + 1, y_top));
    }
  } else if (n.right > 0) {
    const int ref = ~n.left;
    if (ref >= 0 && ref <= s->n_prims - n.right) top = std::max(top, round_up(geometry_top(s, ref, n.right)));
  } else if (~n.left < s->n_objects) {
    const rt_object& ob = s->objects[~n.left];
    const double lift = (ob.xform & RT_XFORM_TRANSLATE) ? (double)ob.offset[1] : 0.0;
    const double inner = (ob.kind == RT_OBJ_BVH || ob.kind == RT_OBJ_MEDIUM_BVH) ? (double)compute_y_top(s, ob.first, depth + 1, y_top) : geometry_top(s, ob.first, ob.count);
    top = std::max(top, round_up(inner + lift));
  }
  (*y_top)[(size_t)idx] = top;
  return top;
}

ScanNode box_node(const rt_node& n, float y_top) {
  ScanNode sn;
  std::memset(&sn, 0, sizeof(sn));
  for (int c = 0; c < 3; ++c) { sn.bmin[c] = n.bmin[c]; sn.bmax[c] = n.bmax[c]; }
  sn.f[5] = y_top;
  return sn;
}

// Depth-first re-layout of the breadth-first node array.  `in_sub` marks nodes
// of a BvhNode used as a child hitable, whose leaves must be primitive leaves;
// `collapsed` is true below the root of a treelet stored as leaves only.
int build_scan(ScanBuilder* B, int idx, bool in_sub, bool collapsed, bool tree_root, int obj) {
  const rt_scene* s = B->s;
  const rt_node& n = s->nodes[idx];
  if (n.left >= 0) {
    const bool small = B->leaves[idx] <= kTreeletLeaves;
    const bool emit = !collapsed && !(small && tree_root);
    const size_t at = B->out.size();
    if (emit) {
      ScanNode sn = box_node(n, B->y_top[(size_t)idx]);
      sn.kind = K_INTERIOR;
      B->out.push_back(sn);
      B->any_interior = true;
    }
    int rc = build_scan(B, n.left, in_sub, collapsed || small, false, obj);
    if (rc != RT_OK) return rc;
    rc = build_scan(B, n.right, in_sub, collapsed || small, false, obj);
    if (rc != RT_OK) return rc;
    if (emit) B->out[at].skip = (int)B->out.size();
    return RT_OK;
  }
  ScanNode sn = box_node(n, B->y_top[(size_t)idx]);
  const int ref = ~n.left;
  if (n.right > 0) {
    if (ref < 0 || ref + n.right > s->n_prims) return rt::set_error(RT_ERR_INVALID, "leaf primitive range out of bounds");
    for (int k = ref; k < ref + n.right; ++k) {
      B->prim_obj[k] = obj;
      if (!in_sub) B->set_leaf_box(k, n);
    }
    if (n.right == 1) {
      const rt_prim& q = s->prims[ref];
      const int type = q.type & RT_PRIM_TYPE_MASK;
      sn.kind = type == RT_PRIM_SPHERE ? K_SPHERE : (type == RT_PRIM_XY_RECT ? K_RECT_XY : (type == RT_PRIM_XZ_RECT ? K_RECT_XZ : K_RECT_YZ));
      for (int k = 0; k < 5; ++k) sn.f[k] = q.f[k];
      sn.a = ref;
      sn.b = 1;
    } else {
      sn.kind = K_LIST;
      sn.a = ref;
      sn.b = n.right | box6_flag(s, ref, n.right);
    }
    const bool box6 = sn.kind == K_LIST && (sn.b & LIST_IS_BOX6);
    sn.skip = (int)B->out.size() + (box6 ? 3 : 1);
    B->out.push_back(sn);
    if (box6) push_box6(s, ref, &B->out);
    return RT_OK;
  }
  if (n.right < 0) return rt::set_error(RT_ERR_INVALID, "leaf with negative count");
  if (ref < 0 || ref >= s->n_objects) return rt::set_error(RT_ERR_INVALID, "leaf object index out of range");
  const rt_object& ob = s->objects[ref];
  if (in_sub) {
    // an object inside an object: the scan does not take it (the stack traversal does)
    B->incomplete = true;
    return RT_OK;
  }
  sn.f[0] = ob.sin_theta; sn.f[1] = ob.cos_theta;
  sn.f[2] = ob.offset[0]; sn.f[3] = ob.offset[1]; sn.f[4] = ob.offset[2];
  const size_t at = B->out.size();
  if (ob.kind == RT_OBJ_LIST || ob.kind == RT_OBJ_MEDIUM) {
    const bool medium = ob.kind == RT_OBJ_MEDIUM;
    for (int k = ob.first; k < ob.first + ob.count; ++k) {
      B->prim_obj[k] = ref;
      B->set_leaf_box(k, n);
    }
    sn.kind = medium ? K_MEDIUM : K_OBJ_LIST;
    sn.a = ob.first;
    sn.b = ob.count | box6_flag(s, ob.first, ob.count) | (ob.xform << 24);
    sn.skip = (int)at + ((sn.b & LIST_IS_BOX6) ? 3 : 1) + (medium ? 1 : 0);
    B->out.push_back(sn);
    if (sn.b & LIST_IS_BOX6) push_box6(s, ob.first, &B->out);
    if (medium) {
      // the medium's own scattering record comes after the primitives; it gets the leaf's box too
      B->leaf_box.resize(((size_t)s->n_prims + B->media.size() + 1) * 8, 0.0f);
      B->set_leaf_box(s->n_prims + (int)B->media.size(), n);
      ScanNode ext;
      std::memset(&ext, 0, sizeof(ext));
      ext.f[0] = ob.density;
      ext.a = ob.medium_index;
      ext.b = s->n_prims + (int)B->media.size();  // its DevPrim
      B->out.push_back(ext);
      B->media.push_back(ref);
    }
    return RT_OK;
  }
  if (ob.kind == RT_OBJ_MEDIUM_BVH) {
    // a medium bounded by a sub-tree: no scan node (media_bvh_pass walks it); its scattering record comes after the primitives like the others'
    B->incomplete = true;
    B->leaf_box.resize(((size_t)s->n_prims + B->media.size() + 1) * 8, 0.0f);
    B->set_leaf_box(s->n_prims + (int)B->media.size(), n);
    B->media.push_back(ref);
    return RT_OK;
  }
  // RT_OBJ_BVH: the sub-tree follows inline, in the object's frame.
  sn.kind = K_OBJ_BVH;
  sn.b = ob.xform << 24;
  B->out.push_back(sn);
  B->any_interior = true;
  const int rc = build_scan(B, ob.first, true, false, true, ref);
  if (rc != RT_OK) return rc;
  B->out[at].skip = (int)B->out.size();
  return RT_OK;
}

// Depth (in nodes) of the tree under idx, following object sub-trees, for sizing the traversal stack.
int tree_depth(const rt_scene* s, int idx) {
  const rt_node& n = s->nodes[idx];
  if (n.left >= 0) {
    const int a = tree_depth(s, n.left), b = tree_depth(s, n.right);
    return 1 + (a > b ? a : b);
  }
  if (n.right == 0 && (s->objects[~n.left].kind == RT_OBJ_BVH || s->objects[~n.left].kind == RT_OBJ_MEDIUM_BVH)) return 2 + tree_depth(s, s->objects[~n.left].first);
  return 1;
}

// Fills the stack traversal's copy of the nodes under idx and gives every
// primitive the depth-first rank of its leaf (what decides an exact tie).
int build_bnodes(const rt_scene* s, int idx, std::vector<BNode>* out, std::vector<int>* prim_rank, int* rank) {
  const rt_node& n = s->nodes[idx];
  BNode& b = (*out)[(size_t)idx];
  for (int c = 0; c < 3; ++c) { b.bmin[c] = n.bmin[c]; b.bmax[c] = n.bmax[c]; }
  b.left = n.left;
  b.right = n.right;
  if (n.left >= 0) {
    if (n.left > BN_INDEX_MASK || n.right > BN_INDEX_MASK) return rt::set_error(RT_ERR_UNSUPPORTED, "too many nodes");
    int rc = build_bnodes(s, n.left, out, prim_rank, rank);
    if (rc != RT_OK) return rc;
    return build_bnodes(s, n.right, out, prim_rank, rank);
  }
  const int ref = ~n.left;
  if (n.right > 0) {
    for (int k = ref; k < ref + n.right; ++k) (*prim_rank)[(size_t)k] = *rank;
  } else if (s->objects[ref].kind == RT_OBJ_LIST || s->objects[ref].kind == RT_OBJ_MEDIUM) {
    for (int k = s->objects[ref].first; k < s->objects[ref].first + s->objects[ref].count; ++k) (*prim_rank)[(size_t)k] = *rank;
  }
  ++*rank;
  if (n.right == 0 && (s->objects[ref].kind == RT_OBJ_BVH || s->objects[ref].kind == RT_OBJ_MEDIUM_BVH))
    return build_bnodes(s, s->objects[ref].first, out, prim_rank, rank);
  return RT_OK;
}

// build_bnodes for a large tree without sub-trees: the leaves under every node are known (count_leaves), so the rank at which
// a subtree starts is known before its left sibling has been walked, and the halves of the top levels go to threads.
int build_bnodes_by_threads(const rt_scene* s, const std::vector<int>& leaves, int idx, std::vector<BNode>* out, std::vector<int>* prim_rank,
                            int rank_start, int depth) {
  const rt_node& n = s->nodes[idx];
  if (n.left < 0 || depth >= 4 || leaves[(size_t)idx] < (1 << 15)) {
    int rank = rank_start;
    return build_bnodes(s, idx, out, prim_rank, &rank);
  }
  BNode& b = (*out)[(size_t)idx];
  for (int c = 0; c < 3; ++c) { b.bmin[c] = n.bmin[c]; b.bmax[c] = n.bmax[c]; }
  b.left = n.left;
  b.right = n.right;
  if (n.left > BN_INDEX_MASK || n.right > BN_INDEX_MASK) return rt::set_error(RT_ERR_UNSUPPORTED, "too many nodes");
  int rc_left = RT_OK;
  std::thread left_half = try_thread([&]() { rc_left = build_bnodes_by_threads(s, leaves, n.left, out, prim_rank, rank_start, depth + 1); });
  const int rc_right = build_bnodes_by_threads(s, leaves, n.right, out, prim_rank, rank_start + leaves[(size_t)n.left], depth + 1);
  if (left_half.joinable()) left_half.join();
  return rc_left != RT_OK ? rc_left : rc_right;
}

// The tree the stack traversal walks.  Its LEAVES are the reference tree's leaves with the reference's boxes, and the
// order that decides an exact tie is the reference's depth-first order (prim_rank); its interior structure is the
// device's own.  That changes no result: BvhNode::hit (bvh.rs:197-228) returns the nearest hit among the leaves that
// the ray reaches, a ray reaches a leaf exactly when it passes the box of every node on the way down, every such box
// contains the leaf's own box and the slab arithmetic is monotonic in the bounds, so "reaches the leaf" is "passes
// the leaf's own box" whatever lies above it.  The reference's interior (a random axis per node, median split,
// bvh.rs:105-147) makes a poor tree to walk; this one is built top-down with the surface area heuristic over 16
// bins per axis, and its interior boxes are the exact unions of the leaf boxes below them.
struct BuildNode {
  float bmin[3], bmax[3];
  float top;     // y_top (see slab_core)
  int left, right;
  int ref_node;  // a leaf: the reference node it stands for; -1 for an interior node
  int n_leaves;  // leaves under the node
};

struct TreeBuilder {
  const rt_scene* s;
  const std::vector<float>* y_top;
  std::vector<BuildNode> nodes;
  bool verbose = false;  // RT_VERBOSE: the parts of a large tree's build are timed on stderr
  bool halve_only;  // no heuristic: halve every range as it stands (a balanced tree; the fallback when the other one is too deep)

  // the reference leaves under reference node idx, in the reference's depth-first order (objects are leaves here)
  // A BvhNode used as an item without Translate or RotateY lives in its parent's frame: its leaves join the parent's
  // tree (the same argument: a leaf of it is reached exactly when the ray passes that leaf's own box).
  // A ConstantMedium is no leaf of this tree: the render kernel walks the media after the tree (they draw random
  // numbers) and a deterministic query skips them, so a lane that reached one would only fetch its records and leave;
  // the global mist of the final scene, whose box is the whole scene, cost every ray one such visit.
  // A BvhNode item without a frame change joins its parent's tree when its root box is an ordinary one.  The sub-tree the
  // flattener writes for a nested HitableList has infinite boxes (every ray tries every item of a list): its object keeps its
  // own leaf, because the box of that leaf (the list's bounding box, which the reference's parent BvhNode tests) is then the
  // only box between a ray and the items, and it need not contain them (a sphere of negative radius has an inverted box).
  static bool joins_parent(const rt_scene* s, int object) {
    const rt_object& ob = s->objects[object];
    if (ob.kind != RT_OBJ_BVH || (ob.xform & 3) != 0) return false;
    const rt_node& root = s->nodes[ob.first];
    for (int c = 0; c < 3; ++c)
      if (!std::isfinite(root.bmin[c]) || !std::isfinite(root.bmax[c])) return false;
    return true;
  }
  void collect(int idx, std::vector<int>* out, bool media = false) const {
    const rt_node& n = s->nodes[idx];
    if (n.left < 0) {
      if (n.right == 0 && joins_parent(s, ~n.left)) collect(s->objects[~n.left].first, out, media);
      else if (media || !(n.right == 0 && s->objects[~n.left].kind >= RT_OBJ_MEDIUM)) out->push_back(idx);
      return;
    }
    collect(n.left, out, media);
    collect(n.right, out, media);
  }
  // A leaf as the builder moves it about: its box beside its reference node, so that the passes over a range of
  // leaves read memory in order instead of following indices into the scene's node array.
  struct LeafRec {
    float bmin[3], bmax[3];
    int ref;
  };
  void make_leaf(const LeafRec& q, size_t at) {
    BuildNode b;
    for (int c = 0; c < 3; ++c) { b.bmin[c] = q.bmin[c]; b.bmax[c] = q.bmax[c]; }
    b.top = (*y_top)[(size_t)q.ref];
    b.left = b.right = -1;
    b.ref_node = q.ref;
    b.n_leaves = 1;
    nodes[at] = b;
  }
  static double half_area(const float* lo, const float* hi) {
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
  // Threads for the passes over one large range (the top levels of the tree, where there are fewer ranges than cores).
  int n_threads = 1;
  std::vector<LeafRec> scratch;  // as long as the leaf array when a range is large enough for a parallel partition
  static const int kBins = 16;
#ifndef RT_SAH_EXACT_LEAVES
#define RT_SAH_EXACT_LEAVES 8
#endif
  static const int kSmall = RT_SAH_EXACT_LEAVES;  // ranges up to this many leaves are split by the exact search
  struct Bins {
    float blo[3][kBins][3], bhi[3][kBins][3];
    int cnt[3][kBins];
    float cmin[3], cmax[3];
  };
  // fn(t, lo, hi) for chunk t of [begin, end) cut into n_chunks, chunk 0 on the calling thread
  template <class F> static void chunks(int begin, int end, int n_chunks, F fn) {
    const long long n = end - begin;
    if (n_chunks <= 1) { fn(0, begin, end); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < n_chunks; ++t) th.push_back(try_thread([=, &fn]() { fn(t, begin + (int)(n * t / n_chunks), begin + (int)(n * (t + 1) / n_chunks)); }));
    fn(0, begin, begin + (int)(n / n_chunks));
    for (size_t t = 0; t < th.size(); ++t)
      if (th[t].joinable()) th[t].join();
  }
  // Builds the subtree over leaves[begin, end) (reordered in place) into nodes[base, base + 2 (end - begin) - 1): the
  // left subtree's nodes, the right subtree's, then the root, which is what it returns.  A subtree's slots are known
  // before it is built, so the two halves of a large range are built by two threads; the passes over a range of
  // 2^17 leaves or more are cut into chunks for the threads the halves do not use yet (same tree whatever the number).
  int build(LeafRec* leaves, int begin, int end, size_t base, int depth = 0) {
    const size_t root_at = base + 2 * (size_t)(end - begin) - 2;
    if (end - begin == 1) {
      make_leaf(leaves[begin], root_at);
      return (int)root_at;
    }
    const bool halve = halve_only || depth > 40;  // (a heuristic split can peel off one leaf at a time: bounded here)
    if (!halve && end - begin <= kSmall) {
      // A range of a few leaves: the binned search below costs a node about 1500 operations whatever the range holds
      // (bins of three axes to clear and to sweep), and nodes over two to eight leaves are six in seven of a tree's
      // nodes.  Here every split of the range along every axis is priced exactly: leaves in the order of their centres,
      // boxes of the first k and of the last n - k.
      const int n = end - begin;
      int mid_s = begin + 1;
      if (n > 2) {
        int order[3][kSmall], best_axis_s = 0, best_k = 1;
        double best_cost_s = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
          int* o = order[axis];
          float key[kSmall];
          for (int i = 0; i < n; ++i) {
            const float ctr = 0.5f * leaves[begin + i].bmin[axis] + 0.5f * leaves[begin + i].bmax[axis];
            int j = i;
            while (j > 0 && key[j - 1] > ctr) { key[j] = key[j - 1]; o[j] = o[j - 1]; --j; }
            key[j] = ctr;
            o[j] = i;
          }
          double right_area[kSmall];
          float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
          for (int k = n - 1; k > 0; --k) {
            const LeafRec& q = leaves[begin + o[k]];
            for (int c = 0; c < 3; ++c) { if (q.bmin[c] < lo[c]) lo[c] = q.bmin[c]; if (q.bmax[c] > hi[c]) hi[c] = q.bmax[c]; }
            right_area[k] = half_area(lo, hi);
          }
          for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
          for (int k = 1; k < n; ++k) {
            const LeafRec& q = leaves[begin + o[k - 1]];
            for (int c = 0; c < 3; ++c) { if (q.bmin[c] < lo[c]) lo[c] = q.bmin[c]; if (q.bmax[c] > hi[c]) hi[c] = q.bmax[c]; }
            double cost = half_area(lo, hi) * k + right_area[k] * (n - k);
            if (!(cost == cost)) cost = 1.0e300;  // a box with an infinite side: still a split
            if (cost < best_cost_s) { best_cost_s = cost; best_axis_s = axis; best_k = k; }
          }
        }
        LeafRec tmp[kSmall];
        for (int i = 0; i < n; ++i) tmp[i] = leaves[begin + order[best_axis_s][i]];
        for (int i = 0; i < n; ++i) leaves[begin + i] = tmp[i];
        mid_s = begin + best_k;
      }
      const int l = build(leaves, begin, mid_s, base, depth + 1);
      const int r = build(leaves, mid_s, end, base + 2 * (size_t)(mid_s - begin) - 1, depth + 1);
      join_children(root_at, l, r, end - begin);
      return (int)root_at;
    }
    int n_chunks = 1;
    if (end - begin >= (1 << 17) && !scratch.empty()) {
      n_chunks = depth < 8 ? n_threads >> depth : 1;
      if (n_chunks < 1) n_chunks = 1;
    }
    Bins one;  // (no allocation for the ranges a single thread walks, which are nearly all of them)
    std::vector<Bins> many(n_chunks > 1 ? (size_t)n_chunks : 0);
    Bins* part = n_chunks > 1 ? many.data() : &one;
    chunks(begin, end, n_chunks, [&](int t, int lo, int hi) {
      Bins& b = part[(size_t)t];
      for (int c = 0; c < 3; ++c) { b.cmin[c] = INFINITY; b.cmax[c] = -INFINITY; }
      for (int i = lo; i < hi; ++i) {
        const LeafRec& n = leaves[i];
        for (int c = 0; c < 3; ++c) {
          const float ctr = 0.5f * n.bmin[c] + 0.5f * n.bmax[c];
          if (ctr < b.cmin[c]) b.cmin[c] = ctr;
          if (ctr > b.cmax[c]) b.cmax[c] = ctr;
        }
      }
    });
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int t = 0; t < n_chunks; ++t)
      for (int c = 0; c < 3; ++c) {
        if (part[(size_t)t].cmin[c] < cmin[c]) cmin[c] = part[(size_t)t].cmin[c];
        if (part[(size_t)t].cmax[c] > cmax[c]) cmax[c] = part[(size_t)t].cmax[c];
      }
    int best_axis = -1, best_split = 0;
    double best_cost = INFINITY;
    if (!halve) {
      // one pass over the leaves fills the bins of all three axes
      float scale[3];
      bool usable[3];
      for (int axis = 0; axis < 3; ++axis) {
        const float extent = cmax[axis] - cmin[axis];
        usable[axis] = extent > 0.0f && std::isfinite(extent);
        scale[axis] = usable[axis] ? (float)kBins / extent : 0.0f;
      }
      chunks(begin, end, n_chunks, [&](int t, int lo, int hi) {
        Bins& b = part[(size_t)t];
        for (int axis = 0; axis < 3; ++axis)
          for (int k = 0; k < kBins; ++k) {
            b.cnt[axis][k] = 0;
            for (int c = 0; c < 3; ++c) { b.blo[axis][k][c] = INFINITY; b.bhi[axis][k][c] = -INFINITY; }
          }
        for (int i = lo; i < hi; ++i) {
          const LeafRec& n = leaves[i];
          for (int axis = 0; axis < 3; ++axis) {
            if (!usable[axis]) continue;
            int k = (int)((0.5f * n.bmin[axis] + 0.5f * n.bmax[axis] - cmin[axis]) * scale[axis]);
            k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
            ++b.cnt[axis][k];
            for (int c = 0; c < 3; ++c) {
              if (n.bmin[c] < b.blo[axis][k][c]) b.blo[axis][k][c] = n.bmin[c];
              if (n.bmax[c] > b.bhi[axis][k][c]) b.bhi[axis][k][c] = n.bmax[c];
            }
          }
        }
      });
      Bins& all = part[0];
      for (int t = 1; t < n_chunks; ++t)
        for (int axis = 0; axis < 3; ++axis)
          for (int k = 0; k < kBins; ++k) {
            all.cnt[axis][k] += part[(size_t)t].cnt[axis][k];
            for (int c = 0; c < 3; ++c) {
              if (part[(size_t)t].blo[axis][k][c] < all.blo[axis][k][c]) all.blo[axis][k][c] = part[(size_t)t].blo[axis][k][c];
              if (part[(size_t)t].bhi[axis][k][c] > all.bhi[axis][k][c]) all.bhi[axis][k][c] = part[(size_t)t].bhi[axis][k][c];
            }
          }
      const auto& blo = all.blo;
      const auto& bhi = all.bhi;
      const auto& cnt = all.cnt;
      for (int axis = 0; axis < 3; ++axis) {
        if (!usable[axis]) continue;
        // sweep from the right, then from the left
        double right_area[kBins];
        int right_cnt[kBins];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int c_acc = 0;
        for (int k = kBins - 1; k > 0; --k) {
          c_acc += cnt[axis][k];
          for (int c = 0; c < 3; ++c) { if (blo[axis][k][c] < lo[c]) lo[c] = blo[axis][k][c]; if (bhi[axis][k][c] > hi[c]) hi[c] = bhi[axis][k][c]; }
          right_cnt[k] = c_acc;
          right_area[k] = c_acc ? half_area(lo, hi) : 0.0;
        }
        for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
        c_acc = 0;
        for (int k = 0; k < kBins - 1; ++k) {
          c_acc += cnt[axis][k];
          for (int c = 0; c < 3; ++c) { if (blo[axis][k][c] < lo[c]) lo[c] = blo[axis][k][c]; if (bhi[axis][k][c] > hi[c]) hi[c] = bhi[axis][k][c]; }
          if (c_acc == 0 || right_cnt[k + 1] == 0) continue;
          double cost = half_area(lo, hi) * c_acc + right_area[k + 1] * right_cnt[k + 1];
          if (!(cost == cost)) cost = INFINITY;  // a box with an infinite side
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k + 1; }
        }
      }
    }
    int mid = begin + (end - begin) / 2;
    if (best_axis >= 0) {
      const float scale = (float)kBins / (cmax[best_axis] - cmin[best_axis]);
      const float base_c = cmin[best_axis];
      const int axis = best_axis, split = best_split;
      auto bin_of = [&](const LeafRec& n) {
        int k = (int)((0.5f * n.bmin[axis] + 0.5f * n.bmax[axis] - base_c) * scale);
        return k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
      };
      if (n_chunks > 1) {
        // the stable partition in chunks: count, then every chunk moves its leaves to the places the counts give them
        std::vector<int> n_left((size_t)n_chunks, 0), lo_of((size_t)n_chunks, 0), hi_of((size_t)n_chunks, 0);
        chunks(begin, end, n_chunks, [&](int t, int lo, int hi) {
          int c = 0;
          for (int i = lo; i < hi; ++i) c += bin_of(leaves[i]) < split ? 1 : 0;
          n_left[(size_t)t] = c;
          lo_of[(size_t)t] = lo;
          hi_of[(size_t)t] = hi;
        });
        int total_left = 0;
        for (int t = 0; t < n_chunks; ++t) total_left += n_left[(size_t)t];
        std::vector<int> at_left((size_t)n_chunks, 0), at_right((size_t)n_chunks, 0);
        int l_acc = begin, r_acc = begin + total_left;
        for (int t = 0; t < n_chunks; ++t) {
          at_left[(size_t)t] = l_acc;
          at_right[(size_t)t] = r_acc;
          l_acc += n_left[(size_t)t];
          r_acc += (hi_of[(size_t)t] - lo_of[(size_t)t]) - n_left[(size_t)t];
        }
        LeafRec* tmp = scratch.data();
        chunks(begin, end, n_chunks, [&](int t, int lo, int hi) {
          int l = at_left[(size_t)t], r = at_right[(size_t)t];
          for (int i = lo; i < hi; ++i) {
            if (bin_of(leaves[i]) < split) tmp[l++] = leaves[i];
            else tmp[r++] = leaves[i];
          }
        });
        chunks(begin, end, n_chunks, [&](int, int lo, int hi) { std::memcpy(leaves + lo, tmp + lo, (size_t)(hi - lo) * sizeof(LeafRec)); });
        mid = begin + total_left;
      } else {
        LeafRec* m = std::stable_partition(leaves + begin, leaves + end, [&](const LeafRec& n) { return bin_of(n) < split; });
        mid = (int)(m - leaves);
      }
      if (mid == begin || mid == end) mid = begin + (end - begin) / 2;
    }
    // (all centres in one point, or nothing to gain: halve the range as it stands)
    int l = 0, r = 0;
    const size_t right_base = base + 2 * (size_t)(mid - begin) - 1;
    if (end - begin >= 32768 && depth < 6) {
      std::thread left_half = try_thread([&]() { l = build(leaves, begin, mid, base, depth + 1); });
      r = build(leaves, mid, end, right_base, depth + 1);
      if (left_half.joinable()) left_half.join();
    } else {
      l = build(leaves, begin, mid, base, depth + 1);
      r = build(leaves, mid, end, right_base, depth + 1);
    }
    join_children(root_at, l, r, end - begin);
    return (int)root_at;
  }
  void join_children(size_t root_at, int l, int r, int n_leaves) {
    BuildNode b;
    for (int c = 0; c < 3; ++c) {
      b.bmin[c] = nodes[(size_t)l].bmin[c] < nodes[(size_t)r].bmin[c] ? nodes[(size_t)l].bmin[c] : nodes[(size_t)r].bmin[c];
      b.bmax[c] = nodes[(size_t)l].bmax[c] > nodes[(size_t)r].bmax[c] ? nodes[(size_t)l].bmax[c] : nodes[(size_t)r].bmax[c];
    }
    b.top = nodes[(size_t)l].top > nodes[(size_t)r].top ? nodes[(size_t)l].top : nodes[(size_t)r].top;
    b.left = l;
    b.right = r;
    b.ref_node = -1;
    b.n_leaves = n_leaves;
    nodes[root_at] = b;
  }
  // RT_OPT_KEEP_TREE: the interior as it was given (the reference's own BvhNode::new tree, or rt_build_lbvh's) instead of
  // one built here.  Same leaves and the same rules as collect(): a BvhNode item without a frame change joins its parent's
  // tree, a ConstantMedium is no leaf.  Returns the root's index in `nodes`, or -1 when there is no leaf under idx.
  bool keep_given = false;
  int mirror(int idx) {
    const rt_node& n = s->nodes[idx];
    if (n.left < 0) {
      if (n.right == 0 && joins_parent(s, ~n.left)) return mirror(s->objects[~n.left].first);
      if (n.right == 0 && s->objects[~n.left].kind >= RT_OBJ_MEDIUM) return -1;
      LeafRec q;
      for (int c = 0; c < 3; ++c) { q.bmin[c] = n.bmin[c]; q.bmax[c] = n.bmax[c]; }
      q.ref = idx;
      nodes.push_back(BuildNode());
      make_leaf(q, nodes.size() - 1);
      return (int)nodes.size() - 1;
    }
    const int l = mirror(n.left), r = mirror(n.right);
    if (l < 0) return r;
    if (r < 0) return l;
    nodes.push_back(BuildNode());
    // (the union of the children's boxes, as the reference's surrounding_box makes it: equal to the given box where the
    // given tree is the reference's, and never larger)
    join_children(nodes.size() - 1, l, r, nodes[(size_t)l].n_leaves + nodes[(size_t)r].n_leaves);
    return (int)nodes.size() - 1;
  }
  // the device's tree over the reference leaves under reference node idx
  int build_under(int idx) {
    if (keep_given && !halve_only) {
      const int root = mirror(idx);
      if (root >= 0) return root;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> leaves;
    auto since = [&](const char* what) {
      if (verbose && leaves.size() >= 65536) std::fprintf(stderr, "  build_under %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    collect(idx, &leaves);
    since("collect");
    if (leaves.empty()) collect(idx, &leaves, true);  // (a world of media only: the tree keeps them so that it has a leaf)
    std::vector<LeafRec> recs(leaves.size());
    for (size_t i = 0; i < leaves.size(); ++i) {
      const rt_node& n = s->nodes[leaves[i]];
      for (int c = 0; c < 3; ++c) { recs[i].bmin[c] = n.bmin[c]; recs[i].bmax[c] = n.bmax[c]; }
      recs[i].ref = leaves[i];
    }
    since("records");
    const size_t base = nodes.size();
    nodes.resize(base + 2 * recs.size() - 1);
    since("nodes resize");
    scratch.clear();
    if (recs.size() >= ((size_t)1 << 17)) {
      unsigned hw = std::thread::hardware_concurrency();
      n_threads = hw >= 16 ? 16 : (hw >= 8 ? 8 : (hw >= 4 ? 4 : (hw >= 2 ? 2 : 1)));  // (a power of two: it is halved with every level)
      if (n_threads > 1) scratch.resize(recs.size());
    }
    since("scratch");
    const int root = build(recs.data(), 0, (int)recs.size(), base);
    since("build");
    scratch.clear();
    scratch.shrink_to_fit();
    return root;
  }
};

// The four-wide view of a built tree (see BNode4): an entry holds up to four descendants of its node.  `need`
// receives the stack entries a lane can have in use below an entry: while it is inside one descendant the others
// it entered wait on the stack, and an object's sub-tree adds its marker.
struct Wide4Builder {
  const rt_scene* s;
  TreeBuilder* T;
  std::vector<BNode4> out;
  std::vector<int> sub_root_ref_own;          // per object (the top builder's)
  std::vector<int>* sub_root_ref = nullptr;   // the top builder's, shared by the builders of its parts
  const std::vector<int>* sub_tree_root = nullptr;  // per object: the built root of an RT_OBJ_BVH's sub-tree (made before the entries: see prebuild)
  const std::vector<int>* box6_index = nullptr;  // per primitive: the six-rect record that starts there, or -1
  std::vector<int> touched;  // objects whose sub_root_ref is an entry of `out`
  // Every sub-tree an object leaf will ask for, built before the four-wide entries are: the entries of a large tree are
  // made by several threads, which then only read the binary nodes.
  static void prebuild(const rt_scene* s, TreeBuilder* T, std::vector<int>* roots) {
    roots->assign((size_t)s->n_objects, -1);
    for (int i = 0; i < s->n_objects; ++i)
      if (s->objects[i].kind == RT_OBJ_BVH && !TreeBuilder::joins_parent(s, i)) (*roots)[(size_t)i] = T->build_under(s->objects[i].first);
  }
  int ref_of_leaf(int ref_node) const {
    const rt_node& c = s->nodes[ref_node];
    if (c.right == 1) return BN2_LEAF | BN2_PRIM | ~c.left;
    // a Cuboid's list goes by its record (one fetch at the leaf instead of node, index and record one after the other)
    if (c.right == 6 && box6_index && (*box6_index)[(size_t)~c.left] >= 0) return BN2_LEAF | BN2_BOX6 | (*box6_index)[(size_t)~c.left];
    return BN2_LEAF | ref_node;
  }
  // the reference of built node `idx` (interior: builds its entry) and the stack it needs
  int ref_of(int idx, int* need, int depth = 0) {
    const BuildNode n = T->nodes[(size_t)idx];
    if (n.ref_node < 0) return build(idx, need, depth);
    *need = 0;
    const rt_node& leaf = s->nodes[n.ref_node];
    if (leaf.right == 0 && s->objects[~leaf.left].kind == RT_OBJ_BVH) {
      int sub = 0;
      const int sub_root = (*sub_tree_root)[(size_t)~leaf.left];
      const int r = ref_of(sub_root, &sub, 99);
      (*sub_root_ref)[2 * (size_t)~leaf.left] = r;
      if (!(r & BN2_LEAF)) touched.push_back(~leaf.left);
      *need = 4 + sub;  // the world ray (three entries) and the marker that switches the lane back to it
    }
    return ref_of_leaf(n.ref_node);
  }
  // The entries another builder made become entries of this one: their references to each other move by the offset.
  int adopt(Wide4Builder& part, int part_root) {
    const int off = (int)out.size();
    out.insert(out.end(), part.out.begin(), part.out.end());
    for (size_t i = (size_t)off; i < out.size(); ++i)
      for (int c = 0; c < 4; ++c)
        if (out[i].ref[c] != BN4_EMPTY && !(out[i].ref[c] & BN2_LEAF)) out[i].ref[c] += off;
    for (size_t k = 0; k < part.touched.size(); ++k) {
      (*sub_root_ref)[2 * (size_t)part.touched[k]] += off;
      touched.push_back(part.touched[k]);
    }
    return part_root + off;
  }
  int build(int idx, int* need, int depth = 0) {
    // Start from the two children and, while there is room, open the interior one with the largest box (the one a
    // ray is most likely to enter anyway) into its own two children, keeping the left-to-right order.
    int kids[4] = {T->nodes[(size_t)idx].left, T->nodes[(size_t)idx].right, 0, 0}, n_kids = 2;
    while (n_kids < 4) {
      int pick = -1;
      double pick_area = -1.0;
      for (int c = 0; c < n_kids; ++c) {
        const BuildNode& k = T->nodes[(size_t)kids[c]];
        if (k.ref_node >= 0) continue;
        const double area = TreeBuilder::half_area(k.bmin, k.bmax);
        if (area > pick_area || pick < 0) { pick_area = area; pick = c; }
      }
      if (pick < 0) break;
      const BuildNode k = T->nodes[(size_t)kids[pick]];
      for (int c = n_kids; c > pick + 1; --c) kids[c] = kids[c - 1];
      kids[pick + 1] = k.right;
      kids[pick] = k.left;
      ++n_kids;
    }
    const int at = (int)out.size();
    out.push_back(BNode4());
    BNode4 b;
    int deepest = 0;
    // a large interior descendant near the top of a large tree gets a builder and a thread of its own
    struct Part {
      Wide4Builder w;
      int root = 0, need = 0;
      std::thread th;
    };
    std::unique_ptr<Part> parts[4];
    for (int c = 0; c < 4; ++c) {
      if (c < n_kids) {
        const BuildNode k = T->nodes[(size_t)kids[c]];
        b.lox[c] = k.bmin[0]; b.loy[c] = k.bmin[1]; b.loz[c] = k.bmin[2];
        b.hix[c] = k.bmax[0]; b.hiy[c] = k.bmax[1]; b.hiz[c] = k.bmax[2];
        b.top[c] = k.top;
        if (k.ref_node < 0 && depth < 2 && k.n_leaves >= (1 << 15)) {
          parts[c].reset(new Part());
          Part* q = parts[c].get();
          q->w.s = s; q->w.T = T; q->w.sub_root_ref = sub_root_ref; q->w.sub_tree_root = sub_tree_root; q->w.box6_index = box6_index;
          q->w.out.reserve((size_t)k.n_leaves * 5 / 8);  // (about half as many entries as leaves; no more than one per interior node)
          const int kid = kids[c];
          q->th = try_thread([q, kid, depth]() { q->root = q->w.build(kid, &q->need, depth + 1); });
          continue;
        }
        int sub = 0;
        b.ref[c] = ref_of(kids[c], &sub, depth + 1);
        if (sub > deepest) deepest = sub;
      } else {
        b.lox[c] = b.loy[c] = b.loz[c] = INFINITY;
        b.hix[c] = b.hiy[c] = b.hiz[c] = -INFINITY;
        b.top[c] = -INFINITY;
        b.ref[c] = BN4_EMPTY;
      }
    }
    for (int c = 0; c < 4; ++c)
      if (parts[c] && parts[c]->th.joinable()) parts[c]->th.join();
    for (int c = 0; c < 4; ++c) {
      if (!parts[c]) continue;
      b.ref[c] = adopt(parts[c]->w, parts[c]->root);
      if (parts[c]->need > deepest) deepest = parts[c]->need;
    }
    out[(size_t)at] = b;
    *need = n_kids - 1 + deepest;
    return at;
  }
};

template <class T> int upload(void** dst, const T* src, size_t n) {
  *dst = nullptr;
  const size_t bytes = (n ? n : 1) * sizeof(T);
  HIP_TRY(hipMalloc(dst, bytes));
  if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return RT_OK;
}

int check_params(const rt_render_params* p) {
  if (!p) return rt::set_error(RT_ERR_INVALID, "params is null");
  if (p->nx <= 0 || p->ny <= 0 || p->nx > 32768 || p->ny > 32768) return rt::set_error(RT_ERR_INVALID, "image size out of range");
  if (p->spp <= 0 || p->spp > (1 << 24) || p->sample_begin < 0) return rt::set_error(RT_ERR_INVALID, "spp out of range (1 .. 2^24)");
  if (p->sample_begin > 0x7fffffff - p->spp) return rt::set_error(RT_ERR_INVALID, "sample_begin + spp exceeds the 31-bit sample index");
  if (p->max_depth < 0 || p->max_depth > 1000) return rt::set_error(RT_ERR_INVALID, "max_depth out of range");
  if (!(p->t_min >= 0.0f) || !(p->t_min < 3.0e38f)) return rt::set_error(RT_ERR_INVALID, "t_min must be finite and not negative");
  if (p->sky != RT_SKY_BLACK && p->sky != RT_SKY_GRADIENT) return rt::set_error(RT_ERR_INVALID, "unknown sky mode");
  if (p->shard_count < 1 || p->shard_index < 0 || p->shard_index >= p->shard_count) return rt::set_error(RT_ERR_INVALID, "bad shard_index/shard_count");
  return RT_OK;
}

DevCamera to_dev_camera(const rt_camera* c) {
  DevCamera d;
  for (int k = 0; k < 3; ++k) {
    d.origin[k] = (float)c->origin[k];
    d.llc[k] = (float)c->lower_left_corner[k];
    d.horizontal[k] = (float)c->horizontal[k];
    d.vertical[k] = (float)c->vertical[k];
    d.u[k] = (float)c->u[k];
    d.v[k] = (float)c->v[k];
  }
  d.lens_radius = (float)c->lens_radius;
  return d;
}

// Grows staging buffer `slot` of the context to at least `bytes` (never shrinks): the host-buffer entry points
// reuse their device buffers from call to call instead of allocating and freeing each time.
int stage(rt_ctx* ctx, int slot, size_t bytes, void** out) {
  if (bytes > ctx->stage_bytes[slot]) {
    if (ctx->d_stage[slot]) HIP_TRY(hipFree(ctx->d_stage[slot]));
    ctx->d_stage[slot] = nullptr;
    ctx->stage_bytes[slot] = 0;
    HIP_TRY(hipMalloc(&ctx->d_stage[slot], bytes));
    ctx->stage_bytes[slot] = bytes;
  }
  *out = ctx->d_stage[slot];
  return RT_OK;
}

// True for memory from hipMalloc on some device; false for host-mapped, managed and unregistered memory.
static bool plain_device_memory(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // the failed query must not show up as the launch's error
    return false;
  }
  return at.type == hipMemoryTypeDevice && !at.isManaged;
}

// no_alloc: the work counter and the slabs are this device's `scratch` array; a launch whose slabs do not fit is refused.
int launch_render(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* p, float* d_sum, float* d_sumsq, bool stats, hipStream_t stream,
                  bool no_alloc = false) {
  RenderArgs A;
  A.nx = p->nx; A.ny = p->ny; A.spp = p->spp; A.sample_begin = p->sample_begin; A.max_depth = p->max_depth; A.sky = p->sky;
  A.t_min = p->t_min; A.inv_nx = 1.0f / (float)p->nx; A.inv_ny = 1.0f / (float)p->ny; A.shard_index = p->shard_index; A.shard_count = p->shard_count;
  A.key0 = (unsigned)p->seed; A.key1 = (unsigned)(p->seed >> 32);
  A.tiles_x = (p->nx + 7) / 8; A.tiles_y = (p->ny + 7) / 8;
  const size_t n = (size_t)p->nx * p->ny * 3;
  int n_tiles = A.tiles_x * A.tiles_y;
  A.shard_width = rt_shard_width(p->nx, p->shard_count);
  if (p->shard_count > 1) {
    const long long numbers = (long long)A.shard_width * A.tiles_y;  // tile numbers t = shard_index (mod shard_count) below this
    if (numbers + p->shard_count > 0x7fffffffLL) return rt::set_error(RT_ERR_UNSUPPORTED, "image too large for tile sharding");
    n_tiles = p->shard_index < numbers ? (int)((numbers - p->shard_index + p->shard_count - 1) / p->shard_count) : 0;
  }
  A.n_tiles = n_tiles;
  if (n_tiles == 0) {  // more shards than tiles: this one has nothing to render
    HIP_TRY(hipMemsetAsync(d_sum, 0, n * sizeof(float), stream));
    if (d_sumsq) HIP_TRY(hipMemsetAsync(d_sumsq, 0, n * sizeof(float), stream));
    return RT_OK;
  }
  // Load balance: a wave keeps its work item for its whole life, so the grid
  // needs many times more items than the chip has wave slots (256 CUs x ~24
  // waves) or the last round leaves most of the chip idle.  Split the samples
  // of each tile until there are about 30 rounds of items, but keep at least 32
  // samples per item (the pool's own tail) and the slabs under 1 GiB.
  const int trav = ctx->dev.stack_mode ? TRAV_STACK : (ctx->dev.flat ? TRAV_FLAT : TRAV_SCAN);
  // The stack-traversal kernel ADDS an item's tile sums into the zeroed output with hardware float adds: no partial
  // images and no reduction pass, so the last chunk of a tile's samples can be halved much further than the partial
  // images of the other kernels allow, and the launch ends on short items (the cost of a tile varies several-fold
  // across an image with media; a short launch, one shard of eight, lost a quarter of its time to its last items).
  // (The order of those adds depends on scheduling, so a pixel's sum can differ in its last bits from run to run;
  // the flat and scan kernels keep their fixed order.)
  // RT_OPT_FIXED_ORDER gives the partial images back, and one open item per wave: bit-reproducible sums.
  // The hardware float adds are only dependable on ordinary device allocations (hipMalloc): on host-mapped, managed or
  // unknown memory they can be lost without an error, so an output buffer of any other kind takes the fixed-order path,
  // which writes it with plain stores.
  bool fixed_order = ctx->opt[RT_OPT_FIXED_ORDER] != 0;
  if (trav == TRAV_STACK && !fixed_order && !(plain_device_memory(d_sum) && (!d_sumsq || plain_device_memory(d_sumsq)))) fixed_order = true;
  A.accumulate = (trav == TRAV_STACK && !fixed_order) ? 1 : 0;
  A.max_slots = fixed_order ? 1 : 8;
  const long long want_items = 256LL * 24 * ctx->opt[RT_OPT_ITEM_ROUNDS];
  int min_chunk_spp = A.accumulate ? 16 : 32;
  // a small launch of the stack kernel (fewer items than a few per wave slot the chip has) takes smaller items, down to
  // 2 samples: its time is the serial chain of paths inside one item (Book-1 scene at 400x225 @ 16: 1.33 -> 0.91 ms)
  if (A.accumulate)
    while (min_chunk_spp > 2 && (long long)n_tiles * (p->spp / min_chunk_spp) < 80LL * (ctx->n_cus > 0 ? ctx->n_cus : 1)) min_chunk_spp /= 2;
  if (ctx->opt[RT_OPT_MIN_CHUNK_SPP] > 0) min_chunk_spp = ctx->opt[RT_OPT_MIN_CHUNK_SPP];
  long long chunks = (want_items + n_tiles - 1) / n_tiles;
  const long long max_by_spp = p->spp / min_chunk_spp > 1 ? p->spp / min_chunk_spp : 1;
  if (chunks > max_by_spp) chunks = max_by_spp;
  if (!A.accumulate && (n % 4) != 0) chunks = 1;  // the slab reduction works on float4
  const size_t per_image = n * (d_sumsq ? 2 : 1);
  while (!A.accumulate && chunks > 1 && (size_t)chunks * per_image * sizeof(float) > ((size_t)1 << 30)) --chunks;
  if (chunks < 1) chunks = 1;
  A.n_chunks = (int)chunks;
  A.chunk_spp = (p->spp + A.n_chunks - 1) / A.n_chunks;
  A.n_chunks = (p->spp + A.chunk_spp - 1) / A.chunk_spp;
  // the last chunk in up to four halving parts of at least 8 samples (see the kernel; sizes measured
  // on five scenes: smaller parts cost more at the end of each item than they save at the end of the launch)
  A.split_parts = 1;
  if (A.accumulate) {
    // no partial images to pay for: halve further, down to RT_STACK_LAST_SPP samples
    const int last = p->spp - (A.n_chunks - 1) * A.chunk_spp;
    const int floor_spp = ctx->opt[RT_OPT_LAST_CHUNK_SPP];
    while (A.split_parts < 8 && (last >> A.split_parts) >= floor_spp) {
      ++A.split_parts;
      ++A.n_chunks;
    }
  } else if ((n % 4) == 0) {
    const int last = p->spp - (A.n_chunks - 1) * A.chunk_spp;
    while (A.split_parts < 4 && (last >> A.split_parts) >= 8 &&
           (size_t)(A.n_chunks + 1) * per_image * sizeof(float) <= ((size_t)1 << 30)) {
      ++A.split_parts;
      ++A.n_chunks;
    }
  }
  // scenes with media or the older materials have the heavier shading phase: it is run for fuller waves
  A.ready_lanes = ctx->opt[RT_OPT_READY_LANES] > 0 ? ctx->opt[RT_OPT_READY_LANES] : (ctx->general != 0 ? 56 : RT_READY_LANES);
  A.hold_lanes = ctx->opt[RT_OPT_HOLD_LANES];
  // (a medium bounded by a sub-tree is walked with the lane's own stack, which the pool kernel's lanes have in use while a batch is shaded)
  A.pool_slots = (trav == TRAV_STACK && ctx->dev.n_media_bvh == 0) ? ctx->opt[RT_OPT_POOL_SLOTS] : 0;
  A.refill_lanes = ctx->opt[RT_OPT_REFILL_LANES];
  A.shade_batch = ctx->opt[RT_OPT_SHADE_BATCH];
  A.tex_batch = ctx->opt[RT_OPT_TEX_BATCH];
  A.hungry_lanes = ctx->opt[RT_OPT_HUNGRY_LANES];
  float* k_sum = d_sum;
  float* k_sumsq = d_sumsq;
  unsigned* d_queue = no_alloc ? (unsigned*)ctx->scratch_ptr : ctx->d_queue;
  if (A.accumulate) {
    // no partial images: the kernel adds into the output itself
  } else if (A.n_chunks > 1 && no_alloc) {
    const size_t need = (size_t)A.n_chunks * per_image;
    const size_t room = (RT_NOALLOC_SCRATCH_BYTES - RT_NOALLOC_SLAB_OFFSET) / sizeof(float);
    if (need > room) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "rt_render_device_noalloc: this launch splits each tile's samples into %d partial images of %zu floats "
                    "(%zu bytes in all); the static scratch holds %zu bytes", A.n_chunks, per_image, need * sizeof(float), room * sizeof(float));
      return rt::set_error(RT_ERR_UNSUPPORTED, msg);
    }
    k_sum = (float*)(ctx->scratch_ptr + RT_NOALLOC_SLAB_OFFSET);
    k_sumsq = d_sumsq ? k_sum + (size_t)A.n_chunks * n : nullptr;
  } else if (A.n_chunks > 1) {
    const size_t need = (size_t)A.n_chunks * per_image;
    if (need > ctx->slab_floats) {
      if (ctx->d_slabs) HIP_TRY(hipFree(ctx->d_slabs));
      ctx->d_slabs = nullptr;
      ctx->slab_floats = 0;
      HIP_TRY(hipMalloc((void**)&ctx->d_slabs, need * sizeof(float)));
      ctx->slab_floats = need;
    }
    k_sum = ctx->d_slabs;
    k_sumsq = d_sumsq ? ctx->d_slabs + (size_t)A.n_chunks * n : nullptr;
  }
  const DevCamera dc = to_dev_camera(cam);
  const int n_items = n_tiles * A.n_chunks;
  const dim3 block(64);
  // One wave for every slot a CU has (8 per SIMD); where the kernel's registers allow fewer, the rest
  // start when the first waves are done, find no item left and leave.  Never more waves than items.
  long long waves = (long long)ctx->opt[RT_OPT_WAVES_PER_CU] * (ctx->n_cus > 0 ? ctx->n_cus : 1);
  if (waves > n_items) waves = n_items;
  // The stack kernel keeps the first entries of a lane's stack in LDS (what decides how many waves a CU holds) and
  // the rest of the scene's worst case in global memory, one region per wave of the grid.
  A.stack_lds = 0;
  A.stack_ovf = nullptr;
  if (trav == TRAV_STACK) {
    const int cap = ctx->opt[RT_OPT_STACK_LDS_ENTRIES];
    A.stack_lds = ctx->dev.stack_depth < cap ? ctx->dev.stack_depth : cap;
    const size_t ovf_bytes = (size_t)waves * (size_t)(ctx->dev.stack_depth - A.stack_lds) * 128 * sizeof(unsigned);
    if (ovf_bytes > 0 && no_alloc) {
      // after the counter, and after the partial images of a launch that has them
      size_t at = RT_NOALLOC_SLAB_OFFSET;
      if (!A.accumulate && A.n_chunks > 1) at += (size_t)A.n_chunks * per_image * sizeof(float);
      at = (at + 255) & ~(size_t)255;
      if (at > RT_NOALLOC_SCRATCH_BYTES)
        return rt::set_error(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: the static scratch cannot hold this launch's partial images");
      // what does not fit the scratch stays in LDS (fewer waves per CU: slower, not refused)
      const size_t per_entry = (size_t)waves * 128 * sizeof(unsigned);
      const int fit = (int)((RT_NOALLOC_SCRATCH_BYTES - at) / per_entry);
      if (ctx->dev.stack_depth - A.stack_lds > fit) A.stack_lds = ctx->dev.stack_depth - fit;
      A.stack_ovf = ctx->dev.stack_depth - A.stack_lds > 0 ? (unsigned*)(ctx->scratch_ptr + at) : nullptr;
    } else if (ovf_bytes > 0) {
      if (ovf_bytes > ctx->stack_ovf_bytes) {
        if (ctx->d_stack_ovf) HIP_TRY(hipFree(ctx->d_stack_ovf));
        ctx->d_stack_ovf = nullptr;
        ctx->stack_ovf_bytes = 0;
        HIP_TRY(hipMalloc((void**)&ctx->d_stack_ovf, ovf_bytes));
        ctx->stack_ovf_bytes = ovf_bytes;
      }
      A.stack_ovf = ctx->d_stack_ovf;
    }
    // the pool kernel's path records: pool_slots x 64 bytes for every wave of the grid
    A.pool_mem = nullptr;
    if (A.pool_slots > 0) {
      const size_t pool_bytes = (size_t)waves * (size_t)A.pool_slots * sizeof(PoolRec);
      if (no_alloc) {
        // after the stack regions in the static scratch, if there is room; otherwise this launch takes the kernel without a pool
        size_t at = RT_NOALLOC_SLAB_OFFSET;
        if (!A.accumulate && A.n_chunks > 1) at += (size_t)A.n_chunks * per_image * sizeof(float);
        at = (at + 255) & ~(size_t)255;
        at += (size_t)waves * (size_t)(ctx->dev.stack_depth - A.stack_lds) * 128 * sizeof(unsigned);
        at = (at + 255) & ~(size_t)255;
        if (at + pool_bytes <= RT_NOALLOC_SCRATCH_BYTES) A.pool_mem = ctx->scratch_ptr + at;
        else A.pool_slots = 0;
      } else {
        if (pool_bytes > ctx->pool_bytes) {
          if (ctx->d_pool) HIP_TRY(hipFree(ctx->d_pool));
          ctx->d_pool = nullptr;
          ctx->pool_bytes = 0;
          HIP_TRY(hipMalloc(&ctx->d_pool, pool_bytes));
          ctx->pool_bytes = pool_bytes;
        }
        A.pool_mem = ctx->d_pool;
      }
    }
  }
  typedef void (*RenderFn)(DevScene, DevCamera, RenderArgs, float*, float*, unsigned long long*, unsigned*);
#define RT_TRIO(ST, SQ, TR) render_kernel<ST, SQ, TR, 0>, render_kernel<ST, SQ, TR, 1>, render_kernel<ST, SQ, TR, 2>, render_kernel<ST, SQ, TR, 3>
#define RT_ROW(ST, SQ) RT_TRIO(ST, SQ, TRAV_SCAN), RT_TRIO(ST, SQ, TRAV_FLAT), nullptr, nullptr, nullptr, nullptr  /* stack scenes: render_stack_kernel */
  static const RenderFn table[48] = {RT_ROW(false, false), RT_ROW(false, true), RT_ROW(true, false), RT_ROW(true, true)};
#undef RT_ROW
#undef RT_TRIO
#define RT_STK(ST, SQ) render_stack_kernel<ST, SQ, 0>, render_stack_kernel<ST, SQ, 1>, render_stack_kernel<ST, SQ, 2>, render_stack_kernel<ST, SQ, 3>
  static const RenderFn stack_table[16] = {RT_STK(false, false), RT_STK(false, true), RT_STK(true, false), RT_STK(true, true)};
#undef RT_STK
  RenderFn fn = table[(stats ? 24 : 0) + (d_sumsq ? 12 : 0) + trav * 4 + ctx->general];
  // scenes on the per-lane stack traversal: the kernel that keeps a lane's traversal going across the shading code
  if (trav == TRAV_STACK) fn = stack_table[(stats ? 8 : 0) + (d_sumsq ? 4 : 0) + ctx->general];
#define RT_POOL(ST, SQ) render_pool_kernel<ST, SQ, 0>, render_pool_kernel<ST, SQ, 1>, render_pool_kernel<ST, SQ, 2>, render_pool_kernel<ST, SQ, 3>
  static const RenderFn pool_table[16] = {RT_POOL(false, false), RT_POOL(false, true), RT_POOL(true, false), RT_POOL(true, true)};
#undef RT_POOL
  // (the same scenes with the wave's paths in a pool and every phase run on a full wave: RT_OPT_POOL_SLOTS)
  if (trav == TRAV_STACK && A.pool_slots > 0) fn = pool_table[(stats ? 8 : 0) + (d_sumsq ? 4 : 0) + ctx->general];
  size_t lds_bytes = trav == TRAV_FLAT ? (size_t)ctx->dev.n_dprims * 6 * sizeof(float4) : 0;
  if (trav == TRAV_STACK) lds_bytes = (size_t)A.stack_lds * 128 * sizeof(unsigned) + (size_t)RT_POOL_LDS_DWORDS(A.pool_slots) * sizeof(unsigned);

  // All launches of a context share its counter and slabs (and all no_alloc launches on a device share `scratch`):
  // a launch on another stream than the last one first waits for that one to finish.
  if (ctx->launched && ctx->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_launch, 0));
  HIP_TRY(hipMemsetAsync(d_queue, 0, sizeof(unsigned), stream));
  if (A.accumulate) {
    HIP_TRY(hipMemsetAsync(k_sum, 0, n * sizeof(float), stream));
    if (k_sumsq) HIP_TRY(hipMemsetAsync(k_sumsq, 0, n * sizeof(float), stream));
  } else if (p->shard_count > 1) {
    // a shard writes its own tiles only: everything else in the image (or in every partial image) is zero
    HIP_TRY(hipMemsetAsync(k_sum, 0, (size_t)A.n_chunks * n * sizeof(float), stream));
    if (k_sumsq) HIP_TRY(hipMemsetAsync(k_sumsq, 0, (size_t)A.n_chunks * n * sizeof(float), stream));
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)waves), block, lds_bytes, stream, ctx->dev, dc, A, k_sum, k_sumsq, ctx->d_stats, d_queue);
  HIP_TRY(hipGetLastError());
  if (A.n_chunks > 1 && !A.accumulate) {
    const size_t n4 = n / 4;
    unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)k_sum, n4, A.n_chunks, (float4*)d_sum);
    if (d_sumsq) hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)k_sumsq, n4, A.n_chunks, (float4*)d_sumsq);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_launch, stream));
  ctx->last_stream = stream;
  ctx->launched = true;
  return RT_OK;
}

}  // namespace

int rt::context_device(const rt_ctx* ctx) { return ctx->device; }

int rt::enqueue_render(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params, float* d_sum, float* d_sumsq, bool stats, void* stream) {
  if (!ctx || !cam || !d_sum) return rt::set_error(RT_ERR_INVALID, "render: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "render: no scene uploaded");
  const int rc = check_params(params);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  return launch_render(ctx, cam, params, d_sum, d_sumsq, stats, (hipStream_t)stream);
}

// On the stream the counting launch will go to, so that the two are ordered: a hipMemset on the null stream can still be
// pending when a kernel on a non-blocking stream has already added its counts.
int rt::reset_counters(rt_ctx* ctx, void* hip_stream) {
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemsetAsync(ctx->d_stats, 0, 32 * sizeof(unsigned long long), (hipStream_t)hip_stream));
  return RT_OK;
}

int rt::read_counters(rt_ctx* ctx, rt_stats* out) {
  HIP_TRY(hipSetDevice(ctx->device));
  unsigned long long v[9];
  HIP_TRY(hipMemcpy(v, ctx->d_stats, sizeof(v), hipMemcpyDeviceToHost));
  std::memset(out, 0, sizeof(*out));
  out->paths = v[0]; out->rays = v[1]; out->hits = v[2]; out->node_tests = v[3]; out->rect_tests = v[4];
  out->sphere_tests = v[5]; out->instance_ops = v[6]; out->shade_events = v[7]; out->light_tests = v[8];
  return RT_OK;
}

extern "C" {

int rt_create(int device_id, rt_ctx** out) {
  if (!out) return rt::set_error(RT_ERR_INVALID, "rt_create: null out");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return rt::set_error(RT_ERR_DEVICE, "rt_create: no HIP device available (this library has no CPU path)");
  if (device_id < 0 || device_id >= count) return rt::set_error(RT_ERR_INVALID, "rt_create: device id out of range");
  HIP_TRY(hipSetDevice(device_id));
  rt_ctx* ctx = new rt_ctx();
  std::memset(ctx, 0, sizeof(*ctx));
  ctx->device = device_id;
  for (int k = 1; k < RT_OPT_COUNT; ++k) {
    ctx->opt[k] = kOptSpec[k].def;
    const char* e = kOptSpec[k].env ? getenv(kOptSpec[k].env) : nullptr;
    if (e && *e) ctx->opt[k] = clamp_opt(k, k == RT_OPT_FIXED_ORDER ? 1 : atoi(e));
  }
  ctx->verbose = getenv("RT_VERBOSE") != nullptr;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  ctx->n_cus = e == hipSuccess ? prop.multiProcessorCount : 0;
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_stats, 32 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_queue, sizeof(unsigned));
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_launch, hipEventDisableTiming);
  // The address of this device's copy of the static scratch array (the device is current here).
  if (e == hipSuccess) e = hipGetSymbolAddress((void**)&ctx->scratch_ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) {
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    delete ctx;
    return hip_fail(e, "rt_create");
  }
  *out = ctx;
  return RT_OK;
}

int rt_set_option(rt_ctx* ctx, int option, int value) {
  if (!ctx) return rt::set_error(RT_ERR_INVALID, "rt_set_option: null context");
  if (option < 1 || option >= RT_OPT_COUNT) return rt::set_error(RT_ERR_INVALID, "rt_set_option: unknown option");
  ctx->opt[option] = clamp_opt(option, value);
  return RT_OK;
}

int rt_get_option(const rt_ctx* ctx, int option, int* value_out) {
  if (!ctx || !value_out) return rt::set_error(RT_ERR_INVALID, "rt_get_option: null argument");
  if (option < 1 || option >= RT_OPT_COUNT) return rt::set_error(RT_ERR_INVALID, "rt_get_option: unknown option");
  *value_out = ctx->opt[option];
  return RT_OK;
}

int rt_shard_width(int nx, int shard_count) {
  int w = (nx + 7) / 8;
  if (shard_count <= 1) return w;
  for (;; ++w) {
    int a = w, b = shard_count;
    while (b) { const int r = a % b; a = b; b = r; }
    if (a == 1) return w;
  }
}

int rt_destroy(rt_ctx* ctx) {
  if (!ctx) return RT_OK;
  (void)hipSetDevice(ctx->device);
  free_scene(ctx);
  if (ctx->d_stats) (void)hipFree(ctx->d_stats);
  if (ctx->d_queue) (void)hipFree(ctx->d_queue);
  if (ctx->d_stack_ovf) (void)hipFree(ctx->d_stack_ovf);
  if (ctx->d_pool) (void)hipFree(ctx->d_pool);
  if (ctx->d_slabs) (void)hipFree(ctx->d_slabs);
  for (int k = 0; k < 6; ++k)
    if (ctx->d_stage[k]) (void)hipFree(ctx->d_stage[k]);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_launch) (void)hipEventDestroy(ctx->ev_launch);
  delete ctx;
  return RT_OK;
}

int rt_upload_scene(rt_ctx* ctx, const rt_scene* s) { return rt_upload_scene_timed(ctx, s, nullptr); }

int rt_upload_scene_timed(rt_ctx* ctx, const rt_scene* s, double* upload_ms) {
  if (!ctx) return rt::set_error(RT_ERR_INVALID, "rt_upload_scene: null context");
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = validate_scene(s);
  if (rc != RT_OK) return rc;
  ScanBuilder B;
  B.s = s;
  B.leaves.assign((size_t)s->n_nodes, 0);
  B.prim_obj.assign((size_t)s->n_prims, -1);
  B.leaf_box.assign((size_t)s->n_prims * 8, 0.0f);
  B.any_interior = false;
  B.incomplete = false;
  FrameWalk FW;
  FW.s = s;
  FW.leaves = &B.leaves;
  FW.prim_obj.assign((size_t)s->n_prims, -1);
  FW.obj_parent.assign((size_t)s->n_objects, -1);
  FW.obj_seen.assign((size_t)s->n_objects, 0);
  FW.nested = false;
  rc = FW.walk_tree(0, -1, 0);
  if (rc < 0) return rc;
  const bool verbose = ctx->verbose;
  auto t_mark = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "rt_upload_scene: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
    t_mark = now;
  };
  lap("checks, frames");
  B.y_top.assign((size_t)s->n_nodes, 0.0f);
  (void)compute_y_top(s, 0, 0, &B.y_top);
  lap("y_top");
  // More than a treelet's worth of leaves (or a sub-tree anywhere: the scan array would hold an interior node):
  // per-lane stack traversal over the device's own four-wide tree; the scan array then keeps only the ConstantMedium
  // nodes, and is only built in full (below) for the scenes that stay on it.
  bool any_sub_tree = false;
  for (int i = 0; i < s->n_objects; ++i) any_sub_tree |= s->objects[i].kind == RT_OBJ_BVH;
  const int depth = tree_depth(s, 0);
  const bool nested = FW.nested;
  const bool want_stack = (B.leaves[0] > kTreeletLeaves || any_sub_tree || nested) && depth <= STACK_DEPTH - 2;
  // the four-wide entries of the interior nodes, and the six-rect records of the list leaves
  TreeBuilder TB;
  TB.s = s;
  TB.y_top = &B.y_top;
  TB.halve_only = false;
  TB.verbose = verbose;
  TB.keep_given = ctx->opt[RT_OPT_KEEP_TREE] != 0;
  Wide4Builder W;
  W.s = s;
  W.T = &TB;
  W.sub_root_ref_own.assign(2 * (size_t)s->n_objects, 0);
  W.sub_root_ref = &W.sub_root_ref_own;
  std::vector<int> sub_tree_root;
  W.sub_tree_root = &sub_tree_root;
  int root_ref = 0, stack_need = 0, world_root_box = 0;
  std::vector<int> box6_index((size_t)s->n_prims, -1);
  std::vector<float> box6_recs;
  if (want_stack) {
    auto note_box6 = [&](int first, int count) {
      if (!box6_flag(s, first, count)) return;
      box6_index[(size_t)first] = (int)(box6_recs.size() / 20);
      float g[20];
      std::memset(g, 0, sizeof(g));
      for (int pair = 0; pair < 3; ++pair) {
        const rt_prim& q0 = s->prims[first + 2 * pair];
        const rt_prim& q1 = s->prims[first + 2 * pair + 1];
        for (int c = 0; c < 4; ++c) g[6 * pair + c] = q0.f[c];
        g[6 * pair + 4] = q0.f[4];
        g[6 * pair + 5] = q1.f[4];
      }
      std::memcpy(&g[18], &first, sizeof(int));  // the list's first primitive, for a lane that comes by the record alone
      box6_recs.insert(box6_recs.end(), g, g + 20);
    };
    for (int i = 0; i < s->n_nodes; ++i)
      if (s->nodes[i].left < 0 && s->nodes[i].right > 1) note_box6(~s->nodes[i].left, s->nodes[i].right);
    for (int i = 0; i < s->n_objects; ++i)
      if (s->objects[i].kind == RT_OBJ_LIST) note_box6(s->objects[i].first, s->objects[i].count);
    if (box6_recs.size() / 20 > (size_t)BN_INDEX_MASK) return rt::set_error(RT_ERR_UNSUPPORTED, "too many cuboids");
    W.box6_index = &box6_index;
    // the world's tree, and the tree of every medium that is bounded by one (walked on its own, after the world's)
    auto build_all = [&]() {
      const int world_root = TB.build_under(0);
      Wide4Builder::prebuild(s, &TB, &sub_tree_root);
      std::vector<int> medium_root((size_t)s->n_objects, -1);
      for (int i = 0; i < s->n_objects; ++i)
        if (s->objects[i].kind == RT_OBJ_MEDIUM_BVH) medium_root[(size_t)i] = TB.build_under(s->objects[i].first);
      lap("SAH tree");
      W.touched.clear();
      W.out.reserve(TB.nodes.size() * 5 / 16);
      root_ref = W.ref_of(world_root, &stack_need);
      for (int i = 0; i < s->n_objects; ++i) {
        if (s->objects[i].kind != RT_OBJ_MEDIUM_BVH) continue;
        int sub = 0;
        W.sub_root_ref_own[2 * (size_t)i] = W.ref_of(medium_root[(size_t)i], &sub);
        if (sub > stack_need) stack_need = sub;
      }
      // The node whose box a walk tests before it takes a tree's root (bvh.rs:198): the given root's, whose box is the
      // union of the leaves' boxes like the device's own root's; but a tree of ONE leaf has no entry that holds the leaf's
      // box, and the leaf need not be the given root (a BvhNode of one item inside a list): then the leaf's own node.
      auto box_node_of = [&](int built_root, int given_root) {
        const BuildNode& b = TB.nodes[(size_t)built_root];
        return b.ref_node >= 0 ? b.ref_node : given_root;
      };
      world_root_box = box_node_of(world_root, 0);
      for (int i = 0; i < s->n_objects; ++i) {
        const int built = s->objects[i].kind == RT_OBJ_MEDIUM_BVH ? medium_root[(size_t)i] : sub_tree_root[(size_t)i];
        if (built >= 0) W.sub_root_ref_own[2 * (size_t)i + 1] = box_node_of(built, s->objects[i].first);
      }
    };
    lap("six-rect records");
    build_all();
    if (stack_need + 2 > RT_STACK_MAX) {
      // the heuristic's tree is too deep for a lane's stack: the balanced one over the same leaves
      TB.nodes.clear();
      TB.halve_only = true;
      W.out.clear();
      build_all();
    }
    if (W.out.size() > (size_t)BN_INDEX_MASK) return rt::set_error(RT_ERR_UNSUPPORTED, "too many nodes");
    lap("four-wide nodes");
  }
  // a tree whose walk could need more stack than a lane has (LDS part + private part) takes the scan
  const bool stack_mode = want_stack && stack_need + 2 <= RT_STACK_MAX;
  if (nested && !stack_mode)
    return rt::set_error(RT_ERR_UNSUPPORTED, "objects inside objects (or a medium bounded by a sub-tree) need the per-lane stack traversal, and this scene's "
                                             "trees are deeper than its stack of 64 entries allows");
  if (stack_mode) {
    // of the scan array only the media's nodes are needed: the world tree's ConstantMedium leaves, in depth-first order
    std::vector<int> world_leaves;
    bool any_medium = false;
    for (int i = 0; i < s->n_objects; ++i) any_medium |= s->objects[i].kind >= RT_OBJ_MEDIUM;
    if (any_medium) TB.collect(0, &world_leaves, true);
    for (size_t i = 0; i < world_leaves.size(); ++i) {
      const rt_node& n = s->nodes[world_leaves[i]];
      if (n.right == 0 && s->objects[~n.left].kind >= RT_OBJ_MEDIUM) {
        rc = build_scan(&B, world_leaves[i], false, true, false, -1);
        if (rc != RT_OK) return rc;
      }
    }
    lap("media nodes");
  } else {
    rc = build_scan(&B, 0, false, false, true, -1);
    if (rc != RT_OK) return rc;
    lap("scan array");
  }
  std::vector<BNode> bnodes;
  std::vector<int> prim_rank((size_t)s->n_prims, 0);
  std::vector<ScanNode> media_scan;
  if (stack_mode) {
    bnodes.assign((size_t)s->n_nodes, BNode());
    int rank = 0;
    bool sub_trees = false;
    for (int i = 0; i < s->n_objects; ++i) sub_trees |= s->objects[i].kind == RT_OBJ_BVH || s->objects[i].kind == RT_OBJ_MEDIUM_BVH;
    if (!sub_trees && s->n_nodes >= (1 << 17)) rc = build_bnodes_by_threads(s, B.leaves, 0, &bnodes, &prim_rank, 0, 0);
    else rc = build_bnodes(s, 0, &bnodes, &prim_rank, &rank);
    if (rc != RT_OK) return rc;
    if (verbose && !sub_trees && s->n_nodes >= (1 << 17)) {
      // (RT_VERBOSE: the ranks the threads made against the ranks of the plain walk)
      std::vector<BNode> check_nodes((size_t)s->n_nodes, BNode());
      std::vector<int> check_rank((size_t)s->n_prims, 0);
      int r2 = 0;
      (void)build_bnodes(s, 0, &check_nodes, &check_rank, &r2);
      size_t differ = 0;
      for (size_t i = 0; i < check_rank.size(); ++i) differ += check_rank[i] != prim_rank[i];
      std::fprintf(stderr, "rt_upload_scene: ranks made by threads: %zu of %zu differ from the plain walk's\n", differ, check_rank.size());
    }
    // (a medium is no leaf of the device's world tree, so build_bnodes, which follows the given one, has met every medium's sub-tree too)
    for (size_t i = 0; i < B.out.size();) {
      const int slots = B.out[i].kind == K_INTERIOR ? 1 : B.out[i].skip - (int)i;
      if (B.out[i].kind == K_MEDIUM) {
        const int at = (int)media_scan.size();
        for (int k = 0; k < slots; ++k) media_scan.push_back(B.out[i + (size_t)k]);
        media_scan[(size_t)at].skip = at + slots;
      }
      i += (B.out[i].kind == K_OBJ_BVH) ? 1 : (size_t)slots;
    }
    lap("node copies, ranks");
  }
  const std::vector<ScanNode>& scan = stack_mode ? media_scan : B.out;
  const auto t_built = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(ctx->device));
  free_scene(ctx);
  if ((rc = upload(&ctx->d_scan, scan.data(), scan.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_prims, s->prims, (size_t)s->n_prims)) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_ref_nodes, s->nodes, (size_t)s->n_nodes)) != RT_OK) return rc;
  // One DevPrim per primitive: geometry + the frame it lives in + its material.
  std::vector<DevPrim> dprims((size_t)s->n_prims + B.media.size());
  int mat_mask = 0, textured = 0, has_chains = 0;
  for (size_t i = 0; i < dprims.size(); ++i) {
    const bool is_medium = i >= (size_t)s->n_prims;
    rt_prim q;
    std::memset(&q, 0, sizeof(q));
    if (is_medium) {
      q.type = DP_TYPE_MEDIUM;
      q.material = s->objects[B.media[i - (size_t)s->n_prims]].material;
    } else {
      q = s->prims[i];
    }
    const rt_material& m = s->materials[q.material];
    mat_mask |= 1 << m.type;
    DevPrim& d = dprims[i];
    std::memset(&d, 0, sizeof(d));
    d.cos_t = 1.0f;
    int xform = 0, chained = 0;
    if (!is_medium && FW.prim_obj[i] >= 0) {
      const int po = FW.prim_obj[i];
      if (FW.obj_parent[(size_t)po] >= 0) {
        chained = DP_CHAINED;  // inside more than one object: finalize_hit goes through them one by one
        has_chains = 1;
      } else {
        const rt_object& ob = s->objects[po];
        xform = ob.xform & 3;
        d.sin_t = ob.sin_theta; d.cos_t = ob.cos_theta;
        d.off[0] = ob.offset[0]; d.off[1] = ob.offset[1]; d.off[2] = ob.offset[2];
      }
    }
    d.flags = (q.type & DP_TYPE_MASK) | ((q.type & RT_PRIM_FLIP) ? DP_FLIP : 0) | (xform << DP_XFORM_SHIFT) |
              (m.type << DP_MAT_SHIFT) | (q.material << DP_INDEX_SHIFT) | chained;
    for (int k = 0; k < 5; ++k) d.f[k] = q.f[k];
    if (m.type == RT_MAT_METAL) {
      for (int k = 0; k < 3; ++k) d.color[k] = m.albedo[k];
    } else if (m.texture >= 0 && m.texture < s->n_textures) {
      const rt_texture& t = s->textures[m.texture];
      if (t.type == RT_TEX_CONSTANT) for (int k = 0; k < 3; ++k) d.color[k] = t.color[k];
      else { d.flags |= DP_TEXTURED; textured = 1; }
    }
    d.fuzz = m.fuzz;
    d.ref_idx = m.ref_idx;
  }
  if ((rc = upload(&ctx->d_dprims, dprims.data(), dprims.size())) != RT_OK) return rc;
  // the objects every object lies in, outermost first and itself last, for the primitives inside more than one
  std::vector<int> obj_chain((size_t)s->n_objects * 2, 0), chain_objs;
  if (has_chains) {
    for (int o = 0; o < s->n_objects; ++o) {
      std::vector<int> up;
      for (int a = o; a >= 0; a = FW.obj_parent[(size_t)a]) up.push_back(a);
      obj_chain[2 * (size_t)o] = (int)chain_objs.size();
      obj_chain[2 * (size_t)o + 1] = (int)up.size();
      chain_objs.insert(chain_objs.end(), up.rbegin(), up.rend());
    }
  }
  std::vector<MediumBvh> media_bvh;
  for (size_t k = 0; k < B.media.size(); ++k) {
    const rt_object& ob = s->objects[B.media[k]];
    if (ob.kind != RT_OBJ_MEDIUM_BVH) continue;
    MediumBvh mb;
    std::memset(&mb, 0, sizeof(mb));
    const float* box = &B.leaf_box[((size_t)s->n_prims + k) * 8];
    for (int c = 0; c < 3; ++c) { mb.bmin[c] = box[c]; mb.bmax[c] = box[4 + c]; }
    mb.object = B.media[k];
    mb.density = ob.density;
    mb.ordinal = ob.medium_index;
    mb.dprim = s->n_prims + (int)k;
    media_bvh.push_back(mb);
  }
  if ((rc = upload(&ctx->d_prim_obj, FW.prim_obj.data(), FW.prim_obj.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_obj_chain, obj_chain.data(), obj_chain.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_chain_objs, chain_objs.data(), chain_objs.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_media_bvh, media_bvh.data(), media_bvh.size())) != RT_OK) return rc;
  // One self-contained record per medium for media_sphere_pass, when the scene is on the stack traversal and every medium
  // left in its scan array is bounded by a single sphere in the world frame.
  std::vector<ScanNode> media_sph;
  if (stack_mode && !scan.empty()) {
    bool all = true;
    for (size_t i = 0; i < scan.size() && all; i = (size_t)scan[i].skip) {
      const ScanNode& n = scan[i];
      if (n.skip <= (int)i) { all = false; break; }
      if (n.kind != K_MEDIUM || (n.b & LIST_IS_BOX6) || (n.b & 0x7fffff) != 1 || (n.b >> 24) != 0 ||
          (s->prims[n.a].type & RT_PRIM_TYPE_MASK) != RT_PRIM_SPHERE) { all = false; break; }
      const ScanNode& ext = scan[(size_t)n.skip - 1];
      ScanNode rec = n;  // the leaf's box, y_top in f[5]
      for (int c = 0; c < 4; ++c) rec.f[c] = s->prims[n.a].f[c];
      rec.f[4] = ext.f[0];
      rec.skip = ext.a;
      rec.b = ext.b;
      media_sph.push_back(rec);
    }
    if (!all || media_sph.size() > 16) media_sph.clear();
  }
  if ((rc = upload(&ctx->d_media_sph, media_sph.data(), media_sph.size())) != RT_OK) return rc;
  B.leaf_box.resize(dprims.size() * 8, 0.0f);
  if ((rc = upload(&ctx->d_leaf_box, B.leaf_box.data(), B.leaf_box.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_bnodes, bnodes.data(), bnodes.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_bnodes4, W.out.data(), W.out.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_sub_root_ref, W.sub_root_ref_own.data(), W.sub_root_ref_own.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_box6_index, box6_index.data(), box6_index.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_box6_recs, box6_recs.data(), box6_recs.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_prim_rank, prim_rank.data(), prim_rank.size())) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_objects, s->objects, (size_t)s->n_objects)) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_materials, s->materials, (size_t)s->n_materials)) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_textures, s->textures, (size_t)s->n_textures)) != RT_OK) return rc;
  if ((rc = upload(&ctx->d_perlins, s->perlins, (size_t)s->n_perlins)) != RT_OK) return rc;
  DevScene& d = ctx->dev;
  std::memset(&d, 0, sizeof(d));
  d.scan = (const ScanNode*)ctx->d_scan;
  d.prims = (const rt_prim*)ctx->d_prims;
  d.ref_nodes = (const rt_node*)ctx->d_ref_nodes;
  d.n_ref_nodes = s->n_nodes;
  d.dprims = (const DevPrim*)ctx->d_dprims;
  d.leaf_box = (const float4*)ctx->d_leaf_box;
  // flat: leaves only, and few enough primitives for the render kernel's LDS tables
  d.flat = (!B.any_interior && !stack_mode && dprims.size() <= (size_t)kFlatMaxPrims) ? 1 : 0;
  d.bnodes = (const BNode*)ctx->d_bnodes;
  d.bnodes4 = (const BNode4*)ctx->d_bnodes4;
  d.root_ref = root_ref;
  d.sub_root_ref = (const int*)ctx->d_sub_root_ref;
  d.root_box = world_root_box;
  d.box6_index = (const int*)ctx->d_box6_index;
  d.box6_recs = (const float4*)ctx->d_box6_recs;
  d.prim_rank = (const int*)ctx->d_prim_rank;
  d.objects = (const rt_object*)ctx->d_objects;
  d.has_chains = has_chains;
  d.prim_obj = (const int*)ctx->d_prim_obj;
  d.obj_chain = (const int*)ctx->d_obj_chain;
  d.chain_objs = (const int*)ctx->d_chain_objs;
  d.media_bvh = (const MediumBvh*)ctx->d_media_bvh;
  d.media_sph = (const ScanNode*)ctx->d_media_sph;
  d.n_media_sph = (int)media_sph.size();
  d.n_media_bvh = (int)media_bvh.size();
  d.stack_mode = stack_mode ? 1 : 0;
  d.no_prune = 0;
  for (int i = 0; i < s->n_prims; ++i)
    if ((s->prims[i].type & RT_PRIM_TYPE_MASK) == RT_PRIM_SPHERE && s->prims[i].f[3] < 0.0f) d.no_prune = 1;
  d.stack_depth = stack_mode ? stack_need + 2 : 0;  // what the deepest walk can have in use, and a margin
  d.stack_lds = d.stack_depth;  // (the trace kernel's; the render kernel's LDS part is set per launch)
  lap("records, copies to the device");
  if (verbose)
    std::fprintf(stderr, "rt_upload_scene: %d nodes, %d primitives, reference tree depth %d; traversal %s; four-wide entries %zu, stack entries per lane %d (%s tree)\n",
                 s->n_nodes, s->n_prims, depth, stack_mode ? "stack" : (d.flat ? "flat" : "scan"), W.out.size(), d.stack_depth,
                 TB.halve_only ? "balanced" : "surface-area");
  d.mat_mask = mat_mask;
  d.textured = textured;
  d.has_spheres = 0;
  for (int i = 0; i < s->n_prims; ++i)
    if ((s->prims[i].type & RT_PRIM_TYPE_MASK) == RT_PRIM_SPHERE) d.has_spheres = 1;
  d.materials = (const rt_material*)ctx->d_materials;
  d.textures = (const rt_texture*)ctx->d_textures;
  d.perlins = (const rt_perlin*)ctx->d_perlins;
  d.n_scan = (int)scan.size();
  d.n_prims = s->n_prims;
  d.n_dprims = (int)dprims.size();
  d.n_lights = s->n_lights;
  if (s->n_lights > 0) d.light = s->lights[0];
  if ((rc = upload(&ctx->d_lights, s->lights, (size_t)s->n_lights)) != RT_OK) return rc;
  d.lights = (const rt_light*)ctx->d_lights;
  d.light_general = (s->n_lights > 1 || (s->n_lights == 1 && s->lights[0].type != RT_LIGHT_XZ_RECT)) ? 1 : 0;
  // (the general light code is compiled into build 3 only, so that the other builds keep their registers; so is the code
  // for primitives inside more than one object and for media bounded by a sub-tree)
  ctx->general = (d.light_general || has_chains || d.n_media_bvh > 0 || (d.no_prune && stack_mode)) ? 3 : textured ? 2 : ((!B.media.empty() || (mat_mask & ~((1 << RT_MAT_LAMBERTIAN) | (1 << RT_MAT_DIFFUSE_LIGHT))) != 0) ? 1 : 0);
  ctx->has_scene = true;
  const auto t_end = std::chrono::steady_clock::now();
  ctx->upload_ms[0] = std::chrono::duration<double, std::milli>(t_built - t_begin).count();
  ctx->upload_ms[1] = std::chrono::duration<double, std::milli>(t_end - t_built).count();
  if (upload_ms) { upload_ms[0] = ctx->upload_ms[0]; upload_ms[1] = ctx->upload_ms[1]; }
  return RT_OK;
}

int rt_render_device(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params, float* d_sum, float* d_sumsq, void* hip_stream) {
  if (!ctx || !cam || !d_sum) return rt::set_error(RT_ERR_INVALID, "rt_render_device: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "rt_render_device: no scene uploaded");
  const int rc = check_params(params);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  return launch_render(ctx, cam, params, d_sum, d_sumsq, false, (hipStream_t)hip_stream);
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params, float* d_sum, float* d_sumsq, void* hip_stream) {
  if (!ctx || !cam || !d_sum) return rt::set_error(RT_ERR_INVALID, "rt_render_device_noalloc: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "rt_render_device_noalloc: no scene uploaded");
  const int rc = check_params(params);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  return launch_render(ctx, cam, params, d_sum, d_sumsq, false, (hipStream_t)hip_stream, true);
}

int rt_render(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params, float* out_sum, float* out_sumsq, rt_stats* stats) {
  if (!ctx || !cam || !out_sum) return rt::set_error(RT_ERR_INVALID, "rt_render: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
  int rc = check_params(params);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)params->nx * params->ny * 3 * sizeof(float);
  float *d_sum = nullptr, *d_sumsq = nullptr;
  if ((rc = stage(ctx, 0, bytes, (void**)&d_sum)) != RT_OK) return rc;
  if (out_sumsq && (rc = stage(ctx, 1, bytes, (void**)&d_sumsq)) != RT_OK) return rc;
  hipError_t e = hipSuccess;
  if (stats) e = hipMemset(ctx->d_stats, 0, 32 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipEventRecord(ctx->ev0, 0);
  rc = RT_OK;
  if (e == hipSuccess) rc = launch_render(ctx, cam, params, d_sum, d_sumsq, stats != nullptr, 0);
  if (e == hipSuccess && rc == RT_OK) e = hipEventRecord(ctx->ev1, 0);
  if (e == hipSuccess && rc == RT_OK) e = hipEventSynchronize(ctx->ev1);
  if (e == hipSuccess && rc == RT_OK) e = hipMemcpy(out_sum, d_sum, bytes, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == RT_OK && out_sumsq) e = hipMemcpy(out_sumsq, d_sumsq, bytes, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == RT_OK && stats) {
    unsigned long long v[32];
    e = hipMemcpy(v, ctx->d_stats, sizeof(v), hipMemcpyDeviceToHost);
#ifdef RT_PHASE_TIMING
    if (e == hipSuccess && ctx->dev.stack_mode) {
      static const char* names[8] = {"texture", "media", "philox+shade", "deposit+update", "refill+camera", "start traversal", "steps", "leaf code"};
      double total = 0;
      for (int k = 0; k < 8; ++k) total += (double)v[16 + k];
      for (int k = 0; k < 8; ++k) std::fprintf(stderr, "  phase %-16s %6.2f %%\n", names[k], 100.0 * (double)v[16 + k] / total);
      std::fprintf(stderr, "  shading phases %llu, lanes per phase %.1f; step trips %llu, lanes per trip %.1f; leaf rounds %llu, lanes per round %.1f\n",
                   v[24], (double)v[25] / (double)(v[24] ? v[24] : 1), v[26], (double)v[27] / (double)(v[26] ? v[26] : 1), v[28], (double)v[29] / (double)(v[28] ? v[28] : 1));
      std::fprintf(stderr, "  of the lanes in leaf rounds, %.1f %% held a leaf of one primitive (the rest a list or an object)\n", 100.0 * (double)v[30] / (double)(v[29] ? v[29] : 1));
    }
#endif
    if (e == hipSuccess) {
      std::memset(stats, 0, sizeof(*stats));
      stats->paths = v[0]; stats->rays = v[1]; stats->hits = v[2]; stats->node_tests = v[3]; stats->rect_tests = v[4];
      stats->sphere_tests = v[5]; stats->instance_ops = v[6]; stats->shade_events = v[7]; stats->light_tests = v[8];
      float ms = 0.0f;
      e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
      stats->kernel_ms = ms;
    }
  }
  if (rc != RT_OK) return rc;
  if (e != hipSuccess) return hip_fail(e, "rt_render");
  return RT_OK;
}

int rt_trace_closest(rt_ctx* ctx, const float* o, const float* d, size_t n, float t_min, float t_max, int32_t* prim_id, float* t,
                     float* normal, float* p) {
  if (!ctx || !o || !d || !prim_id || !t) return rt::set_error(RT_ERR_INVALID, "rt_trace_closest: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "rt_trace_closest: no scene uploaded");
  if (n == 0) return RT_OK;
  if (!(t_min <= t_max)) {
    // an empty (or NaN) interval: every test rejects (hitable.rs:165), and the kernels assume t_min <= t_max
    for (size_t i = 0; i < n; ++i) { prim_id[i] = -1; t[i] = 0.0f; }
    if (normal) std::memset(normal, 0, n * 3 * sizeof(float));
    if (p) std::memset(p, 0, n * 3 * sizeof(float));
    return RT_OK;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr, *d_n = nullptr, *d_p = nullptr;
  int* d_id = nullptr;
  int rc;
  if ((rc = stage(ctx, 0, n * 3 * sizeof(float), (void**)&d_o)) != RT_OK) return rc;
  if ((rc = stage(ctx, 1, n * 3 * sizeof(float), (void**)&d_d)) != RT_OK) return rc;
  if ((rc = stage(ctx, 2, n * sizeof(float), (void**)&d_t)) != RT_OK) return rc;
  if ((rc = stage(ctx, 3, n * sizeof(int), (void**)&d_id)) != RT_OK) return rc;
  if (normal && (rc = stage(ctx, 4, n * 3 * sizeof(float), (void**)&d_n)) != RT_OK) return rc;
  if (p && (rc = stage(ctx, 5, n * 3 * sizeof(float), (void**)&d_p)) != RT_OK) return rc;
  hipError_t e = hipMemcpy(d_o, o, n * 3 * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_d, d, n * 3 * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (ctx->dev.stack_mode) {
      // the whole stack of every lane in LDS: as many waves per workgroup (at most 4) as 64 KB hold
      const size_t per_wave = (size_t)ctx->dev.stack_lds * 128 * sizeof(unsigned);
      int wpb = per_wave ? (int)(65536 / per_wave) : 4;
      wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);
      size_t sblocks = (n + (size_t)wpb * 64 - 1) / ((size_t)wpb * 64);
      if (sblocks > 16384u / (unsigned)wpb) sblocks = 16384u / (unsigned)wpb;
      hipLaunchKernelGGL(trace_kernel<TRAV_STACK>, dim3((unsigned)sblocks), dim3((unsigned)wpb * 64), per_wave * (size_t)wpb, 0, ctx->dev, d_o, d_d, n, t_min, t_max, d_id, d_t, d_n, d_p);
    }
    else if (ctx->dev.flat) hipLaunchKernelGGL(trace_kernel<TRAV_FLAT>, dim3((unsigned)blocks), dim3(256), 0, 0, ctx->dev, d_o, d_d, n, t_min, t_max, d_id, d_t, d_n, d_p);
    else hipLaunchKernelGGL(trace_kernel<TRAV_SCAN>, dim3((unsigned)blocks), dim3(256), 0, 0, ctx->dev, d_o, d_d, n, t_min, t_max, d_id, d_t, d_n, d_p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(prim_id, d_id, n * sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess && normal) e = hipMemcpy(normal, d_n, n * 3 * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess && p) e = hipMemcpy(p, d_p, n * 3 * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "rt_trace_closest");
  return RT_OK;
}

int rt_trace_closest_f64(rt_ctx* ctx, const double* o, const double* d, size_t n, double t_min, double t_max, int32_t* prim_id, double* t,
                         double* normal, double* p) {
  if (!ctx || !o || !d || !prim_id || !t) return rt::set_error(RT_ERR_INVALID, "rt_trace_closest_f64: null argument");
  if (!ctx->has_scene) return rt::set_error(RT_ERR_NO_SCENE, "rt_trace_closest_f64: no scene uploaded");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  double *d_o = nullptr, *d_d = nullptr, *d_t = nullptr, *d_n = nullptr, *d_p = nullptr;
  int* d_id = nullptr;
  int rc;
  if ((rc = stage(ctx, 0, n * 3 * sizeof(double), (void**)&d_o)) != RT_OK) return rc;
  if ((rc = stage(ctx, 1, n * 3 * sizeof(double), (void**)&d_d)) != RT_OK) return rc;
  if ((rc = stage(ctx, 2, n * sizeof(double), (void**)&d_t)) != RT_OK) return rc;
  if ((rc = stage(ctx, 3, n * sizeof(int), (void**)&d_id)) != RT_OK) return rc;
  if (normal && (rc = stage(ctx, 4, n * 3 * sizeof(double), (void**)&d_n)) != RT_OK) return rc;
  if (p && (rc = stage(ctx, 5, n * 3 * sizeof(double), (void**)&d_p)) != RT_OK) return rc;
  hipError_t e = hipMemcpy(d_o, o, n * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_d, d, n * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    size_t blocks = (n + 63) / 64;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(trace64_kernel, dim3((unsigned)blocks), dim3(64), 0, 0, ctx->dev, d_o, d_d, n, t_min, t_max, d_id, d_t, d_n, d_p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(prim_id, d_id, n * sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(t, d_t, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && normal) e = hipMemcpy(normal, d_n, n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && p) e = hipMemcpy(p, d_p, n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "rt_trace_closest_f64");
  return RT_OK;
}

int rt_selftest_reciprocal(rt_ctx* ctx, uint64_t* mismatches, uint32_t* examples16) {
  if (!ctx || !mismatches) return rt::set_error(RT_ERR_INVALID, "rt_selftest_reciprocal: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  unsigned long long* d_counts = nullptr;
  unsigned* d_bad = nullptr;
  hipError_t e = hipMalloc((void**)&d_counts, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&d_bad, 17 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(d_counts, 0, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(d_bad, 0, 17 * sizeof(unsigned));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(selftest_rcp_kernel, dim3(256 * 32), dim3(256), 0, 0, d_counts, d_bad);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long counts[2] = {0, 0};
  unsigned bad[17];
  std::memset(bad, 0, sizeof(bad));
  if (e == hipSuccess) e = hipMemcpy(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
  if (d_counts) (void)hipFree(d_counts);
  if (d_bad) (void)hipFree(d_bad);
  if (e != hipSuccess) return hip_fail(e, "rt_selftest_reciprocal");
  mismatches[0] = counts[0] + counts[1];
  if (examples16) std::memcpy(examples16, bad, 16 * sizeof(unsigned));
  return RT_OK;
}

int rt_tonemap_device(rt_ctx* ctx, const float* d_sum_rgb, int32_t nx, int32_t ny, int32_t spp_total, uint8_t* d_rgb8, void* hip_stream) {
  if (!ctx || !d_sum_rgb || !d_rgb8 || nx <= 0 || ny <= 0 || spp_total <= 0) return rt::set_error(RT_ERR_INVALID, "rt_tonemap_device: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = (size_t)nx * ny * 3;
  hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, d_sum_rgb, n, (float)spp_total, d_rgb8);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

int rt_tonemap(rt_ctx* ctx, const float* sum_rgb, int32_t nx, int32_t ny, int32_t spp_total, uint8_t* rgb8) {
  if (!ctx || !sum_rgb || !rgb8 || nx <= 0 || ny <= 0 || spp_total <= 0) return rt::set_error(RT_ERR_INVALID, "rt_tonemap: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = (size_t)nx * ny * 3;
  float* d_in = nullptr;
  unsigned char* d_out = nullptr;
  int rc;
  if ((rc = stage(ctx, 0, n * sizeof(float), (void**)&d_in)) != RT_OK) return rc;
  if ((rc = stage(ctx, 1, n, (void**)&d_out)) != RT_OK) return rc;
  HIP_TRY(hipMemcpy(d_in, sum_rgb, n * sizeof(float), hipMemcpyHostToDevice));
  if ((rc = rt_tonemap_device(ctx, d_in, nx, ny, spp_total, d_out, nullptr)) != RT_OK) return rc;
  HIP_TRY(hipMemcpy(rgb8, d_out, n, hipMemcpyDeviceToHost));
  return RT_OK;
}

}  // extern "C"
