/*
 * rt_abi.h -- C ABI of the MI355X path-tracing hot path.
 *
 * The reference (jesseyoungmann/raytracing) is a Rust binary crate with no FFI
 * of its own (src/main.rs:10-20 declares only private modules, there is no
 * lib.rs / build.rs / extern block).  The seam this library replaces is the
 * in-process region src/main.rs:67-133: the 8-thread fan-out, the per-thread
 * BvhNode::new (src/bvh.rs:105-147), the pixel/sample triple loop
 * (src/main.rs:96-116), color() (src/main.rs:165-209) and the 8-way image sum
 * (src/main.rs:128-133).  Inputs at that seam are a Camera (src/camera.rs:5-14),
 * a HitableList (src/hitable.rs:94-97) and nx, ny, ns; the output is nx*ny
 * linear-radiance RGB values, top row first.
 *
 * Everything here is plain-old-data: pointers, sizes, fixed-width scalars.
 * All functions return RT_OK (0) or a negative error code; the text of the
 * last error of the calling thread is available from rt_last_error().
 * The caller owns every host buffer it passes in; the library copies what it
 * needs during the call and keeps no pointer.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* bad argument or inconsistent scene */
#define RT_ERR_DEVICE (-2)      /* a HIP call failed */
#define RT_ERR_UNSUPPORTED (-3) /* valid input this build cannot run */
#define RT_ERR_NO_SCENE (-4)    /* render/trace before rt_upload_scene */

/* ------------------------------------------------------------------------
 * Flattened scene.
 *
 * Nodes are the BvhNode tree of src/bvh.rs:97-103 in breadth-first order,
 * root at index 0.  An interior node has left >= 0 and right >= 0 (child
 * indices).  A leaf has left < 0 and refers to ~left:
 *   right  > 0 : prims[~left .. ~left+right) tested as a HitableList
 *                (src/hitable.rs:105-117); right == 1 is a bare primitive.
 *   right == 0 : objects[~left].
 * bmin/bmax are the node's Aabb (src/bvh.rs:9-12) rounded to fp32.
 * ---------------------------------------------------------------------- */
typedef struct rt_node {
  float bmin[3];
  int32_t left;
  float bmax[3];
  int32_t right;
} rt_node; /* 32 bytes */

/* Primitive classes.  Rect parameter order follows the constructors
 * XYRect::new(x0,x1,y0,y1,k) / XZRect::new(x0,x1,z0,z1,k) /
 * YZRect::new(y0,y1,z0,z1,k) (src/hitable.rs:151,204,278):
 * f = {a0, a1, b0, b1, k, 0}.  Sphere (src/hitable.rs:40): f = {cx,cy,cz,r,0,0}.
 * RT_PRIM_FLIP marks a rect wrapped in FlipNormals (src/hitable.rs:320-361). */
#define RT_PRIM_SPHERE 0
#define RT_PRIM_XY_RECT 1
#define RT_PRIM_XZ_RECT 2
#define RT_PRIM_YZ_RECT 3
#define RT_PRIM_TYPE_MASK 0xff
#define RT_PRIM_FLIP 0x100

typedef struct rt_prim {
  int32_t type;     /* RT_PRIM_* | optional RT_PRIM_FLIP */
  int32_t material; /* index into materials */
  float f[6];
} rt_prim; /* 32 bytes */

/* An object is what the reference nests as
 *   Translate::new(RotateY::new(inner, angle), offset)
 * (src/hitable.rs:433,465; either wrapper may be absent) around
 *   RT_OBJ_LIST  : prims[first .. first+count) as a HitableList (a Cuboid,
 *                  src/hitable.rs:380-424, is its 6 rects in source order);
 *   RT_OBJ_BVH   : the sub-tree rooted at nodes[first] (a BvhNode used as a
 *                  child hitable).  Its leaves may be objects again, to any
 *                  depth the traversal stack allows (64 entries; each level
 *                  costs four plus its tree's own), which is how every other
 *                  nesting the constructors allow is written:
 *                  RotateY(Translate(x)) is a rotate-only object around a
 *                  sub-tree of one leaf that holds the translate-only object
 *                  around x; a HitableList of such things is a sub-tree with
 *                  one leaf per item, in the list's order.  A node that stands
 *                  for no BvhNode of the source carries an infinite box (the
 *                  source makes no box test there);
 *   RT_OBJ_MEDIUM: ConstantMedium (src/hitable.rs:542-604, commented out in
 *                  the reference) whose boundary is prims[first..first+count)
 *                  under the same optional rotate/translate;
 *   RT_OBJ_MEDIUM_BVH: the same with the sub-tree rooted at nodes[first] as the
 *                  boundary (a BvhNode, or any nesting as for RT_OBJ_BVH).
 * A medium must be a leaf of the world tree itself (the kernels walk the media
 * after the tree, in the world's frame). */
#define RT_OBJ_LIST 0
#define RT_OBJ_BVH 1
#define RT_OBJ_MEDIUM 2
#define RT_OBJ_MEDIUM_BVH 3
#define RT_XFORM_ROTATE_Y 1
#define RT_XFORM_TRANSLATE 2

typedef struct rt_object {
  int32_t kind;  /* RT_OBJ_* */
  int32_t first; /* first prim, or root node for RT_OBJ_BVH */
  int32_t count; /* prim count (LIST / MEDIUM) */
  int32_t xform; /* RT_XFORM_* bits */
  float sin_theta, cos_theta; /* RotateY (src/hitable.rs:466-469) */
  float offset[3];            /* Translate */
  float density;              /* medium */
  int32_t material;           /* medium phase function (Isotropic) */
  int32_t medium_index;       /* ordinal among media: selects the RNG stream */
} rt_object; /* 48 bytes */

/* Material enum of src/material.rs:11-17 (live: Lambertian, DiffuseLight;
 * Metal :85-113, Dielectric :115-160, Isotropic :206-223 are commented out
 * there and are restated from that text). */
#define RT_MAT_LAMBERTIAN 0
#define RT_MAT_DIFFUSE_LIGHT 1
#define RT_MAT_METAL 2
#define RT_MAT_DIELECTRIC 3
#define RT_MAT_ISOTROPIC 4

typedef struct rt_material {
  int32_t type;    /* RT_MAT_* */
  int32_t texture; /* albedo / emit texture (Lambertian, DiffuseLight, Isotropic) */
  float albedo[3]; /* Metal */
  float fuzz;      /* Metal, already clamped to <= 1 (src/material.rs:96) */
  float ref_idx;   /* Dielectric */
  int32_t pad;
} rt_material; /* 32 bytes */

/* Texture enum of src/texture.rs:5-17. */
#define RT_TEX_CONSTANT 0
#define RT_TEX_CHECKER 1
#define RT_TEX_NOISE 2

typedef struct rt_texture {
  int32_t type;   /* RT_TEX_* */
  float color[3]; /* constant */
  int32_t odd;    /* checker: texture indices */
  int32_t even;
  float scale;    /* noise */
  int32_t perlin; /* noise: index into perlins */
} rt_texture; /* 32 bytes */

/* Perlin tables (src/perlin.rs:6-11).  The tables cross the boundary instead
 * of being regenerated so both sides evaluate identical noise. */
typedef struct rt_perlin {
  float ranvec[256][3];
  int32_t perm_x[256];
  int32_t perm_y[256];
  int32_t perm_z[256];
} rt_perlin;

/* Light shapes for HitablePDF (src/pdf.rs:33-50).  Only XZRect overrides
 * pdf_value/random in the reference (src/hitable.rs:245-264); color() builds
 * XZRect(213,343,227,332,554) inline (src/main.rs:177-178): one RT_LIGHT_XZ_RECT,
 * f = {x0, x1, z0, z1, k, 0}, reproduces that.  Zero lights means CosinePDF alone.
 * Beyond the reference (it has no code for them): XY and YZ rects with the same
 * formulas on their own axes (f = constructor order a0,a1,b0,b1,k); spheres
 * (f = {cx,cy,cz,r}) with the pdf_value/random of "Ray Tracing: The Rest of Your
 * Life" (uniform over the cone the sphere subtends); and up to RT_MAX_LIGHTS
 * shapes at once, as a HitableList whose random() picks one shape uniformly and
 * whose pdf_value() is the mean of the shapes'.  The MixturePDF stays 50/50
 * between that HitablePDF and the CosinePDF (src/pdf.rs:52-76). */
#define RT_LIGHT_XZ_RECT 0
#define RT_LIGHT_XY_RECT 1
#define RT_LIGHT_YZ_RECT 2
#define RT_LIGHT_SPHERE 3
#define RT_MAX_LIGHTS 64

typedef struct rt_light {
  int32_t type;
  int32_t pad;
  float f[6];
} rt_light; /* 32 bytes */

typedef struct rt_scene {
  const rt_node* nodes;
  int32_t n_nodes; /* node 0 is the root of the world BVH */
  const rt_prim* prims;
  int32_t n_prims;
  const rt_object* objects;
  int32_t n_objects;
  const rt_material* materials;
  int32_t n_materials;
  const rt_texture* textures;
  int32_t n_textures;
  const rt_perlin* perlins;
  int32_t n_perlins;
  const rt_light* lights;
  int32_t n_lights; /* 0 .. RT_MAX_LIGHTS */
} rt_scene;

/* Camera (src/camera.rs:5-14) in f64 exactly as Camera::new computes it;
 * the device rounds to fp32 on upload. */
typedef struct rt_camera {
  double origin[3];
  double lower_left_corner[3];
  double horizontal[3];
  double vertical[3];
  double u[3];
  double v[3];
  double w[3];
  double lens_radius;
} rt_camera;

/* Camera::new(lookfrom, lookat, vup, vfov_deg, aspect, aperture, focus_dist),
 * src/camera.rs:17-51. */
int rt_camera_init(rt_camera* out, const double lookfrom[3], const double lookat[3],
                   const double vup[3], double vfov_deg, double aspect, double aperture,
                   double focus_dist);

#define RT_SKY_BLACK 0    /* SKYBOX_COLOR = 0, src/main.rs:32,208 */
#define RT_SKY_GRADIENT 1 /* the commented-out gradient, src/main.rs:203-205 */

typedef struct rt_render_params {
  int32_t nx, ny;       /* image size; v is measured from the bottom row (src/main.rs:103) */
  int32_t spp;          /* samples per pixel rendered by THIS call */
  int32_t sample_begin; /* index of the first of them (sample-range sharding) */
  int32_t max_depth;    /* 50 in the reference (src/main.rs:173) */
  int32_t sky;          /* RT_SKY_* */
  float t_min;          /* 0.001 in the reference (src/main.rs:166) */
  int32_t shard_index;  /* pixel-tile sharding: this call renders only the 8x8-pixel */
  int32_t shard_count;  /* tiles t with t % shard_count == shard_index (see below)  */
  uint64_t seed;        /* Philox key */
} rt_render_params;

/* Pixel-tile sharding deals 8x8-pixel tiles (tile (0,0) is the image's top left
 * corner).  Tiles are numbered row by row over a grid rt_shard_width() tiles wide:
 * the smallest width >= ceil(nx/8) that has no common factor with shard_count, so
 * that the tiles t = shard_index (mod shard_count) of one shard never line up in
 * columns; numbers whose column lies beyond the image's edge stand for no tile.
 * Tile t covers pixels x in [8*(t % w), +8), y in [8*(t / w), +8) (y from the top). */
int rt_shard_width(int nx, int shard_count);

/* Random numbers.  The reference draws everything from rand::thread_rng()
 * (OS seeded, not replayable).  This ABI fixes a counter-based stream instead:
 * Philox4x32-10 with key (seed low word, seed high word) and counter
 *   (pixel, sample, depth, sub),   pixel = j*nx + i with j counted from the bottom row.
 * One call, sub 0, serves a whole path vertex at that depth:
 *   word 0  (depth 0 only) the pixel jitter of src/main.rs:102-103: its high
 *           16 bits / 2^16 is added to i, its low 16 bits / 2^16 to j;
 *   word 1  top bit: the MixturePDF coin (src/pdf.rs:70); bits 30..7 as a 24-bit
 *           uniform: the Dielectric reflect draw (src/material.rs:152);
 *   words 2,3  the two draws of whichever generator the coin picks
 *           (src/hitable.rs:259-261 or src/main.rs:241-242).
 * sub 1+a at depth 0, words 0,1: attempt a of random_in_unit_disk (src/camera.rs:65-75).
 * sub 256+a, words 0..2: attempt a of random_in_unit_sphere (src/main.rs:211-221).
 * sub 4096+m, word 0: the ConstantMedium draw of medium m (src/hitable.rs:582).
 * sub 8192, word 0 (only with more than one light shape): the shape HitablePDF
 *           generates from is number (word * n_lights) >> 32.
 * A 32-bit word x becomes a uniform in [0,1) as (x >> 8) * 2^-24. */

typedef struct rt_stats {
  uint64_t paths;          /* camera samples (src/main.rs:101-110) */
  uint64_t rays;           /* closest-hit queries (src/main.rs:166) */
  uint64_t hits;           /* of which found a surface (HitRecord built) */
  uint64_t node_tests;     /* Aabb::hit evaluations (src/bvh.rs:46-61) */
  uint64_t rect_tests;     /* XY/XZ/YZRect::hit evaluations */
  uint64_t sphere_tests;   /* Sphere::hit evaluations */
  uint64_t instance_ops;   /* rays moved into an object's frame */
  uint64_t shade_events;   /* scattering vertices (mixture sample + both pdf values) */
  uint64_t light_tests;    /* XZRect::pdf_value rect tests (src/hitable.rs:246) */
  double kernel_ms;        /* device time of the render kernel */
} rt_stats;

typedef struct rt_ctx rt_ctx;

const char* rt_last_error(void);

/* One context per (host thread, device).  Not thread-safe. */
int rt_create(int device_id, rt_ctx** out);
int rt_destroy(rt_ctx* ctx);

/* Context options.  No counterpart in the reference (its only knobs are constants in
 * the source, src/main.rs:70,173).  A value outside an option's range is clamped to
 * it; an unknown option is RT_ERR_INVALID.  Each option's default can also be set
 * from the environment variable named beside it, which rt_create reads ONCE (and
 * clamps the same way); no render call reads the environment.
 *   RT_OPT_FIXED_ORDER (RT_NO_ACCUMULATE; 0 or 1, default 0): scenes on the per-lane
 *     stack traversal normally add each work item's tile sums into the output with
 *     hardware float adds, in whatever order the items finish, so the last bits of a
 *     pixel's sum can differ from run to run.  With 1, every work item writes a
 *     partial image of its own, a wave has one item open at a time, and the partial
 *     images are added in a fixed order: equal inputs give equal bits on every run
 *     (at some cost in speed and up to 1 GiB of partial images).  Scenes of at most
 *     16 leaves and scenes on the skip-link scan always work that way.
 *   RT_OPT_READY_LANES (RT_READY_LANES; 1..64, default 0 = chosen per scene): finished
 *     rays in 64 live lanes that end a traversal phase of the stack kernel.
 *   RT_OPT_HOLD_LANES (RT_HOLD_LANES; 1..64, default 16): lanes holding a leaf, in 64
 *     traversing ones, that make the wave run the leaf code.
 *   RT_OPT_STACK_LDS_ENTRIES (RT_STACK_LDS_CAP; 1..64, default 10): entries of a lane's
 *     traversal stack kept in LDS; deeper ones live in global memory.
 *   RT_OPT_WAVES_PER_CU (RT_WAVES_PER_CU; 1..32, default 32): waves launched per CU.
 *   RT_OPT_ITEM_ROUNDS (RT_WANT_ROUNDS; 1..1000, default 30): work items the launch
 *     aims for, per wave slot of the chip.
 *   RT_OPT_MIN_CHUNK_SPP (RT_MIN_CHUNK_SPP; 1..65536, default 0 = chosen per launch):
 *     fewest samples of a tile that make one work item.
 *   RT_OPT_LAST_CHUNK_SPP (RT_STACK_LAST_SPP; 1..65536, default 4): the stack kernel's
 *     last sample chunk of a tile is halved down to parts of this many samples.
 *   RT_OPT_POOL_SLOTS (RT_POOL_SLOTS; 0 or 64..512, default 0): with a value from 64
 *     on, scenes on the per-lane stack traversal keep that many paths per wave in a
 *     pool (64-byte records in a region of global memory of the wave's own, queues of
 *     slot numbers in LDS) and run every phase of the sample loop (tree walk, shading,
 *     Texture::value) for a full wave of them; 0, the default, is the kernel that
 *     keeps a path in one lane from start to end, which is the faster one on every
 *     scene measured (the pool buys fuller waves with two more memory round trips
 *     per ray and fewer resident waves).
 *   RT_OPT_SHADE_BATCH (RT_SHADE_BATCH; 1..64, default 64), RT_OPT_TEX_BATCH
 *     (RT_TEX_BATCH; 1..64, default 24): paths waiting that make a wave of the pool
 *     kernel run its shading / texture code; RT_OPT_REFILL_LANES (RT_REFILL_LANES;
 *     1..64, default 16): finished walks that make it hand them over and take new rays;
 *     RT_OPT_HUNGRY_LANES (RT_HUNGRY_LANES; 1..64, default 8): idle lanes, with no ray
 *     waiting, that make it run a waiting batch of any size.
 *   RT_OPT_KEEP_TREE (RT_KEEP_TREE; 0 or 1, default 0; read by rt_upload_scene): with 1,
 *     scenes on the per-lane stack traversal walk the interior nodes AS GIVEN (the
 *     reference's own BvhNode::new tree, src/bvh.rs:105-147, or the one rt_build_lbvh
 *     made on the device), collapsed to four-wide entries, instead of the
 *     surface-area tree rt_upload_scene builds over the leaves by default: a cheaper
 *     upload, slower walks.  Nearest hits are the same either way.
 * None of these changes which paths are traced or what they return: a pixel's sum
 * changes at most in the order of its fp32 additions. */
#define RT_OPT_FIXED_ORDER 1
#define RT_OPT_READY_LANES 2
#define RT_OPT_HOLD_LANES 3
#define RT_OPT_STACK_LDS_ENTRIES 4
#define RT_OPT_WAVES_PER_CU 5
#define RT_OPT_ITEM_ROUNDS 6
#define RT_OPT_MIN_CHUNK_SPP 7
#define RT_OPT_LAST_CHUNK_SPP 8
#define RT_OPT_POOL_SLOTS 9
#define RT_OPT_SHADE_BATCH 10
#define RT_OPT_REFILL_LANES 11
#define RT_OPT_TEX_BATCH 12
#define RT_OPT_HUNGRY_LANES 13
#define RT_OPT_KEEP_TREE 14
int rt_set_option(rt_ctx* ctx, int option, int value);
int rt_get_option(const rt_ctx* ctx, int option, int* value_out);

/* Validates and copies the scene to the device (replaces any previous one).
 * upload_ms_or_null of rt_upload_scene_timed receives the wall time of the call's two
 * parts: [0] building the device's own views of the tree on the host (the SAH /
 * four-wide interior over the given leaves, the scan array, per-primitive records),
 * [1] allocation and copies to the device. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene* scene);
int rt_upload_scene_timed(rt_ctx* ctx, const rt_scene* scene, double* upload_ms_or_null);

/* Replaces main.rs:67-133.  Writes per-pixel SUMS (not means) of de_nan'ed
 * sample radiance, nx*ny*3 floats, top row first, x ascending (main.rs:96-97,
 * 114).  out_sumsq (optional) receives per-channel sums of squared samples.
 * stats (optional) selects the event-counting build of the kernel.  Blocking. */
int rt_render(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params,
              float* out_sum_rgb, float* out_sumsq_rgb_or_null, rt_stats* stats_or_null);

/* Same computation with device-resident outputs, enqueued on `hip_stream`
 * (a hipStream_t passed as void*; NULL is the default stream).  The buffers are
 * overwritten, not accumulated.  Returns after enqueueing.  All launches of one
 * context share its work counter and partial-image buffer, so they run one after
 * the other: a launch on another stream than the context's previous launch first
 * waits (on the device) for that launch to end.  To overlap renders, use one
 * context per stream.  A launch that needs a larger partial-image buffer than any
 * before it on this context allocates it during the call (which synchronises).
 * Scenes of at most 16 leaves and scenes on the skip-link scan give the same bits
 * on every run.  Scenes on the per-lane stack traversal (larger trees) have no
 * partial images: their work items add tile sums into the zeroed output with
 * hardware float adds, so the last bits of a sum can differ from run to run
 * (RT_OPT_FIXED_ORDER gives fixed bits).  Those adds are only dependable on ordinary
 * device allocations, so the call asks the runtime what kind of memory each output
 * buffer is: anything but plain hipMalloc memory (host-mapped, managed, unknown to
 * the runtime) is written the RT_OPT_FIXED_ORDER way, with plain stores.  Device
 * memory allocated fine-grained (hipExtMallocWithFlags) reports as plain device
 * memory: set RT_OPT_FIXED_ORDER for such buffers. */
int rt_render_device(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params,
                     float* d_sum_rgb, float* d_sumsq_rgb_or_null, void* hip_stream);

/* No counterpart in the reference.  rt_render_device for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): the same kernels and the same results, bit
 * for bit, but the work counter and the partial images of a chunked launch live
 * in a static 64 MiB device array instead of buffers the context allocates, and
 * the call itself allocates and frees nothing.  Limit: the partial images of the
 * launch (number of sample chunks x nx*ny*3 floats, twice that with d_sumsq; the
 * library splits a tile's samples into chunks of at least 32 until the launch has
 * about 184,000 work items) must fit in 64 MiB - 256 B, otherwise the call
 * returns RT_ERR_UNSUPPORTED and the text says how many bytes it needed:
 * 256x256 at 1000 spp fits (34 partial images, 27 MB) and so does 1024x1024 at
 * 100 spp (5 partial images, 63 MB); 1024x1024 at 1000 spp does not (15 partial
 * images, 189 MB); any image at fewer than 64 spp does (one chunk, no partial
 * images), and so does any launch of a scene on the per-lane stack traversal (it
 * has no partial images; see rt_render_device for what that means for its bits;
 * the deep ends of its lanes' traversal stacks live in the scratch too, and what
 * of them does not fit stays in LDS, which costs speed, not the launch).
 * Every call on a GPU, from any context, uses that GPU's one array: calls that
 * overlap on one GPU race (calls of one context are ordered as for
 * rt_render_device; calls of different contexts on one device are the caller's
 * to order). */
int rt_render_device_noalloc(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params,
                             float* d_sum_rgb, float* d_sumsq_rgb_or_null, void* hip_stream);

/* ------------------------------------------------------------------------
 * Several GPUs of one node from ONE process: what main.rs:70-133 does with its
 * 8 threads (each renders its share, the shares are combined into one image),
 * with GPUs for threads.  device_ids lists one device per rank; an id may
 * repeat, which gives ranks that share a GPU (the way to check a split on a
 * machine with one GPU).  The scene is replicated on every rank.
 *   RT_SHARD_SAMPLES  rank g renders samples [g*spp/N, (g+1)*spp/N) of every pixel
 *                     (main.rs:74-78) and the frames are summed onto rank 0
 *                     (main.rs:128-133): nx*ny*12 bytes per rank.
 *   RT_SHARD_TILES    rank g renders the 8x8-pixel tiles t with t % N == g (numbered
 *                     as for rt_render_params.shard_index) at the full sample
 *                     count; each rank sends only its own tiles and rank 0 puts
 *                     them in place: nx*ny*12/N bytes per rank, no arithmetic.
 * With more than one rank and distinct device ids the exchange is RCCL on the
 * ranks' streams (one communicator per rank from ncclCommInitAll; ncclReduce, or
 * grouped ncclSend/ncclRecv); librccl.so is opened when such a context is created,
 * and if it cannot be loaded or initialised the context falls back to the copy
 * path (rt_multi_uses_rccl() then returns 0).  With repeated ids, one rank, or
 * RT_MULTI_NO_RCCL, rank 0's stream copies the other ranks' buffers over and adds
 * or places them in rank order.  The RCCL exchange between DIFFERENT devices has
 * not run on hardware yet (no multi-GPU machine was available to this build); the
 * copy path and the one-rank RCCL calls have.  Results equal a
 * one-rank render up to the order of fp32 additions (tile mode: up to how each
 * launch splits a tile's samples into partial sums), because every (pixel,
 * sample) has its own random-number counter.  Not thread-safe. */
#define RT_SHARD_SAMPLES 0
#define RT_SHARD_TILES 1
#define RT_MULTI_NO_RCCL 1u
#define RT_MULTI_FORCE_RCCL 2u /* RCCL even for a single rank (the calls then run on one GPU: a reduce onto itself,
                                * an empty send/receive group), and an error instead of the fallback when it cannot be used */
typedef struct rt_multi rt_multi;
int rt_multi_create(const int* device_ids, int n_devices, unsigned flags, rt_multi** out);
int rt_multi_destroy(rt_multi* m);
int rt_multi_uses_rccl(const rt_multi* m); /* 1 when the exchange goes through RCCL */
int rt_multi_upload_scene(rt_multi* m, const rt_scene* scene);
/* rt_render over the ranks.  params describes the whole image (shard_count 1);
 * stats_per_rank (optional, n_devices entries) receives each rank's event counts
 * and the device time of its render kernel.  Blocking. */
int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_render_params* params, int shard_mode,
                    float* out_sum_rgb, float* out_sumsq_rgb_or_null, rt_stats* stats_per_rank_or_null);

/* Nearest-hit query on a batch of rays: the world.hit(r, t_min, t_max) of
 * src/main.rs:166 without any random numbers (media are skipped).  o and d are
 * n*3 floats.  prim_id[i] is the index into prims of the primitive hit, or -1;
 * t[i] its ray parameter.  normal and p (n*3 floats each) are optional. */
int rt_trace_closest(rt_ctx* ctx, const float* o, const float* d, size_t n, float t_min,
                     float t_max, int32_t* prim_id, float* t, float* normal_or_null,
                     float* p_or_null);

/* The same query in the reference's own arithmetic, as a check on the fp32 path
 * ("parity mode"): f64 throughout, a division where the source divides
 * (src/bvh.rs:49, src/hitable.rs:59-61,165), no fused multiply-adds, and the
 * source's traversal: every node of the tree as passed in gets its Aabb::hit, both
 * children are visited with the same t_max and the right one wins an exact tie
 * (src/bvh.rs:197-228).  None of the fp32 path's own steps applies.  Scene data are
 * the uploaded fp32 arrays widened to f64 (what a Rust caller that flattens its f64
 * scene into rt_scene has handed over).  Media are skipped.  One ray per lane, no
 * pruning: it is a checker, orders of magnitude slower than rt_trace_closest. */
int rt_trace_closest_f64(rt_ctx* ctx, const double* o, const double* d, size_t n, double t_min,
                         double t_max, int32_t* prim_id, double* t, double* normal_or_null,
                         double* p_or_null);

/* main.rs:136-155: divide the sums by spp_total, sqrt, clamp high to 1,
 * (255.99*c) truncated to u8.  sum_rgb and rgb8 are host buffers. */
int rt_tonemap(rt_ctx* ctx, const float* sum_rgb, int32_t nx, int32_t ny, int32_t spp_total,
               uint8_t* rgb8);

/* The same on device-resident buffers (the sums rt_render_device wrote; nx*ny*3
 * bytes out), enqueued on `hip_stream`: render, tonemap and one copy of the 8-bit
 * image to the host are then the whole of main.rs:67-155. */
int rt_tonemap_device(rt_ctx* ctx, const float* d_sum_rgb, int32_t nx, int32_t ny, int32_t spp_total,
                      uint8_t* d_rgb8, void* hip_stream);

/* An alternative to BvhNode::new (src/bvh.rs:105-147) for the world tree, built on
 * the device: a linear BVH (Morton order of the leaf-box centres, hierarchy from the
 * sorted keys, boxes fitted bottom-up) over the same leaves.  nodes_out receives
 * scene->n_nodes nodes: the scene's own array with the interior nodes of the world
 * tree replaced (leaves, their boxes and the sub-trees inside objects are kept, so
 * every other array of the scene still applies).  Use it by uploading a copy of the
 * scene whose `nodes` points at nodes_out.  The tree is not the reference's, so box
 * test counts and the winner of an exact tie differ: a separately reported variant,
 * never the parity configuration. */
int rt_build_lbvh(rt_ctx* ctx, const rt_scene* scene, rt_node* nodes_out);

/* Device self-test with no counterpart in the reference: the kernels replace
 * the three divisions 1/d of Aabb::hit (src/bvh.rs:47) by a short reciprocal
 * sequence that must equal the correctly rounded quotient.  This runs the
 * sequence against the division for all 2^32 float bit patterns on the device
 * and returns the number of inputs where they differ (0 is the only pass) and up
 * to 16 example bit patterns (optional). */
int rt_selftest_reciprocal(rt_ctx* ctx, uint64_t* mismatches, uint32_t* examples16_or_null);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
