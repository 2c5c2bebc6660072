/*
 * rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C interface of the CPU restatement of the reference's path-tracing loop.
 * Only tests, the smoke check and the benchmark's cpu_baseline leg may load
 * this library; the product library (librt_amd.so) never links or calls it.
 *
 * Parity status: the reference (a Rust crate) ships no tests, golden vectors
 * or fixtures, and cannot be compiled here (no Rust toolchain, crates not
 * vendored).  The only artefact of the reference that constrains this oracle
 * is examples/cornell_box.png (600x600, unknown sample count), which the
 * test-suite compares against block-wise.  Everything finer than that is
 * PARITY UNPINNED by the reference: it is pinned only by this restatement.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic modes.
 * ORC_F64      : IEEE f64, operation order of the Rust source (the reference).
 * ORC_F32      : the same operation order in fp32.
 * ORC_F32_FUSED: fp32 with the device's documented deviations in the
 *                intersection routines: reciprocal ray direction instead of a
 *                division per rect, and single-rounding multiply-adds where
 *                the device uses them.  Exists so that device nearest-hit
 *                results can be checked bit for bit. */
#define ORC_F64 0
#define ORC_F32 1
#define ORC_F32_FUSED 2

/* Traversal for orc_trace_closest. */
#define ORC_TRAVERSE_BVH 0  /* BvhNode::hit, src/bvh.rs:197-228 */
#define ORC_TRAVERSE_LIST 1 /* every leaf as one HitableList, src/hitable.rs:105-117 */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

int orc_trace_closest(const rt_scene* scene, int mode, int traversal, const float* o,
                      const float* d, size_t n, double t_min, double t_max, int32_t* prim_id,
                      double* t, double* normal_or_null, double* p_or_null);

/* split_samples != 0 renders the way src/main.rs:70-133 does: every thread
 * renders the whole image for its share of the samples and the images are
 * summed.  Otherwise threads split the rows.  Outputs are per-pixel sums and
 * sums of squares over the samples, nx*ny*3 doubles, top row first. */
int orc_render(const rt_scene* scene, const rt_camera* cam, const rt_render_params* params,
               int mode, int n_threads, int split_samples, double* out_sum,
               double* out_sumsq_or_null, rt_stats* stats_or_null);

/* src/main.rs:136-155 on per-pixel sums. */
void orc_tonemap(const double* sum_rgb, int32_t nx, int32_t ny, int32_t spp_total, uint8_t* rgb8);

/* Component hooks for unit tests. */
double orc_perlin_noise(const rt_perlin* perlin, const double p[3], int mode);
double orc_perlin_turb(const rt_perlin* perlin, const double p[3], int depth, int mode);
void orc_texture_value(const rt_scene* scene, int texture, double u, double v, const double p[3],
                       int mode, double out_rgb[3]);
void orc_camera_ray(const rt_camera* cam, double s, double t, const rt_render_params* params,
                    int pixel, int sample, int mode, double out_o[3], double out_d[3]);
void orc_random_cosine_direction(double r1, double r2, double out[3]);
void orc_onb_local(const double n[3], const double a[3], double out[3]);
/* Hitable::pdf_value(o, dir) and Hitable::random(o) of one light shape (f64), for n directions / n pairs of uniforms. */
void orc_light_pdf_value(const rt_light* light, const double o[3], const double* dirs, size_t n, double* out);
void orc_light_random(const rt_light* light, const double o[3], const double* xi, size_t n, double* out_dirs);

#ifdef __cplusplus
}
#endif
#endif
