/*
 * rt_scene_api.h -- C handles over the host scene-construction classes.
 *
 * The reference builds scenes in Rust (src/scenes.rs) out of constructors such
 * as Sphere::new / XZRect::new / Translate::new.  A foreign host reaches the
 * same constructors through these functions; each one names the constructor it
 * stands for.  Textures, materials and hitables are small non-negative integer
 * handles local to one builder.  Functions return a handle (>= 0) or a negative
 * RT_ERR_* code; rt_last_error() has the text.
 */
#ifndef RT_SCENE_API_H
#define RT_SCENE_API_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_builder rt_builder;

/* `seed` drives the host generator that replaces rand::thread_rng() during
 * construction: BVH split axes (src/bvh.rs:107) and Perlin gradients
 * (src/perlin.rs:67-79). */
int rtb_create(uint64_t seed, rt_builder** out);
int rtb_destroy(rt_builder* b);

/* src/texture.rs:36-52 */
int rtb_texture_constant(rt_builder* b, double r, double g, double bl);
int rtb_texture_checker(rt_builder* b, int odd, int even);
int rtb_texture_noise(rt_builder* b, double scale);

/* src/material.rs:57, 93, 121, 189, 213 */
int rtb_lambertian(rt_builder* b, int texture);
int rtb_metal(rt_builder* b, double r, double g, double bl, double fuzz);
int rtb_dielectric(rt_builder* b, double ref_idx);
int rtb_diffuse_light(rt_builder* b, int texture);
int rtb_isotropic(rt_builder* b, int texture);

/* src/hitable.rs:40, 151, 204, 278 */
int rtb_sphere(rt_builder* b, const double center[3], double radius, int material);
int rtb_xy_rect(rt_builder* b, double x0, double x1, double y0, double y1, double k, int material);
int rtb_xz_rect(rt_builder* b, double x0, double x1, double z0, double z1, double k, int material);
int rtb_yz_rect(rt_builder* b, double y0, double y1, double z0, double z1, double k, int material);
/* src/hitable.rs:330-338 (new_xy / new_xz / new_yz: the handle must be a rect) */
int rtb_flip_normals(rt_builder* b, int rect);
/* src/hitable.rs:380, 433, 465, 551, 100; src/bvh.rs:106 */
int rtb_cuboid(rt_builder* b, const double p0[3], const double p1[3], int material);
int rtb_translate(rt_builder* b, int hitable, const double offset[3]);
int rtb_rotate_y(rt_builder* b, int hitable, double angle_degrees);
int rtb_constant_medium(rt_builder* b, int boundary, double density, int texture);
int rtb_hitable_list(rt_builder* b, const int* hitables, int n);
int rtb_bvh_node(rt_builder* b, const int* hitables, int n);

/* The light shape color() hands to HitablePDF (src/main.rs:177-178): an XZRect. */
int rtb_light_xz_rect(rt_builder* b, double x0, double x1, double z0, double z1, double k);
/* The same, read off the scene instead of typed in again: the first item of `world_list` that is an XZRect
 * (bare or under FlipNormals) with a DiffuseLight material becomes the light shape, replacing any set before.
 * Returns how many such items the list holds (0: nothing changed).  For the Cornell box this registers exactly
 * the numbers src/main.rs:178 hard-codes, which are those of src/scenes.rs:310-317. */
int rtb_light_from_world(rt_builder* b, int world_list);
/* Beyond the reference (whose color() samples one XZRect): APPENDS a shape to the list HitablePDF samples.
 * type is RT_LIGHT_*; rects take a0,a1,b0,b1,k in their constructor's order, a sphere cx,cy,cz,r (e unused).
 * Returns the number of shapes now registered. */
int rtb_light_add(rt_builder* b, int type, double a, double bb, double c, double d, double e);
/* Registers EVERY item of `world_list` that can be sampled and carries a DiffuseLight material: XY, XZ and YZ
 * rects (bare or under FlipNormals) and spheres, in list order, replacing any set before (at most RT_MAX_LIGHTS;
 * further ones are left out).  Returns how many were registered.  random_scene's emissive spheres
 * (src/scenes.rs:87-92 and the small ones) become light shapes this way. */
int rtb_lights_from_world(rt_builder* b, int world_list);

/* Builds the world BvhNode over the items of `world_list` (src/main.rs:85-90)
 * and flattens.  *out points into memory owned by the builder, valid until the
 * next rtb_flatten or rtb_destroy. */
int rtb_flatten(rt_builder* b, int world_list, rt_scene* out);

#ifdef __cplusplus
}
#endif
#endif
