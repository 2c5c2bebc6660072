// cornell_main.cpp -- the reference's main() (src/main.rs:34-163) over librt_amd.so.
//
// The same program flow with the host classes of rt_scene.hpp: build
// cornell_box_scene (src/scenes.rs:276-360), render, tonemap (main.rs:136-155),
// write an 8-bit RGB PNG (main.rs:157-160).  The 8-thread sample loop of
// main.rs:67-133 is the one rt_render call.
//
//   cornell_main [factor [quality [out.png [seed]]]]
//
// factor and quality are the reference's two arguments (main.rs:47-62): the image
// is 100*factor pixels square at `quality` samples per pixel.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_scene.hpp"

namespace {

using rt::Vec3;

// src/scenes.rs:276-360, value for value.
rt::HitableList cornell_box_scene(rt::Tables& t, double ratio, rt_camera* camera) {
  const rt::Material red = rt::Lambertian::new_(t, rt::Texture::new_constant(t, Vec3(0.65, 0.05, 0.05)));
  const rt::Material white = rt::Lambertian::new_(t, rt::Texture::new_constant(t, Vec3(0.73, 0.73, 0.73)));
  const rt::Material green = rt::Lambertian::new_(t, rt::Texture::new_constant(t, Vec3(0.12, 0.45, 0.15)));
  const rt::Material light = rt::DiffuseLight::new_(t, rt::Texture::new_constant(t, rt::scalar(15.0)));
  std::vector<rt::HitablePtr> list;
  list.push_back(std::make_shared<rt::FlipNormals>(rt::YZRect(0.0, 555.0, 0.0, 565.0, 555.0, green)));  // left wall
  list.push_back(rt::YZRect(0.0, 555.0, 0.0, 555.0, 0.0, red));                                          // right wall
  list.push_back(std::make_shared<rt::FlipNormals>(rt::XZRect(213.0, 343.0, 227.0, 332.0, 554.0, light)));
  list.push_back(std::make_shared<rt::FlipNormals>(rt::XZRect(0.0, 555.0, 0.0, 555.0, 555.0, white)));  // ceiling
  list.push_back(rt::XZRect(0.0, 555.0, 0.0, 555.0, 0.0, white));                                        // floor
  list.push_back(std::make_shared<rt::FlipNormals>(rt::XYRect(0.0, 555.0, 0.0, 555.0, 555.0, white)));  // back wall
  list.push_back(std::make_shared<rt::Translate>(
      std::make_shared<rt::RotateY>(std::make_shared<rt::Cuboid>(rt::scalar(0.0), rt::scalar(165.0), white), -18.0), Vec3(130.0, 0.0, 65.0)));
  list.push_back(std::make_shared<rt::Translate>(
      std::make_shared<rt::RotateY>(std::make_shared<rt::Cuboid>(rt::scalar(0.0), Vec3(165.0, 330.0, 165.0), white), 15.0),
      Vec3(265.0, 0.0, 295.0)));
  const double lookfrom[3] = {278.0, 278.0, -800.0}, lookat[3] = {278.0, 278.0, 0.0}, vup[3] = {0.0, 1.0, 0.0};
  rt_camera_init(camera, lookfrom, lookat, vup, 40.0, ratio, 0.0, 10.0);
  return rt::HitableList(list);
}

void put_u32(std::vector<unsigned char>* out, unsigned v) {
  for (int s = 24; s >= 0; s -= 8) out->push_back((unsigned char)(v >> s));
}
void put_chunk(std::vector<unsigned char>* out, const char tag[4], const std::vector<unsigned char>& payload) {
  put_u32(out, (unsigned)payload.size());
  std::vector<unsigned char> body(tag, tag + 4);
  body.insert(body.end(), payload.begin(), payload.end());
  out->insert(out->end(), body.begin(), body.end());
  put_u32(out, (unsigned)crc32(0L, body.data(), (uInt)body.size()));
}

// image::save_buffer(.., RGB(8)) of main.rs:160: non-interlaced truecolour, filter 0 on every row.
bool write_png(const std::string& path, const std::vector<unsigned char>& rgb, int nx, int ny) {
  std::vector<unsigned char> rows;
  rows.reserve((size_t)ny * ((size_t)nx * 3 + 1));
  for (int y = 0; y < ny; ++y) {
    rows.push_back(0);
    rows.insert(rows.end(), rgb.begin() + (size_t)y * nx * 3, rgb.begin() + (size_t)(y + 1) * nx * 3);
  }
  uLongf bound = compressBound((uLong)rows.size());
  std::vector<unsigned char> packed(bound);
  if (compress2(packed.data(), &bound, rows.data(), (uLong)rows.size(), 6) != Z_OK) return false;
  packed.resize(bound);
  std::vector<unsigned char> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  std::vector<unsigned char> ihdr;
  put_u32(&ihdr, (unsigned)nx);
  put_u32(&ihdr, (unsigned)ny);
  const unsigned char tail[5] = {8, 2, 0, 0, 0};
  ihdr.insert(ihdr.end(), tail, tail + 5);
  put_chunk(&file, "IHDR", ihdr);
  put_chunk(&file, "IDAT", packed);
  put_chunk(&file, "IEND", std::vector<unsigned char>());
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
  return std::fclose(f) == 0 && ok;
}

int fail(const char* what) {
  std::fprintf(stderr, "cornell_main: %s: %s\n", what, rt_last_error());
  return 1;
}

}  // namespace

int main(int argc, char** argv) {
  const long factor = argc > 1 ? std::atol(argv[1]) : 1;     // main.rs:48-52
  const long quality = argc > 2 ? std::atol(argv[2]) : 100;  // main.rs:54-58
  const std::string path = argc > 3 ? argv[3] : "cornell_box.png";
  const unsigned long long seed = argc > 4 ? std::strtoull(argv[4], nullptr, 0) : 0x5EED0001ull;
  // The reference spreads the samples over 8 threads (main.rs:70); here over `gpus` GPUs of the node (devices 0..gpus-1).
  // A negative count gives that many ranks that all share GPU 0; "tiles" splits the image by 8x8 tiles instead of samples.
  const long gpus = argc > 5 ? std::atol(argv[5]) : 1;
  const bool by_tiles = argc > 6 && std::strcmp(argv[6], "tiles") == 0;
  if (factor <= 0 || quality <= 0 || gpus == 0 || gpus > 64 || gpus < -64) {
    std::fprintf(stderr, "usage: cornell_main [factor [quality [out.png [seed [gpus [samples|tiles]]]]]]\n");
    return 2;
  }
  const int nx = (int)(100 * factor), ny = (int)(100 * factor), ns = (int)quality;

  rt::Tables tables;
  rt::HostRng rng(seed);
  rt_camera camera;
  rt::FlatScene flat;
  try {
    const rt::HitableList world = cornell_box_scene(tables, (double)nx / (double)ny, &camera);
    // the light shape color() builds for its HitablePDF, main.rs:177-178
    rt_light shape;
    std::memset(&shape, 0, sizeof(shape));
    shape.type = RT_LIGHT_XZ_RECT;
    shape.f[0] = 213.0f; shape.f[1] = 343.0f; shape.f[2] = 227.0f; shape.f[3] = 332.0f; shape.f[4] = 554.0f;
    tables.lights.push_back(shape);
    rt::flatten_world(world, tables, rng, &flat);  // BvhNode::new over world.list, main.rs:85-90
  } catch (const rt::SceneError& e) {
    std::fprintf(stderr, "cornell_main: %s\n", e.what());
    return 1;
  }
  const rt_scene scene = flat.view();

  rt_render_params params;
  std::memset(&params, 0, sizeof(params));
  params.nx = nx; params.ny = ny; params.spp = ns; params.max_depth = 50; params.sky = RT_SKY_BLACK; params.t_min = 0.001f;
  params.shard_count = 1; params.seed = seed;
  std::vector<float> sums((size_t)nx * ny * 3);
  rt_stats stats;
  std::memset(&stats, 0, sizeof(stats));
  rt_ctx* ctx = nullptr;
  if (rt_create(0, &ctx) != RT_OK) return fail("rt_create");
  if (gpus == 1) {
    if (rt_upload_scene(ctx, &scene) != RT_OK) return fail("rt_upload_scene");
    if (rt_render(ctx, &camera, &params, sums.data(), nullptr, &stats) != RT_OK) return fail("rt_render");
  } else {
    const int n = (int)(gpus < 0 ? -gpus : gpus);
    std::vector<int> ids((size_t)n, 0);
    if (gpus > 0)
      for (int g = 0; g < n; ++g) ids[(size_t)g] = g;
    rt_multi* multi = nullptr;
    if (rt_multi_create(ids.data(), n, 0, &multi) != RT_OK) return fail("rt_multi_create");
    if (rt_multi_upload_scene(multi, &scene) != RT_OK) return fail("rt_multi_upload_scene");
    std::vector<rt_stats> per_rank((size_t)n);
    if (rt_multi_render(multi, &camera, &params, by_tiles ? RT_SHARD_TILES : RT_SHARD_SAMPLES, sums.data(), nullptr, per_rank.data()) != RT_OK)
      return fail("rt_multi_render");
    for (const rt_stats& r : per_rank) {  // the job: counts add up, the time is the slowest rank's
      stats.paths += r.paths;
      stats.rays += r.rays;
      if (r.kernel_ms > stats.kernel_ms) stats.kernel_ms = r.kernel_ms;
    }
    rt_multi_destroy(multi);
  }
  std::vector<unsigned char> rgb((size_t)nx * ny * 3);
  if (rt_tonemap(ctx, sums.data(), nx, ny, ns, rgb.data()) != RT_OK) return fail("rt_tonemap");
  rt_destroy(ctx);
  if (!write_png(path, rgb, nx, ny)) {
    std::fprintf(stderr, "cornell_main: cannot write %s\n", path.c_str());
    return 1;
  }
  std::printf("%s: %dx%d, %d samples per pixel, %llu paths, %llu rays, %.3f ms on the device\n", path.c_str(), nx, ny, ns,
              (unsigned long long)stats.paths, (unsigned long long)stats.rays, stats.kernel_ms);
  return 0;
}
