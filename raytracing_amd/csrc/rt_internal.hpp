// Internal helpers shared by the host-side sources of librt_amd.so.
#ifndef RT_INTERNAL_HPP
#define RT_INTERNAL_HPP

#include <string>

struct rt_ctx;

namespace rt {
// Records the text rt_last_error() returns for the calling thread; returns code.
int set_error(int code, const std::string& text);
// The HIP device a context was created on.
int context_device(const struct ::rt_ctx* ctx);
// What rt_render_device does, with the choice of the counting build of the kernel (`stats`): for the entry points
// that drive several contexts.  `stream` is a hipStream_t.
int enqueue_render(struct ::rt_ctx* ctx, const struct rt_camera* cam, const struct rt_render_params* params, float* d_sum,
                   float* d_sumsq_or_null, bool stats, void* stream);
// Zeroes / reads the event counters of the context's counting build (the device must be idle for the read).
int reset_counters(struct ::rt_ctx* ctx, void* hip_stream);
int read_counters(struct ::rt_ctx* ctx, struct rt_stats* out);
}  // namespace rt

#endif
