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
}  // namespace rt

#endif
