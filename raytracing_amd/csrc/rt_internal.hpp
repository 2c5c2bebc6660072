// Internal helpers shared by the host-side sources of librt_amd.so.
#ifndef RT_INTERNAL_HPP
#define RT_INTERNAL_HPP

#include <string>

namespace rt {
// Records the text rt_last_error() returns for the calling thread; returns code.
int set_error(int code, const std::string& text);
}  // namespace rt

#endif
