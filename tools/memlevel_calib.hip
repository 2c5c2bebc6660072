// Calibration for tools/pmc_memlevel.sh: random 128-byte-line reads (one 16-byte load per lane, every lane of a wave
// in a line of its own) from a table of a chosen size, with as many waves per CU as the stack render kernel has.
// A 64 MiB table sits in the 256 MiB Infinity Cache after the first launch; an 8 GiB table is served by HBM.
// The time a fabric read is outstanding (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) under each gives the two figures the
// render kernel's own is compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int MIB>
__global__ void __launch_bounds__(64) gather_lines(const float4* __restrict__ table, unsigned long long n_lines, int rounds, float* out) {
  unsigned long long s = ((unsigned long long)blockIdx.x * 64 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // four independent loads in flight per lane, as a walk of a four-wide tree has few
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned long long line = (s >> 20) % n_lines;
      v[k] = table[line * 8 + (threadIdx.x & 7)];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

// One read at a time per lane, each address made from the value the last read returned (the table is all zeros):
// with two waves per CU this is the latency of a read with nothing queued in front of it.
template <int MIB>
__global__ void __launch_bounds__(64) chase_lines(const float4* __restrict__ table, unsigned long long n_lines, int rounds, float* out) {
  unsigned long long s = ((unsigned long long)blockIdx.x * 64 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  float acc = 0.0f;
  for (int r = 0; r < rounds * 4; ++r) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const float4 v = table[((s >> 20) % n_lines) * 8 + (threadIdx.x & 7)];
    s += __float_as_uint(v.y);
    acc += v.x;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MIB>
static void run(int waves, int rounds, bool chase = false) {
  const size_t bytes = (size_t)MIB << 20;
  float4* t = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc((void**)&t, bytes));
  CHECK(hipMalloc((void**)&out, 256));
  CHECK(hipMemset(t, 0, bytes));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0.0f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a, 0));
    if (chase) hipLaunchKernelGGL(chase_lines<MIB>, dim3(waves), dim3(64), 0, 0, t, (unsigned long long)(bytes / 128), rounds, out);
    else hipLaunchKernelGGL(gather_lines<MIB>, dim3(waves), dim3(64), 0, 0, t, (unsigned long long)(bytes / 128), rounds, out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double lines = (double)waves * 64 * rounds * 4;
  std::printf("%s table %5d MiB: %.3f ms, %.3g line reads, %.0f GB/s of 128-byte lines\n", chase ? "one read at a time, 2 waves per CU:" : "4 reads in flight, 20 waves per CU:", MIB, ms, lines, lines * 128 / (ms * 1e6));
  CHECK(hipFree(t)); CHECK(hipFree(out));
}

int main() {
  const int waves = 256 * 20, rounds = 128;
  run<64>(waves, rounds);
  run<192>(waves, rounds);
  run<1024>(waves, rounds);
  run<8192>(waves, rounds);
  // the 64 MiB table is warmed by a loaded pass first: 512 waves x 64 x 2048 reads cover it 128 times over anyway
  run<64>(256 * 2, 512, true);
  run<192>(256 * 2, 512, true);
  run<8192>(256 * 2, 512, true);
  return 0;
}
