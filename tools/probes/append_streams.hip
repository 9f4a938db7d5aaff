// append_streams.hip -- what do small-granule appends to many streams cost on MI355X?
//
// Question behind it (DESIGN.md, config C5 backward): a one-pass partition of the gradient records
// into the 2048 slices of a level leaves a tile of 1024 points with ~4 records (80 bytes) per slice
// and level.  If those 80-byte pieces are appended to per-slice streams -- neighbouring pieces of a
// stream come from different workgroups, at different times -- do they reach HBM as whole lines
// (the XCD's L2 merges them) or as partial writes?  The split pass exists only because the answer
// was assumed to be "partial".
//
// Every tile writes G bytes to each of D streams at its own offset (no atomics, no binning: the
// memory system alone).  Variants: streams shared by all workgroups (tile t and t+1 write
// neighbouring pieces from different XCDs) or one stream per (destination, blockIdx % 8) so that a
// stream is only written through one XCD's L2 (round-robin placement, a speed assumption only).
//
//   hipcc --offload-arch=gfx950 -O3 -o append_streams append_streams.hip && ./append_streams
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
      return 1;                                                 \
    }                                                           \
  } while (0)

// G = bytes per (tile, stream), multiple of 16.  PER_XCD: stream = (dest, blockIdx % 8).
// Thread i of a tile's sweep writes the 16-byte piece (i % (G/16)) of stream (i / (G/16)).
template <int G, bool PER_XCD>
__global__ __launch_bounds__(1024) void append_kernel(
  uint4 * __restrict__ dst, int D, int64_t stream_bytes, int n_tiles)
{
  constexpr int P = G / 16;
  const int group = blockIdx.x & 7;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int64_t off = PER_XCD ? (int64_t)(t >> 3) * G : (int64_t)t * G;
    for (int i = threadIdx.x; i < D * P; i += 1024) {
      const int s = i / P, p = i - s * P;
      const int64_t stream = PER_XCD ? (int64_t)s * 8 + group : (int64_t)s;
      uint4 v = make_uint4((uint32_t)t, (uint32_t)s, (uint32_t)p, 0x5a5a5a5au);
      dst[(stream * stream_bytes + off) / 16 + p] = v;
    }
  }
}

// the same bytes as one plain streaming store (the ceiling)
__global__ __launch_bounds__(1024) void stream_kernel(uint4 * __restrict__ dst, int64_t n16)
{
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 1024)
    dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

template <int G, bool PER_XCD>
int run(uint4 * buf, int64_t buf_bytes, int D, int n_tiles, int grid)
{
  // bytes per stream: tiles * G (shared) or tiles / 8 * G (per XCD group)
  const int64_t stream_bytes = PER_XCD ? (int64_t)((n_tiles + 7) / 8) * G : (int64_t)n_tiles * G;
  const int64_t total = (int64_t)D * n_tiles * G;
  if ((PER_XCD ? 8 : 1) * (int64_t)D * stream_bytes > buf_bytes) {
    printf("G=%d skipped (buffer)\n", G);
    return 0;
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((append_kernel<G, PER_XCD>), dim3(grid), dim3(1024), 0, 0, buf, D, stream_bytes, n_tiles);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  printf("G=%4d B  D=%5d  %-8s grid=%5d : %8.3f ms  %7.1f GB/s  (%.2f GB)\n", G, D,
         PER_XCD ? "per-xcd" : "shared", grid, best, total / best / 1e6, total / 1e9);
  return 0;
}

int main(int argc, char ** argv)
{
  const int64_t buf_bytes = (int64_t)6 << 30;
  uint4 * buf;
  CHECK(hipMalloc(&buf, buf_bytes));
  CHECK(hipMemset(buf, 0, buf_bytes));
  {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 3; it++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(1024), 0, 0, buf, (int64_t)(4ll << 30) / 16);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("streaming store 4 GiB: %.3f ms  %.1f GB/s\n", ms, (4ll << 30) / ms / 1e6);
    }
  }
  // about 2.7 GB per case: tiles * D * G
  const int grids[2] = {256, 2048};
  for (int gi = 0; gi < 2; gi++) {
    const int grid = grids[gi];
#define CASE(G_, D_)                                                            \
  {                                                                             \
    const int tiles = (int)(((int64_t)2700 << 20) / ((int64_t)(D_) * (G_)));    \
    if (run<G_, false>(buf, buf_bytes, D_, tiles, grid)) return 1;              \
    if (run<G_, true>(buf, buf_bytes, D_, tiles, grid)) return 1;               \
  }
    CASE(32, 2048)
    CASE(64, 2048)
    CASE(80, 2048)
    CASE(128, 2048)
    CASE(160, 2048)
    CASE(256, 2048)
    CASE(512, 2048)
    CASE(1280, 64)
#undef CASE
  }
  CHECK(hipFree(buf));
  return 0;
}
