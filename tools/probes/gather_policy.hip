// gather_policy.hip -- what does one random 16-byte (or 4-byte) table gather cost on MI355X, by cache
// policy bits and by how the lanes' rows share 64-byte / 128-byte lines?
//
// Question behind it: with T = 2^22 rows of F = 8 f16 (16-byte rows, 64 MiB per
// level, Infinity-Cache resident) the forward gathers 65 G rows/s; x 128 B that is the Infinity
// Cache's whole bandwidth.  If an L2 miss moves a 128-byte line for a 16-byte row, a load flavour
// that moves less (sc0/sc1/nt) or a layout that puts several wanted rows in one line would pay.
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip && ./gather_policy
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                           \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// 8 independent 16-byte gathers per lane and iteration, issued together, policy bits as text
#define GATHER8_X4(POLICY)                                                                       \
  asm volatile(                                                                                  \
    "global_load_dwordx4 %0, %8, off " POLICY "\n\t"                                             \
    "global_load_dwordx4 %1, %9, off " POLICY "\n\t"                                             \
    "global_load_dwordx4 %2, %10, off " POLICY "\n\t"                                            \
    "global_load_dwordx4 %3, %11, off " POLICY "\n\t"                                            \
    "global_load_dwordx4 %4, %12, off " POLICY "\n\t"                                            \
    "global_load_dwordx4 %5, %13, off " POLICY "\n\t"                                            \
    "global_load_dwordx4 %6, %14, off " POLICY "\n\t"                                            \
    "global_load_dwordx4 %7, %15, off " POLICY "\n\t"                                            \
    "s_waitcnt vmcnt(0)"                                                                         \
    : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), \
      "=&v"(r[7])                                                                                \
    : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])     \
    : "memory")

#define GATHER8_X1(POLICY)                                                                       \
  asm volatile(                                                                                  \
    "global_load_dword %0, %8, off " POLICY "\n\t"                                               \
    "global_load_dword %1, %9, off " POLICY "\n\t"                                               \
    "global_load_dword %2, %10, off " POLICY "\n\t"                                              \
    "global_load_dword %3, %11, off " POLICY "\n\t"                                              \
    "global_load_dword %4, %12, off " POLICY "\n\t"                                              \
    "global_load_dword %5, %13, off " POLICY "\n\t"                                              \
    "global_load_dword %6, %14, off " POLICY "\n\t"                                              \
    "global_load_dword %7, %15, off " POLICY "\n\t"                                              \
    "s_waitcnt vmcnt(0)"                                                                         \
    : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), \
      "=&v"(q[7])                                                                                \
    : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])     \
    : "memory")

// PATTERN: 0 every lane its own random row; 1 lanes 2j, 2j+1 in the two 64-byte halves of one random
// 128-byte line; 2 lanes 2j, 2j+1 in one random 64-byte sector (different 16-byte rows);
// 3 lanes 4j..4j+3 in the four 32-byte... (16-byte rows 0,2,4,6 of one 128-byte line)
template <int POLICY, int PATTERN, int ROWBYTES>
__global__ __launch_bounds__(256) void gather_kernel(
  const uint8_t * __restrict__ table, uint32_t row_mask, int iters, uint32_t * __restrict__ sink)
{
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    const uint8_t * a[8];
#pragma unroll
    for (int d = 0; d < 8; d++) {
      uint32_t key = tid, sub = 0;
      constexpr uint32_t rows_per_128 = 128 / ROWBYTES, rows_per_64 = 64 / ROWBYTES;
      if (PATTERN == 1) {
        key = tid >> 1;
        sub = (lane & 1u) * rows_per_64;
      } else if (PATTERN == 2) {
        key = tid >> 1;
        sub = (lane & 1u);
      } else if (PATTERN == 3) {
        key = tid >> 2;
        sub = (lane & 3u) * (rows_per_128 / 4);
      }
      uint32_t row = mix(key * 0x9e3779b9u + (uint32_t)(it * 8 + d) * 0x85ebca6bu) & row_mask;
      if (PATTERN != 0) row = (row & ~(rows_per_128 - 1u)) | sub;
      a[d] = table + (size_t)row * ROWBYTES;
    }
    if constexpr (ROWBYTES == 16) {
      uint4 r[8];
      if (POLICY == 0) GATHER8_X4("");
      if (POLICY == 1) GATHER8_X4("nt");
      if (POLICY == 2) GATHER8_X4("sc0");
      if (POLICY == 3) GATHER8_X4("sc1");
      if (POLICY == 4) GATHER8_X4("sc0 sc1");
      if (POLICY == 5) GATHER8_X4("sc1 nt");
      if (POLICY == 6) GATHER8_X4("sc0 sc1 nt");
#pragma unroll
      for (int d = 0; d < 8; d++) acc += r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
    } else {
      uint32_t q[8];
      if (POLICY == 0) GATHER8_X1("");
      if (POLICY == 1) GATHER8_X1("nt");
      if (POLICY == 2) GATHER8_X1("sc0");
      if (POLICY == 3) GATHER8_X1("sc1");
      if (POLICY == 4) GATHER8_X1("sc0 sc1");
      if (POLICY == 5) GATHER8_X1("sc1 nt");
      if (POLICY == 6) GATHER8_X1("sc0 sc1 nt");
#pragma unroll
      for (int d = 0; d < 8; d++) acc += q[d];
    }
  }
  if (acc == 0x12345678u) sink[tid] = acc;  // keeps the loads alive, practically never taken
}

static const char * kPolicy[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
static const char * kPattern[] = {"random rows", "pairs: 2 halves of a 128-B line", "pairs: one 64-B sector",
                                  "quads: one 128-B line"};

template <int POLICY, int PATTERN, int ROWBYTES>
int run(const uint8_t * table, size_t table_bytes, uint32_t * sink, const char * where)
{
  const uint32_t rows = (uint32_t)(table_bytes / ROWBYTES);
  const int blocks = 256 * 8, iters = 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(
      (gather_kernel<POLICY, PATTERN, ROWBYTES>), dim3(blocks), dim3(256), 0, 0, table, rows - 1u, iters,
      sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double gathers = (double)blocks * 256 * iters * 8;
  printf("%-8s %2d-B rows  %-34s %-11s %8.3f ms  %7.1f G rows/s  (x128 B = %5.2f TB/s, x64 B = %5.2f)\n",
         where, ROWBYTES, kPattern[PATTERN], kPolicy[POLICY], best, gathers / best / 1e6,
         gathers * 128 / best / 1e9, gathers * 64 / best / 1e9);
  return 0;
}

template <int ROWBYTES>
int sweep(const uint8_t * table, size_t bytes, uint32_t * sink, const char * where, bool all_policies)
{
  if (run<0, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
  if (run<0, 1, ROWBYTES>(table, bytes, sink, where)) return 1;
  if (run<0, 2, ROWBYTES>(table, bytes, sink, where)) return 1;
  if (run<0, 3, ROWBYTES>(table, bytes, sink, where)) return 1;
  if (all_policies) {
    if (run<1, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<2, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<3, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<4, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<5, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<6, 0, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<4, 1, ROWBYTES>(table, bytes, sink, where)) return 1;
    if (run<1, 1, ROWBYTES>(table, bytes, sink, where)) return 1;
  }
  return 0;
}

int main()
{
  const size_t max_bytes = (size_t)1 << 30;
  uint8_t * table = nullptr;
  uint32_t * sink = nullptr;
  CHECK(hipMalloc(&table, max_bytes));
  CHECK(hipMalloc(&sink, sizeof(uint32_t) * 256 * 8 * 256));
  CHECK(hipMemset(table, 1, max_bytes));
  CHECK(hipDeviceSynchronize());
  // 64 MiB: one C5 level (Infinity-Cache resident); 2 MiB: one C2 level (L2 resident); 1 GiB: HBM
  if (sweep<16>(table, (size_t)64 << 20, sink, "64 MiB", true)) return 1;
  if (sweep<16>(table, (size_t)1 << 30, sink, "1 GiB", true)) return 1;
  if (sweep<16>(table, (size_t)2 << 20, sink, "2 MiB", false)) return 1;
  if (sweep<4>(table, (size_t)2 << 20, sink, "2 MiB", true)) return 1;
  if (sweep<4>(table, (size_t)64 << 20, sink, "64 MiB", false)) return 1;
  printf("done\n");
  return 0;
}
