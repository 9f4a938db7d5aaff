// Probe: throughput of LDS atomics on gfx950 with random addresses (what the hash backward does).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>

constexpr int kWords = 32768;
constexpr int kIters = 2048;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const uint32_t * __restrict__ idx, float * out, int span)
{
  __shared__ float acc[kWords];
  for (int i = threadIdx.x; i < kWords; i += 1024) acc[i] = 0.f;
  __syncthreads();
  uint32_t h = idx[blockIdx.x * 1024 + threadIdx.x];
  for (int it = 0; it < kIters; it++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t a = (h >> 8) % (uint32_t)span;
    if (MODE == 0) atomicAdd(&acc[a], 1.0f);                                   // ds_add_f32
    if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t *>(acc) + a, 1u);        // ds_add_u32
    if (MODE == 2) acc[a] += 1.0f;                                             // racy ld/st
    if (MODE == 3) {
      __half2 v = __floats2half2_rn(1.f, 1.f);
      unsafeAtomicAdd(reinterpret_cast<__half2 *>(acc) + a, v);                 // ds_pk_add_f16
    }
    if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + (a >> 1), 1ull);  // ds_add_u64
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < kWords; i += 1024) s += acc[i];
  if (s == 12345.f) out[0] = s;
}

int main()
{
  const int blocks = 512;
  std::vector<uint32_t> h(blocks * 1024);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u);
  uint32_t * d_idx;
  float * d_out;
  hipMalloc(&d_idx, h.size() * 4);
  hipMalloc(&d_out, 4);
  hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char * names[5] = {"ds_add_f32", "ds_add_u32", "racy ld+st", "ds_pk_add_f16", "ds_add_u64"};
  for (int span : {32768, 1024, 8}) {
    for (int mode = 0; mode < 5; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: probe<0><<<blocks, 1024>>>(d_idx, d_out, span); break;
          case 1: probe<1><<<blocks, 1024>>>(d_idx, d_out, span); break;
          case 2: probe<2><<<blocks, 1024>>>(d_idx, d_out, span); break;
          case 3: probe<3><<<blocks, 1024>>>(d_idx, d_out, span); break;
          default: probe<4><<<blocks, 1024>>>(d_idx, d_out, span / 2 * 2); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
          const double ops = (double)blocks * 1024 * kIters;
          printf("span %6d  %-14s %8.3f ms  %7.2f G lane-ops/s  (%.1f cycles per wave-instr per CU at 2.1 GHz)\n",
                 span, names[mode], ms, ops / ms / 1e6, ms * 1e-3 * 2.1e9 / (ops / 64 / 256));
        }
      }
    }
  }
  return 0;
}
