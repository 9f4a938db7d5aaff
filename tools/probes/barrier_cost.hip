// barrier_cost.hip -- what does one __syncthreads() cost a 1024-thread workgroup that owns a CU?
// (The bin pass of the binned backward spends ~2 us per tile and level on three barriers and almost
// no work when the gradients are sparse: tools/ab_hash_bwd.py --zero-rays.)
//   hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip && ./barrier_cost
#include <hip/hip_runtime.h>

#include <cstdio>

template <int BLOCK, int LDS_WORDS, int BARRIERS>
__global__ __launch_bounds__(BLOCK) void k(int iters, unsigned * out, const float * in, int stride)
{
  __shared__ unsigned lds[LDS_WORDS];
  unsigned acc = 0;
  for (int i = threadIdx.x; i < 64; i += BLOCK) lds[i] = i;
  float g = in ? in[(size_t)blockIdx.x * BLOCK + threadIdx.x] : 0.f;
  for (int it = 0; it < iters; it++) {
    float gn = in ? in[(size_t)(it + 1) * stride + (size_t)blockIdx.x * BLOCK + threadIdx.x] : 0.f;
#pragma unroll
    for (int b = 0; b < BARRIERS; b++) {
      if (threadIdx.x < 64) lds[threadIdx.x] += 1u;
      __syncthreads();
    }
    acc += lds[(threadIdx.x + it) & 63] + (g != 0.f);
    g = gn;
  }
  if (acc == 0x7fffffffu) out[0] = acc;
}

template <int BLOCK, int LDS_WORDS, int BARRIERS>
void run(const char * what, int blocks, int iters, const float * in, int stride)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  unsigned * out;
  hipMalloc(&out, 4);
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BLOCK, LDS_WORDS, BARRIERS>), dim3(blocks), dim3(BLOCK), 0, 0, iters, out, in, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double per_cu_blocks = blocks / 256.0;
  printf("%-58s %8.3f ms  = %7.3f us per block, %6.3f us per iteration (%d barriers)\n", what, best,
         best * 1e3 / per_cu_blocks, best * 1e3 / per_cu_blocks / iters, BARRIERS);
  hipFree(out);
}

int main()
{
  float * in;
  const int stride = 8192 * 1024;
  hipMalloc(&in, sizeof(float) * (size_t)stride * 18);
  hipMemset(in, 0, sizeof(float) * (size_t)stride * 18);
  run<1024, 32768, 3>("1024 threads, 128 KiB LDS (1 per CU), 16 x 3 barriers", 8192, 16, nullptr, 0);
  run<1024, 32768, 1>("1024 threads, 128 KiB LDS (1 per CU), 16 x 1 barrier", 8192, 16, nullptr, 0);
  run<1024, 32768, 3>("1024 threads, 128 KiB LDS, 1 x 3 barriers", 8192, 1, nullptr, 0);
  run<1024, 32768, 3>("  + a prefetched global load per iteration", 8192, 16, in, stride);
  run<512, 16384, 3>("512 threads, 64 KiB LDS (2 per CU), 16 x 3 barriers", 16384, 16, nullptr, 0);
  run<256, 8192, 3>("256 threads, 32 KiB LDS (4 per CU), 16 x 3 barriers", 32768, 16, nullptr, 0);
  run<1024, 1024, 3>("1024 threads, 4 KiB LDS (2 per CU), 16 x 3 barriers", 8192, 16, nullptr, 0);
  return 0;
}
