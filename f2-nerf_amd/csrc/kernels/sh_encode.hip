// sh_encode.hip -- real spherical-harmonics direction encoding.
// Replaces SHKernel<<<ceil(n/512),512>>> (reference src/sh_shader.cu:11-115); basis definition and
// sign convention per that file (:32-102; degrees 1..8, SHShader uses 4) -- see sh_basis.hiph.
//
// One thread per direction.  Each thread owns a full degree^2-float output row (64 B at degree 4 =
// one cache line), written with 16-byte stores.
#include "sh_basis.hiph"

namespace
{

template <int DEGREE>
__global__ __launch_bounds__(F2N_BLOCK) void sh_encode_kernel(
  const float * __restrict__ dirs, float * __restrict__ out, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
  constexpr int C = DEGREE * DEGREE;
  float o[C];
  sh_basis<DEGREE>(x, y, z, o);
  float * dst = out + i * C;
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int c = 0; c < C; c += 4)
      *reinterpret_cast<float4 *>(dst + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
  } else {
#pragma unroll
    for (int c = 0; c < C; c++) dst[c] = o[c];
  }
}

}  // namespace

extern "C" int f2n_sh_encode(const float * dirs, float * out, int64_t n, int degree, void * stream)
{
  if (!dirs || !out || n < 0 || degree < 1) return F2N_E_INVALID_ARG;
  if (degree > 8) return F2N_E_UNSUPPORTED;  // the reference codes degrees up to 8 (:52-102)
  if (n == 0) return F2N_OK;
  if (reinterpret_cast<uintptr_t>(out) & 15u) return F2N_E_INVALID_ARG;
  const dim3 grid(f2n_div_up(n, F2N_BLOCK)), block(F2N_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  switch (degree) {
    case 1: hipLaunchKernelGGL(sh_encode_kernel<1>, grid, block, 0, s, dirs, out, n); break;
    case 2: hipLaunchKernelGGL(sh_encode_kernel<2>, grid, block, 0, s, dirs, out, n); break;
    case 3: hipLaunchKernelGGL(sh_encode_kernel<3>, grid, block, 0, s, dirs, out, n); break;
    case 4: hipLaunchKernelGGL(sh_encode_kernel<4>, grid, block, 0, s, dirs, out, n); break;
    case 5: hipLaunchKernelGGL(sh_encode_kernel<5>, grid, block, 0, s, dirs, out, n); break;
    case 6: hipLaunchKernelGGL(sh_encode_kernel<6>, grid, block, 0, s, dirs, out, n); break;
    case 7: hipLaunchKernelGGL(sh_encode_kernel<7>, grid, block, 0, s, dirs, out, n); break;
    default: hipLaunchKernelGGL(sh_encode_kernel<8>, grid, block, 0, s, dirs, out, n); break;
  }
  return f2n_launch_status();
}
