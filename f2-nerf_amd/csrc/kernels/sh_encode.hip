// sh_encode.hip -- real spherical-harmonics direction encoding.
// Replaces SHKernel<<<ceil(n/512),512>>> (reference src/sh_shader.cu:11-115); basis definition and
// sign convention per that file (:32-50 for degree <= 4, the degree SHShader uses).
//
// One thread per direction.  Each thread owns a full degree^2-float output row (64 B at degree 4 =
// one cache line), written with 16-byte stores.
#include "common.hiph"

namespace
{

// Basis constants: Y_l^m normalisation factors, in the order the reference emits them.
constexpr float kY00 = 0.28209479177387814f;   // 1/(2 sqrt(pi))
constexpr float kY1 = 0.48860251190291987f;    // sqrt(3)/(2 sqrt(pi))
constexpr float kY2a = 1.0925484305920792f;    // sqrt(15)/(2 sqrt(pi))
constexpr float kY20s = 0.94617469575755997f;  // 3 sqrt(5)/(4 sqrt(pi))
constexpr float kY20o = 0.31539156525251999f;  // sqrt(5)/(4 sqrt(pi))
constexpr float kY22 = 0.54627421529603959f;   // sqrt(15)/(4 sqrt(pi))
constexpr float kY33 = 0.59004358992664352f;   // sqrt(70)/(8 sqrt(pi))
constexpr float kY32 = 2.8906114426405538f;    // sqrt(105)/(2 sqrt(pi))
constexpr float kY31 = 0.45704579946446572f;   // sqrt(42)/(8 sqrt(pi))
constexpr float kY30 = 0.3731763325901154f;    // sqrt(7)/(4 sqrt(pi))
constexpr float kY32b = 1.4453057213202769f;   // sqrt(105)/(4 sqrt(pi))

template <int DEGREE>
__global__ __launch_bounds__(F2N_BLOCK) void sh_encode_kernel(
  const float * __restrict__ dirs, float * __restrict__ out, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
  constexpr int C = DEGREE * DEGREE;
  float o[C];
  o[0] = kY00;
  if constexpr (DEGREE >= 2) {
    o[1] = -kY1 * y;
    o[2] = kY1 * z;
    o[3] = -kY1 * x;
  }
  if constexpr (DEGREE >= 3) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[4] = kY2a * xy;
    o[5] = -kY2a * yz;
    o[6] = fmaf(kY20s, z2, -kY20o);
    o[7] = -kY2a * xz;
    o[8] = fmaf(kY22, x2, -(kY22 * y2));
    if constexpr (DEGREE >= 4) {
      o[9] = kY33 * y * fmaf(-3.0f, x2, y2);
      o[10] = kY32 * xy * z;
      o[11] = kY31 * y * fmaf(-5.0f, z2, 1.0f);
      o[12] = kY30 * z * fmaf(5.0f, z2, -3.0f);
      o[13] = kY31 * x * fmaf(-5.0f, z2, 1.0f);
      o[14] = kY32b * z * (x2 - y2);
      o[15] = kY33 * x * fmaf(3.0f, y2, -x2);
    }
  }
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
  if (degree > 4) return F2N_E_UNSUPPORTED;
  if (n == 0) return F2N_OK;
  if (reinterpret_cast<uintptr_t>(out) & 15u) return F2N_E_INVALID_ARG;
  const dim3 grid(f2n_div_up(n, F2N_BLOCK)), block(F2N_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  switch (degree) {
    case 1: hipLaunchKernelGGL(sh_encode_kernel<1>, grid, block, 0, s, dirs, out, n); break;
    case 2: hipLaunchKernelGGL(sh_encode_kernel<2>, grid, block, 0, s, dirs, out, n); break;
    case 3: hipLaunchKernelGGL(sh_encode_kernel<3>, grid, block, 0, s, dirs, out, n); break;
    default: hipLaunchKernelGGL(sh_encode_kernel<4>, grid, block, 0, s, dirs, out, n); break;
  }
  return f2n_launch_status();
}
