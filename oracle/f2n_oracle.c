/*
 * f2n_oracle.c -- CPU restatement of the F2-NeRF rendering hot path's device kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (f2-nerf_amd/) may link, load or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference (SakodaShintaro/f2-nerf) ships no golden vectors, known-answer
 * tests or fixtures for this path and its CUDA sources cannot be built
 * here (nvcc, cuda_runtime.h, OpenCV absent).  This file restates the arithmetic of the reference's
 * 14 CUDA kernels from their source text; each function cites the file:line it follows
 * (paths relative to the reference checkout).
 *
 * Conventions that ARE the spec:
 *   - a*b+c that nvcc contracts (-fmad=true) is written fmaf() explicitly;
 *   - float -> unsigned conversion saturates at 0 for negatives (CUDA cvt.rzi.u32.f32);
 *   - f16 storage is emulated with a software RNE f32<->f16 conversion;
 *   - per-ray loops keep the reference's serial summation order.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -march=x86-64-v3 -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- f16 emulation (RNE) --------- */

static inline uint32_t f32_bits(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float bits_f32(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* IEEE-754 binary32 -> binary16, round-to-nearest-even, NaN stays NaN, overflow -> inf. */
uint16_t f2no_f32_to_f16(float f)
{
  const uint32_t x = f32_bits(f);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
  }
  if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25 exactly: ties to even -> 0) */
    return (uint16_t)sign;
  }
  const int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  int shift;
  uint32_t he;
  if (e < -14) { /* subnormal half */
    shift = 13 + (-14 - e);
    he = 0;
  } else {
    shift = 13;
    he = (uint32_t)(e + 15);
  }
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u);
  const uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t h;
  if (he == 0) {
    h = q; /* may carry into exponent 1: correct by construction */
  } else {
    h = ((he - 1u) << 10) + q; /* q includes the hidden bit (0x400) -> adds 1 to exponent */
  }
  return (uint16_t)(sign | h);
}

float f2no_f16_to_f32(uint16_t h)
{
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1fu;
  const uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return bits_f32(sign);
    /* subnormal: m * 2^-24 */
    float v = (float)m * 5.9604644775390625e-08f;
    return sign ? -v : v;
  }
  if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

static inline float round_f16(float f) { return f2no_f16_to_f32(f2no_f32_to_f16(f)); }

/* feat_pool.to(kFloat16): src/hash_3d_anchored.cu:169,198 */
void f2no_cast_f32_to_f16(const float * in, uint16_t * out, int64_t n)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) out[i] = f2no_f32_to_f16(in[i]);
}

void f2no_cast_f16_to_f32(const uint16_t * in, float * out, int64_t n)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) out[i] = f2no_f16_to_f32(in[i]);
}

/* ---------------------------------------------------------------- hash grid (A1/A2) ----------- */

/* Per-level scale: src/hash_3d_anchored.cu:72-73
 *   mul = exp2f((RES_FINE_POW_2 - RES_BASE_POW_2) * float(l) / float(N_LEVELS - 1) + RES_BASE_POW_2)
 * evaluated left to right in f32 (glibc exp2f).  L == 1 would divide by zero in the reference;
 * here it yields the base resolution. */
void f2no_level_mul(int L, float * mul)
{
  for (int l = 0; l < L; l++) {
    float e = (L > 1) ? ((10.f - 3.f) * (float)l / (float)(L - 1) + 3.f) : 3.f;
    mul[l] = exp2f(e);
  }
}

/* static_cast<unsigned>(floorf(x)) with CUDA/gfx950 saturating semantics (quirk Q1). */
static inline uint32_t sat_u32(float fl)
{
  if (!(fl > 0.f)) return 0u; /* negatives and NaN -> 0 */
  if (fl >= 4294967296.f) return 0xffffffffu;
  return (uint32_t)fl;
}

/* calculate_pos_and_w: src/hash_3d_anchored.cu:25-58.  pt already scaled+biased. */
static inline void pos_and_w(
  const float pt[3], uint32_t T, const int32_t * prim, uint32_t pos[8], float w[8])
{
  const uint32_t pa = (uint32_t)prim[0], pb = (uint32_t)prim[1], pc = (uint32_t)prim[2];
  const float fx = floorf(pt[0]), fy = floorf(pt[1]), fz = floorf(pt[2]);
  const uint32_t px = sat_u32(fx), py = sat_u32(fy), pz = sat_u32(fz);
  pos[0] = ((px * pa) ^ (py * pb) ^ (pz * pc)) % T;
  pos[1] = ((px * pa) ^ (py * pb) ^ ((pz + 1u) * pc)) % T;
  pos[2] = ((px * pa) ^ ((py + 1u) * pb) ^ (pz * pc)) % T;
  pos[3] = ((px * pa) ^ ((py + 1u) * pb) ^ ((pz + 1u) * pc)) % T;
  pos[4] = (((px + 1u) * pa) ^ (py * pb) ^ (pz * pc)) % T;
  pos[5] = (((px + 1u) * pa) ^ (py * pb) ^ ((pz + 1u) * pc)) % T;
  pos[6] = (((px + 1u) * pa) ^ ((py + 1u) * pb) ^ (pz * pc)) % T;
  pos[7] = (((px + 1u) * pa) ^ ((py + 1u) * pb) ^ ((pz + 1u) * pc)) % T;
  const float a = pt[0] - fx, b = pt[1] - fy, c = pt[2] - fz;
  w[0] = (1.f - a) * (1.f - b) * (1.f - c);
  w[1] = (1.f - a) * (1.f - b) * c;
  w[2] = (1.f - a) * b * (1.f - c);
  w[3] = (1.f - a) * b * c;
  w[4] = a * (1.f - b) * (1.f - c);
  w[5] = a * (1.f - b) * c;
  w[6] = a * b * (1.f - c);
  w[7] = a * b * c;
}

/* pt = points[p] * mul + bias[l]  (src/hash_3d_anchored.cu:74, contracted to FMA by nvcc). */
static inline void scale_point(const float * p, float mul, const float * bias, float pt[3])
{
  pt[0] = fmaf(p[0], mul, bias[0]);
  pt[1] = fmaf(p[1], mul, bias[1]);
  pt[2] = fmaf(p[2], mul, bias[2]);
}

/* Hash3DAnchoredForwardKernel + Function::forward: src/hash_3d_anchored.cu:60-93,150-179.
 *   table_f16 : f16 bits of the (already cast) pool, length >= level_stride*(L-1) + T*F
 *   out       : [n, L*F] f32 holding the f16-rounded result (the reference returns out.to(f32))
 *   idx_out   : optional [n, L, 8] u32 hash rows (for the bit-exact index parity test)
 * level base = table + level_stride*l ELEMENTS (reference: level_stride == T, quirk Q2). */
void f2no_hash_fwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, float * out, uint32_t * idx_out, int64_t n, int L, int F, uint32_t T,
  int64_t level_stride)
{
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n; p++) {
    for (int l = 0; l < L; l++) {
      const uint16_t * base = table_f16 + level_stride * l;
      float pt[3], w[8];
      uint32_t pos[8];
      scale_point(pts + 3 * p, mul[l], bias + 3 * l, pt);
      pos_and_w(pt, T, primes + 3 * l, pos, w);
      if (idx_out) memcpy(idx_out + ((int64_t)p * L + l) * 8, pos, sizeof(pos));
      for (int k = 0; k < F; k++) {
        /* ws[0]*f0 + ws[1]*f1 + ... left to right, each '+' contracted with its product */
        float acc = w[0] * f2no_f16_to_f32(base[(int64_t)pos[0] * F + k]);
        for (int d = 1; d < 8; d++)
          acc = fmaf(w[d], f2no_f16_to_f32(base[(int64_t)pos[d] * F + k]), acc);
        out[p * (int64_t)(L * F) + l * F + k] = round_f16(acc);
      }
    }
  }
}

/* Hash3DAnchoredBackwardKernel + Function::backward: src/hash_3d_anchored.cu:95-145,181-218.
 *   grad_out   : [n, L*F] f32
 *   table_grad : f32, same indexing as the table; ACCUMULATED INTO (caller zeroes).  Each
 *                contribution is f16(f16(grad_scale*g) * w) as in the reference (:130-137) but the
 *                running sum is kept in f32 (the reference's f16 atomics are order-dependent, so
 *                it has no single right answer; this is the f32-accumulated statement of it).
 *   pts_grad   : [n,3] f32 or NULL; contributions f16(sign * f(d,k)*mul*g) (:138-143, quirk Q5)
 * pts_grad is divided by grad_scale here (:214); the caller divides table_grad (:215) once all
 * calls that accumulate into it are done (f2no_div_inplace). */
static void hash_bwd_point(
  int64_t p, const float * pts, const uint16_t * table_f16, const int32_t * primes,
  const float * bias, const float * mul, const float * grad_out, float * table_grad,
  float * pts_grad, int L, int F, uint32_t T, int64_t level_stride, float grad_scale, int atomic)
{
  static const float sx[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
  static const float sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
  static const float sz[8] = {-1, 1, -1, 1, -1, 1, -1, 1};
  float gp[3] = {0.f, 0.f, 0.f};
  for (int l = 0; l < L; l++) {
    const uint16_t * base = table_f16 + level_stride * l;
    float * gbase = table_grad + level_stride * l;
    float pt[3], w[8];
    uint32_t pos[8];
    scale_point(pts + 3 * p, mul[l], bias + 3 * l, pt);
    pos_and_w(pt, T, primes + 3 * l, pos, w);
    const float * g = grad_out + p * (int64_t)(L * F) + l * F;
    for (int d = 0; d < 8; d++) {
      for (int k = 0; k < F; k++) {
        const float gk = round_f16(g[k] * grad_scale); /* (g*128).to(f16), :200 */
        const float c = round_f16(gk * w[d]);          /* (__half)(w0 * ws[d]), :134 */
        float * dst = gbase + (int64_t)pos[d] * F + k;
        if (atomic) {
#pragma omp atomic
          *dst += c;
        } else {
          *dst += c;
        }
        if (pts_grad) {
          const float norm = f2no_f16_to_f32(base[(int64_t)pos[d] * F + k]) * mul[l] * gk;
          gp[0] += round_f16(sx[d] * norm);
          gp[1] += round_f16(sy[d] * norm);
          gp[2] += round_f16(sz[d] * norm);
        }
      }
    }
  }
  if (pts_grad) {
    pts_grad[3 * p + 0] = gp[0] / grad_scale;
    pts_grad[3 * p + 1] = gp[1] / grad_scale;
    pts_grad[3 * p + 2] = gp[2] / grad_scale;
  }
}

void f2no_hash_bwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, const float * grad_out, float * table_grad, float * pts_grad, int64_t n,
  int L, int F, uint32_t T, int64_t level_stride, float grad_scale, int parallel)
{
  if (parallel) {
    /* same arithmetic, table sums in thread-arrival order (used for the timed CPU baseline) */
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++)
      hash_bwd_point(
        p, pts, table_f16, primes, bias, mul, grad_out, table_grad, pts_grad, L, F, T,
        level_stride, grad_scale, 1);
  } else {
    /* serial over points: deterministic accumulation order (sample, level, corner) */
    for (int64_t p = 0; p < n; p++)
      hash_bwd_point(
        p, pts, table_f16, primes, bias, mul, grad_out, table_grad, pts_grad, L, F, T,
        level_stride, grad_scale, 0);
  }
}

/* embeds_grad.to(f32) / grad_scale: src/hash_3d_anchored.cu:215 */
void f2no_div_inplace(float * x, int64_t n, float div)
{
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) x[i] = x[i] / div;
}

/* ---------------------------------------------------------------- SH encode (A6) -------------- */

/* SHKernel: src/sh_shader.cu:11-103.  Real spherical-harmonics basis, degree <= 8 -> degree^2 values.
 * Products/sums follow the reference expressions; a*b+c forms use fmaf as nvcc would. */
void f2no_sh_encode(const float * dirs, float * out, int64_t n, int degree)
{
  const int C = degree * degree;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    float * o = out + i * C;
    const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    if (degree <= 1) continue;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    if (degree <= 2) continue;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = fmaf(0.94617469575755997f, z2, -0.31539156525251999f);
    o[7] = -1.0925484305920792f * xz;
    o[8] = fmaf(0.54627421529603959f, x2, -(0.54627421529603959f * y2));
    if (degree <= 3) continue;
    o[9] = 0.59004358992664352f * y * fmaf(-3.0f, x2, y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * fmaf(-5.0f, z2, 1.0f);
    o[12] = 0.3731763325901154f * z * fmaf(5.0f, z2, -3.0f);
    o[13] = 0.45704579946446572f * x * fmaf(-5.0f, z2, 1.0f);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * fmaf(3.0f, y2, -x2);
    if (degree <= 4) continue;
    /* Bands 4..7 (degree 5..8): coded in the reference (:52-102) as expanded polynomials "based on
     * the recurrence relations in appendix A1 of" Sloan's Stupid SH Tricks (:30), used by nothing
     * (SHShader::DEGREE == 4, sh_shader.hpp:20).  Restated here from the definition those
     * polynomials expand, in double precision:
     *   out[l*l + l + m] = (-1)^m sqrt2(m) N(l,|m|) Q(l,|m|)(z) {Re, Im}((x + iy)^|m|)
     * (same ordering and signs as bands 0..3 above).  tests/test_oracle_cpu.py checks this against
     * the closed forms the reference's comments give for a dozen of the entries. */
    {
      double c[8], s[8];
      c[0] = 1.0;
      s[0] = 0.0;
      for (int m = 1; m < degree; m++) {
        c[m] = (double)x * c[m - 1] - (double)y * s[m - 1];
        s[m] = (double)x * s[m - 1] + (double)y * c[m - 1];
      }
      for (int m = 0; m < degree; m++) {
        double q2 = 0.0, q1 = 1.0;
        for (int k = 1; k <= m; k++) q1 *= (double)(2 * k - 1);
        for (int l = m; l < degree; l++) {
          double q = q1;
          if (l > m) {
            q = ((2.0 * l - 1.0) * (double)z * q1 - (double)(l + m - 1) * q2) / (double)(l - m);
            q2 = q1;
            q1 = q;
          }
          if (l < 4) continue;
          double ratio = 1.0;
          for (int k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
          double nrm = sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio) * (m ? sqrt(2.0) : 1.0);
          if (m & 1) nrm = -nrm;
          o[l * l + l + m] = (float)(nrm * q * c[m]);
          if (m > 0) o[l * l + l - m] = (float)(nrm * q * s[m]);
        }
      }
    }
  }
}

/* ---------------------------------------------------------------- ragged segment ops (A7) ----- */

/* FlexSumForwardKernel: src/CustomOps/FlexOps.cu:6-16 */
void f2no_seg_sum_fwd(const float * val, const int32_t * idx, float * sum, int n_rays)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++) {
    float acc = 0.f;
    for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++) acc += val[i];
    sum[r] = acc;
  }
}

/* FlexSumBackwardKernel: src/CustomOps/FlexOps.cu:18-27 (only covered samples are written) */
void f2no_seg_sum_bwd(const float * dsum, const int32_t * idx, float * dval, int n_rays)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++)
    for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++) dval[i] = dsum[r];
}

/* FlexSumVecForwardKernel: src/CustomOps/FlexOps.cu:29-41 */
void f2no_seg_sum_vec_fwd(const float * val, const int32_t * idx, float * sum, int n_rays, int vec)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++)
    for (int j = 0; j < vec; j++) {
      float acc = 0.f;
      for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++) acc += val[(int64_t)i * vec + j];
      sum[(int64_t)r * vec + j] = acc;
    }
}

/* FlexSumVecBackwardKernel: src/CustomOps/FlexOps.cu:43-54 */
void f2no_seg_sum_vec_bwd(const float * dsum, const int32_t * idx, float * dval, int n_rays, int vec)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++)
    for (int j = 0; j < vec; j++)
      for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++)
        dval[(int64_t)i * vec + j] = dsum[(int64_t)r * vec + j];
}

/* FlexAccumulateSumForwardKernel: src/CustomOps/FlexOps.cu:56-74 */
void f2no_seg_scan_fwd(
  const float * val, const int32_t * idx, float * sum, int n_rays, int include_this)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++) {
    float acc = 0.f;
    for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++) {
      if (include_this) {
        acc += val[i];
        sum[i] = acc;
      } else {
        sum[i] = acc;
        acc += val[i];
      }
    }
  }
}

/* FlexAccumulateSumBackwardKernel: src/CustomOps/FlexOps.cu:76-94 */
void f2no_seg_scan_bwd(
  const float * dsum, const int32_t * idx, float * dval, int n_rays, int include_this)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++) {
    float wp = 0.f;
    for (int i = idx[2 * r + 1] - 1; i >= idx[2 * r]; i--) {
      if (include_this) {
        wp += dsum[i];
        dval[i] = wp;
      } else {
        dval[i] = wp;
        wp += dsum[i];
      }
    }
  }
}

/* ---------------------------------------------------------------- WeightVar (A10) ------------- */

/* WeightVarLossForwardKernel: src/CustomOps/CustomOps.cu:13-36 (SCALE = 16, :9) */
void f2no_weight_var_fwd(const float * w, const int32_t * idx, float * out_vars, int n_rays)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++) {
    const int s = idx[2 * r], e = idx[2 * r + 1];
    if (s >= e) {
      out_vars[r] = 0.f;
      continue;
    }
    float mean = 0.f, wsum = 1e-6f;
    const float len = 16.f;
    for (int i = 0; i + s < e; i++) {
      mean = fmaf(w[i + s], (float)i / len, mean);
      wsum += w[i + s];
    }
    mean /= wsum;
    float var = 0.f;
    for (int i = 0; i + s < e; i++) {
      const float b = (float)i / len - mean;
      var = fmaf(w[i + s] * b, b, var);
    }
    out_vars[r] = var;
  }
}

/* WeightVarLossBackwardKernel: src/CustomOps/CustomOps.cu:39-67 (quirk Q9 kept as coded) */
void f2no_weight_var_bwd(
  const float * w, const int32_t * idx, const float * dvars, float * dw, int n_rays)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++) {
    const int s = idx[2 * r], e = idx[2 * r + 1];
    if (s >= e) continue;
    float mean = 0.f, wsum = 1e-6f;
    const float len = 16.f;
    for (int i = 0; i + s < e; i++) {
      mean = fmaf(w[i + s], (float)i / len, mean);
      wsum += w[i + s];
    }
    mean /= wsum;
    float tmp = 0.f;
    for (int i = 0; i + s < e; i++) {
      const float b = (float)i / len - mean;
      tmp = fmaf(w[i + s] * 2.f, b, tmp);
    }
    for (int i = 0; i + s < e; i++) {
      const float b = (float)i / len - mean;
      const float grad = fmaf(b, b, tmp * -((float)i / len) / wsum);
      dw[i + s] = dvars[r] * grad;
    }
  }
}

/* ---------------------------------------------------------------- scatter (A9) ---------------- */

/* ScatterIdxKernal: src/CustomOps/Scatter.cu:111-121 */
void f2no_scatter_idx(const int32_t * idx, const int32_t * emb_idx, int32_t * all_emb_idx, int n_rays)
{
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < n_rays; r++)
    for (int i = idx[2 * r]; i < idx[2 * r + 1]; i++) all_emb_idx[i] = emb_idx[r];
}

/* ScatterAddFuncForward: src/CustomOps/Scatter.cu:11-19 (sum is a clone of to_add, :63) */
void f2no_scatter_add_fwd(
  const float * emb, const int32_t * scatter_idx, const float * to_add, float * sum, int64_t n_all,
  int C)
{
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n_all; p++)
    for (int c = 0; c < C; c++)
      sum[p * C + c] = to_add[p * C + c] + emb[(int64_t)scatter_idx[p] * C + c];
}

/* ScatterAddFuncBackwardBlock + torch::sum(dim=1): src/CustomOps/Scatter.cu:21-41,83-97.
 * Same blocking as the reference: block_size = (floor(sqrt(n_all+1024)) >> 5) << 5, per-(image,
 * block) partial sums in index order, then the partials summed over blocks in block order. */
void f2no_scatter_add_bwd(
  const int32_t * scatter_idx, const float * dsum, float * demb, int64_t n_all, int n_emb, int C)
{
  int block_size = (((int)sqrt((double)(n_all + 1024))) >> 5) << 5;
  if (block_size < 32) block_size = 32;
  const int64_t n_blocks = (n_all + block_size - 1) / block_size;
  float * pool = (float *)calloc((size_t)n_emb * n_blocks * C, sizeof(float));
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < n_blocks; b++) {
    const int64_t lo = b * block_size;
    const int64_t hi = (lo + block_size > n_all) ? n_all : lo + block_size;
    for (int64_t i = lo; i < hi; i++) {
      const int e = scatter_idx[i];
      if (e < 0 || e >= n_emb) continue;
      float * dst = pool + ((int64_t)e * n_blocks + b) * C;
      for (int c = 0; c < C; c++) dst[c] += dsum[i * C + c];
    }
  }
  for (int e = 0; e < n_emb; e++)
    for (int c = 0; c < C; c++) {
      float acc = 0.f;
      for (int64_t b = 0; b < n_blocks; b++) acc += pool[((int64_t)e * n_blocks + b) * C + c];
      demb[(int64_t)e * C + c] = acc;
    }
  free(pool);
}

/* ---------------------------------------------------------------- misc ------------------------ */

int f2no_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void f2no_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
