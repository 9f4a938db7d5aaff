// sampler.hip -- ray sampling, the fused early-termination march and sample compaction
// (SURVEY.md rows A4, A5).
//
// Replaces PtsSampler::get_samples (about 20 ATen launches, reference src/points_sampler.cpp:20-64)
// and the early-stop block of Renderer::render (src/renderer.cpp:58-90), which in the reference
// evaluates the whole field on all n_rays*S samples only to build a keep-mask and then runs
// where() + four index() gathers.
//
// Mapping: one 64-lane wavefront per ray, lanes = 64 consecutive samples (a "stride").  Cumulative
// step noise and the exclusive optical-depth scan are DPP wave scans with a scalar carry between
// strides; the keep-mask is a wave ballot; a ray whose transmittance has dropped to <= t_thresh
// stops issuing strides, so a terminated ray costs ceil(kept/64) strides, not S/64.
#include "hash_grid.hiph"

namespace
{

struct RayFrame
{
  float ox, oy, oz;
  float dx, dy, dz;  // unit direction
};

// rays_d / linalg_norm(rays_d, 2, -1, true)  (points_sampler.cpp:24)
__device__ __forceinline__ RayFrame load_ray(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d, int r)
{
  RayFrame rf;
  rf.ox = rays_o[3 * r];
  rf.oy = rays_o[3 * r + 1];
  rf.oz = rays_o[3 * r + 2];
  const float x = rays_d[3 * r], y = rays_d[3 * r + 1], z = rays_d[3 * r + 2];
  const float nrm = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
  rf.dx = x / nrm;
  rf.dy = y / nrm;
  rf.dz = z / nrm;
  return rf;
}

struct StrideCarry
{
  float noise;       // cumulative noise up to the previous stride
  float lx, ly, lz;  // last sample point of the previous stride
};

struct StrideSample
{
  float t, px, py, pz, dt;
  bool valid;
};

// Samples k0 .. k0+63 of one ray (points_sampler.cpp:31-48):
//   t_k = cumsum(noise)_k * step ; p_k = o + d*t_k (mul, then add -- two ATen ops, not fused) ;
//   dt_0 = 0, dt_k = |p_k - p_{k-1}|   (differences of points, quirk Q7)
// The same function feeds f2n_sample_rays, f2n_density_march and f2n_sample_compact so that all
// three see bit-identical samples.
__device__ __forceinline__ StrideSample make_stride(
  const RayFrame & rf, const float * __restrict__ noise_row, int k0, int S, float step,
  StrideCarry & carry, int lane)
{
  StrideSample sm;
  const int k = k0 + lane;
  sm.valid = k < S;
  float cum;
  if (noise_row) {
    const float nz = sm.valid ? noise_row[k] : 0.f;
    cum = carry.noise + wave_incl_scan(nz);
  } else {
    cum = (float)(min(k, S - 1) + 1);  // cumsum of ones is exact
  }
  sm.t = cum * step;
  const float mx = rf.dx * sm.t, my = rf.dy * sm.t, mz = rf.dz * sm.t;
  sm.px = rf.ox + mx;
  sm.py = rf.oy + my;
  sm.pz = rf.oz + mz;
  const float qx = wave_shift_up1(sm.px, carry.lx);
  const float qy = wave_shift_up1(sm.py, carry.ly);
  const float qz = wave_shift_up1(sm.pz, carry.lz);
  const float ex = sm.px - qx, ey = sm.py - qy, ez = sm.pz - qz;
  sm.dt = (k == 0) ? 0.f : sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
  carry.noise = wave_bcast_last(cum);
  carry.lx = wave_bcast_last(sm.px);
  carry.ly = wave_bcast_last(sm.py);
  carry.lz = wave_bcast_last(sm.pz);
  return sm;
}

__device__ __forceinline__ int ray_of_wave()
{
  return (int)blockIdx.x * F2N_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
}

// ---- f2n_sample_rays ----------------------------------------------------------------------------

__global__ __launch_bounds__(F2N_BLOCK) void sample_rays_kernel(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d,
  const float * __restrict__ noise, float * __restrict__ pts, float * __restrict__ dirs,
  float * __restrict__ dt, float * __restrict__ t, int32_t * __restrict__ bounds, int n_rays, int S,
  float step)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const RayFrame rf = load_ray(rays_o, rays_d, r);
  const float * nrow = noise ? noise + (int64_t)r * S : nullptr;
  StrideCarry carry = {0.f, 0.f, 0.f, 0.f};
  const int64_t base = (int64_t)r * S;
  for (int k0 = 0; k0 < S; k0 += F2N_WAVE) {
    const StrideSample sm = make_stride(rf, nrow, k0, S, step, carry, lane);
    if (sm.valid) {
      const int64_t i = base + k0 + lane;
      pts[3 * i] = sm.px;
      pts[3 * i + 1] = sm.py;
      pts[3 * i + 2] = sm.pz;
      dirs[3 * i] = rf.dx;
      dirs[3 * i + 1] = rf.dy;
      dirs[3 * i + 2] = rf.dz;
      dt[i] = sm.dt;
      t[i] = sm.t;
    }
  }
  if (lane == 0) {
    bounds[2 * r] = (int32_t)base;
    bounds[2 * r + 1] = (int32_t)(base + S);
  }
}

// ---- f2n_sample_compact -------------------------------------------------------------------------

__global__ __launch_bounds__(F2N_BLOCK) void sample_compact_kernel(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d,
  const float * __restrict__ noise, const int32_t * __restrict__ bounds, float * __restrict__ pts,
  float * __restrict__ dirs, float * __restrict__ dt, float * __restrict__ t, int n_rays, int S,
  float step)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int start = bounds[2 * r];
  const int cnt = bounds[2 * r + 1] - start;
  if (cnt <= 0) return;
  const RayFrame rf = load_ray(rays_o, rays_d, r);
  const float * nrow = noise ? noise + (int64_t)r * S : nullptr;
  StrideCarry carry = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < cnt; k0 += F2N_WAVE) {
    const StrideSample sm = make_stride(rf, nrow, k0, S, step, carry, lane);
    if (k0 + lane < cnt) {
      const int64_t i = (int64_t)start + k0 + lane;
      pts[3 * i] = sm.px;
      pts[3 * i + 1] = sm.py;
      pts[3 * i + 2] = sm.pz;
      dirs[3 * i] = rf.dx;
      dirs[3 * i + 1] = rf.dy;
      dirs[3 * i + 2] = rf.dz;
      dt[i] = sm.dt;
      t[i] = sm.t;
    }
  }
}

// ---- f2n_density_march --------------------------------------------------------------------------

template <int F, bool POW2>
__global__ __launch_bounds__(F2N_BLOCK) void density_march_kernel(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d,
  const float * __restrict__ noise, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, const float * __restrict__ w0, const float * __restrict__ b0,
  int32_t * __restrict__ kept, int n_rays, int S, float step, int L, uint32_t T,
  int64_t level_stride, float t_thresh, float density_shift)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const RayFrame rf = load_ray(rays_o, rays_d, r);
  const float * nrow = noise ? noise + (int64_t)r * S : nullptr;
  const float bias0 = b0[0];
  StrideCarry carry = {0.f, 0.f, 0.f, 0.f};
  float depth_carry = 0.f;  // optical depth accumulated by earlier strides
  int n_kept = 0;
  for (int k0 = 0; k0 < S; k0 += F2N_WAVE) {
    const StrideSample sm = make_stride(rf, nrow, k0, S, step, carry, lane);
    float x = sm.px, y = sm.py, z = sm.pz;
    contract_point(x, y, z);
    // density logit = row 0 of Linear(L*F -> 16) applied to the f16-rounded encoding
    float logit = bias0;
    for (int l = 0; l < L; l++) {
      const LevelParams lp = load_level(primes, bias, mul, l);
      uint32_t row[8];
      float w[8], acc[F];
      corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);
      gather_blend<F>(table + level_stride * l, row, w, acc);
#pragma unroll
      for (int k = 0; k < F; k++) logit = fmaf(round_f16(acc[k]), w0[l * F + k], logit);
    }
    const float sigma = expf(logit - density_shift);   // TruncExp forward
    const float sec = sm.valid ? sigma * sm.dt : 0.f;  // sigma * dt
    const float incl = wave_incl_scan(sec);
    const float depth = depth_carry + wave_shift_up1(incl, 0.f);  // exclusive scan
    const float trans = expf(-depth);
    const bool keep = sm.valid && (trans > t_thresh);
    const unsigned long long m = __ballot(keep);
    n_kept += __popcll(m);
    const int n_valid = min(F2N_WAVE, S - k0);
    if (__popcll(m) < n_valid) break;  // the mask is a prefix: nothing later survives
    depth_carry += wave_bcast_last(incl);
  }
  if (lane == 0) kept[r] = n_kept;
}

// Four rays per wavefront, one per 16-lane DPP row, strides of 16 samples: a ray that stops after a
// handful of samples (a trained scene seen from close by; the bench's terminating regime keeps 3.5
// samples per ray) costs a 16-sample stride instead of a 64-sample one -- the wave scans are row
// scans anyway, and the gathers of a stride touch as many lines either way.
//
// Bit-identical to density_march_kernel (and to make_stride, which f2n_sample_compact uses to
// re-create the kept samples): the 64-lane Kogge-Stone scan adds, for the lanes of its row j,
//   row 0: rs            row 1: rs + T0          row 2: rs + (T1 + T0)     row 3: (rs + T2) + (T1 + T0)
// (rs = scan inside the row, Tj = total of row j) and carries carry + ((T3 + T2) + (T1 + T0)) to the
// next 64 samples; stride j of a 64-sample block does exactly those additions here.
struct RowScan
{
  float carry;    // inclusive total of the completed 64-sample blocks
  float t0, t1p, t2;  // T0, T1 + T0, T2 of the current block
  float last;     // inclusive value (within the block) of the previous stride's last lane
};

__device__ __forceinline__ float row_incl_scan(float v)
{
  v += dpp_get<0x111, 0xf, 0xf>(v, 0.f);
  v += dpp_get<0x112, 0xf, 0xf>(v, 0.f);
  v += dpp_get<0x114, 0xf, 0xf>(v, 0.f);
  v += dpp_get<0x118, 0xf, 0xf>(v, 0.f);
  return v;
}
// lane 15 of the own row in every lane of the row (row_newbcast:15)
__device__ __forceinline__ float row_last(float v) { return dpp_get<0x15F, 0xf, 0xf>(v, 0.f); }
// previous lane of the own row, `fill` in the row's first lane (row_shr:1)
__device__ __forceinline__ float row_shift_up1(float v, float fill) { return dpp_get<0x111, 0xf, 0xf>(v, fill); }

// inclusive value, within its 64-sample block, of this lane's element of stride j (0..3); updates st
__device__ __forceinline__ float row_scan_step(float v, int j, RowScan & st, float & incl_in_block)
{
  const float rs = row_incl_scan(v);
  const float tj = row_last(rs);
  float inner;
  if (j == 0) {
    inner = rs;
    st.t0 = tj;
  } else if (j == 1) {
    inner = rs + st.t0;
    st.t1p = tj + st.t0;
  } else if (j == 2) {
    inner = rs + st.t1p;
    st.t2 = tj;
  } else {
    inner = (rs + st.t2) + st.t1p;
  }
  incl_in_block = inner;
  return st.carry + inner;
}

template <int F, bool POW2>
__global__ __launch_bounds__(F2N_BLOCK) void density_march16_kernel(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d,
  const float * __restrict__ noise, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, const float * __restrict__ w0, const float * __restrict__ b0,
  int32_t * __restrict__ kept, int n_rays, int S, float step, int L, uint32_t T,
  int64_t level_stride, float t_thresh, float density_shift)
{
  const int lane = lane_id(), q = lane >> 4, m = lane & 15;
  const int r_raw = (ray_of_wave() << 2) + q;
  const bool has_ray = r_raw < n_rays;
  const int r = has_ray ? r_raw : n_rays - 1;  // (the spare rows of the last wave redo the last ray)
  const RayFrame rf = load_ray(rays_o, rays_d, r);
  const float * nrow = noise ? noise + (int64_t)r * S : nullptr;
  const float bias0 = b0[0];
  RowScan ns = {0.f, 0.f, 0.f, 0.f, 0.f}, ds = {0.f, 0.f, 0.f, 0.f, 0.f};
  float lx = 0.f, ly = 0.f, lz = 0.f;  // last sample point of the previous stride
  int n_kept = 0;
  bool done = !has_ray;
  for (int k0 = 0; k0 < S; k0 += 16) {
    const int j = (k0 >> 4) & 3;
    const int k = k0 + m;
    const bool valid = k < S;
    // ---- the samples of this stride: make_stride, row by row
    float cum, dummy;
    if (nrow) {
      const float nz = valid ? nrow[k] : 0.f;
      cum = row_scan_step(nz, j, ns, dummy);
      const float block_last = row_last(dummy);  // (T3 + T2) + (T1 + T0) once j = 3
      if (j == 3) ns.carry = ns.carry + block_last;
    } else {
      cum = (float)(min(k, S - 1) + 1);
    }
    const float t = cum * step;
    const float mx = rf.dx * t, my = rf.dy * t, mz = rf.dz * t;
    const float px = rf.ox + mx, py = rf.oy + my, pz = rf.oz + mz;
    const float qx = row_shift_up1(px, lx), qy = row_shift_up1(py, ly), qz = row_shift_up1(pz, lz);
    const float ex = px - qx, ey = py - qy, ez = pz - qz;
    const float dt = (k == 0) ? 0.f : sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
    lx = row_last(px);
    ly = row_last(py);
    lz = row_last(pz);
    // ---- density, as in density_march_kernel
    float x = px, y = py, z = pz;
    contract_point(x, y, z);
    float logit = bias0;
    for (int l = 0; l < L; l++) {
      const LevelParams lp = load_level(primes, bias, mul, l);
      uint32_t row[8];
      float w[8], acc[F];
      corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);
      gather_blend<F>(table + level_stride * l, row, w, acc);
#pragma unroll
      for (int kk = 0; kk < F; kk++) logit = fmaf(round_f16(acc[kk]), w0[l * F + kk], logit);
    }
    const float sigma = expf(logit - density_shift);
    const float sec = valid ? sigma * dt : 0.f;
    // ---- exclusive optical depth: depth_carry + wave_shift_up1(incl, 0) of the 64-lane scan
    float incl;
    row_scan_step(sec, j, ds, incl);
    const float prev = row_shift_up1(incl, (j == 0) ? 0.f : ds.last);
    const float depth = ds.carry + prev;
    ds.last = row_last(incl);
    if (j == 3) ds.carry = ds.carry + ds.last;
    const float trans = expf(-depth);
    const bool keep = valid && !done && (trans > t_thresh);
    const unsigned long long mk = __ballot(keep);
    const int cnt = __popc((uint32_t)(mk >> (16 * q)) & 0xffffu);
    if (!done) {
      n_kept += cnt;
      if (cnt < min(16, S - k0)) done = true;  // the mask is a prefix: nothing later survives
    }
    if (__ballot(!done) == 0ull) break;
  }
  if (has_ray && m == 0) kept[r_raw] = n_kept;
}

// ---- eight rays per wavefront --------------------------------------------------------------------
// A ray that stops after three or four samples still pays for the 16 of its first stride above.
// Here a ray owns HALF a DPP row (8 lanes) and strides are 8 samples: half the evaluations where
// rays stop early.  The 64-lane scan's additions are reproduced as before, now half a row at a
// time: the first half of a row is the row scan's steps 1, 2, 4 (step 8 adds nothing below lane 8);
// the second half needs, for its first lanes, what the first half's lanes 7 / 6,7 / 4..7 held after
// steps 0 / 1 / 2 (kept in registers, fetched with row_shl) and adds the first half's own result at
// step 8.  Same counts, bit for bit (tests/test_gpu_fused.py).
struct HalfScan
{
  float carry, t0, t1p, t2, last;  // as RowScan
  float pv, pa, pb, pc;            // the first half's values after steps 0, 1, 2, 4 (per lane)
};

// lane 7 of the own 8-lane group in every lane of the group
__device__ __forceinline__ float group_last(float v, int lane)
{
  const float lo = dpp_get<0x157, 0xf, 0xf>(v, 0.f), hi = dpp_get<0x15F, 0xf, 0xf>(v, 0.f);
  return (lane & 8) ? hi : lo;
}

// value after the row scan (16 lanes of the 64-lane scan) of the element this lane holds: m = lane
// within the group = lane within the half row, h = which half of the row this stride is
__device__ __forceinline__ float half_row_scan(float v, int m, int h, HalfScan & st)
{
  // (every DPP move is executed by ALL lanes and selected afterwards: inside a conditional
  // expression it would run with the other lanes switched off, and a lane that reads a switched-off
  // lane gets the fill value)
  float a, b, c;
  if (h == 0) {  // (wave-uniform)
    const float s1 = dpp_get<0x111, 0xf, 0xf>(v, 0.f);
    a = v + (m >= 1 ? s1 : 0.f);
    const float s2 = dpp_get<0x112, 0xf, 0xf>(a, 0.f);
    b = a + (m >= 2 ? s2 : 0.f);
    const float s4 = dpp_get<0x114, 0xf, 0xf>(b, 0.f);
    c = b + (m >= 4 ? s4 : 0.f);
    st.pv = v;
    st.pa = a;
    st.pb = b;
    st.pc = c;
    return c;
  }
  const float s1 = dpp_get<0x111, 0xf, 0xf>(v, 0.f), f1 = dpp_get<0x107, 0xf, 0xf>(st.pv, 0.f);
  a = v + (m >= 1 ? s1 : f1);
  const float s2 = dpp_get<0x112, 0xf, 0xf>(a, 0.f), f2 = dpp_get<0x106, 0xf, 0xf>(st.pa, 0.f);
  b = a + (m >= 2 ? s2 : f2);
  const float s4 = dpp_get<0x114, 0xf, 0xf>(b, 0.f), f4 = dpp_get<0x104, 0xf, 0xf>(st.pb, 0.f);
  c = b + (m >= 4 ? s4 : f4);
  return c + st.pc;
}

// as row_scan_step, for half j * 2 + h of the 64-sample block
__device__ __forceinline__ float half_scan_step(
  float v, int j, int h, int m, int lane, HalfScan & st, float & incl_in_block)
{
  const float rs = half_row_scan(v, m, h, st);
  float inner;
  if (j == 0) inner = rs;
  else if (j == 1) inner = rs + st.t0;
  else if (j == 2) inner = rs + st.t1p;
  else inner = (rs + st.t2) + st.t1p;
  if (h == 1) {  // the row is complete: its total
    const float tj = group_last(rs, lane);
    if (j == 0) st.t0 = tj;
    else if (j == 1) st.t1p = tj + st.t0;
    else if (j == 2) st.t2 = tj;
  }
  incl_in_block = inner;
  return st.carry + inner;
}

template <int F, bool POW2>
__global__ __launch_bounds__(F2N_BLOCK) void density_march8_kernel(
  const float * __restrict__ rays_o, const float * __restrict__ rays_d,
  const float * __restrict__ noise, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, const float * __restrict__ w0, const float * __restrict__ b0,
  int32_t * __restrict__ kept, int n_rays, int S, float step, int L, uint32_t T,
  int64_t level_stride, float t_thresh, float density_shift)
{
  const int lane = lane_id(), g = lane >> 3, m = lane & 7;
  const int r_raw = (ray_of_wave() << 3) + g;
  const bool has_ray = r_raw < n_rays;
  const int r = has_ray ? r_raw : n_rays - 1;  // (the spare groups of the last wave redo the last ray)
  const RayFrame rf = load_ray(rays_o, rays_d, r);
  const float * nrow = noise ? noise + (int64_t)r * S : nullptr;
  const float bias0 = b0[0];
  HalfScan ns = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ds = ns;
  float lx = 0.f, ly = 0.f, lz = 0.f;  // last sample point of the previous stride
  int n_kept = 0;
  bool done = !has_ray;
  for (int k0 = 0; k0 < S; k0 += 8) {
    const int j = (k0 >> 4) & 3, h = (k0 >> 3) & 1;
    const int k = k0 + m;
    const bool valid = k < S;
    float cum, dummy;
    if (nrow) {
      const float nz = valid ? nrow[k] : 0.f;
      cum = half_scan_step(nz, j, h, m, lane, ns, dummy);
      if (j == 3 && h == 1) ns.carry = ns.carry + group_last(dummy, lane);  // (T3 + T2) + (T1 + T0)
    } else {
      cum = (float)(min(k, S - 1) + 1);
    }
    const float t = cum * step;
    const float mx = rf.dx * t, my = rf.dy * t, mz = rf.dz * t;
    const float px = rf.ox + mx, py = rf.oy + my, pz = rf.oz + mz;
    const float sx = dpp_get<0x111, 0xf, 0xf>(px, 0.f), sy = dpp_get<0x111, 0xf, 0xf>(py, 0.f),
                sz = dpp_get<0x111, 0xf, 0xf>(pz, 0.f);  // (moved by all lanes, then selected)
    const float qx = m >= 1 ? sx : lx, qy = m >= 1 ? sy : ly, qz = m >= 1 ? sz : lz;
    const float ex = px - qx, ey = py - qy, ez = pz - qz;
    const float dt = (k == 0) ? 0.f : sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
    lx = group_last(px, lane);
    ly = group_last(py, lane);
    lz = group_last(pz, lane);
    float x = px, y = py, z = pz;
    contract_point(x, y, z);
    float logit = bias0;
    for (int l = 0; l < L; l++) {
      const LevelParams lp = load_level(primes, bias, mul, l);
      uint32_t row[8];
      float w[8], acc[F];
      corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);
      gather_blend<F>(table + level_stride * l, row, w, acc);
#pragma unroll
      for (int kk = 0; kk < F; kk++) logit = fmaf(round_f16(acc[kk]), w0[l * F + kk], logit);
    }
    const float sigma = expf(logit - density_shift);
    const float sec = valid ? sigma * dt : 0.f;
    float incl;
    half_scan_step(sec, j, h, m, lane, ds, incl);
    const float before = (j == 0 && h == 0) ? 0.f : ds.last;
    const float shifted = dpp_get<0x111, 0xf, 0xf>(incl, 0.f);
    const float prev = m >= 1 ? shifted : before;
    const float depth = ds.carry + prev;
    ds.last = group_last(incl, lane);
    if (j == 3 && h == 1) ds.carry = ds.carry + ds.last;
    const float trans = expf(-depth);
    const bool keep = valid && !done && (trans > t_thresh);
    const unsigned long long mk = __ballot(keep);
    const int cnt = __popc((uint32_t)(mk >> (8 * g)) & 0xffu);
    if (!done) {
      n_kept += cnt;
      if (cnt < min(8, S - k0)) done = true;  // the mask is a prefix: nothing later survives
    }
    if (__ballot(!done) == 0ull) break;
  }
  if (has_ray && m == 0) kept[r_raw] = n_kept;
}

// ---- f2n_density_scan ---------------------------------------------------------------------------
// The keep-prefix of every ray from an ALREADY COMPUTED encoding of all its samples (channel-major
// [C, n_all]): same logit FMA chain, same scan and same threshold test as density_march_kernel, so
// both give identical counts.  Used when most samples survive anyway: the encoding is then computed
// once by the level-major f2n_hash_fwd (L2-resident table levels) and reused by the shading pass,
// instead of being evaluated by the march and again by the second pass.
template <int C>
__global__ __launch_bounds__(F2N_BLOCK) void density_scan_kernel(
  const float * __restrict__ enc, const float * __restrict__ dt, const float * __restrict__ w0,
  const float * __restrict__ b0, int32_t * __restrict__ kept, int n_rays, int S, int64_t n_all,
  float t_thresh, float density_shift)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const float bias0 = b0[0];
  const int64_t base = (int64_t)r * S;
  float depth_carry = 0.f;
  int n_kept = 0;
  for (int k0 = 0; k0 < S; k0 += F2N_WAVE) {
    const int k = k0 + lane;
    const bool valid = k < S;
    const int64_t i = base + (valid ? k : S - 1);
    float logit = bias0;
#pragma unroll
    for (int c = 0; c < C; c++) logit = fmaf(enc[(int64_t)c * n_all + i], w0[c], logit);
    const float sigma = expf(logit - density_shift);
    const float sec = valid ? sigma * dt[i] : 0.f;
    const float incl = wave_incl_scan(sec);
    const float depth = depth_carry + wave_shift_up1(incl, 0.f);
    const float trans = expf(-depth);
    const bool keep = valid && (trans > t_thresh);
    const unsigned long long m = __ballot(keep);
    n_kept += __popcll(m);
    const int n_valid = min(F2N_WAVE, S - k0);
    if (__popcll(m) < n_valid) break;
    depth_carry += wave_bcast_last(incl);
  }
  if (lane == 0) kept[r] = n_kept;
}

// ---- f2n_density_margin -------------------------------------------------------------------------
// Does every ray of a dense [n_rays, S] grid keep all its samples with room to spare?  One wavefront
// per ray sums the optical depth of its S density logits; a ray whose total reaches `depth_limit`
// (or is not a number) sets flag[0].  The caller chooses depth_limit = -ln(threshold) - margin, so a
// clear flag means the exact early-stop scan (f2n_density_scan, whose FMA order differs in the last
// bits) would keep everything too.
__global__ __launch_bounds__(F2N_BLOCK) void density_margin_kernel(
  const float * __restrict__ logit, const float * __restrict__ dt, int32_t * __restrict__ flag,
  int n_rays, int S, float density_shift, float depth_limit)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int64_t base = (int64_t)r * S;
  float sum = 0.f;
  for (int k = lane; k < S; k += F2N_WAVE) sum += expf(logit[base + k] - density_shift) * dt[base + k];
  sum = wave_sum(sum);
  if (lane == 0 && !(sum < depth_limit)) atomicOr(flag, 1);
}

// ---- f2n_compact_rows_cm ------------------------------------------------------------------------
// Channel-major compaction of per-ray prefixes: dst[c, new_start_r + k] = src[c, r*S + k], k < cnt_r.
__global__ __launch_bounds__(F2N_BLOCK) void compact_rows_cm_kernel(
  const float * __restrict__ src, int64_t n_src, float * __restrict__ dst, int64_t n_dst, int C,
  const int32_t * __restrict__ bounds, int n_rays, int S)
{
  const int r = ray_of_wave();
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int start = bounds[2 * r];
  const int cnt = bounds[2 * r + 1] - start;
  const int64_t sbase = (int64_t)r * S;
  for (int k = lane; k < cnt; k += F2N_WAVE)
    for (int c = 0; c < C; c++)
      dst[(int64_t)c * n_dst + start + k] = src[(int64_t)c * n_src + sbase + k];
}

// ---- f2n_bounds_from_counts ---------------------------------------------------------------------

constexpr int kScanBlock = 1024;
constexpr int kScanItems = 16;  // per thread and pass: the single workgroup is bound by the latency of
                                // its passes (65 536 rays: 4 passes instead of 16, 39 -> ~12 us)

__global__ __launch_bounds__(kScanBlock) void bounds_from_counts_kernel(
  const int32_t * __restrict__ kept, int32_t * __restrict__ bounds, int32_t * __restrict__ total,
  int n_rays)
{
  __shared__ int wave_tot[kScanBlock / F2N_WAVE];
  __shared__ int tile_tot;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool pair_aligned = (reinterpret_cast<uintptr_t>(bounds) & 7u) == 0;
  int carry = 0;
  for (int base = 0; base < n_rays; base += kScanBlock * kScanItems) {
    const int i0 = base + tid * kScanItems;
    int v[kScanItems];
    int local = 0;
    if (i0 + kScanItems <= n_rays && (reinterpret_cast<uintptr_t>(kept) & 15u) == 0) {
#pragma unroll
      for (int j = 0; j < kScanItems; j += 4) {
        const int4 q = *reinterpret_cast<const int4 *>(kept + i0 + j);
        v[j] = q.x;
        v[j + 1] = q.y;
        v[j + 2] = q.z;
        v[j + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kScanItems; j++) v[j] = (i0 + j < n_rays) ? kept[i0 + j] : 0;
    }
#pragma unroll
    for (int j = 0; j < kScanItems; j++) local += v[j];
    const int incl = wave_incl_scan_i32(local);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    if (wv == 0) {
      const int wt = (lane < kScanBlock / F2N_WAVE) ? wave_tot[lane] : 0;
      const int wi = wave_incl_scan_i32(wt);
      if (lane < kScanBlock / F2N_WAVE) wave_tot[lane] = wi - wt;  // exclusive wave offsets
      if (lane == 63) tile_tot = wi;
    }
    __syncthreads();
    int run = carry + wave_tot[wv] + (incl - local);
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
      if (i0 + j < n_rays) {
        if (pair_aligned) {
          *reinterpret_cast<int2 *>(bounds + 2 * (i0 + j)) = make_int2(run, run + v[j]);
        } else {
          bounds[2 * (i0 + j)] = run;
          bounds[2 * (i0 + j) + 1] = run + v[j];
        }
      }
      run += v[j];
    }
    carry += tile_tot;
    __syncthreads();
  }
  if (tid == 0 && total) total[0] = carry;
}

// Up to 2^17 rays: one workgroup per 1024 rays.  Each re-derives its own offset -- the sum of all
// counts before its rays, read straight from `kept` (256 KiB for 65 536 rays: L2-resident; 8 MB of L2
// reads over all workgroups) -- so there is neither a second launch nor a hand-off between
// workgroups, and the 0.75 MB of loads and stores no longer pass through one CU (65 536 rays:
// 42 -> ~6 us).  Integer sums: the same bounds whatever the order.
constexpr int kMultiScanBlock = 256, kMultiScanItems = 4;
constexpr int kMultiScanRays = kMultiScanBlock * kMultiScanItems;
constexpr int kMultiScanMaxRays = 1 << 17;

__global__ __launch_bounds__(kMultiScanBlock) void bounds_from_counts_multi_kernel(
  const int32_t * __restrict__ kept, int32_t * __restrict__ bounds, int32_t * __restrict__ total,
  int n_rays)
{
  constexpr int kWaves = kMultiScanBlock / F2N_WAVE;
  __shared__ int wave_part[kWaves], wave_tot[kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int first = (int)blockIdx.x * kMultiScanRays;  // (a multiple of 1024: int4 loads stay aligned)
  const bool vec = (reinterpret_cast<uintptr_t>(kept) & 15u) == 0;
  // ---- everything before this workgroup's rays
  int before = 0;
  if (vec) {
    for (int i = 4 * tid; i < first; i += 4 * kMultiScanBlock) {
      const int4 q = *reinterpret_cast<const int4 *>(kept + i);
      before += (q.x + q.y) + (q.z + q.w);
    }
  } else {
    for (int i = tid; i < first; i += kMultiScanBlock) before += kept[i];
  }
  before = __builtin_amdgcn_readlane(wave_incl_scan_i32(before), 63);
  // ---- this workgroup's rays
  const int i0 = first + tid * kMultiScanItems;
  int v[kMultiScanItems];
  int local = 0;
#pragma unroll
  for (int j = 0; j < kMultiScanItems; j++) {
    v[j] = (i0 + j < n_rays) ? kept[i0 + j] : 0;
    local += v[j];
  }
  const int incl = wave_incl_scan_i32(local);
  if (lane == 63) wave_tot[wv] = incl;
  if (lane == 0) wave_part[wv] = before;
  __syncthreads();
  int offset = 0, lower = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    offset += wave_part[w];
    if (w < wv) lower += wave_tot[w];
    all += wave_tot[w];
  }
  int run = offset + lower + (incl - local);
  const bool pair_aligned = (reinterpret_cast<uintptr_t>(bounds) & 7u) == 0;
#pragma unroll
  for (int j = 0; j < kMultiScanItems; j++) {
    if (i0 + j < n_rays) {
      if (pair_aligned) {
        *reinterpret_cast<int2 *>(bounds + 2 * (i0 + j)) = make_int2(run, run + v[j]);
      } else {
        bounds[2 * (i0 + j)] = run;
        bounds[2 * (i0 + j) + 1] = run + v[j];
      }
    }
    run += v[j];
  }
  if (total && blockIdx.x == gridDim.x - 1 && tid == 0) total[0] = offset + all;
}

inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1u)); }

}  // namespace

extern "C" int f2n_sample_rays(
  const float * rays_o, const float * rays_d, const float * noise, float * pts, float * dirs,
  float * dt, float * t, int32_t * bounds, int n_rays, int S, float step, void * stream)
{
  if (n_rays < 0 || S < 1) return F2N_E_INVALID_ARG;
  if ((int64_t)n_rays * S > INT32_MAX) return F2N_E_INVALID_ARG;  // bounds are int32
  if (n_rays == 0) return F2N_OK;
  if (!rays_o || !rays_d || !pts || !dirs || !dt || !t || !bounds) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    sample_rays_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, rays_o, rays_d, noise, pts, dirs, dt, t, bounds, n_rays, S, step);
  return f2n_launch_status();
}

extern "C" int f2n_sample_compact(
  const float * rays_o, const float * rays_d, const float * noise, const int32_t * bounds,
  float * pts, float * dirs, float * dt, float * t, int n_rays, int S, float step, void * stream)
{
  if (n_rays < 0 || S < 1) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  if (!rays_o || !rays_d || !bounds) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    sample_compact_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, rays_o, rays_d, noise, bounds, pts, dirs, dt, t, n_rays, S, step);
  return f2n_launch_status();
}

extern "C" int f2n_density_march(
  const float * rays_o, const float * rays_d, const float * noise, const uint16_t * table_f16,
  const int32_t * primes, const float * bias, const float * mul, const float * w0, const float * b0,
  int32_t * kept, int n_rays, int S, float step, int L, int F, uint32_t T, int64_t level_stride,
  float t_thresh, float density_shift, void * stream)
{
  if (n_rays < 0 || S < 1 || L < 1 || L > F2N_MAX_LEVELS || T < 1 || level_stride < 0)
    return F2N_E_INVALID_ARG;
  if (F != 1 && F != 2 && F != 4 && F != 8) return F2N_E_UNSUPPORTED;
  if (level_stride % F) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  if (!rays_o || !rays_d || !table_f16 || !primes || !bias || !mul || !w0 || !b0 || !kept)
    return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(table_f16) % (2u * F)) return F2N_E_INVALID_ARG;
  // eight rays per wavefront (8-sample strides) unless F2N_OPT_MARCH asks for one (1: 64-sample
  // strides, round 2) or four (2: 16-sample strides)
  const int route = f2n_get_option(F2N_OPT_MARCH);
  const bool rows = route == 2;
  const int rays_per_wave = route == 1 ? 1 : route == 2 ? 4 : 8;
  const dim3 grid(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK * rays_per_wave)), block(F2N_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  const bool p2 = is_pow2(T);
#define F2N_MARCH(FF, P2)                                                                         \
  if (route == 0)                                                                                 \
    hipLaunchKernelGGL(                                                                           \
      (density_march8_kernel<FF, P2>), grid, block, 0, s, rays_o, rays_d, noise, table_f16,       \
      primes, bias, mul, w0, b0, kept, n_rays, S, step, L, T, level_stride, t_thresh,             \
      density_shift);                                                                             \
  else if (rows)                                                                                  \
    hipLaunchKernelGGL(                                                                           \
      (density_march16_kernel<FF, P2>), grid, block, 0, s, rays_o, rays_d, noise, table_f16,      \
      primes, bias, mul, w0, b0, kept, n_rays, S, step, L, T, level_stride, t_thresh,             \
      density_shift);                                                                             \
  else                                                                                            \
    hipLaunchKernelGGL(                                                                           \
      (density_march_kernel<FF, P2>), grid, block, 0, s, rays_o, rays_d, noise, table_f16, primes,\
      bias, mul, w0, b0, kept, n_rays, S, step, L, T, level_stride, t_thresh, density_shift)
  switch (F) {
    case 1: if (p2) F2N_MARCH(1, true); else F2N_MARCH(1, false); break;
    case 2: if (p2) F2N_MARCH(2, true); else F2N_MARCH(2, false); break;
    case 4: if (p2) F2N_MARCH(4, true); else F2N_MARCH(4, false); break;
    default: if (p2) F2N_MARCH(8, true); else F2N_MARCH(8, false); break;
  }
#undef F2N_MARCH
  return f2n_launch_status();
}

extern "C" int f2n_density_scan(
  const float * enc_cm, int C, const float * dt, const float * w0, const float * b0, int32_t * kept,
  int n_rays, int S, float t_thresh, float density_shift, void * stream)
{
  if (n_rays < 0 || S < 1) return F2N_E_INVALID_ARG;
  if (C != 8 && C != 16 && C != 32 && C != 64 && C != 128) return F2N_E_UNSUPPORTED;
  if (n_rays == 0) return F2N_OK;
  if (!enc_cm || !dt || !w0 || !b0 || !kept) return F2N_E_INVALID_ARG;
  const dim3 grid(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), block(F2N_BLOCK);
  const int64_t n_all = (int64_t)n_rays * S;
  hipStream_t s = (hipStream_t)stream;
#define F2N_SCAN(CC)                                                                              \
  hipLaunchKernelGGL(                                                                             \
    (density_scan_kernel<CC>), grid, block, 0, s, enc_cm, dt, w0, b0, kept, n_rays, S, n_all,     \
    t_thresh, density_shift)
  switch (C) {
    case 8: F2N_SCAN(8); break;
    case 16: F2N_SCAN(16); break;
    case 32: F2N_SCAN(32); break;
    case 64: F2N_SCAN(64); break;
    default: F2N_SCAN(128); break;
  }
#undef F2N_SCAN
  return f2n_launch_status();
}

extern "C" int f2n_density_margin(
  const float * logit, const float * dt, int32_t * flag, int n_rays, int S, float density_shift,
  float depth_limit, void * stream)
{
  if (n_rays < 0 || S < 1) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  if (!logit || !dt || !flag) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    density_margin_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, logit, dt, flag, n_rays, S, density_shift, depth_limit);
  return f2n_launch_status();
}

extern "C" int f2n_compact_rows_cm(
  const float * src, int64_t n_src, float * dst, int64_t n_dst, int C, const int32_t * bounds,
  int n_rays, int S, void * stream)
{
  if (n_rays < 0 || S < 1 || C < 1 || n_src < 0 || n_dst < 0) return F2N_E_INVALID_ARG;
  if (n_rays == 0 || n_dst == 0) return F2N_OK;
  if (!src || !dst || !bounds) return F2N_E_INVALID_ARG;
  hipLaunchKernelGGL(
    compact_rows_cm_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, src, n_src, dst, n_dst, C, bounds, n_rays, S);
  return f2n_launch_status();
}

extern "C" int f2n_bounds_from_counts(
  const int32_t * kept, int32_t * bounds, int32_t * total, int n_rays, void * stream)
{
  if (n_rays < 0 || n_rays > (1 << 24)) return F2N_E_INVALID_ARG;
  if (n_rays > 0 && (!kept || !bounds)) return F2N_E_INVALID_ARG;
  if (n_rays > kMultiScanRays && n_rays <= kMultiScanMaxRays)
    hipLaunchKernelGGL(
      bounds_from_counts_multi_kernel, dim3(f2n_div_up(n_rays, kMultiScanRays)),
      dim3(kMultiScanBlock), 0, (hipStream_t)stream, kept, bounds, total, n_rays);
  else
    hipLaunchKernelGGL(
      bounds_from_counts_kernel, dim3(1), dim3(kScanBlock), 0, (hipStream_t)stream, kept, bounds,
      total, n_rays);
  return f2n_launch_status();
}
