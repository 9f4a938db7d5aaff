// hash_grid.hip -- multi-resolution hash-grid encode, forward and backward.
//
// Replaces Hash3DAnchoredForwardKernel / Hash3DAnchoredBackwardKernel and the dtype glue around
// them (reference src/hash_3d_anchored.cu:60-218) plus the contraction of
// Hash3DAnchored::query (src/hash_3d_anchored.cpp:79-82).
//
// Launch shape: grid = (ceil(n/256), L), block = 256 (4 wavefronts).  blockIdx.y = level, so a
// workgroup is level-uniform: mul/bias/primes/level base live in SGPRs (scalar loads), and because
// x is the fast grid dimension the chip works through one level at a time -- the live part of the
// table is one level (T*F*2 bytes: 2 MiB at the reference size), which an XCD's 4 MiB L2 holds.
#include "hash_grid.hiph"


namespace
{

// ---------------------------------------------------------------------------- table cast -------

__global__ __launch_bounds__(F2N_BLOCK) void table_to_f16_kernel(
  const float * __restrict__ in, uint16_t * __restrict__ out, int64_t n)
{
  // 8 elements per thread: 2 x 16-B loads, 1 x 16-B store
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4 *>(in + i);
      const float4 b = *reinterpret_cast<const float4 *>(in + i + 4);
      uint4 o;
      o.x = (uint32_t)__half_as_ushort(__float2half_rn(a.x)) |
            ((uint32_t)__half_as_ushort(__float2half_rn(a.y)) << 16);
      o.y = (uint32_t)__half_as_ushort(__float2half_rn(a.z)) |
            ((uint32_t)__half_as_ushort(__float2half_rn(a.w)) << 16);
      o.z = (uint32_t)__half_as_ushort(__float2half_rn(b.x)) |
            ((uint32_t)__half_as_ushort(__float2half_rn(b.y)) << 16);
      o.w = (uint32_t)__half_as_ushort(__float2half_rn(b.z)) |
            ((uint32_t)__half_as_ushort(__float2half_rn(b.w)) << 16);
      *reinterpret_cast<uint4 *>(out + i) = o;
    } else {
      for (int64_t j = i; j < n; j++) out[j] = __half_as_ushort(__float2half_rn(in[j]));
    }
  }
}

// ---------------------------------------------------------------------------- forward ----------

template <int F, bool POW2>
__global__ __launch_bounds__(F2N_BLOCK) void hash_fwd_kernel(
  const float * __restrict__ pts, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, float * __restrict__ out, int64_t out_ld_point,
  int64_t out_ld_chan, uint32_t * __restrict__ idx_out, int64_t n, int L, uint32_t T,
  int64_t level_stride)
{
  const int l = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (p >= n) return;
  const LevelParams lp = load_level(primes, bias, mul, l);
  const float x = pts[3 * p + 0], y = pts[3 * p + 1], z = pts[3 * p + 2];
  uint32_t row[8];
  float w[8];
  corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);
  float acc[F];
  gather_blend<F>(table + level_stride * l, row, w, acc);
  float * o = out + p * out_ld_point + (int64_t)(l * F) * out_ld_chan;
#pragma unroll
  for (int k = 0; k < F; k++) o[k * out_ld_chan] = round_f16(acc[k]);
  if (idx_out) {
    uint32_t * io = idx_out + (p * L + l) * 8;
#pragma unroll
    for (int d = 0; d < 8; d++) io[d] = row[d];
  }
}


// Forward for a dense [n_rays, S] sample grid (ray-major in memory, as the sampler emits it) whose
// rays are neighbouring pixels.  Consecutive samples of ONE ray are ~40 finest-level cells apart,
// the same sample index on ADJACENT pixels ~1-4 cells: with lane = ray at a fixed depth the
// lanes of a wave share corners (identical rows coalesce in the texture unit) and mid levels stay in
// L1, which halves the 128-byte line requests that bound hash_fwd_kernel (2.4 -> 1.2 ms per 8.4 M
// samples of an 800-wide view).  Random ray batches gain nothing and lose nothing.
//
// One workgroup = 64 rays x 16 consecutive samples of one level: wave w walks samples 4w..4w+3 with
// lane = ray.  The [n] axis of the channel-major output is ray-major, so results go through an LDS
// tile and leave as 64-byte runs (16 samples of one ray); points are read directly (12 bytes per
// lane, 1/16 of the gather's line requests).
// One workgroup = 64 rays x SAMPLES consecutive samples of one level; wave w walks samples
// w, w+4, ... with lane = ray.  Points come in through LDS (coalesced runs of 12 * SAMPLES bytes per
// ray), results leave through LDS as runs of 4 * SAMPLES bytes per ray and channel.  (Wider ray
// tiles with shorter sample runs are slower: 128 x 8 1.73 ms, 256 x 4 2.5 ms vs 64 x 16 1.46 ms --
// the stores then leave as 32- and 16-byte pieces.)
template <int F, bool POW2, int SAMPLES>
__global__ __launch_bounds__(F2N_BLOCK) void hash_fwd_raytile_kernel(
  const float * __restrict__ pts, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, float * __restrict__ out, int n_rays, int S, uint32_t T,
  int64_t level_stride, int walk)
{
  static_assert(F2N_BLOCK == 256 && SAMPLES % 16 == 0, "4 waves, float4 runs");
  constexpr int RAYS = 64, kPitch = SAMPLES + 4, kPPitch = 3 * SAMPLES + 1;
  __shared__ __attribute__((aligned(16))) float tile[F][RAYS][kPitch];
  __shared__ float ptile[RAYS][kPPitch];
  const int l = blockIdx.y;
  const int tiles_s = S / SAMPLES;
  const int r0 = (int)(blockIdx.x / tiles_s) * RAYS, k0 = (int)(blockIdx.x % tiles_s) * SAMPLES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)n_rays * S;
#pragma unroll
  for (int i = 0; i < RAYS * 3 * SAMPLES / 256; i++) {
    const int idx = threadIdx.x + 256 * i;
    const int ray = idx / (3 * SAMPLES), off = idx % (3 * SAMPLES);
    const int r = min(r0 + ray, n_rays - 1);  // clamped: tail rays redo the last ray, never stored
    ptile[ray][off] = pts[((int64_t)r * S + k0) * 3 + off];
  }
  const LevelParams lp = load_level(primes, bias, mul, l);
  __syncthreads();
  // Which way do the 64 lanes of a gather walk the tile?  The cost of a gather instruction is its
  // number of distinct 128-byte lines (tools/probes/gather_policy.hip), i.e. of distinct cells:
  //   across  lane = ray, all at one sample index: neighbouring pixels of a view sit 1-4 fine cells
  //           apart at one depth (image-ordered batches, the renderer's whole-view chunks);
  //   along   lane = consecutive samples of one ray (64 / SAMPLES rays per instruction): the
  //           reference's 1024 steps of 1/256 put up to 32 consecutive samples in one coarse cell
  //           (random training rays: neighbouring rays share nothing).
  // Decided per tile from its own points: the walk with the smaller extent touches fewer cells.
  bool along, drifted;
  {
    // Distinct cells of size c touched by one gather: across ~ (e / c + 1) for rays strung along a
    // pixel row of extent e, times the depth cloud the TRAIN jitter adds (neighbouring rays then sit
    // a random e1 apart at one sample index); along ~ 2 rays x (s / c + 1) for a ray segment of
    // length s.  Across wins when e + 8 e1 < 2 s: image-ordered batches, jittered (0.15 + 8 x 0.07
    // against 2 x 1.0 at 128 samples) or not, near or far; random rays (e, e1 ~ 1.5) walk along.
    constexpr int m = SAMPLES / 2;
    auto dist = [&](int ra, int sa, int rb, int sb) {
      const float dx = ptile[ra][3 * sa] - ptile[rb][3 * sb], dy = ptile[ra][3 * sa + 1] - ptile[rb][3 * sb + 1],
                  dz = ptile[ra][3 * sa + 2] - ptile[rb][3 * sb + 2];
      return sqrtf(dx * dx + dy * dy + dz * dz);
    };
    const float e = dist(RAYS - 1, m, 0, m), e1 = dist(1, m, 0, m), sl = dist(0, SAMPLES - 1, 0, 0);
    along = !(e + 8.f * e1 < 2.f * sl);
    if (walk) along = (walk == 2);
    // neighbouring rays more than a third of a step apart at one sample index: jittered depths
    // (un-jittered views sit a pixel apart there, 0.06 steps: nothing to sort, 4 % to lose)
    // (three pairs of neighbours: one pair is that close by chance in a fifth of the jittered tiles)
    const float apart = fmaxf(e1, fmaxf(dist(RAYS / 2 + 1, m, RAYS / 2, m), dist(RAYS - 1, m, RAYS - 2, m)));
    drifted = apart * (float)(SAMPLES - 1) > 0.35f * sl;
  }
  // Across, third variant (walk 3, and the default when a tile that walks across has drifted): the tile's
  // 64 x SAMPLES (ray, sample) pairs in DEPTH order.  TRAIN jitter lets the rays of a tile drift
  // +-2.5 steps apart in depth by mid-ray, so "all rays at one sample index" is a cloud 5 steps deep
  // and the mid levels (cells of 0.3-1 step) stop sharing lines; 64 pairs taken from a counting sort
  // by depth along the middle ray (64 bins of half a step) form a slab one step thick instead.  The
  // order of the pairs means nothing to the result.
  __shared__ uint16_t order[RAYS * SAMPLES];
  __shared__ uint32_t bin_at[64];
  const bool sorted = !along && (walk == 3 || (walk == 0 && drifted));
  if (sorted) {
    constexpr int kPairs = RAYS * SAMPLES / 256;  // per thread
    constexpr int mr = RAYS / 2;
    const float ax = ptile[mr][3 * (SAMPLES - 1)] - ptile[mr][0],
                ay = ptile[mr][3 * (SAMPLES - 1) + 1] - ptile[mr][1],
                az = ptile[mr][3 * (SAMPLES - 1) + 2] - ptile[mr][2];
    const float inv = 64.f / fmaxf(ax * ax + ay * ay + az * az, 1e-30f);
    const float ox = ptile[mr][0], oy = ptile[mr][1], oz = ptile[mr][2];
    if (threadIdx.x < 64) bin_at[threadIdx.x] = 0u;
    __syncthreads();
    int my_bin[kPairs];
#pragma unroll
    for (int i = 0; i < kPairs; i++) {
      const int pair = threadIdx.x + 256 * i, ray = pair / SAMPLES, ks = pair % SAMPLES;
      const float d = (ptile[ray][3 * ks] - ox) * ax + (ptile[ray][3 * ks + 1] - oy) * ay +
                      (ptile[ray][3 * ks + 2] - oz) * az;
      my_bin[i] = min(max((int)(d * inv), 0), 63);  // (a NaN point lands in bin 0)
      atomicAdd(&bin_at[my_bin[i]], 1u);
    }
    __syncthreads();
    if (wave == 0) {
      const int c = (int)bin_at[lane];
      bin_at[lane] = (uint32_t)(wave_incl_scan_i32(c) - c);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPairs; i++)
      order[atomicAdd(&bin_at[my_bin[i]], 1u)] = (uint16_t)(threadIdx.x + 256 * i);
    __syncthreads();
  }
  constexpr int kRaysPerIter = 64 / SAMPLES;  // along: rays one wave instruction covers
#pragma unroll
  for (int j = 0; j < SAMPLES / 4; j++) {
    int ks = along ? (lane % SAMPLES) : (j * 4 + wave);
    int ray = along ? ((j * 4 + wave) * kRaysPerIter + lane / SAMPLES) : lane;
    if (sorted) {
      const int pair = order[(j * 4 + wave) * 64 + lane];
      ray = pair / SAMPLES;
      ks = pair % SAMPLES;
    }
    const float x = ptile[ray][3 * ks], y = ptile[ray][3 * ks + 1], z = ptile[ray][3 * ks + 2];
    uint32_t row[8];
    float w[8];
    corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);
    float acc[F];
    gather_blend<F>(table + level_stride * l, row, w, acc);
#pragma unroll
    for (int k = 0; k < F; k++) tile[k][ray][ks] = round_f16(acc[k]);
  }
  __syncthreads();
  // RAYS x (SAMPLES / 4) float4 per channel, SAMPLES / 16 per thread
#pragma unroll
  for (int i = 0; i < SAMPLES / 16; i++) {
    const int idx = threadIdx.x + 256 * i;
    const int ray = idx / (SAMPLES / 4), quad = idx % (SAMPLES / 4);
    if (r0 + ray < n_rays) {
#pragma unroll
      for (int k = 0; k < F; k++) {
        const float4 v = *reinterpret_cast<const float4 *>(&tile[k][ray][4 * quad]);
        *reinterpret_cast<float4 *>(
          out + (int64_t)(l * F + k) * n + (int64_t)(r0 + ray) * S + k0 + 4 * quad) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------- backward ---------

// v1: one thread per (point, level); 8*F f32 atomics into the table gradient.  Contributions are
// f16(f16(scale*g) * w_d) exactly as the reference forms them; the running sum is f32 and already
// divided by the (power-of-two) scale, so no epilogue pass over the table is needed.
template <int F, bool POW2, bool WITH_PTS_GRAD>
__global__ __launch_bounds__(F2N_BLOCK) void hash_bwd_kernel(
  const float * __restrict__ pts, const uint16_t * __restrict__ table,
  const int32_t * __restrict__ primes, const float * __restrict__ bias,
  const float * __restrict__ mul, const float * __restrict__ grad_out, int64_t g_ld_point,
  int64_t g_ld_chan, float * __restrict__ table_grad, float * __restrict__ pts_grad, int64_t n,
  uint32_t T, int64_t level_stride, float grad_scale, float inv_scale)
{
  const int l = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (p >= n) return;
  const LevelParams lp = load_level(primes, bias, mul, l);
  const float x = pts[3 * p + 0], y = pts[3 * p + 1], z = pts[3 * p + 2];
  uint32_t row[8];
  float w[8];
  corner_rows_and_weights<POW2>(x, y, z, lp, T, row, w);

  const float * g = grad_out + p * g_ld_point + (int64_t)(l * F) * g_ld_chan;
  float gk[F];
  bool any = false;
#pragma unroll
  for (int k = 0; k < F; k++) {
    gk[k] = round_f16(g[k * g_ld_chan] * grad_scale);
    any |= (gk[k] != 0.f);
  }
  float * gbase = table_grad + level_stride * l;
  if (any) {  // reference skips all-zero channel pairs (:133); adding zeros changes nothing
#pragma unroll
    for (int d = 0; d < 8; d++) {
#pragma unroll
      for (int k = 0; k < F; k++) {
        const float c = round_f16(gk[k] * w[d]) * inv_scale;
        atomicAdd(gbase + (int64_t)row[d] * F + k, c);
      }
    }
  }
  if (WITH_PTS_GRAD) {
    // quirk Q5: sign * feature * mul * g per corner and channel, each rounded to f16 (:138-143)
    using Row = typename RowBits<F>::type;
    const Row * rows = reinterpret_cast<const Row *>(table + level_stride * l);
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      float f[F];
      const Row r = rows[row[d]];
      unpack_row<F>(r, f);
#pragma unroll
      for (int k = 0; k < F; k++) {
        const float nrm = f[k] * lp.mul * gk[k];
        const float pos = round_f16(nrm), neg = round_f16(-nrm);
        gx += (d & 4) ? pos : neg;
        gy += (d & 2) ? pos : neg;
        gz += (d & 1) ? pos : neg;
      }
    }
    atomicAdd(pts_grad + 3 * p + 0, gx * inv_scale);
    atomicAdd(pts_grad + 3 * p + 1, gy * inv_scale);
    atomicAdd(pts_grad + 3 * p + 2, gz * inv_scale);
  }
}

// v2 ("sliced"): the table gradient is built in LDS, not with global atomics.
//
// Scattered global float atomics run at ~2e10 requests/s on MI355X (each lane's add is its own
// 64-B request to the memory side; MI355X_MICROARCH.md "Global float atomics"), which made v1 the
// dominant kernel of a training step.  Here a workgroup owns one (level, slice) pair, a slice being
// kSliceFloats/F consecutive table rows held as f32 accumulators in 128 KiB of LDS.  It walks ALL
// points of its sample partition, recomputes the 8 hashed rows (integer VALU work, the points stream
// from L2 because the co-resident workgroups of an XCD walk the same range together) and applies only
// the corners that land in its slice with ds_add_f32.  At the end the slice is added into the global
// gradient with contiguous atomics (256 B per wave instruction = the full atomic rate; atomics
// because the reference's level windows overlap, quirk Q2).  The hash work is repeated n_slices
// times (32x at the reference size) -- it is cheap next to 2e9 scattered atomics.
constexpr int kSliceFloats = 32768;  // 128 KiB of f32 accumulators
constexpr int kSliceBlock = 1024;

template <int F, bool POW2>
__global__ __launch_bounds__(kSliceBlock) void hash_bwd_sliced_kernel(
  const float * __restrict__ pts, const int32_t * __restrict__ primes,
  const float * __restrict__ bias, const float * __restrict__ mul,
  const float * __restrict__ grad_out, int64_t g_ld_point, int64_t g_ld_chan,
  float * __restrict__ table_grad, int64_t n, uint32_t T, int64_t level_stride, float grad_scale,
  float inv_scale, int n_parts)
{
  __shared__ float acc[kSliceFloats];
  constexpr uint32_t kRows = kSliceFloats / F;  // rows per slice (power of two)
  const uint32_t slice = blockIdx.x;
  const int l = blockIdx.y;
  const int part = blockIdx.z;
  const uint32_t row_lo = slice * kRows;

  for (int i = threadIdx.x; i < kSliceFloats; i += kSliceBlock) acc[i] = 0.f;
  __syncthreads();

  const LevelParams lp = load_level(primes, bias, mul, l);
  const int64_t per = (n + n_parts - 1) / n_parts;
  const int64_t p_begin = per * part;
  const int64_t p_end = (p_begin + per < n) ? p_begin + per : n;
  const float * gcol = grad_out + (int64_t)(l * F) * g_ld_chan;

  // Software-pipelined walk: the point and its F gradient channels for iteration i+1 are requested
  // before iteration i is processed, so no load sits between the hash arithmetic and the LDS adds
  // (with one workgroup per CU there are only 4 waves per SIMD to hide memory latency otherwise).
  struct Item
  {
    float x, y, z;
    float g[F];
  };
  auto fetch = [&](int64_t p, Item & it) {
    it.x = pts[3 * p + 0];
    it.y = pts[3 * p + 1];
    it.z = pts[3 * p + 2];
    const float * g = gcol + p * g_ld_point;
#pragma unroll
    for (int k = 0; k < F; k++) it.g[k] = g[k * g_ld_chan];
  };
  auto apply = [&](const Item & it) {
    uint32_t row[8];
    float w[8];
    corner_rows_and_weights<POW2>(it.x, it.y, it.z, lp, T, row, w);
    float gk[F];
#pragma unroll
    for (int k = 0; k < F; k++) gk[k] = round_f16(it.g[k] * grad_scale);
#pragma unroll
    for (int d = 0; d < 8; d++) {
      const uint32_t local = row[d] - row_lo;
      if (local < kRows) {
#pragma unroll
        for (int k = 0; k < F; k++) {
          const float c = round_f16(gk[k] * w[d]);
          if (c != 0.f) atomicAdd(&acc[local * F + k], c);
        }
      }
    }
  };

  int64_t p = p_begin + threadIdx.x;
  if (p < p_end) {
    Item cur, nxt;
    fetch(p, cur);
    for (; p < p_end; p += kSliceBlock) {
      // unconditional (clamped) prefetch: no branch around the loads, so the only wait for them
      // is at the first use of `cur` in the next iteration
      const int64_t pn = (p + kSliceBlock < p_end) ? p + kSliceBlock : p;
      fetch(pn, nxt);
      apply(cur);
      cur = nxt;
    }
  }
  __syncthreads();

  // flush: slice rows [row_lo, row_lo + kRows) of level l, clipped to T
  float * gbase = table_grad + level_stride * l + (int64_t)row_lo * F;
  const uint32_t rows_here = (row_lo + kRows <= T) ? kRows : (T > row_lo ? T - row_lo : 0u);
  const int n_flush = (int)rows_here * F;
  for (int i = threadIdx.x; i < n_flush; i += kSliceBlock) {
    const float v = acc[i];
    if (v != 0.f) atomicAdd(gbase + i, v * inv_scale);
  }
}

// ---------------------------------------------------------------------------- contraction ------

__global__ __launch_bounds__(F2N_BLOCK) void contract_fwd_kernel(
  const float * __restrict__ pts, float * __restrict__ out, int64_t n)
{
  const int64_t p = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (p >= n) return;
  float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
  contract_point(x, y, z);
  out[3 * p] = x;
  out[3 * p + 1] = y;
  out[3 * p + 2] = z;
}

// d/dp of x = a(|p|) p with a = 2/n - 1/n^2 outside the unit ball, identity inside.
__global__ __launch_bounds__(F2N_BLOCK) void contract_bwd_kernel(
  const float * __restrict__ pts, const float * __restrict__ dx, float * __restrict__ dp, int64_t n)
{
  const int64_t p = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (p >= n) return;
  const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
  const float gx = dx[3 * p], gy = dx[3 * p + 1], gz = dx[3 * p + 2];
  const float n2 = fmaf(z, z, fmaf(y, y, x * x));
  const float nrm = sqrtf(n2);
  float ox, oy, oz;
  if (nrm <= 1.f) {
    const float poison = (nrm == 0.f) ? __builtin_nanf("") : 0.f;
    ox = gx + poison;
    oy = gy + poison;
    oz = gz + poison;
  } else {
    const float inv = 1.f / nrm;
    const float a = (2.f - inv) * inv;                       // 2/n - 1/n^2
    const float da_over_n = (2.f * inv - 2.f) * inv * inv * inv;  // a'(n)/n = (-2/n^2 + 2/n^3)/n
    const float pg = fmaf(z, gz, fmaf(y, gy, x * gx));
    const float c = da_over_n * pg;
    ox = fmaf(c, x, a * gx);
    oy = fmaf(c, y, a * gy);
    oz = fmaf(c, z, a * gz);
  }
  dp[3 * p] = ox;
  dp[3 * p + 1] = oy;
  dp[3 * p + 2] = oz;
}

inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1u)); }

}  // namespace

extern "C" int f2n_table_to_f16(
  const float * table_f32, uint16_t * table_f16, int64_t n, void * stream)
{
  if (!table_f32 || !table_f16 || n < 0) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  if ((reinterpret_cast<uintptr_t>(table_f32) & 15u) || (reinterpret_cast<uintptr_t>(table_f16) & 15u))
    return F2N_E_INVALID_ARG;
  const int64_t work = (n + 7) / 8;
  const unsigned grid = (unsigned)std::min<int64_t>((work + F2N_BLOCK - 1) / F2N_BLOCK, 256 * 16);
  hipLaunchKernelGGL(
    table_to_f16_kernel, dim3(grid), dim3(F2N_BLOCK), 0, (hipStream_t)stream, table_f32, table_f16,
    n);
  return f2n_launch_status();
}

#define F2N_DISPATCH_F(F_, ...)      \
  switch (F_) {                      \
    case 1: { constexpr int FF = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int FF = 2; __VA_ARGS__; } break; \
    case 4: { constexpr int FF = 4; __VA_ARGS__; } break; \
    case 8: { constexpr int FF = 8; __VA_ARGS__; } break; \
    default: return F2N_E_UNSUPPORTED; \
  }

extern "C" int f2n_hash_fwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, float * out, int64_t out_ld_point, int64_t out_ld_chan, uint32_t * idx_out,
  int64_t n, int L, int F, uint32_t T, int64_t level_stride, void * stream)
{
  if (!pts || !table_f16 || !primes || !bias || !mul || !out) return F2N_E_INVALID_ARG;
  if (F != 1 && F != 2 && F != 4 && F != 8) return F2N_E_UNSUPPORTED;
  if (!f2n_hash_args_ok(n, L, F, T, level_stride)) return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(table_f16) % (2u * F)) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const dim3 grid(f2n_div_up(n, F2N_BLOCK), (unsigned)L), block(F2N_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  const bool p2 = is_pow2(T);
  F2N_DISPATCH_F(F, {
    if (p2)
      hipLaunchKernelGGL(
        (hash_fwd_kernel<FF, true>), grid, block, 0, s, pts, table_f16, primes, bias, mul, out,
        out_ld_point, out_ld_chan, idx_out, n, L, T, level_stride);
    else
      hipLaunchKernelGGL(
        (hash_fwd_kernel<FF, false>), grid, block, 0, s, pts, table_f16, primes, bias, mul, out,
        out_ld_point, out_ld_chan, idx_out, n, L, T, level_stride);
  })
  return f2n_launch_status();
}

extern "C" int f2n_hash_fwd_raytile(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, float * out_cm, int n_rays, int S, int L, int F, uint32_t T,
  int64_t level_stride, void * stream)
{
  if (!pts || !table_f16 || !primes || !bias || !mul || !out_cm) return F2N_E_INVALID_ARG;
  if (F != 1 && F != 2 && F != 4 && F != 8) return F2N_E_UNSUPPORTED;
  if (n_rays < 0 || S <= 0) return F2N_E_INVALID_ARG;
  if (S % 16) return F2N_E_UNSUPPORTED;  // callers fall back to f2n_hash_fwd
  const int64_t n = (int64_t)n_rays * S;
  if (!f2n_hash_args_ok(n, L, F, T, level_stride)) return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(table_f16) % (2u * F)) return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(out_cm) & 15u) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const int walk = f2n_get_option(F2N_OPT_RAYTILE_WALK);
  const int want = f2n_get_option(F2N_OPT_RAYTILE);  // samples per tile: 16 | 32 (measurements)
  const int ts = (want == 16) ? 16 : (S % 32 == 0 ? 32 : 16);
  const int64_t tiles = (int64_t)f2n_div_up(n_rays, 64) * (S / ts);
  if (tiles > 0x7fffffff) return F2N_E_INVALID_ARG;
  const dim3 grid((unsigned)tiles, (unsigned)L), block(F2N_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  const bool p2 = is_pow2(T);
#define F2N_RT_LAUNCH(P2, SS)                                                                      \
  hipLaunchKernelGGL(                                                                              \
    (hash_fwd_raytile_kernel<FF, P2, SS>), grid, block, 0, s, pts, table_f16, primes, bias, mul,   \
    out_cm, n_rays, S, T, level_stride, walk)
  F2N_DISPATCH_F(F, {
    if (p2) {
      if (ts == 16) F2N_RT_LAUNCH(true, 16);
      else F2N_RT_LAUNCH(true, 32);
    } else {
      if (ts == 16) F2N_RT_LAUNCH(false, 16);
      else F2N_RT_LAUNCH(false, 32);
    }
  })
#undef F2N_RT_LAUNCH
  return f2n_launch_status();
}

extern "C" int f2n_hash_bwd(
  const float * pts, const uint16_t * table_f16, const int32_t * primes, const float * bias,
  const float * mul, const float * grad_out, int64_t g_ld_point, int64_t g_ld_chan,
  float * table_grad, float * pts_grad, int64_t n, int L, int F, uint32_t T, int64_t level_stride,
  float grad_scale, void * stream)
{
  if (!pts || !table_f16 || !primes || !bias || !mul || !grad_out || !table_grad)
    return F2N_E_INVALID_ARG;
  if (F != 1 && F != 2 && F != 4 && F != 8) return F2N_E_UNSUPPORTED;
  if (!f2n_hash_args_ok(n, L, F, T, level_stride)) return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(table_f16) % (2u * F)) return F2N_E_INVALID_ARG;
  int e = 0;
  const float m = frexpf(grad_scale, &e);
  if (!(grad_scale > 0.f) || m != 0.5f) return F2N_E_INVALID_ARG;  // power of two only
  if (n == 0) return F2N_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool p2 = is_pow2(T);
  const float inv = 1.f / grad_scale;

  // Training case (no point gradient) on a big batch: LDS-sliced accumulation.  The hash work is
  // repeated once per slice, so it only pays while the slice count stays moderate.
  const int64_t rows_per_slice = kSliceFloats / F;
  const int64_t n_slices = ((int64_t)T + rows_per_slice - 1) / rows_per_slice;
  const int force = f2n_get_option(F2N_OPT_HASH_BWD);  // 1 atomic | 2 sliced: A/B switch
  bool sliced = !pts_grad && n_slices <= 64 && n >= 32768;
  if (force && !pts_grad && n_slices <= 1024) sliced = (force == 2);
  if (sliced) {
    // enough sample partitions to put >= 2 workgroups on every CU when levels x slices is small
    int n_parts = 1;
    while ((int64_t)L * n_slices * n_parts < 512 && (n / (n_parts * 2)) >= 65536) n_parts *= 2;
    const dim3 grid((unsigned)n_slices, (unsigned)L, (unsigned)n_parts), block(kSliceBlock);
#define F2N_BWD_SLICED(P2)                                                                        \
  hipLaunchKernelGGL(                                                                             \
    (hash_bwd_sliced_kernel<FF, P2>), grid, block, 0, s, pts, primes, bias, mul, grad_out,        \
    g_ld_point, g_ld_chan, table_grad, n, T, level_stride, grad_scale, inv, n_parts)
    F2N_DISPATCH_F(F, {
      if (p2) F2N_BWD_SLICED(true);
      else F2N_BWD_SLICED(false);
    })
#undef F2N_BWD_SLICED
    return f2n_launch_status();
  }

  if (pts_grad) {
    if (hipMemsetAsync(pts_grad, 0, sizeof(float) * 3 * n, s) != hipSuccess) return F2N_E_LAUNCH;
  }
  const dim3 grid(f2n_div_up(n, F2N_BLOCK), (unsigned)L), block(F2N_BLOCK);
#define F2N_BWD_LAUNCH(P2, PG)                                                                    \
  hipLaunchKernelGGL(                                                                             \
    (hash_bwd_kernel<FF, P2, PG>), grid, block, 0, s, pts, table_f16, primes, bias, mul, grad_out, \
    g_ld_point, g_ld_chan, table_grad, pts_grad, n, T, level_stride, grad_scale, inv)
  F2N_DISPATCH_F(F, {
    if (p2) {
      if (pts_grad) F2N_BWD_LAUNCH(true, true);
      else F2N_BWD_LAUNCH(true, false);
    } else {
      if (pts_grad) F2N_BWD_LAUNCH(false, true);
      else F2N_BWD_LAUNCH(false, false);
    }
  })
#undef F2N_BWD_LAUNCH
  return f2n_launch_status();
}

extern "C" int f2n_contract_fwd(const float * pts, float * x, int64_t n, void * stream)
{
  if (!pts || !x || n < 0) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(
    contract_fwd_kernel, dim3(f2n_div_up(n, F2N_BLOCK)), dim3(F2N_BLOCK), 0, (hipStream_t)stream,
    pts, x, n);
  return f2n_launch_status();
}

extern "C" int f2n_contract_bwd(
  const float * pts, const float * dx, float * dpts, int64_t n, void * stream)
{
  if (!pts || !dx || !dpts || n < 0) return F2N_E_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(
    contract_bwd_kernel, dim3(f2n_div_up(n, F2N_BLOCK)), dim3(F2N_BLOCK), 0, (hipStream_t)stream,
    pts, dx, dpts, n);
  return f2n_launch_status();
}
