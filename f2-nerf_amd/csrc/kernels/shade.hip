// shade.hip -- the per-sample network between the hash encode and the compositing, fused:
//   h      = Linear(C -> 16)(enc)                          (Hash3DAnchored::mlp_, reference
//                                                           src/hash_3d_anchored.cpp:86)
//   logit  = h[0]                                          (density logit, src/renderer.cpp:93)
//   X      = [1, h[1..15]] (+ app_emb[image]) ++ SH16(dir) (src/renderer.cpp:95-104,
//                                                           src/sh_shader.cpp:24-25)
//   rgb    = (1+2e) * sigmoid(Linear(64->3)(relu(Linear(32->64)(X)))) - e   (src/sh_shader.cpp:26-28)
// forward and backward, including every parameter gradient.
//
// Why: at 8.4 M samples per chunk the three skinny GEMMs (and their transposed/weight-gradient
// forms) run at ~1 TFLOP/s in rocBLAS and, with the cat / ScatterAdd / bias-reduce kernels around
// them, cost more than all hash kernels together (profiles/r01_*_kernel_stats.csv).  The layers
// are 16/64/3 wide, so one lane can carry one sample through the whole network in registers:
// weights are wave-uniform and stream through SGPRs (scalar loads), activations never touch memory.
//
// Backward = recompute forward per sample, propagate, and reduce the weight gradients inside the
// wavefront without atomics: sample-major vectors are transposed through LDS so that lane j owns
// row j of the weight gradient and accumulates it in registers over every stride the wave processes;
// one atomic flush per wave at the end.  Weights are staged in LDS there (see shade_bwd_kernel).
//
// Activations enc / d_enc are channel-major [C, n] (the layout f2n_hash_fwd / f2n_hash_bwd use).
#include "sh_basis.hiph"
#include "shade_mfma.hiph"

#include <cstdlib>
#include <cstring>

namespace
{

constexpr int kOut1 = 16;   // field head width
constexpr int kIn2 = 32;    // shader input = 16 shading features + 16 SH
constexpr int kHid = 64;    // shader hidden width
constexpr float kEps = 1e-3f;

struct ShadeParams
{
  const float * __restrict__ w_h;  // [16, C]
  const float * __restrict__ b_h;  // [16]
  const float * __restrict__ w1;   // [64, 32]
  const float * __restrict__ b1;   // [64]
  const float * __restrict__ w2;   // [3, 64]
  const float * __restrict__ b2;   // [3]
  const float * __restrict__ emb;  // [E, 16] or null
};

// One sample through the network.  All indices are compile-time constants after unrolling, so the
// arrays live in VGPRs and the weights are fetched with scalar loads.
template <int C>
__device__ __forceinline__ void shade_forward(
  const float (&e)[C], float dx, float dy, float dz, const float * __restrict__ emb_row,
  const ShadeParams & P, float (&h)[kOut1], float (&X)[kIn2], float (&pre)[kHid], float (&o)[3])
{
#pragma unroll
  for (int i = 0; i < kOut1; i++) {
    float acc = P.b_h[i];
#pragma unroll
    for (int c = 0; c < C; c++) acc = fmaf(e[c], P.w_h[i * C + c], acc);
    h[i] = acc;
  }
  X[0] = 1.f;
#pragma unroll
  for (int i = 1; i < kOut1; i++) X[i] = h[i];
  if (emb_row) {
#pragma unroll
    for (int i = 0; i < kOut1; i++) X[i] += emb_row[i];
  }
  sh_basis<4>(dx, dy, dz, &X[kOut1]);
#pragma unroll
  for (int j = 0; j < kHid; j++) {
    float acc = P.b1[j];
#pragma unroll
    for (int i = 0; i < kIn2; i++) acc = fmaf(X[i], P.w1[j * kIn2 + i], acc);
    pre[j] = acc;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float acc = P.b2[c];
#pragma unroll
    for (int j = 0; j < kHid; j++) acc = fmaf(fmaxf(pre[j], 0.f), P.w2[c * kHid + j], acc);
    o[c] = acc;
  }
}

template <int C>
__global__ __launch_bounds__(F2N_BLOCK) void shade_fwd_kernel(
  const float * __restrict__ enc, const float * __restrict__ dirs,
  const int32_t * __restrict__ sample_img, const float * __restrict__ p_w_h,
  const float * __restrict__ p_b_h, const float * __restrict__ p_w1,
  const float * __restrict__ p_b1, const float * __restrict__ p_w2,
  const float * __restrict__ p_b2, const float * __restrict__ p_emb, float * __restrict__ logit,
  float * __restrict__ rgb, float * __restrict__ pre_out, int64_t n)
{
  const int64_t p = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (p >= n) return;
  const ShadeParams P{p_w_h, p_b_h, p_w1, p_b1, p_w2, p_b2, p_emb};
  float e[C];
#pragma unroll
  for (int c = 0; c < C; c++) e[c] = enc[(int64_t)c * n + p];
  const float * emb_row = (P.emb && sample_img) ? P.emb + (int64_t)sample_img[p] * kOut1 : nullptr;
  float h[kOut1], X[kIn2], pre[kHid], o[3];
  shade_forward<C>(e, dirs[3 * p], dirs[3 * p + 1], dirs[3 * p + 2], emb_row, P, h, X, pre, o);
  logit[p] = h[0];
#pragma unroll
  for (int c = 0; c < 3; c++)
    rgb[3 * p + c] = (1.f + 2.f * kEps) / (1.f + expf(-o[c])) - kEps;
  // optional: hidden pre-activations, channel-major [64, n], so the backward need not recompute the
  // 64x32 layer at its low occupancy (coalesced 256-byte stores per wave and neuron)
  if (pre_out) {
#pragma unroll
    for (int j = 0; j < kHid; j++) pre_out[(int64_t)j * n + p] = pre[j];
  }
}

// ---- backward -----------------------------------------------------------------------------------
//
// Structure of one stride (64 samples, lane = sample) -- everything wave-private, no workgroup
// barriers after the weights are staged:
//   1. forward recompute with ROLLED loops over the output neuron: the neuron's weight row comes from
//      LDS (wave-uniform ds_read_b128 = broadcast), the result goes to a [neuron][65] LDS tile (own
//      column: conflict-free), so no register array is indexed dynamically and the code stays small.
//      (Fully unrolled variants -- weights through SGPRs or through LDS -- made hipcc hoist hundreds
//      of weight loads to the top of a 10 K-instruction block and spill 8-13 KB per lane.)
//   2. d w2 += d_o (x) relu(pre)            lane j owns column j of w2: transposed reads of the tile
//   3. backward over the hidden layer, rolled over j: d_hid[j] replaces pre[j] in the tile in place,
//      dX[0:16] accumulates in registers
//   4. d w1[j][:] += d_hid[s][j] * X[s][:]  lane j owns row j: transposed tile reads + broadcast X
//   5. d_enc = w_h^T d_h (rolled over the 16 head outputs), d w_h, biases, embedding
// The [64][65] tile is read conflict-free both ways: element (s, j) sits at j*65 + s.

constexpr int kTP = 65;                 // row pitch of the [neuron][sample] tile
constexpr int kDoOff = kHid * kTP;      // [64 samples][3] d_o rows live in the tile's tail
constexpr int kHP = kOut1 + 4;          // row pitch of the [sample][16] rows of step 5b

// Per-wave LDS tile.  Occupancy is what this kernel lives on: with one wave per SIMD every LDS read
// stalls the SIMD (4.4 ms per 8.4 M samples); a second wave hides it.  The tile is therefore the only
// per-wave LDS object (17 KiB at C <= 32: eight waves + the weights = 151 KiB), and the broadcast of
// X[s][:] in step 4 uses v_readlane instead of a second tile.
template <int C>
struct BwdShape
{
  static constexpr int kERowP = C + 4;                 // [sample][C] rows of step 5b
  static constexpr int kHOff = 64 * kERowP;            // then [sample][16+4] rows
  static constexpr int kTile =
    (kHOff + 64 * kHP > kDoOff + 64 * 3) ? kHOff + 64 * kHP : kDoOff + 64 * 3;
  static constexpr int kWeights = kOut1 * (C + 4) + kHid * (kIn2 + 4) + kHid * 4 + 4;
  static constexpr int kWaves = ((kWeights + 8 * kTile) * 4 <= 160 * 1024) ? 8 : 4;
};

__device__ __forceinline__ void wave_lds_sync()
{
  // one wavefront's LDS operations execute in order; this only stops the compiler from moving the
  // transposed reads above the writes of other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int C>
__global__ __launch_bounds__(BwdShape<C>::kWaves * 64) void shade_bwd_kernel(
  const float * __restrict__ enc, const float * __restrict__ dirs,
  const int32_t * __restrict__ sample_img, const float * __restrict__ p_w_h,
  const float * __restrict__ p_b_h, const float * __restrict__ p_w1,
  const float * __restrict__ p_b1, const float * __restrict__ p_w2,
  const float * __restrict__ p_b2, const float * __restrict__ p_emb,
  const float * __restrict__ d_logit, const float * __restrict__ d_rgb, float * __restrict__ d_enc,
  float * __restrict__ g_w_h, float * __restrict__ g_b_h, float * __restrict__ g_w1,
  float * __restrict__ g_b1, float * __restrict__ g_w2, float * __restrict__ g_b2,
  float * __restrict__ g_emb, const float * __restrict__ pre_in, int64_t n)
{
  static_assert(C % 4 == 0 && C <= kHid, "C must be a multiple of 4 and at most 64");
  constexpr int kBwdWaves = BwdShape<C>::kWaves;
  constexpr int kWavelds = BwdShape<C>::kTile;
  // weights, staged once per workgroup: rows padded by 4 floats with the bias in the first pad slot
  constexpr int kWhP = C + 4, kW1P = kIn2 + 4;
  constexpr int kOffWh = 0, kOffW1 = kOffWh + kOut1 * kWhP, kOffW2T = kOffW1 + kHid * kW1P,
                kOffB2 = kOffW2T + kHid * 4, kWTotal = kOffB2 + 4;
  // one LDS object, weights first: their byte offsets stay below 64 KiB, the range of the DS
  // instructions' immediate offset field (a second __shared__ array landed above 0x1a000 and every
  // weight address needed its own v_add)
  __shared__ __attribute__((aligned(16))) float lds_all[kWTotal + kBwdWaves * kWavelds];
  float * lds_w = lds_all;
  float * lds = lds_all + kWTotal;
  for (int i = threadIdx.x; i < kOut1 * C; i += kBwdWaves * 64)
    lds_w[kOffWh + (i / C) * kWhP + (i % C)] = p_w_h[i];
  for (int i = threadIdx.x; i < kHid * kIn2; i += kBwdWaves * 64)
    lds_w[kOffW1 + (i / kIn2) * kW1P + (i % kIn2)] = p_w1[i];
  for (int i = threadIdx.x; i < 3 * kHid; i += kBwdWaves * 64)
    lds_w[kOffW2T + (i % kHid) * 4 + (i / kHid)] = p_w2[i];  // transposed: [j][c]
  if (threadIdx.x < kOut1) lds_w[kOffWh + threadIdx.x * kWhP + C] = p_b_h[threadIdx.x];
  if (threadIdx.x < kHid) {
    lds_w[kOffW1 + threadIdx.x * kW1P + kIn2] = p_b1[threadIdx.x];
    lds_w[kOffW2T + threadIdx.x * 4 + 3] = 0.f;
  }
  if (threadIdx.x < 4) lds_w[kOffB2 + threadIdx.x] = (threadIdx.x < 3) ? p_b2[threadIdx.x] : 0.f;
  __syncthreads();

  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  // A zero that lives in a VGPR and that the compiler cannot see through.  Added to the (wave-uniform)
  // neuron index it makes the weight-row address VGPR-based, so the eight ds_read_b128 of a row use
  // ONE address register + immediate offsets; with a scalar base hipcc rebuilt every address with
  // s_add + v_mov (41 of the 145 instructions of a two-neuron iteration).
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  float * tileP = lds + wave * kWavelds;  // [64 neurons][65]: pre-activations, then d_hid
  const bool has_emb = (p_emb != nullptr) && (sample_img != nullptr);

  // per-lane accumulators that live across all strides of this wave
  float acc_w1[kIn2];  // lane j: d w1[j][:]
  float acc_b1 = 0.f;  // lane j: d b1[j]
  float acc_w2[3];     // lane j: d w2[:][j]
  constexpr int CQ = C / 4;
  float acc_wh[CQ];    // lane (i = lane&15, q = lane>>4): d w_h[i][q*CQ .. q*CQ+CQ)
  float acc_bh[kOut1]; // lane = sample partial sums, wave-reduced at the end
  float acc_b2[3];
#pragma unroll
  for (int i = 0; i < kIn2; i++) acc_w1[i] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) acc_w2[c] = acc_b2[c] = 0.f;
#pragma unroll
  for (int k = 0; k < CQ; k++) acc_wh[k] = 0.f;
#pragma unroll
  for (int i = 0; i < kOut1; i++) acc_bh[i] = 0.f;

  const int64_t n_strides = (n + 63) / 64;
  const int64_t wave_global = (int64_t)blockIdx.x * kBwdWaves + wave;
  const int64_t wave_count = (int64_t)gridDim.x * kBwdWaves;
  for (int64_t st = wave_global; st < n_strides; st += wave_count) {
    const int64_t p = st * 64 + lane;
    const bool valid = p < n;
    const int64_t pc = valid ? p : n - 1;  // clamped: tail lanes recompute a real sample, weight 0

    // ---- 1. forward recompute
    float e[C];
#pragma unroll
    for (int c = 0; c < C; c++) e[c] = enc[(int64_t)c * n + pc];
    const int img = has_emb ? sample_img[pc] : 0;
    float X[kIn2];
    {
      // field head: h[i] -> column i of the (still unused) tile viewed as [16][64]
#pragma unroll 2
      for (int i = 0; i < kOut1; i++) {
        const float * row = lds_w + kOffWh + (i + vzero) * kWhP;
        float acc = row[C];
#pragma unroll
        for (int c = 0; c < C; c += 4) {
          const float4 w = *reinterpret_cast<const float4 *>(row + c);
          acc = fmaf(e[c], w.x, acc);
          acc = fmaf(e[c + 1], w.y, acc);
          acc = fmaf(e[c + 2], w.z, acc);
          acc = fmaf(e[c + 3], w.w, acc);
        }
        tileP[i * 64 + lane] = acc;
      }
      X[0] = 1.f;
#pragma unroll
      for (int i = 1; i < kOut1; i++) X[i] = tileP[i * 64 + lane];
      if (has_emb) {
        const float * emb_row = p_emb + (int64_t)img * kOut1;
#pragma unroll
        for (int i = 0; i < kOut1; i++) X[i] += emb_row[i];
      }
      sh_basis<4>(dirs[3 * pc], dirs[3 * pc + 1], dirs[3 * pc + 2], &X[kOut1]);
    }
    float o[3];
    {
      const float4 b2 = *reinterpret_cast<const float4 *>(lds_w + kOffB2);
      o[0] = b2.x;
      o[1] = b2.y;
      o[2] = b2.z;
      if (pre_in) {
        // pre-activations saved by the forward kernel: 64 coalesced loads instead of 2 K FMAs
#pragma unroll 8
        for (int j = 0; j < kHid; j++) {
          const float acc = pre_in[(int64_t)j * n + pc];
          tileP[j * kTP + lane] = acc;
          const float hj = fmaxf(acc, 0.f);
          const float4 w2 = *reinterpret_cast<const float4 *>(lds_w + kOffW2T + (j + vzero) * 4);
          o[0] = fmaf(hj, w2.x, o[0]);
          o[1] = fmaf(hj, w2.y, o[1]);
          o[2] = fmaf(hj, w2.z, o[2]);
        }
      } else {
#pragma unroll 2
        for (int j = 0; j < kHid; j++) {
          const float * row = lds_w + kOffW1 + (j + vzero) * kW1P;
          float acc = row[kIn2];
#pragma unroll
          for (int i = 0; i < kIn2; i += 4) {
            const float4 w = *reinterpret_cast<const float4 *>(row + i);
            acc = fmaf(X[i], w.x, acc);
            acc = fmaf(X[i + 1], w.y, acc);
            acc = fmaf(X[i + 2], w.z, acc);
            acc = fmaf(X[i + 3], w.w, acc);
          }
          tileP[j * kTP + lane] = acc;  // pre-activation
          const float hj = fmaxf(acc, 0.f);
          const float4 w2 = *reinterpret_cast<const float4 *>(lds_w + kOffW2T + (j + vzero) * 4);
          o[0] = fmaf(hj, w2.x, o[0]);
          o[1] = fmaf(hj, w2.y, o[1]);
          o[2] = fmaf(hj, w2.z, o[2]);
        }
      }
    }

    float d_o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float sg = 1.f / (1.f + expf(-o[c]));
      const float g = valid ? d_rgb[3 * pc + c] : 0.f;
      d_o[c] = g * (1.f + 2.f * kEps) * sg * (1.f - sg);
      acc_b2[c] += d_o[c];
    }

    // ---- 2. d w2[c][j] += sum_s d_o[s][c] * relu(pre[s][j])     (lane = j)
    tileP[kDoOff + lane * 3 + 0] = d_o[0];
    tileP[kDoOff + lane * 3 + 1] = d_o[1];
    tileP[kDoOff + lane * 3 + 2] = d_o[2];
    wave_lds_sync();
#pragma unroll 4
    for (int s = 0; s < 64; s++) {
      const float a = fmaxf(tileP[lane * kTP + s], 0.f);
      const float * g = tileP + kDoOff + (s + vzero) * 3;
      acc_w2[0] = fmaf(g[0], a, acc_w2[0]);
      acc_w2[1] = fmaf(g[1], a, acc_w2[1]);
      acc_w2[2] = fmaf(g[2], a, acc_w2[2]);
    }
    wave_lds_sync();

    // ---- 3. back through the hidden layer (lane = sample): d_hid replaces pre in the tile
    float dX[kOut1];
#pragma unroll
    for (int i = 0; i < kOut1; i++) dX[i] = 0.f;
#pragma unroll 2
    for (int j = 0; j < kHid; j++) {
      const float pre = tileP[j * kTP + lane];
      const float4 w2 = *reinterpret_cast<const float4 *>(lds_w + kOffW2T + (j + vzero) * 4);
      float dh = d_o[0] * w2.x;
      dh = fmaf(d_o[1], w2.y, dh);
      dh = fmaf(d_o[2], w2.z, dh);
      dh = (pre > 0.f) ? dh : 0.f;
      tileP[j * kTP + lane] = dh;
      const float * row = lds_w + kOffW1 + (j + vzero) * kW1P;
#pragma unroll
      for (int i = 0; i < kOut1; i += 4) {  // only the 16 non-SH input columns carry gradient
        const float4 w = *reinterpret_cast<const float4 *>(row + i);
        dX[i] = fmaf(dh, w.x, dX[i]);
        dX[i + 1] = fmaf(dh, w.y, dX[i + 1]);
        dX[i + 2] = fmaf(dh, w.z, dX[i + 2]);
        dX[i + 3] = fmaf(dh, w.w, dX[i + 3]);
      }
    }
    // d_h: gradient w.r.t. the 16 head outputs; dX[0] only feeds the embedding (X[0] = 1 + emb[0])
    float d_h[kOut1];
    d_h[0] = valid ? d_logit[pc] : 0.f;
#pragma unroll
    for (int i = 1; i < kOut1; i++) d_h[i] = dX[i];
#pragma unroll
    for (int i = 0; i < kOut1; i++) acc_bh[i] += d_h[i];

    // ---- 4. d w1[j][i] += sum_s d_hid[s][j] * X[s][i] ; d b1[j] += sum_s d_hid[s][j]   (lane = j)
    // X[s][:] is wave-uniform per s: v_readlane broadcasts it from lane s's registers into SGPRs
    wave_lds_sync();
#pragma unroll 2
    for (int s = 0; s < 64; s++) {
      const float a = tileP[lane * kTP + s];
      acc_b1 += a;
#pragma unroll
      for (int i = 0; i < kIn2; i++) {
        const float xs =
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X[i]), s));
        acc_w1[i] = fmaf(a, xs, acc_w1[i]);
      }
    }
    wave_lds_sync();

    // ---- 5a. d_enc[c] = sum_i d_h[i] * w_h[i][c]
    {
      float d_e[C];
#pragma unroll
      for (int c = 0; c < C; c++) d_e[c] = 0.f;
      // d_h through LDS ([i][64] columns) so the rolled loop over i needs no dynamic register index
#pragma unroll
      for (int i = 0; i < kOut1; i++) tileP[i * 64 + lane] = d_h[i];
#pragma unroll 2
      for (int i = 0; i < kOut1; i++) {
        const float a = tileP[i * 64 + lane];
        const float * row = lds_w + kOffWh + (i + vzero) * kWhP;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
          const float4 w = *reinterpret_cast<const float4 *>(row + c);
          d_e[c] = fmaf(a, w.x, d_e[c]);
          d_e[c + 1] = fmaf(a, w.y, d_e[c + 1]);
          d_e[c + 2] = fmaf(a, w.z, d_e[c + 2]);
          d_e[c + 3] = fmaf(a, w.w, d_e[c + 3]);
        }
      }
      if (valid) {
#pragma unroll
        for (int c = 0; c < C; c++) d_enc[(int64_t)c * n + p] = d_e[c];
      }
    }

    // ---- 5b. d w_h[i][c] += sum_s d_h[s][i] * enc[s][c]
    {
      constexpr int kEP = BwdShape<C>::kERowP;  // [sample][C] rows at the start of the tile
      float * rowsH = tileP + BwdShape<C>::kHOff;  // [sample][16+4] rows behind them
      wave_lds_sync();
#pragma unroll
      for (int c = 0; c < C; c += 4)
        *reinterpret_cast<float4 *>(tileP + lane * kEP + c) =
          make_float4(e[c], e[c + 1], e[c + 2], e[c + 3]);
      // row of {dX[0] (embedding only), d_h[1..15]} plus d_h[0] in the pad slot
      *reinterpret_cast<float4 *>(rowsH + lane * kHP) =
        make_float4(valid ? dX[0] : 0.f, d_h[1], d_h[2], d_h[3]);
#pragma unroll
      for (int i = 4; i < kOut1; i += 4)
        *reinterpret_cast<float4 *>(rowsH + lane * kHP + i) =
          make_float4(d_h[i], d_h[i + 1], d_h[i + 2], d_h[i + 3]);
      rowsH[lane * kHP + kOut1] = d_h[0];
      wave_lds_sync();
      const int wi = lane & 15, wq = lane >> 4;
      const int col = (wi == 0) ? kOut1 : wi;  // head output 0's gradient lives in the pad slot
#pragma unroll 4
      for (int s = 0; s < 64; s++) {
        const float a = rowsH[s * kHP + col];
#pragma unroll
        for (int k = 0; k < CQ; k += 4) {
          const float4 x = *reinterpret_cast<const float4 *>(tileP + s * kEP + wq * CQ + k);
          acc_wh[k] = fmaf(a, x.x, acc_wh[k]);
          if (k + 1 < CQ) acc_wh[k + 1] = fmaf(a, x.y, acc_wh[k + 1]);
          if (k + 2 < CQ) acc_wh[k + 2] = fmaf(a, x.z, acc_wh[k + 2]);
          if (k + 3 < CQ) acc_wh[k + 3] = fmaf(a, x.w, acc_wh[k + 3]);
        }
      }
      // ---- 5c. appearance embedding: d emb[img][i] += dX[i]
      if (has_emb) {
        const int img0 = __builtin_amdgcn_readfirstlane(img);
        if (__all(img == img0)) {
          // lane (i = lane&15, part = lane>>4) sums 16 of the 64 samples, then the 4 parts combine
          float part = 0.f;
#pragma unroll
          for (int s = 0; s < 16; s++) part += rowsH[(wq * 16 + s) * kHP + wi];
          part += __shfl_xor(part, 16);
          part += __shfl_xor(part, 32);
          if (lane < 16) atomicAdd(g_emb + (int64_t)img0 * kOut1 + lane, part);
        } else if (valid) {
#pragma unroll
          for (int i = 0; i < kOut1; i++) atomicAdd(g_emb + (int64_t)img * kOut1 + i, dX[i]);
        }
      }
      wave_lds_sync();
    }
  }

  // ---- flush this wave's accumulators
#pragma unroll
  for (int i = 0; i < kIn2; i++) atomicAdd(g_w1 + lane * kIn2 + i, acc_w1[i]);
  atomicAdd(g_b1 + lane, acc_b1);
#pragma unroll
  for (int c = 0; c < 3; c++) atomicAdd(g_w2 + c * kHid + lane, acc_w2[c]);
  {
    const int wi = lane & 15, wq = lane >> 4;
#pragma unroll
    for (int k = 0; k < CQ; k++) atomicAdd(g_w_h + wi * C + wq * CQ + k, acc_wh[k]);
  }
#pragma unroll
  for (int i = 0; i < kOut1; i++) {
    const float t = wave_sum(acc_bh[i]);
    if (lane == 0) atomicAdd(g_b_h + i, t);
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float t = wave_sum(acc_b2[c]);
    if (lane == 0) atomicAdd(g_b2 + c, t);
  }
}

}  // namespace

#define F2N_DISPATCH_C(C_, ...)                                   \
  switch (C_) {                                                   \
    case 8: { constexpr int CC = 8; __VA_ARGS__; } break;         \
    case 16: { constexpr int CC = 16; __VA_ARGS__; } break;       \
    case 32: { constexpr int CC = 32; __VA_ARGS__; } break;       \
    case 64: { constexpr int CC = 64; __VA_ARGS__; } break;       \
    default: return F2N_E_UNSUPPORTED;                            \
  }

extern "C" int f2n_shade_fwd(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, float * logit, float * rgb, float * pre_cm, int64_t n, void * stream)
{
  if (n < 0) return F2N_E_INVALID_ARG;
  if (C != 8 && C != 16 && C != 32 && C != 64) return F2N_E_UNSUPPORTED;
  if (n == 0) return F2N_OK;
  if (!enc_cm || !dirs || !w_h || !b_h || !w1 || !b1 || !w2 || !b2 || !logit || !rgb)
    return F2N_E_INVALID_ARG;
  // matrix-core forward (shade_mfma.hip) unless F2N_OPT_SHADE_FWD asks for the one-sample-per-lane
  // kernel below (A/B measurements) or n is beyond its 32-bit per-sample offsets (2^28 samples)
  if (f2n_get_option(F2N_OPT_SHADE_FWD) == 0 && f2n_detail::shade_bwd_mfma_supports(64, n))
    return f2n_detail::launch_shade_fwd_mfma(
      enc_cm, C, dirs, sample_img, w_h, b_h, w1, b1, w2, b2, app_emb, logit, rgb, pre_cm, n,
      (hipStream_t)stream);
  const dim3 grid(f2n_div_up(n, F2N_BLOCK)), block(F2N_BLOCK);
  F2N_DISPATCH_C(C, hipLaunchKernelGGL(
                      (shade_fwd_kernel<CC>), grid, block, 0, (hipStream_t)stream, enc_cm, dirs,
                      sample_img, w_h, b_h, w1, b1, w2, b2, app_emb, logit, rgb, pre_cm, n))
  return f2n_launch_status();
}

extern "C" int f2n_shade_bwd(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, const float * d_logit, const float * d_rgb, float * d_enc_cm,
  float * g_w_h, float * g_b_h, float * g_w1, float * g_b1, float * g_w2, float * g_b2,
  float * g_app_emb, const float * pre_cm, int64_t n, void * stream)
{
  if (n < 0) return F2N_E_INVALID_ARG;
  if (C != 8 && C != 16 && C != 32 && C != 64) return F2N_E_UNSUPPORTED;
  if (n == 0) return F2N_OK;
  if (!enc_cm || !dirs || !w_h || !b_h || !w1 || !b1 || !w2 || !b2 || !d_logit || !d_rgb ||
      !d_enc_cm || !g_w_h || !g_b_h || !g_w1 || !g_b1 || !g_w2 || !g_b2)
    return F2N_E_INVALID_ARG;
  if (app_emb && sample_img && !g_app_emb) return F2N_E_INVALID_ARG;
  // matrix-core kernel (shade_mfma.hip) unless the saved pre-activations are offered, the width has
  // no MFMA tiling, or F2N_OPT_SHADE_BWD asks for the vector kernel (A/B measurements)
  const bool force_valu = f2n_get_option(F2N_OPT_SHADE_BWD) == 1;
  if (!force_valu && !pre_cm && f2n_detail::shade_bwd_mfma_supports(C, n))
    return f2n_detail::launch_shade_bwd_mfma(
      enc_cm, C, dirs, sample_img, w_h, b_h, w1, b1, w2, b2, app_emb, d_logit, d_rgb, d_enc_cm,
      g_w_h, g_b_h, g_w1, g_b1, g_w2, g_b2, g_app_emb, n, (hipStream_t)stream);
  const int64_t n_strides = (n + 63) / 64;
  // persistent waves: one workgroup per CU (its LDS holds the weights + 8 wave tiles); fewer if n is small
  F2N_DISPATCH_C(C, constexpr int kW = BwdShape<CC>::kWaves;
                    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_strides + kW - 1) / kW);
                    hipLaunchKernelGGL(
                      (shade_bwd_kernel<CC>), dim3(grid), dim3(kW * 64), 0,
                      (hipStream_t)stream, enc_cm, dirs, sample_img, w_h, b_h, w1, b1, w2, b2,
                      app_emb, d_logit, d_rgb, d_enc_cm,
                      g_w_h, g_b_h, g_w1, g_b1, g_w2, g_b2, g_app_emb, pre_cm, n))
  return f2n_launch_status();
}
