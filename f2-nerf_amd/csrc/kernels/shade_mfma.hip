// shade_mfma.hip -- backward of the per-sample network (see shade.hip for the network and its
// reference sites) on the f32 matrix cores: v_mfma_f32_16x16x4_f32, exact f32 (a k-ordered fmaf
// chain), so the parity bar of the VALU kernel carries over.
//
// Why: the VALU backward (shade.hip) reads its wave-uniform weights from LDS as broadcast
// ds_read_b128 -- 1 KiB of LDS bandwidth per 4 FMA instructions -- and measures 25 % of the f32 FMA
// rate; it is LDS-bound, not FMA-bound.  An MFMA takes both operands from VGPRs: one 256-byte weight
// read feeds 4 MFMAs = 4096 MACs, 64x less LDS traffic per MAC, and the weight-gradient products
// (sums over samples) become real GEMMs instead of broadcast loops.
//
// Layouts.  lane l = (q = l >> 4, m = l & 15).  One 16x16x4 MFMA: D[4q+r][m] += sum_k A[m][k] B[k][m]
// with lane (q, m) supplying a = A[row m][k = q], b = B[k = q][col m] and holding D rows 4q+r, r<4.
//   Q-layout (activations of 64 samples): lane (q, m) holds feature 4q+r (+16 per M-tile) of
//     samples 16T+m, T = 0..3.  An MFMA output IS in Q-layout (rows = features, cols = samples), and
//     is the next layer's B operand when that layer walks its k index in the order (M, r) with
//     k = 16M + 4q + r -- so the whole forward and the data-gradient chain need no lane movement.
//     The weight (A) operands are staged in LDS once per workgroup, pre-arranged per lane in
//     exactly that k order ("slots" of 64 floats, conflict-free ds_read_b32).
//   S-layout (for sums over samples: d w1, d w2, d w_h): operands need the SAMPLE on the k axis and
//     the feature on the lane, i.e. the transpose.  Each wave has an LDS tile [feature][72]: Q-layout
//     registers are written with ds_write_b32 (lanes of a quarter = consecutive samples) and read back
//     as float4 = four consecutive k-steps (k-step t of quarter q = sample 16(t/4) + 4q + t%4).
//
// The 3-wide output layer is NOT on the matrix cores: as a 16-row MFMA operand 13 of its 16 rows are
// padding (round 2: 23-29 % of all issued products).  Each lane holds 16 hidden neurons of 4 samples
// in the Q-layout, so o[c][s] = sum_j w2[c][j] relu(pre[j][s]) is 48 vector FMAs per 16-neuron tile
// into per-quarter partial sums, and ONE product per (c, sample tile) with an all-ones A operand adds
// the four quarters (the MFMA's k axis) and hands the sum to every lane: 12 products instead of 64.
// Its two backward products (d w2, and d_hid = w2^T d_o) are vector FMAs on the same registers.
//
// Per 64-sample stride: 428 MFMAs + ~580 vector FMAs, ~130 KB of LDS traffic.  Accumulators of all
// parameter gradients stay in registers across the strides of a (persistent) wave; one atomic flush
// per wave at the end.
#include "shade_mfma.hiph"

#include "sh_basis.hiph"

#include <cstdlib>

namespace
{

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kOut1 = 16;   // field head width
constexpr int kIn2 = 32;    // shader input = 16 shading features + 16 SH
constexpr int kHid = 64;    // shader hidden width
constexpr float kEps = 1e-3f;

template <int C>
struct MShape
{
  static_assert(C % 8 == 0 && C <= 64, "MFMA path: C must be 8, 16, 32 or 64");
  static constexpr int kS1 = C / 4;    // k-steps of the head layer (quarter q owns channels q*kS1 ..)
  static constexpr int kM6 = (C + 15) / 16;  // 16-channel tiles of enc (C = 8: half a tile, zero padded)
  // weight operand slots (64 floats each, one per lane)
  static constexpr int oWA1 = 0;               // [t]        w_h[m][q*kS1 + t]
  static constexpr int oWA2 = oWA1 + kS1;      // [M*8 + t]  w1[16M+m][kappa2(t, q)]
  static constexpr int oWA5 = oWA2 + 32;       // [M*4 + r]  w1[16M+4q+r][m]
  static constexpr int oWA6 = oWA5 + 16;       // [M'*4 + r] w_h[4q+r][16M'+m]
  static constexpr int kSlots = oWA6 + kM6 * 4;
  static constexpr int oW2Q = kSlots * 64;     // [M][q][c] float4 over r: w2[c][16M+4q+r] (vector output layer)
  static constexpr int oBh = oW2Q + 4 * 4 * 3 * 4;  // b_h[16]
  static constexpr int oB1 = oBh + 16;         // b1[64]
  static constexpr int oB2 = oB1 + 64;         // b2[3], 0
  static constexpr int kWFloats = oB2 + 4;
  // per-wave S-layout tiles
  static constexpr int kP = 72;                // row pitch: 16-byte aligned rows; with 72 the float4
                                               // reads (row m, column 16u+4q) are conflict-free in
                                               // ds_read_b128's four 16-lane groups (68 is 2-way)
  static constexpr int oXS = 0;                // [32][kP]  X
  static constexpr int oES = oXS + 32 * kP;    // [C][kP]   enc
  static constexpr int oPS = oES + kM6 * 16 * kP;  // [16][kP]  one M-tile of relu(pre) / d_hid, then d_h
  static constexpr int oDS = oPS + 16 * kP;    // [4][kP]   d_o rows 0..2
  static constexpr int kAccs = 89 + 4 * kM6;   // accumulator registers a wave hands to wave 0 at the end
  static constexpr int kWaveFloats = (oDS + 4 * kP > kAccs * 64) ? oDS + 4 * kP : kAccs * 64;
  // One wave per SIMD.  The live state of a stride (64 pre-activations, 65 accumulators, operands in
  // flight) does not fit the 256 registers a wave gets at two per SIMD: hipcc then parks the
  // accumulators in scratch and every reload drains vmcnt (2.0-2.6 ms per 8.4 M samples); with 512
  // registers there is no spill and room to prefetch the next stride's inputs (measured 1.8 ms
  // before, see DESIGN.md, the prefetch).
  static constexpr int kWaves = 4;
  static constexpr int kLdsFloats = kWFloats + kWaves * kWaveFloats;
  static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
};

// V & 1: the scheduler may mix the phases of a stride (the default since the output layer moved to
// the vector pipe: 1.47 vs 1.54 ms per 8.4 M samples -- its vector phases then fill the gaps of the
// matrix phases around them; with every phase on the matrix cores, round 2, the fenced form was 5 %
// faster).  F2N_OPT_SHADE_VARIANT = 1 puts the fences back, for measurements.
template <int V>
__device__ __forceinline__ void phase_fence_v()
{
  if constexpr (!(V & 1)) __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void wave_lds_sync()
{
  // one wavefront's LDS operations execute in order; this only stops the compiler from moving the
  // transposed reads above the writes of other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row[byte_off / 4] with a wave-uniform row pointer: global_load_dword v, v_off, s[row:row+1]
// (Off = uint32_t: the saddr + 32-bit voffset form above; uint64_t, the WIDE kernels for C * n * 4 >=
// 2^32: a 64-bit add per access and the vaddr form -- ~6 % slower, against the 2-3x of falling back
// to the vector kernels)
template <typename T, typename Off>
__device__ __forceinline__ T ld_row(const T * row, Off byte_off)
{
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(row) + byte_off);
}
template <typename Off>
__device__ __forceinline__ void st_row(float * row, Off byte_off, float v)
{
  *reinterpret_cast<float *>(reinterpret_cast<char *>(row) + byte_off) = v;
}
template <bool WIDE>
struct RowOffset
{
  typedef uint32_t type;
};
template <>
struct RowOffset<true>
{
  typedef uint64_t type;
};

// Pins global loads where they are written: without it the scheduler sinks them to their first use,
// ~200 MFMAs later, and the wave then waits a full memory latency there.
__device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); }

// max(x, 0) in one instruction: as integers, positive floats order like floats and every negative
// float (and -0) is a negative integer.  fmaxf / v_med3 put a canonicalising v_max x, x in front.
// (Not inline asm: the hazard recogniser does not see an asm's read of a register an MFMA is
// still writing.)
__device__ __forceinline__ float relu(float x)
{
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}

// sum over the 16 lanes of a quarter (a DPP row), result in every lane of the quarter
__device__ __forceinline__ float quarter_sum(float v)
{
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

template <int C, int V, bool WIDE>
__global__ __launch_bounds__(MShape<C>::kWaves * 64) void shade_bwd_mfma_kernel(
  const float * __restrict__ enc, const float * __restrict__ dirs,
  const int32_t * __restrict__ sample_img, const float * __restrict__ p_w_h,
  const float * __restrict__ p_b_h, const float * __restrict__ p_w1,
  const float * __restrict__ p_b1, const float * __restrict__ p_w2,
  const float * __restrict__ p_b2, const float * __restrict__ p_emb,
  const float * __restrict__ d_logit, const float * __restrict__ d_rgb, float * __restrict__ d_enc,
  float * __restrict__ g_w_h, float * __restrict__ g_b_h, float * __restrict__ g_w1,
  float * __restrict__ g_b1, float * __restrict__ g_w2, float * __restrict__ g_b2,
  float * __restrict__ g_emb, int64_t n)
{
  using S = MShape<C>;
  constexpr int kS1 = S::kS1, kM6 = S::kM6, kP = S::kP;
  __shared__ __attribute__((aligned(16))) float lds_all[S::kLdsFloats];
  float * lds_w = lds_all;

  // ---- stage the weight operands, pre-arranged per lane
  for (int i = threadIdx.x; i < S::kSlots * 64; i += S::kWaves * 64) {
    const int slot = i >> 6, l = i & 63, q = l >> 4, m = l & 15;
    float v;
    if (slot < S::oWA2) {
      const int t = slot - S::oWA1;
      v = p_w_h[m * C + q * kS1 + t];
    } else if (slot < S::oWA5) {
      const int M = (slot - S::oWA2) >> 3, t = (slot - S::oWA2) & 7;
      const int k = (t < 4) ? 4 * q + t : 16 + 4 * q + (t - 4);
      v = p_w1[(16 * M + m) * kIn2 + k];
    } else if (slot < S::oWA6) {
      const int M = (slot - S::oWA5) >> 2, r = (slot - S::oWA5) & 3;
      v = p_w1[(16 * M + 4 * q + r) * kIn2 + m];
    } else {
      const int M = (slot - S::oWA6) >> 2, r = (slot - S::oWA6) & 3;
      v = (16 * M + m < C) ? p_w_h[(4 * q + r) * C + 16 * M + m] : 0.f;
    }
    lds_w[i] = v;
  }
  if (threadIdx.x < 192) {  // output-layer weights of the vector path: [M][q][c][r] = w2[c][16M+4q+r]
    const int i = threadIdx.x, r = i & 3, c = (i >> 2) % 3, Mq = i / 12;
    lds_w[S::oW2Q + i] = p_w2[c * kHid + 16 * (Mq >> 2) + 4 * (Mq & 3) + r];
  }
  if (threadIdx.x < 16) lds_w[S::oBh + threadIdx.x] = p_b_h[threadIdx.x];
  if (threadIdx.x < 64) lds_w[S::oB1 + threadIdx.x] = p_b1[threadIdx.x];
  if (threadIdx.x < 4) lds_w[S::oB2 + threadIdx.x] = (threadIdx.x < 3) ? p_b2[threadIdx.x] : 0.f;

  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int q = lane >> 4, m = lane & 15;
  float * tile = lds_all + S::kWFloats + wave * S::kWaveFloats;
  float * XS = tile + S::oXS;
  float * ES = tile + S::oES;
  float * PS = tile + S::oPS;
  float * DS = tile + S::oDS;
  if constexpr (C % 16 != 0) {  // enc rows beyond C (read as zeros by the d w_h product)
    for (int r = C; r < kM6 * 16; r++) ES[r * kP + lane] = 0.f;
  }
  __syncthreads();

  const float * wop = lds_w + lane;  // slot s of this lane: wop[s * 64]
  // With 512 registers the weight operands stay in registers for the whole kernel (C <= 32: 84 of
  // them); read just in time from LDS they cost a full LDS latency every eight products.
  // (the head and hidden layers' operands, slots below oWA5; the data-gradient operands of the last
  // two phases are read when needed: the vector phases of the output layer need their registers)
  constexpr bool kWReg = C <= 32;
  constexpr int kRegLo = S::oWA2, kRegHi = S::oWA5;  // the hidden layer's 32 slots
  float wreg[kWReg ? kRegHi - kRegLo : 1];
  if constexpr (kWReg) {
#pragma unroll
    for (int i = kRegLo; i < kRegHi; i++) wreg[i - kRegLo] = wop[i * 64];
  }
  auto W = [&](int slot) {
    const bool in_reg = kWReg && slot >= kRegLo && slot < kRegHi;
    return in_reg ? wreg[in_reg ? slot - kRegLo : 0] : wop[slot * 64];
  };
  // byte offsets of this quarter's first enc / d_enc row (quarter q owns channels q*kS1.., rows 4q..)
  using Off = typename RowOffset<WIDE>::type;
  const Off cE = (Off)((int64_t)(q * kS1) * n * 4), cD = (Off)((int64_t)(4 * q) * n * 4);
  const bool has_emb = (p_emb != nullptr) && (sample_img != nullptr);

  // per-lane accumulators that live across all strides of this wave
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc_w1[4][2];   // d w1[16M+4q+r][16N+m]
  f32x4 acc_w2[3][4];   // d w2[c][16M+4q+r], partial over this lane's samples (vector FMAs)
  f32x4 acc_wh[kM6];    // d w_h[4q+r][16N+m]
  float acc_b1[4];      // d b1[16M+m], partial over this quarter's k-steps (S-layout reads)
  f32x4 acc_bh = zero4; // d b_h[4q+r]
  float acc_b2 = 0.f;   // d b2[q]
  f32x4 acc_emb = zero4;  // d emb[emb_img][4q+r], partial over this lane's samples
  int emb_img = -1;
  auto flush_emb = [&]() {
    if (emb_img >= 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float t = quarter_sum(acc_emb[r]);
        if (m == 0) atomicAdd(g_emb + (int64_t)emb_img * kOut1 + 4 * q + r, t);
      }
    }
    acc_emb = zero4;
  };
#pragma unroll
  for (int M = 0; M < 4; M++) {
    acc_w1[M][0] = acc_w1[M][1] = zero4;
    acc_w2[0][M] = acc_w2[1][M] = acc_w2[2][M] = zero4;
    acc_b1[M] = 0.f;
  }
#pragma unroll
  for (int N = 0; N < kM6; N++) acc_wh[N] = zero4;

  const int64_t n_strides = (n + 63) / 64;
  const int64_t wave_global = (int64_t)blockIdx.x * S::kWaves + wave;
  const int64_t wave_count = (int64_t)gridDim.x * S::kWaves;
  // inputs of one stride that depend on nothing: issued together (one memory latency), and for
  // the NEXT stride while the current one is in its hidden layer
  float eB[kS1][4];
  int img[4] = {0, 0, 0, 0};
  float dir[3];
  auto load_inputs = [&](int64_t st_, float (&eB_)[kS1][4], int (&img_)[4], float (&dir_)[3]) {
    const int64_t s0_ = st_ * 64;  // may lie beyond the end: every index is clamped to a real sample
    uint32_t off_[4];
#pragma unroll
    for (int T = 0; T < 4; T++) off_[T] = (uint32_t)(((s0_ + 16 * T + m < n) ? s0_ + 16 * T + m : n - 1) * 4);
#pragma unroll
    for (int t = 0; t < kS1; t++)
#pragma unroll
      for (int T = 0; T < 4; T++) eB_[t][T] = ld_row(enc + (int64_t)t * n, off_[T] + cE);
    if (has_emb) {
#pragma unroll
      for (int T = 0; T < 4; T++) img_[T] = ld_row(sample_img, off_[T]);
    }
    const uint32_t offL = (uint32_t)(((s0_ + lane < n) ? s0_ + lane : n - 1) * 12);
#pragma unroll
    for (int k = 0; k < 3; k++) dir_[k] = ld_row(dirs + k, offL);
  };
  load_inputs(wave_global, eB, img, dir);
  issue_fence();

  for (int64_t st = wave_global; st < n_strides; st += wave_count) {
    const int64_t s0 = st * 64;
    // this lane's Q-layout samples 16T+m (clamped: invalid ones read a real sample and get zero
    // gradients).  Everything per sample is addressed as wave-uniform pointer + 32-bit byte offset
    // (the launcher guarantees C * n * 4 < 2^32): four offset registers serve all rows.
    bool vT[4];
    uint32_t offS[4];
#pragma unroll
    for (int T = 0; T < 4; T++) {
      const int64_t s = s0 + 16 * T + m;
      vT[T] = s < n;
      offS[T] = (uint32_t)((vT[T] ? s : n - 1) * 4);
    }

    // ---- head layer: h[4q+r][s] = w_h . enc + b_h
    f32x4 h[4];
#pragma unroll
    for (int T = 0; T < 4; T++) h[T] = *reinterpret_cast<const f32x4 *>(lds_w + S::oBh + 4 * q);
#pragma unroll
    for (int t = 0; t < kS1; t++) {
      const float a = W(S::oWA1 + t);
#pragma unroll
      for (int T = 0; T < 4; T++) h[T] = mfma16(a, eB[t][T], h[T]);
    }
    // enc's S-layout image for d w_h at the end of the stride
#pragma unroll
    for (int t = 0; t < kS1; t++)
#pragma unroll
      for (int T = 0; T < 4; T++) ES[(q * kS1 + t) * kP + 16 * T + m] = eB[t][T];

    phase_fence_v<V>();
    // ---- shader input X: rows 0..15 = [1, h[1..15]] (+ embedding), rows 16..31 = SH16(dir)
    f32x4 Xh[4];
    int img_cur[4];
#pragma unroll
    for (int T = 0; T < 4; T++) img_cur[T] = img[T];
    {
      f32x4 e4[4];
      if (has_emb) {
#pragma unroll
        for (int T = 0; T < 4; T++)
          e4[T] = *reinterpret_cast<const f32x4 *>(p_emb + (int64_t)img_cur[T] * kOut1 + 4 * q);
      }
      // SH: lane l evaluates sample s0 + l and writes its column of the S-layout tile directly
      float sh[16];
      sh_basis<4>(dir[0], dir[1], dir[2], sh);
#pragma unroll
      for (int k = 0; k < 16; k++) XS[(16 + k) * kP + lane] = sh[k];
#pragma unroll
      for (int T = 0; T < 4; T++) {
        Xh[T] = h[T];
        if (q == 0) Xh[T][0] = 1.f;
        if (has_emb) Xh[T] += e4[T];
#pragma unroll
        for (int r = 0; r < 4; r++) XS[(4 * q + r) * kP + 16 * T + m] = Xh[T][r];
      }
    }
    wave_lds_sync();

    phase_fence_v<V>();
    // ---- hidden layer: pre[16M+4q+r][s] = w1 . X + b1
    f32x4 pre[4][4];
    {
      float Xs[4][4];  // SH rows 16+4q+t' of this lane's samples
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int T = 0; T < 4; T++) Xs[t][T] = XS[(16 + 4 * q + t) * kP + 16 * T + m];
#pragma unroll
      for (int M = 0; M < 4; M++) {
#pragma unroll
        for (int T = 0; T < 4; T++)
          pre[M][T] = *reinterpret_cast<const f32x4 *>(lds_w + S::oB1 + 16 * M + 4 * q);
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const float a = W(S::oWA2 + M * 8 + t);
#pragma unroll
          for (int T = 0; T < 4; T++)
            pre[M][T] = mfma16(a, (t < 4) ? Xh[T][t] : Xs[t - 4][T], pre[M][T]);
        }
      }
    }

    phase_fence_v<V>();
    // ---- output layer on the vector pipe: per-quarter partial sums over this lane's 16 hidden
    // neurons (pre becomes relu(pre): its sign is all the ReLU's backward needs), the four quarters
    // summed by one all-ones product per (c, sample tile); then d_o for c = q
    float d_o[4];
    {
      float g_rgb[4];  // d_rgb[s][c = q]: in flight during the sums below
#pragma unroll
      for (int T = 0; T < 4; T++) g_rgb[T] = ld_row(d_rgb, 3 * offS[T] + ((q < 3) ? 4 * q : 8));
      issue_fence();
      float po[3][4];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int T = 0; T < 4; T++) po[c][T] = 0.f;
#pragma unroll
      for (int M = 0; M < 4; M++) {
        f32x4 wq[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
          wq[c] = *reinterpret_cast<const f32x4 *>(lds_w + S::oW2Q + ((M * 4 + q) * 3 + c) * 4);
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float p = relu(pre[M][T][r]);
            pre[M][T][r] = p;
#pragma unroll
            for (int c = 0; c < 3; c++) po[c][T] = __builtin_fmaf(wq[c][r], p, po[c][T]);
          }
      }
      const float b2q = lds_w[S::oB2 + q];
#pragma unroll
      for (int T = 0; T < 4; T++) {
        const float s0 = mfma16(1.f, po[0][T], zero4)[0];
        const float s1 = mfma16(1.f, po[1][T], zero4)[0];
        const float s2 = mfma16(1.f, po[2][T], zero4)[0];
        const float o = ((q == 0) ? s0 : (q == 1) ? s1 : s2) + b2q;
        const float sg = 1.f / (1.f + expf(-o));
        const float g = (vT[T] && q < 3) ? g_rgb[T] : 0.f;
        d_o[T] = g * (1.f + 2.f * kEps) * sg * (1.f - sg);
        acc_b2 += d_o[T];
        if (q < 3) DS[q * kP + 16 * T + m] = d_o[T];
      }
    }
    wave_lds_sync();

    phase_fence_v<V>();
    // ---- d w2[c][j] += sum_s d_o[c][s] relu(pre)[j][s] and d_hid[j][s] = [pre > 0] sum_c w2[c][j]
    // d_o[c][s], both on the registers that hold relu(pre): d_o of all three colours comes back from
    // the LDS tile (lane (q, m) computed colour q only); d_hid replaces pre
    {
      float doc[3][4];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int T = 0; T < 4; T++) doc[c][T] = DS[c * kP + 16 * T + m];
#pragma unroll
      for (int M = 0; M < 4; M++) {
        f32x4 wq[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
          wq[c] = *reinterpret_cast<const f32x4 *>(lds_w + S::oW2Q + ((M * 4 + q) * 3 + c) * 4);
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            float t = doc[c][0] * pre[M][0][r];
#pragma unroll
            for (int T = 1; T < 4; T++) t = __builtin_fmaf(doc[c][T], pre[M][T][r], t);
            acc_w2[c][M][r] += t;
          }
#pragma unroll
          for (int T = 0; T < 4; T++) {
            float dh = wq[0][r] * doc[0][T];
            dh = __builtin_fmaf(wq[1][r], doc[1][T], dh);
            dh = __builtin_fmaf(wq[2][r], doc[2][T], dh);
            pre[M][T][r] = (pre[M][T][r] > 0.f) ? dh : 0.f;
          }
        }
      }
    }

    phase_fence_v<V>();
    // ---- d w1[j][i] += sum_s d_hid[j][s] X[i][s], the four neuron tiles streaming through the LDS
    // tile as above
    {
      auto put = [&](int M) {
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
          for (int r = 0; r < 4; r++) PS[(4 * q + r) * kP + 16 * T + m] = pre[M][T][r];
      };
      put(0);
      wave_lds_sync();
#pragma unroll
      for (int M = 0; M < 4; M++) {
        f32x4 a4[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
          a4[u] = *reinterpret_cast<const f32x4 *>(PS + m * kP + 16 * u + 4 * q);
        wave_lds_sync();
        if (M < 3) {
          put(M + 1);
          wave_lds_sync();
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc_b1[M] += (a4[u][0] + a4[u][1]) + (a4[u][2] + a4[u][3]);
          f32x4 b4[2];
#pragma unroll
          for (int N = 0; N < 2; N++)
            b4[N] = *reinterpret_cast<const f32x4 *>(XS + (16 * N + m) * kP + 16 * u + 4 * q);
#pragma unroll
          for (int k = 0; k < 4; k++)
#pragma unroll
            for (int N = 0; N < 2; N++) acc_w1[M][N] = mfma16(a4[u][k], b4[N][k], acc_w1[M][N]);
        }
      }
    }

    // ---- the next stride's inputs (their registers are free since the head layer / the SH block;
    // issued behind the vector phases and the d w1 product, which leave no registers for them: the
    // 128 products of the last three phases, ~4 K cycles, cover the latency -- asking one phase
    // earlier, 260 products ahead, measured the same within noise, 1.45-1.50 ms, and spills the WIDE
    // instantiation)
    load_inputs(st + wave_count, eB, img, dir);
    issue_fence();

    phase_fence_v<V>();
    // ---- d_X rows 0..15 = w1[:, 0:16]^T . d_hid   (the SH inputs carry no gradient)
    float g_logit[4];  // in flight during the products below
#pragma unroll
    for (int T = 0; T < 4; T++) g_logit[T] = ld_row(d_logit, offS[T]);
    issue_fence();
    f32x4 dX[4];
#pragma unroll
    for (int T = 0; T < 4; T++) dX[T] = zero4;
#pragma unroll
    for (int M = 0; M < 4; M++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float a = W(S::oWA5 + M * 4 + r);
#pragma unroll
        for (int T = 0; T < 4; T++) dX[T] = mfma16(a, pre[M][T][r], dX[T]);
      }

    phase_fence_v<V>();
    // ---- appearance embedding: d emb[img][i] += d_X[i]  (row 0 included: X[0] = 1 + emb[0]).
    // A chunk of rays usually comes from ONE image: its row's gradient is accumulated in registers
    // and flushed when the image changes / at the end.  (One atomic per stride on the same 64 bytes,
    // from every wave of the chip, serialises in L2: 6.9 ms instead of 1.7 ms per launch.)
    if (has_emb) {
      const int img0 = __builtin_amdgcn_readfirstlane(img_cur[0]);
      if (__all(img_cur[0] == img0 && img_cur[1] == img0 && img_cur[2] == img0 && img_cur[3] == img0)) {
        if (img0 != emb_img) {
          flush_emb();
          emb_img = img0;
        }
        acc_emb += (dX[0] + dX[1]) + (dX[2] + dX[3]);  // clamped samples carry zeros
      } else {
#pragma unroll
        for (int T = 0; T < 4; T++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (vT[T]) atomicAdd(g_emb + (int64_t)img_cur[T] * kOut1 + 4 * q + r, dX[T][r]);
      }
    }

    // ---- d_h: the head outputs' gradient; row 0 is the density logit's
    f32x4 d_h[4];
#pragma unroll
    for (int T = 0; T < 4; T++) {
      d_h[T] = dX[T];
      if (q == 0) d_h[T][0] = vT[T] ? g_logit[T] : 0.f;
      acc_bh += d_h[T];
#pragma unroll
      for (int r = 0; r < 4; r++) PS[(4 * q + r) * kP + 16 * T + m] = d_h[T][r];
    }

    phase_fence_v<V>();
    // ---- d_enc[c][s] = w_h^T . d_h
#pragma unroll
    for (int M = 0; M < kM6; M++) {
      f32x4 de[4];
#pragma unroll
      for (int T = 0; T < 4; T++) de[T] = zero4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float a = W(S::oWA6 + M * 4 + r);
#pragma unroll
        for (int T = 0; T < 4; T++) de[T] = mfma16(a, d_h[T][r], de[T]);
      }
#pragma unroll
      for (int T = 0; T < 4; T++)
        if (vT[T]) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (C % 16 == 0 || 16 * M + 4 * q + r < C)
              st_row(d_enc + (int64_t)(16 * M + r) * n, offS[T] + cD, de[T][r]);
        }
    }

    phase_fence_v<V>();
    // ---- d w_h[i][c] += sum_s d_h[i][s] enc[c][s]
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(PS + m * kP + 16 * u + 4 * q);
#pragma unroll
      for (int N = 0; N < kM6; N++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(ES + (16 * N + m) * kP + 16 * u + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; k++) acc_wh[N] = mfma16(a4[k], b4[k], acc_wh[N]);
      }
    }
    wave_lds_sync();
  }

  // ---- flush the accumulators: the four waves' sums meet in wave 0 first (its LDS tile regions are
  // free now), so a workgroup sends one set of atomics instead of four -- with few strides per wave
  // (a 512-ray training batch: 8) the 4 160 atomics of every wave on the same 2 400 addresses were a
  // third of the kernel
  if (has_emb) flush_emb();
  {
    auto each_acc = [&](auto && fn) {
      int i = 0;
#pragma unroll
      for (int M = 0; M < 4; M++) {
#pragma unroll
        for (int N = 0; N < 2; N++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc_w1[M][N][r] = fn(acc_w1[M][N][r], i++);
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc_w2[c][M][r] = fn(acc_w2[c][M][r], i++);
        acc_b1[M] = fn(acc_b1[M], i++);
      }
#pragma unroll
      for (int N = 0; N < kM6; N++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc_wh[N][r] = fn(acc_wh[N][r], i++);
#pragma unroll
      for (int r = 0; r < 4; r++) acc_bh[r] = fn(acc_bh[r], i++);
      acc_b2 = fn(acc_b2, i++);
    };
    static_assert(S::kAccs * 64 <= S::kWaveFloats, "the accumulators fit a wave's tile region");
    wave_lds_sync();
    if (wave != 0) each_acc([&](float v, int i) { tile[i * 64 + lane] = v; return v; });
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < S::kWaves; w++) {
      const float * other = lds_all + S::kWFloats + w * S::kWaveFloats;
      each_acc([&](float v, int i) { return v + other[i * 64 + lane]; });
    }
  }
#pragma unroll
  for (int M = 0; M < 4; M++) {
#pragma unroll
    for (int N = 0; N < 2; N++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        atomicAdd(g_w1 + (16 * M + 4 * q + r) * kIn2 + 16 * N + m, acc_w1[M][N][r]);
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float t = quarter_sum(acc_w2[c][M][r]);  // over the quarter's 16 sample columns
        if (m == 0) atomicAdd(g_w2 + c * kHid + 16 * M + 4 * q + r, t);
      }
    {
      float t = acc_b1[M];  // lane (q, m): neuron 16M+m, this quarter's samples
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (q == 0) atomicAdd(g_b1 + 16 * M + m, t);
    }
  }
#pragma unroll
  for (int N = 0; N < kM6; N++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (C % 16 == 0 || 16 * N + m < C) atomicAdd(g_w_h + (4 * q + r) * C + 16 * N + m, acc_wh[N][r]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float t = quarter_sum(acc_bh[r]);
    if (m == 0) atomicAdd(g_b_h + 4 * q + r, t);
  }
  {
    const float t = quarter_sum(acc_b2);
    if (m == 0 && q < 3) atomicAdd(g_b2 + q, t);
  }
}


// ---- forward on the matrix cores ------------------------------------------------------------------
// The forward third of the kernel above (head, hidden and output layer in the Q-layout, no lane
// movement), four waves per SIMD (three until the end of round 3; two, three, four: 0.584, 0.577,
// 0.551 ms per 8.4 M samples alone): no accumulators live across strides and the other waves hide one
// wave's loads.  The output layer STAYS on the matrix cores here (rows 0, 4, 8 of a 16-row operand):
// the vector form the backward uses was measured here too and lost, 0.58 against 0.55 ms per 8.2 M
// samples inside the bench -- 52 products fewer but 200 vector instructions more and, at 168
// registers, ten spilled; this kernel waits on latencies, not on the matrix pipe.  LDS: the three forward weight operands (56 slots at
// C = 32) + a 16-row tile per wave that moves SH16(dir) from lane = sample into the Q-layout.
template <int C, int W = 12>
struct FShape
{
  static_assert(C % 8 == 0 && C <= 64, "MFMA path: C must be 8, 16, 32 or 64");
  static constexpr int kS1 = C / 4;
  static constexpr int oWA1 = 0, oWA2 = oWA1 + kS1, oWA3 = oWA2 + 32, kSlots = oWA3 + 16;
  static constexpr int oBh = kSlots * 64, oB1 = oBh + 16, oB2 = oB1 + 64, kWFloats = oB2 + 4;
  static constexpr int kP = 68;             // [16][kP] SH tile per wave (b32 accesses only)
  static constexpr int kWaveFloats = 16 * kP;
  static constexpr int kWaves = W;   // 16 = four per SIMD (the default: the allocator fits 128 registers
                                     // without spilling; 0.552 -> 0.536 ms per 8.2 M samples), 12 = three, 8 = two
  static constexpr int kLdsFloats = kWFloats + kWaves * kWaveFloats;
};

template <int C, bool WIDE, int W>
__global__ __launch_bounds__(W * 64) void shade_fwd_mfma_kernel(
  const float * __restrict__ enc, const float * __restrict__ dirs,
  const int32_t * __restrict__ sample_img, const float * __restrict__ p_w_h,
  const float * __restrict__ p_b_h, const float * __restrict__ p_w1,
  const float * __restrict__ p_b1, const float * __restrict__ p_w2,
  const float * __restrict__ p_b2, const float * __restrict__ p_emb, float * __restrict__ logit,
  float * __restrict__ rgb, float * __restrict__ pre_out, int64_t n)
{
  using S = FShape<C, W>;
  constexpr int kS1 = S::kS1, kP = S::kP;
  __shared__ __attribute__((aligned(16))) float lds_all[S::kLdsFloats];
  float * lds_w = lds_all;
  for (int i = threadIdx.x; i < S::kSlots * 64; i += S::kWaves * 64) {
    const int slot = i >> 6, l = i & 63, q = l >> 4, m = l & 15;
    float v;
    if (slot < S::oWA2) {
      v = p_w_h[m * C + q * kS1 + (slot - S::oWA1)];
    } else if (slot < S::oWA3) {
      const int M = (slot - S::oWA2) >> 3, t = (slot - S::oWA2) & 7;
      v = p_w1[(16 * M + m) * kIn2 + ((t < 4) ? 4 * q + t : 16 + 4 * q + (t - 4))];
    } else {
      const int M = (slot - S::oWA3) >> 2, r = (slot - S::oWA3) & 3;
      v = ((m & 3) == 0 && m < 12) ? p_w2[(m >> 2) * kHid + 16 * M + 4 * q + r] : 0.f;
    }
    lds_w[i] = v;
  }
  if (threadIdx.x < 16) lds_w[S::oBh + threadIdx.x] = p_b_h[threadIdx.x];
  if (threadIdx.x < 64) lds_w[S::oB1 + threadIdx.x] = p_b1[threadIdx.x];
  if (threadIdx.x < 4) lds_w[S::oB2 + threadIdx.x] = (threadIdx.x < 3) ? p_b2[threadIdx.x] : 0.f;
  __syncthreads();

  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int q = lane >> 4, m = lane & 15;
  float * XS = lds_all + S::kWFloats + wave * S::kWaveFloats;  // SH rows 0..15 of this wave
  const float * wop = lds_w + lane;
  using Off = typename RowOffset<WIDE>::type;
  const Off cE = (Off)((int64_t)(q * kS1) * n * 4), cP = (Off)((int64_t)(4 * q) * n * 4);
  const bool has_emb = (p_emb != nullptr) && (sample_img != nullptr);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const int64_t n_strides = (n + 63) / 64;
  const int64_t wave_count = (int64_t)gridDim.x * S::kWaves;
  for (int64_t st = (int64_t)blockIdx.x * S::kWaves + wave; st < n_strides; st += wave_count) {
    const int64_t s0 = st * 64;
    bool vT[4];
    uint32_t offS[4];
#pragma unroll
    for (int T = 0; T < 4; T++) {
      const int64_t s = s0 + 16 * T + m;
      vT[T] = s < n;
      offS[T] = (uint32_t)((vT[T] ? s : n - 1) * 4);
    }
    float eB[kS1][4];
#pragma unroll
    for (int t = 0; t < kS1; t++)
#pragma unroll
      for (int T = 0; T < 4; T++) eB[t][T] = ld_row(enc + (int64_t)t * n, offS[T] + cE);
    int img[4] = {0, 0, 0, 0};
    if (has_emb) {
#pragma unroll
      for (int T = 0; T < 4; T++) img[T] = ld_row(sample_img, offS[T]);
    }
    float dir[3];
    {
      const uint32_t offL = (uint32_t)(((s0 + lane < n) ? s0 + lane : n - 1) * 12);
#pragma unroll
      for (int k = 0; k < 3; k++) dir[k] = ld_row(dirs + k, offL);
    }
    issue_fence();

    // ---- head layer
    f32x4 h[4];
#pragma unroll
    for (int T = 0; T < 4; T++) h[T] = *reinterpret_cast<const f32x4 *>(lds_w + S::oBh + 4 * q);
#pragma unroll
    for (int t = 0; t < kS1; t++) {
      const float a = wop[(S::oWA1 + t) * 64];
#pragma unroll
      for (int T = 0; T < 4; T++) h[T] = mfma16(a, eB[t][T], h[T]);
    }
    if (q == 0) {
#pragma unroll
      for (int T = 0; T < 4; T++)
        if (vT[T]) st_row(logit, offS[T], h[T][0]);
    }

    // ---- shader input
    f32x4 Xh[4];
    {
      f32x4 e4[4];
      if (has_emb) {
#pragma unroll
        for (int T = 0; T < 4; T++)
          e4[T] = *reinterpret_cast<const f32x4 *>(p_emb + (int64_t)img[T] * kOut1 + 4 * q);
      }
      float sh[16];
      sh_basis<4>(dir[0], dir[1], dir[2], sh);
#pragma unroll
      for (int k = 0; k < 16; k++) XS[k * kP + lane] = sh[k];
#pragma unroll
      for (int T = 0; T < 4; T++) {
        Xh[T] = h[T];
        if (q == 0) Xh[T][0] = 1.f;
        if (has_emb) Xh[T] += e4[T];
      }
    }
    wave_lds_sync();
    float Xs[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int T = 0; T < 4; T++) Xs[t][T] = XS[(4 * q + t) * kP + 16 * T + m];
    wave_lds_sync();

    // ---- hidden layer, then the output layer on rows 0, 4, 8
    f32x4 o[4];
#pragma unroll
    for (int T = 0; T < 4; T++) {
      o[T] = zero4;
      o[T][0] = lds_w[S::oB2 + q];
    }
#pragma unroll
    for (int M = 0; M < 4; M++) {
      f32x4 pre[4];
#pragma unroll
      for (int T = 0; T < 4; T++)
        pre[T] = *reinterpret_cast<const f32x4 *>(lds_w + S::oB1 + 16 * M + 4 * q);
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const float a = wop[(S::oWA2 + M * 8 + t) * 64];
#pragma unroll
        for (int T = 0; T < 4; T++)
          pre[T] = mfma16(a, (t < 4) ? Xh[T][t] : Xs[t - 4][T], pre[T]);
      }
      if (pre_out) {
#pragma unroll
        for (int T = 0; T < 4; T++)
          if (vT[T]) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              st_row(pre_out + (int64_t)(16 * M + r) * n, offS[T] + cP, pre[T][r]);
          }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float a = wop[(S::oWA3 + M * 4 + r) * 64];
#pragma unroll
        for (int T = 0; T < 4; T++) o[T] = mfma16(a, relu(pre[T][r]), o[T]);
      }
    }
    if (q < 3) {
#pragma unroll
      for (int T = 0; T < 4; T++)
        if (vT[T])
          st_row(rgb + q, 3 * offS[T], (1.f + 2.f * kEps) / (1.f + expf(-o[T][0])) - kEps);
    }
  }
}

}  // namespace

namespace f2n_detail
{

bool shade_bwd_mfma_supports(int C, int64_t n)
{
  // (per-sample offsets such as 12 s for d_rgb stay 32-bit: n < 2^28; the row offsets C n 4 are
  // 32-bit below 2^30 elements and 64-bit, the WIDE kernels, above)
  return (C == 8 || C == 16 || C == 32 || C == 64) && n < ((int64_t)1 << 28);
}

int launch_shade_bwd_mfma(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, const float * d_logit, const float * d_rgb, float * d_enc_cm,
  float * g_w_h, float * g_b_h, float * g_w1, float * g_b1, float * g_w2, float * g_b2,
  float * g_app_emb, int64_t n, hipStream_t stream)
{
  const int64_t n_strides = (n + 63) / 64;
  // the embedding rows are read as float4
  if (app_emb && (reinterpret_cast<uintptr_t>(app_emb) & 15u)) return F2N_E_INVALID_ARG;
  if (n >= ((int64_t)1 << 28)) return F2N_E_UNSUPPORTED;  // 32-bit per-sample offsets
  const bool wide = (int64_t)C * n >= ((int64_t)1 << 30);   // row offsets beyond 32 bits
  const int variant = f2n_get_option(F2N_OPT_SHADE_VARIANT);
#define F2N_LAUNCH_MFMA_V(CC, VV, WW)                                                                \
  {                                                                                                  \
    constexpr int kW = MShape<CC>::kWaves;                                                             \
    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_strides + kW - 1) / kW);               \
    hipLaunchKernelGGL(                                                                              \
      (shade_bwd_mfma_kernel<CC, VV, WW>), dim3(grid), dim3(kW * 64), 0, stream, enc_cm, dirs,        \
      sample_img, w_h, b_h, w1, b1, w2, b2, app_emb, d_logit, d_rgb, d_enc_cm, g_w_h, g_b_h, g_w1,    \
      g_b1, g_w2, g_b2, g_app_emb, n);                                                               \
  }
#define F2N_LAUNCH_MFMA(CC)                                \
  if (wide) F2N_LAUNCH_MFMA_V(CC, 1, true)                 \
  else switch (variant) {                                  \
      case 1: F2N_LAUNCH_MFMA_V(CC, 0, false) break;       \
      default: F2N_LAUNCH_MFMA_V(CC, 1, false) break;      \
    }
  switch (C) {
    case 8: F2N_LAUNCH_MFMA(8) break;
    case 16: F2N_LAUNCH_MFMA(16) break;
    case 32: F2N_LAUNCH_MFMA(32) break;
    case 64: F2N_LAUNCH_MFMA(64) break;
    default: return F2N_E_UNSUPPORTED;
  }
#undef F2N_LAUNCH_MFMA
#undef F2N_LAUNCH_MFMA_V
  return f2n_launch_status();
}


int launch_shade_fwd_mfma(
  const float * enc_cm, int C, const float * dirs, const int32_t * sample_img, const float * w_h,
  const float * b_h, const float * w1, const float * b1, const float * w2, const float * b2,
  const float * app_emb, float * logit, float * rgb, float * pre_cm, int64_t n, hipStream_t stream)
{
  const int64_t n_strides = (n + 63) / 64;
  if (app_emb && (reinterpret_cast<uintptr_t>(app_emb) & 15u)) return F2N_E_INVALID_ARG;
  if (n >= ((int64_t)1 << 28)) return F2N_E_UNSUPPORTED;  // 32-bit per-sample offsets
  const bool wide = (int64_t)(pre_cm ? 64 : C) * n >= ((int64_t)1 << 30);  // row offsets beyond 32 bits
  const bool two_per_simd = f2n_get_option(F2N_OPT_SHADE_VARIANT) == 2;
#define F2N_LAUNCH_FWD_W(CC, WW, KW)                                                               \
  {                                                                                                \
    constexpr int kW = KW;                                                                         \
    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_strides + kW - 1) / kW);             \
    hipLaunchKernelGGL(                                                                            \
      (shade_fwd_mfma_kernel<CC, WW, KW>), dim3(grid), dim3(kW * 64), 0, stream, enc_cm, dirs,     \
      sample_img, w_h, b_h, w1, b1, w2, b2, app_emb, logit, rgb, pre_cm, n);                       \
  }
#define F2N_LAUNCH_FWD(CC)                                                   \
  if (wide) F2N_LAUNCH_FWD_W(CC, true, 12)                                   \
  else if (two_per_simd) F2N_LAUNCH_FWD_W(CC, false, 8)                      \
  else if (f2n_get_option(F2N_OPT_SHADE_VARIANT) == 3) F2N_LAUNCH_FWD_W(CC, false, 12) \
  else F2N_LAUNCH_FWD_W(CC, false, 16)
  switch (C) {
    case 8: F2N_LAUNCH_FWD(8) break;
    case 16: F2N_LAUNCH_FWD(16) break;
    case 32: F2N_LAUNCH_FWD(32) break;
    case 64: F2N_LAUNCH_FWD(64) break;
    default: return F2N_E_UNSUPPORTED;
  }
#undef F2N_LAUNCH_FWD
#undef F2N_LAUNCH_FWD_W
  return f2n_launch_status();
}

}  // namespace f2n_detail
