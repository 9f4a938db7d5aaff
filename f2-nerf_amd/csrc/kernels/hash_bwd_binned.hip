// hash_bwd_binned.hip -- table gradient of the hash-grid encode without scattered atomics
// (replaces Hash3DAnchoredBackwardKernel's half2 atomics,
// reference src/hash_3d_anchored.cu:95-145,181-218, for training-sized batches).
//
// Every contribution is f16(f16(scale*g) * w_d) per channel, exactly as the reference forms it, i.e.
// an integer multiple of 2^-24 below 2^16.  The pipeline moves those contributions as RECORDS
// {row, F halves} to the workgroup that owns the row and sums them EXACTLY in 64-bit fixed point
// (ds_add_u64: LDS integer atomics run 12x faster than float ones on gfx950), so the table gradient
// does not depend on summation order.  A "slice" is the range of table rows whose accumulators fit
// one workgroup's LDS: kBinAcc / F rows (128 KiB of 64-bit sums).
//
//   pass A (bin)     one 1024-thread workgroup per tile of 1024 points walks all levels; records are
//                    staged per BUCKET in 128 KiB of LDS queues and leave as contiguous regions of the
//                    workspace.  Up to 64 slices per level a bucket IS a slice.  Coarse levels, where
//                    the points of a tile share cells, are first COMBINED per tile: a direct-mapped
//                    LDS table keyed by row (tag, then add: no compare-and-swap loop) sums equal-row
//                    contributions exactly, and only the sums leave, re-expressed as f16 pieces so the
//                    record format does not change.
//   pass B (split)   only when a level has more than 64 slices (T*F > 2^20, e.g. T = 2^22, F = 8:
//                    2048 slices): a bucket is 2^k slices; pass B streams a bucket's regions and
//                    splits them by slice through LDS queues into long per-slice runs.
//   pass C (reduce)  one workgroup per (level, slice): stream the slice's regions / runs, ds_add_u64,
//                    then add the slice into the table gradient with contiguous float atomics (the
//                    reference's level windows overlap, quirk Q2, so it must be an add).
//
// Capacities never affect results.  Pass B (two-level tables): a record that finds its sub-slice
// queue or its run full goes to the ARENA of its (level, bucket) -- an append buffer in the workspace
// that the reduce pass of every slice of that bucket scans after its own runs -- and is summed exactly
// like the rest; only a record that finds the arena full as well is applied directly with a global
// float atomic.  Pass A applies a record that finds its LDS queue full directly: a tile whose points
// pile onto a few rows (the per-tile combine exists for those; measured: routing them through the
// arena costs 1-3 ms per 8.4 M dense samples on single-level tables and 9 % of pass A on config C5).
// A direct add is order-dependent in the last bit, like the reference's own atomics, and counted
// (f2n_hash_bwd_set_overflow_counter).
#include "hash_grid.hiph"

#include <algorithm>
#include <atomic>

namespace
{

constexpr int kBinBlock = 1024;        // threads = points per tile (pass A), threads of pass C
constexpr int kBinAcc = 16384;         // 64-bit accumulators per slice (128 KiB)
// LDS record staging per tile (pass A): 156 KiB, all a CU has next to the counters.  F = 8 with 64
// buckets: 124 records per queue for a mean of 64 per round, 7.5 sigma of an even spread -- at
// 32768 words (100 records, 4.5 sigma) config C5 sent ~2000 of its 2.1e9 records past a full queue.
constexpr int kBinQueueWords = 39936;
constexpr int kCombAccWords = 8192 + 16;  // combine mode: 4096 (+ pitch padding) 64-bit sums, M = 4096 / F slots
constexpr int kCombTagWords = 4096;    //               M row tags (sized for F = 1)
constexpr int kCombQueueWords = kBinQueueWords - kCombAccWords - kCombTagWords;  // 108 KiB stay a queue
constexpr int kMaxPieces = 6;          // f16 pieces per combined sum before the remainder goes atomic
// A combined level costs about twice a plain one (two ds_add_u64 per contribution, 4-way same-address
// inside a wave at the coarsest levels), so it must shrink the record stream a lot to pay:
constexpr int kCombineMinRunQ8 = 5 * 256;  // mean samples per level-0-scaled cell along a ray (x256)
constexpr int kCombinePaysAt = 4500;       // ... or records saved per tile-level, predicted
constexpr int kCombineRepeats = 448;       // ... or lanes (of the 768 that have four lanes before them in
                                           // their DPP row) with a gradient above the f16 underflow whose
                                           // level-0 cell is that of one of those four: samples piled
                                           // onto few cells without forming runs
constexpr int kCombinePiledPerCell = 64;   // ... or non-zero level-0 contributions (estimated from the gradient's
                                           // magnitude) per distinct level-0 cell of the tile: a pile
constexpr int kSplitBlock = 512;       // pass B: four workgroups per CU (it is latency-bound)
constexpr int kSplitQueueWords = 9216;   // 36 KiB: four workgroups per CU
constexpr int kSplitRegions = 2;       // regions a wave of pass B ingests per round
constexpr int kSplitSpill = 128;       // pass B: records past a full sub-slice queue wait here (exactly
                                       // summed like the rest) instead of becoming float atomics
constexpr int kArenaRecords = 32768;   // overflow arena per (level, bucket): records, at most
constexpr int kMaxBuckets = 64;
constexpr int kMaxLog2Sub = 6;

// Record of F channels: word 0 = row inside the bucket, then the F halves packed two per word.
// F <= 2: array of {row, value} pairs (8 bytes).  F >= 4: structure of arrays inside each region
// (all rows, then all value groups of 8 / 16 bytes) so both parts move as full-width vector accesses.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int F>
struct Rec
{
  static constexpr int kValWords = (F >= 2 ? F / 2 : 1);
  static constexpr int kWords = 1 + kValWords;
  static constexpr bool kSoA = (F >= 4);
};

// f16 bits -> value * 2^24 as a signed 64-bit integer (exact for every finite f16)
__device__ __forceinline__ long long f16_bits_to_fixed(uint32_t hbits)
{
  const uint32_t e = (hbits >> 10) & 31u, m = hbits & 0x3ffu;
  const unsigned long long mag =
    e ? ((unsigned long long)(0x400u | m) << (e - 1u)) : (unsigned long long)m;
  return (hbits & 0x8000u) ? -(long long)mag : (long long)mag;
}

__device__ __forceinline__ bool f16_bits_nonfinite(uint32_t hbits) { return (hbits & 0x7c00u) == 0x7c00u; }

// Largest-magnitude f16 (towards zero) of a fixed-point sum; subtracts it from S.  Repeated calls
// re-express any |S| < 2^63 as f16 values whose exact sum is S (11 bits per piece).
__device__ __forceinline__ uint32_t take_f16_piece(long long & S)
{
  if (S == 0) return 0u;
  const bool neg = S < 0;
  const unsigned long long mag = neg ? (unsigned long long)(-S) : (unsigned long long)S;
  const int hb = 63 - __clzll((long long)mag);
  unsigned long long piece;
  uint32_t bits;
  if (hb >= 40) {  // beyond the f16 range: peel off 65504 at a time
    piece = 2047ull << 29;
    bits = 0x7bffu;
  } else {
    const int shift = hb > 10 ? hb - 10 : 0;
    const unsigned long long m = mag >> shift;  // 11 significant bits
    piece = m << shift;
    bits = (uint32_t)((shift << 10) + (int)m);  // exponent field = shift + 1 once bit 10 of m is set
  }
  S = neg ? S + (long long)piece : S - (long long)piece;
  return bits | (neg ? 0x8000u : 0u);
}

template <int F>
__device__ __forceinline__ uint32_t val_channel_bits(const uint32_t * val, int k)
{
  const uint32_t word = val[F >= 2 ? k / 2 : 0];
  return (k & 1) ? (word >> 16) : (word & 0xffffu);
}

// One record straight into the table gradient (queue / region full, non-finite halves).
template <int F>
__device__ __forceinline__ void apply_record_atomic(
  float * __restrict__ gbase, uint32_t row, const uint32_t * val, float inv_scale)
{
#pragma unroll
  for (int k = 0; k < F; k++) {
    const float v = h2f((uint16_t)val_channel_bits<F>(val, k));
    if (v != 0.f) atomicAdd(gbase + (int64_t)row * F + k, v * inv_scale);
  }
}

// The overflow arena of one (level, bucket): records {sub-slice << 16 | row in slice, values}, array of
// structures.  Space is reserved per WAVE and bucket (one global atomic for all lanes of a wave that
// overflow into the same bucket: a tile whose samples pile onto one cell overflows thousands of
// records into one bucket, and one returning atomic per record on one word serialises at ~90 per
// microsecond), and an arena that is already full is not asked again.
struct Arena
{
  uint32_t * counts;   // [L][n_buckets]
  uint32_t * records;  // [L][n_buckets][cap][KW]
  int n_buckets, cap;
};

// Called by any subset of a wave's lanes (divergent callers welcome); false = no room, the caller
// applies the record directly.
template <int F>
__device__ __forceinline__ bool arena_push(
  const Arena & ar, int l, uint32_t bucket, uint32_t sub, uint32_t local, const uint32_t * val)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  if (!ar.counts) return false;
  bool done = false, ok = false;
  while (!done) {  // one turn per distinct bucket among the calling lanes
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bucket);
    if (bucket == b0) {
      const unsigned long long m = __ballot(true);  // the calling lanes with this bucket
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
        (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      const size_t slot = (size_t)l * ar.n_buckets + b0;
      uint32_t base = (uint32_t)ar.cap;
      if (rank == 0) {
        // (an L2-coherent load: never larger than the true count, so a stale value only costs the add)
        if (__hip_atomic_load(ar.counts + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)ar.cap)
          base = atomicAdd(ar.counts + slot, (uint32_t)__popcll(m));
      }
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);  // the first calling lane has rank 0
      const uint32_t pos = base + rank;
      if (pos < (uint32_t)ar.cap) {
        uint32_t * rec = ar.records + (slot * (size_t)ar.cap + pos) * KW;
        rec[0] = (sub << 16) | local;
#pragma unroll
        for (int j = 0; j < VW; j++) rec[1 + j] = val[j];
        ok = true;
      }
      done = true;
    }
  }
  return ok;
}

template <int F>
__device__ __forceinline__ bool val_all_zero(const uint32_t * val)
{
  uint32_t any = 0u;
#pragma unroll
  for (int j = 0; j < Rec<F>::kValWords; j++) any |= val[j];
  return (any & 0x7fff7fffu) == 0u;
}

// LDS / workspace addressing of record `slot` of a region with capacity `cap` starting at `base`.
template <int F>
__device__ __forceinline__ void store_record(uint32_t * base, int cap, uint32_t slot, uint32_t row, const uint32_t * val)
{
  constexpr int VW = Rec<F>::kValWords;
  if constexpr (!Rec<F>::kSoA) {
    *reinterpret_cast<uint2 *>(base + (size_t)slot * 2) = make_uint2(row, val[0]);
  } else {
    base[slot] = row;
    uint32_t * v = base + cap + (size_t)slot * VW;
    if constexpr (VW == 2) *reinterpret_cast<uint2 *>(v) = make_uint2(val[0], val[1]);
    else *reinterpret_cast<uint4 *>(v) = make_uint4(val[0], val[1], val[2], val[3]);
  }
}

template <int F>
__device__ __forceinline__ void load_record(const uint32_t * base, int cap, uint32_t slot, uint32_t & row, uint32_t * val)
{
  constexpr int VW = Rec<F>::kValWords;
  if constexpr (!Rec<F>::kSoA) {
    const uint2 r = *reinterpret_cast<const uint2 *>(base + (size_t)slot * 2);
    row = r.x;
    val[0] = r.y;
  } else {
    row = base[slot];
    const uint32_t * v = base + cap + (size_t)slot * VW;
    if constexpr (VW == 2) {
      const uint2 r = *reinterpret_cast<const uint2 *>(v);
      val[0] = r.x;
      val[1] = r.y;
    } else {
      const uint4 r = *reinterpret_cast<const uint4 *>(v);
      val[0] = r.x;
      val[1] = r.y;
      val[2] = r.z;
      val[3] = r.w;
    }
  }
}

// Copy the first `cnt` records of an LDS queue (capacity src_cap) into a workspace region (capacity
// dst_cap, records dst_off..) with the 64 lanes of one wave.
// ROWS4: the caller owns the whole destination region and appends nothing behind these records (pass
// A's flush), so the row array may be copied in whole uint4s.
template <int F, bool ROWS4 = false>
__device__ __forceinline__ void copy_records(
  const uint32_t * src, int src_cap, uint32_t * dst, int dst_cap, uint32_t dst_off, uint32_t cnt, int lane)
{
  constexpr int VW = Rec<F>::kValWords;
  if constexpr (!Rec<F>::kSoA) {
    const uint2 * s = reinterpret_cast<const uint2 *>(src);
    uint2 * d = reinterpret_cast<uint2 *>(dst) + dst_off;
    for (uint32_t i = lane; i < cnt; i += 64) d[i] = s[i];
  } else {
    if (ROWS4 && (dst_off & 3u) == 0u) {
      // rows four at a time (queue and region capacities are multiples of four records, so the whole
      // uint4 that holds the last row exists on both sides): a 4-byte store per lane moves 256 bytes
      // per instruction, a fifth of the record bytes but half of the flush's store instructions
      const uint4 * s4 = reinterpret_cast<const uint4 *>(src);
      uint4 * d4 = reinterpret_cast<uint4 *>(dst + dst_off);
      for (uint32_t i = lane; i < (cnt + 3u) / 4u; i += 64) d4[i] = s4[i];
    } else {
      for (uint32_t i = lane; i < cnt; i += 64) dst[dst_off + i] = src[i];
    }
    if constexpr (VW == 2) {
      const uint2 * s = reinterpret_cast<const uint2 *>(src + src_cap);
      uint2 * d = reinterpret_cast<uint2 *>(dst + dst_cap) + dst_off;
      for (uint32_t i = lane; i < cnt; i += 64) d[i] = s[i];
    } else {
      const uint4 * s = reinterpret_cast<const uint4 *>(src + src_cap);
      uint4 * d = reinterpret_cast<uint4 *>(dst + dst_cap) + dst_off;
      for (uint32_t i = lane; i < cnt; i += 64) d[i] = s[i];
    }
  }
}

// ------------------------------------------------------------------------------ pass A: bin ----

// scalars of pass A (pointers stay plain __restrict__ kernel arguments: wave-uniform loads through
// them are then scalar loads)
struct BinArgs
{
  int64_t g_ld_point, g_ld_chan;
  int64_t n;        // points of this chunk (pts / grad_out already point at its first point)
  int64_t n_tiles;  // tiles of this chunk
  int64_t level_stride;
  uint32_t T;
  int L;
  float grad_scale, inv_scale;
  int n_buckets, bshift, groups, qcap, qcap_comb, combine;
  uint32_t * stats;  // optional [L][4] u32 counters (tools/ab_hash_bwd.py), else NULL
  unsigned long long * overflow;  // optional: += records applied with float atomics, else NULL
};

// SAT: sum the saturated cell (0,0,0) in LDS (tables that take the split pass); a template
// parameter so that the single-level kernel keeps its code and registers exactly.
template <int F, bool POW2, bool SAT>
__global__ __launch_bounds__(kBinBlock) void hash_bwd_bin_kernel(
  const float * __restrict__ pts, const int32_t * __restrict__ primes,
  const float * __restrict__ bias, const float * __restrict__ mul,
  const float * __restrict__ grad_out, float * __restrict__ table_grad,
  uint32_t * __restrict__ ws_records, uint32_t * __restrict__ ws_counts, const BinArgs a)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  constexpr uint32_t kSlots = 4096 / F;  // combine table: slots (tags), F 64-bit sums each
  // channel-major with an odd pitch, for the same bank reason as SliceAcc (F >= 2: 4096 + F words)
  // slot of a row in the combine table: a multiplicative hash, not the row's low bits (EXPERIMENT)
  auto comb_slot = [](uint32_t r) { return (r ^ (r >> __builtin_ctz(kSlots))) & (kSlots - 1u); };
  auto comb_index = [](uint32_t slot, int k) { return (uint32_t)k * (kSlots + (F > 1 ? 1u : 0u)) + slot; };
  __shared__ __attribute__((aligned(16))) uint32_t queue[kBinQueueWords];
  __shared__ uint32_t qcount[kMaxBuckets];
  __shared__ unsigned long long sat_acc[8 * F];  // cell (0,0,0): exact sums per corner and channel
  // [0] combine the coming level, [1] non-zero contributions of this level, [2] lanes of the tile
  // whose level-0 cell equals the previous sample's, [3] lanes that have a previous sample,
  // [4] mean run length at level 0 (x256), [5] records that overflowed this level, [6] estimate of
  // the tile's non-zero level-0 contributions
  __shared__ uint32_t comb_state[7];
  unsigned long long * const comb_acc =
    reinterpret_cast<unsigned long long *>(queue + kBinQueueWords - kCombAccWords);
  uint32_t * const comb_tag = queue + kBinQueueWords - kCombAccWords - kCombTagWords;

  // One workgroup owns a tile of points for ALL levels: the point is read once, and the gradient
  // channels of level l+1 are requested before level l is processed -- with a 128 KiB LDS stage only
  // one workgroup fits a CU, so nothing else would hide that load.
  const int64_t tile = blockIdx.x;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  constexpr int kWaves = kBinBlock / 64;
  // (not const: a tile that combines re-deals its points to the threads below)
  bool valid = tile * kBinBlock + threadIdx.x < a.n;
  int64_t pc = valid ? tile * kBinBlock + threadIdx.x : a.n - 1;
  float x = pts[3 * pc + 0], y = pts[3 * pc + 1], z = pts[3 * pc + 2];
  const uint32_t bmask = (1u << a.bshift) - 1u;
  const int64_t n_tiles_g = a.n_tiles * a.groups;

  float g_cur[F], g_nxt[F];
#pragma unroll
  for (int k = 0; k < F; k++) g_cur[k] = grad_out[pc * a.g_ld_point + (int64_t)k * a.g_ld_chan];

  const bool comb_allowed = a.combine && a.groups == 1;
  if (threadIdx.x < 7) comb_state[threadIdx.x] = 0u;
  if (threadIdx.x < 32) queue[threadIdx.x] = 0u;  // prologue: bit set of the tile's level-0 cells
  __syncthreads();
  if (comb_allowed) {
    // how many consecutive samples share a level-0 cell?  (points are ray-major: the lanes of a wave
    // are consecutive samples of a ray.)  Dense sampling (the reference's 1024 steps of 1/256: 32
    // per cell) makes long runs of identical rows: that is where combining pays and where plain
    // binning overflows its queues.
    const LevelParams lp0 = load_level(primes, bias, mul, 0);
    const int cx = (int)floorf(fmaf(x, lp0.mul, lp0.bx)), cy = (int)floorf(fmaf(y, lp0.mul, lp0.by)),
              cz = (int)floorf(fmaf(z, lp0.mul, lp0.bz));
    // (one packed id per cell for the comparisons: 10 bits per axis tell neighbouring samples apart;
    // every instruction here is executed by all 16 waves of a tile that has the CU to itself)
    const int cid = (cx & 1023) | ((cy & 1023) << 10) | ((cz & 1023) << 20);
    const bool has_prev = valid && lane > 0;
    const unsigned long long m_prev = __ballot(has_prev);
    const unsigned long long m_same = __ballot(has_prev && cid == dpp_get_i<0x138, 0xf, 0xf>(cid));
    // ... or do the samples pile onto a few cells without forming runs?  (Rays that stop right in
    // front of one camera keep three or four samples each, all in the same two or three level-0
    // cells: A B B B A B B B ...  Plain binning sends such a tile's 2400 contributions to a dozen
    // rows, overflows its queues and ends in same-address global atomics, 80 us per tile and level.)
    // A lane repeats if its cell is that of one of the four lanes before it (16-lane DPP rows).
    const bool repeat = valid && (lane & 15) >= 4 &&  // (the first lanes of a DPP row see zeros)
                        (cid == dpp_get_i<0x111, 0xf, 0xf>(cid) || cid == dpp_get_i<0x112, 0xf, 0xf>(cid) ||
                         cid == dpp_get_i<0x113, 0xf, 0xf>(cid) || cid == dpp_get_i<0x114, 0xf, 0xf>(cid));
    // (only worth acting on when the gradient is dense as well: a point whose scaled gradient reaches
    // 2^-16 keeps nearly all of its 8 corner products above the f16 underflow, one far below keeps
    // none -- the bench's dense regime has this geometry too, but 90 % of its products underflow and
    // combining them is a loss.  An estimate: counting the products costs a level-0 hash per tile.)
    float big = 0.f;
#pragma unroll
    for (int k = 0; k < F; k++) big = fmaxf(big, fabsf(g_cur[k] * a.grad_scale));
    const unsigned long long m_repeat = __ballot(repeat && big >= 1.52587890625e-05f);
    // ... or is the whole tile a pile on a handful of cells, small gradients or not?  (Rays that
    // stop in front of one camera, gradients below the magnitude asked for above: level 0 was taken
    // plain, its queues overflowed into same-address global atomics, and only that overflow switched
    // the combine on, from level 1 -- half the backward's time in the bench's terminating regime.)
    // Contributions that survive the f16 underflow, estimated per lane from its gradient (a product
    // g w_d survives when w_d >= 2^-25 / g: all eight corners, about four, about one, none), against
    // the tile's distinct level-0 cells (a bit per cell of a 16 x 8 x 8 block; a long thin tile
    // aliases and looks smaller than it is: the price is one combined level, after which the
    // numbers measured on that level decide).
    const float big_s = big * 33554432.f;  // x 2^25
    const unsigned long long m8 = __ballot(valid && big_s >= 64.f), m4 = __ballot(valid && big_s >= 8.f),
                             m1 = __ballot(valid && big_s >= 2.f);
    const uint32_t nnz_est = 4u * (uint32_t)__popcll(m8) + 3u * (uint32_t)__popcll(m4) + (uint32_t)__popcll(m1);
    if (valid) {
      const uint32_t h = (uint32_t)(cx & 15) | ((uint32_t)(cy & 7) << 4) | ((uint32_t)(cz & 7) << 7);
      atomicOr(&queue[h >> 5], 1u << (h & 31u));
    }
    if (lane == 0) {
      atomicAdd(&comb_state[2], (uint32_t)__popcll(m_same));
      atomicAdd(&comb_state[3], (uint32_t)__popcll(m_prev));
      atomicAdd(&comb_state[5], (uint32_t)__popcll(m_repeat));
      atomicAdd(&comb_state[6], nnz_est);
    }
    __syncthreads();
    uint32_t n_cells = 0u;
    if (wave == 0) {
      n_cells = lane < 32 ? (uint32_t)__popc(queue[lane]) : 0u;
      n_cells = (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan_i32((int)n_cells), 63);
    }
    if (threadIdx.x == 0) {
      const uint32_t breaks = comb_state[3] - comb_state[2] + (uint32_t)kWaves;  // runs in the tile
      // mean run (x256) = 256 (lanes + waves) / breaks; the comparison needs no division, and the
      // quotient itself only matters to a tile that combines (a 64-bit division is ~150 dependent
      // instructions of this one thread while 1023 wait)
      const uint64_t lanes_q8 = (uint64_t)(comb_state[3] + kWaves) << 8;
      // (at most 768 of 1024 lanes can repeat; 640 = five in six of those with a look-back)
      const bool on = lanes_q8 >= (uint64_t)kCombineMinRunQ8 * breaks ||
                      comb_state[5] >= (uint32_t)kCombineRepeats ||
                      comb_state[6] >= (uint32_t)kCombinePiledPerCell * n_cells;
      comb_state[0] = on ? 1u : 0u;
      if (on) comb_state[4] = (uint32_t)(lanes_q8 / breaks);
      if (a.stats) {
        atomicAdd(a.stats + 3, comb_state[5]);
        atomicAdd(a.stats + 7, 1u);
      }
    }
    __syncthreads();
    if (comb_state[0] != 0u) {
      // A tile that combines is (part of) a densely sampled ray: runs of consecutive samples share
      // a cell, so the 64 lanes of one LDS atomic below would hit a handful of addresses and the LDS
      // serialises them (32 samples per level-0 cell at the reference's 1024 steps of 1/256: 19 us per
      // tile and level instead of 4).  Deal the samples out 16 apart instead -- lane l of wave w takes
      // sample 16 l + w -- so that one instruction's lanes spread over the whole tile.  The order of
      // the points means nothing to the sums; the price is uncoalesced (but cached, and prefetched a
      // level ahead) gradient loads.
      const int64_t p = tile * kBinBlock + (int64_t)(lane * kWaves + wave);
      valid = p < a.n;
      pc = valid ? p : a.n - 1;
      x = pts[3 * pc + 0];
      y = pts[3 * pc + 1];
      z = pts[3 * pc + 2];
#pragma unroll
      for (int k = 0; k < F; k++) g_cur[k] = grad_out[pc * a.g_ld_point + (int64_t)k * a.g_ld_chan];
    }
  }


  for (int l = 0; l < a.L; l++) {
    {
      const int ln = (l + 1 < a.L) ? l + 1 : l;  // clamped: the last prefetch re-reads level L-1
#pragma unroll
      for (int k = 0; k < F; k++)
        g_nxt[k] = grad_out[pc * a.g_ld_point + (int64_t)(ln * F + k) * a.g_ld_chan];
    }
    float gk[F];
    bool any = false;
#pragma unroll
    for (int k = 0; k < F; k++) {
      gk[k] = round_f16(g_cur[k] * a.grad_scale);
      any |= (gk[k] != 0.f);
    }
    // The prefetched gradient is taken over BEFORE this level's flush issues its stores.  vmcnt counts
    // loads and stores alike, in order: left at the top of the next level, the wait for these
    // loads was a wait for every store of the flush as well.
    auto take_prefetch = [&]() {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): only the prefetch (and older stores) are out
#pragma unroll
      for (int k = 0; k < F; k++) {
        g_cur[k] = g_nxt[k];
        asm volatile("" : "+v"(g_cur[k]));  // (the copy happens here, not after the stores)
      }
    };
    const bool active = valid && any;  // the reference skips all-zero channel groups (:133)
    uint32_t row[8];
    float w[8];
    float * gbase = table_grad + a.level_stride * l;

    // contribution of corner d as packed halves
    auto corner_value = [&](int d, uint32_t * val) {
      if constexpr (F == 1) {
        float c0 = gk[0] * w[d];
        asm volatile("" : "+v"(c0));  // keep the f32 rounding before the f16 one (see round_f16)
        val[0] = (uint32_t)__half_as_ushort(__float2half_rn(c0));
      } else {
        // two channels at a time: v_pk_mul_f32 + v_cvt_pk_f16_f32 (gfx950: two round-to-nearest-even
        // conversions and the pack in one instruction) instead of two multiplies, two conversions, a
        // shift and an or -- a third of this kernel's vector instructions were this expression
#pragma unroll
        for (int k = 0; k < F; k += 2) {
          f32x2 c = {gk[k], gk[k + 1]};
          c = c * w[d];
          asm volatile("" : "+v"(c));  // keep the f32 rounding before the f16 one (see round_f16)
          val[k / 2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(c, f16x2));
        }
      }
    };
    // stage one record in its bucket's LDS queue (layout capacity `cap`), or apply it directly
    auto enqueue = [&](uint32_t r, const uint32_t * val, int cap) {
      const uint32_t bucket = r >> a.bshift;
      const uint32_t slot = atomicAdd(&qcount[bucket], 1u);
      if (slot < (uint32_t)cap)
        // (bucket < 64, cap * KW <= 32768: a 24-bit multiply is full rate, v_mul_lo_u32 a quarter)
        store_record<F>(queue + __umul24(bucket, (uint32_t)(cap * KW)), cap, slot, r & bmask, val);
      else {
        // (not to the overflow arena: its code at the eight corners cost the two-level kernel 17
        // spilled registers and 9 % of its time, and the single-level kernel 2 %; instead the queues
        // are large enough that evenly spread points never get here, see kBinQueueWords)
        apply_record_atomic<F>(gbase, r, val, a.inv_scale);
        if (a.overflow) atomicAdd(a.overflow, 1ull);
      }
    };
    // wave w copies the queues of buckets w, w+16, ... to their workspace regions and records the
    // counts ([level][bucket][tile] so that pass B / C read them coalesced)
    auto flush = [&](int cap, int64_t tile_g, bool watch_overflow) {
      // (the thread id is laundered: everything derived from it below is loop-invariant, and the
      // compiler would otherwise compute it all before the level loop and keep -- spill -- it)
      uint32_t tid_f = (uint32_t)threadIdx.x;
      asm volatile("" : "+v"(tid_f));
      const int lane = (int)(tid_f & 63u), wave = (int)(tid_f >> 6);
      uint32_t * region0 =
        ws_records + ((size_t)l * n_tiles_g + tile_g) * a.n_buckets * (size_t)a.qcap * KW;
      if constexpr (!Rec<F>::kSoA) {
        if (a.n_buckets == 4 * kWaves) {
          // 64 buckets, four per wave (w, w+16, w+32, w+48), one per 16-lane group: lane (u, i)
          // copies records i, i+16, ... of queue u.  The first two records and the count are read
          // together, speculatively -- with sparse gradients (a dozen records per queue) that is the
          // whole flush: one LDS round trip, then the stores.  All sixteen waves of the tile do this
          // at the same moment with nothing else to run, so the length of the dependent chain is what
          // the flush costs, not its idle lanes: walking the four queues as one flat list (counts to
          // scalars, a prefix, a search per lane: round 2) kept every lane busy and was slower (bin
          // pass of the bench workload 2.02 -> 1.89 ms together with the register fixes below).
          const uint32_t u = (uint32_t)lane >> 4, i0 = (uint32_t)lane & 15u;
          const uint32_t b = (uint32_t)wave + (uint32_t)kWaves * u;
          const uint2 * src = reinterpret_cast<const uint2 *>(queue) + __umul24(b, (uint32_t)cap);
          uint2 * dst = reinterpret_cast<uint2 *>(region0) + __umul24(b, (uint32_t)a.qcap);
          const uint32_t asked = qcount[b];
          const uint2 r0 = src[i0], r1 = src[min(i0 + 16u, (uint32_t)cap - 1u)];  // (inside the queue)
          const uint32_t c = min(asked, (uint32_t)cap);
          // a queue overflowed (its records left as global atomics): this tile's contributions pile
          // onto few rows here (e.g. the samples of rays that stop right in front of one camera: a
          // dozen rows per tile and level, 80 us of same-address atomics) -- combine from the next
          // level on
          if (watch_overflow && __ballot(asked > (uint32_t)cap) != 0ull && lane == 0) comb_state[0] = 1u;
          if (i0 < c) dst[i0] = r0;
          if (i0 + 16u < c) dst[i0 + 16u] = r1;
          for (uint32_t i = i0 + 32u; __ballot(i < c) != 0ull; i += 32u) {
            const bool p0 = i < c, p1 = i + 16u < c;
            uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
            if (p0) v0 = src[i];
            if (p1) v1 = src[i + 16u];
            if (p0) dst[i] = v0;
            if (p1) dst[i + 16u] = v1;
          }
          if (i0 == 0u) ws_counts[((size_t)l * a.n_buckets + b) * n_tiles_g + tile_g] = c;
          return;
        }
        // the LDS reads of up to four buckets are issued before the first store
        for (int b0 = wave; b0 < a.n_buckets; b0 += 4 * kWaves) {
          uint2 v[4][4];
          uint32_t cnt[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int b = b0 + u * kWaves;
            cnt[u] = b < a.n_buckets ? min(qcount[b], (uint32_t)cap) : 0u;
            const uint2 * q = reinterpret_cast<const uint2 *>(queue + (size_t)b * cap * KW);
#pragma unroll
            for (int k = 0; k < 4; k++)
              if ((uint32_t)(lane + 64 * k) < cnt[u]) v[u][k] = q[lane + 64 * k];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int b = b0 + u * kWaves;
            uint2 * dst = reinterpret_cast<uint2 *>(region0 + (size_t)b * a.qcap * KW);
#pragma unroll
            for (int k = 0; k < 4; k++)
              if ((uint32_t)(lane + 64 * k) < cnt[u]) dst[lane + 64 * k] = v[u][k];
            for (uint32_t i = lane + 256; i < cnt[u]; i += 64)  // capacities above 256 (few buckets)
              dst[i] = reinterpret_cast<const uint2 *>(queue + (size_t)b * cap * KW)[i];
          }
        }
      } else {
        for (int b = wave; b < a.n_buckets; b += kWaves) {
          const uint32_t cnt = min(qcount[b], (uint32_t)cap);
          copy_records<F, true>(
            queue + (size_t)b * cap * KW, cap, region0 + (size_t)b * a.qcap * KW, a.qcap, 0u, cnt, lane);
        }
      }
      // every wave records the counts of the buckets it copied (w, w+16, ...: at most four)
      if (lane < kMaxBuckets / kWaves) {
        const int b = wave + kWaves * lane;
        if (b < a.n_buckets)
          ws_counts[((size_t)l * a.n_buckets + b) * n_tiles_g + tile_g] = min(qcount[b], (uint32_t)cap);
      }
    };

    __syncthreads();  // the previous level's flush has read the queues and counters
    const bool comb = comb_state[0] != 0u;
    if (threadIdx.x < kMaxBuckets) qcount[threadIdx.x] = 0u;
    if (SAT && threadIdx.x < 8 * F) sat_acc[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) comb_state[1] = 0u;
    __syncthreads();

    if (comb) {
      // ---- combine: equal rows of this tile are summed in LDS before they become records --------
      {  // zero the sums (32 KiB); tags need no reset: a slot is only read after a write of this level
        uint4 * zp = reinterpret_cast<uint4 *>(comb_acc);
        uint32_t zero = 0u;  // (laundered: four registers of zeros were being kept across the level loop)
        asm volatile("" : "+v"(zero));
        const uint4 z4 = make_uint4(zero, zero, zero, zero);
        zp[threadIdx.x] = z4;
        zp[threadIdx.x + kBinBlock] = z4;
        if (threadIdx.x < kCombAccWords / 4 - 2 * kBinBlock) zp[threadIdx.x + 2 * kBinBlock] = z4;
      }
      if (active) {
        const LevelParams lp = load_level(primes, bias, mul, l);
        corner_rows_and_weights<POW2>(x, y, z, lp, a.T, row, w);
#pragma unroll
        for (int d = 0; d < 8; d++) comb_tag[comb_slot(row[d])] = row[d];  // some writer wins
      }
      __syncthreads();
      uint32_t n_nz = 0u;
      if (active) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
          uint32_t val[VW];
          corner_value(d, val);
          if (val_all_zero<F>(val)) continue;
          n_nz++;
          const uint32_t slot = comb_slot(row[d]);
          bool finite = true;
#pragma unroll
          for (int k = 0; k < F; k++) finite &= !f16_bits_nonfinite(val_channel_bits<F>(val, k));
          if (comb_tag[slot] == row[d] && finite) {
#pragma unroll
            for (int k = 0; k < F; k++) {
              const uint32_t hb = val_channel_bits<F>(val, k);
              if (hb & 0x7fffu)
                atomicAdd(&comb_acc[comb_index(slot, k)], (unsigned long long)f16_bits_to_fixed(hb));
            }
          } else {
            enqueue(row[d], val, a.qcap_comb);  // lost the slot to another row: an ordinary record
            if (a.stats) atomicAdd(a.stats + 4 * (32 + l), 1u);
          }
        }
      }
      {
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan_i32((int)n_nz), 63);
        if (lane == 0 && tot) atomicAdd(&comb_state[1], tot);
      }
      __syncthreads();
      // emit the sums as f16 pieces (same record format, exact)
      for (uint32_t s = threadIdx.x; s < kSlots; s += kBinBlock) {
        long long S[F];
        bool nz = false;
#pragma unroll
        for (int k = 0; k < F; k++) {
          S[k] = (long long)comb_acc[comb_index(s, k)];
          nz |= (S[k] != 0);
        }
        if (!nz) continue;
        const uint32_t r = comb_tag[s];
        for (int piece = 0; piece < kMaxPieces && nz; piece++) {
          uint32_t val[VW];
#pragma unroll
          for (int j = 0; j < VW; j++) val[j] = 0u;
          nz = false;
#pragma unroll
          for (int k = 0; k < F; k++) {
            const uint32_t hb = take_f16_piece(S[k]);
            val[F >= 2 ? k / 2 : 0] |= (k & 1) ? (hb << 16) : hb;
            nz |= (S[k] != 0);
          }
          enqueue(r, val, a.qcap_comb);
        }
        if (nz) {  // sums beyond kMaxPieces * 11 bits (near the f16 overflow): the rest goes direct
          if (a.overflow) atomicAdd(a.overflow, 1ull);
#pragma unroll
          for (int k = 0; k < F; k++)
            if (S[k] != 0)
              atomicAdd(
                gbase + (int64_t)r * F + k,
                (float)((double)S[k] * ((double)a.inv_scale * (1.0 / 16777216.0))));
        }
      }
      __syncthreads();
      take_prefetch();
      flush(a.qcap_comb, tile, false);
      // Combine the next level too?  Yes while the sampling is still dense relative to its cells, or
      // while the records saved (measured here, scaled by the ~1.6x more distinct rows a finer level
      // has) outweigh the cost of the mode.  Finer levels only get worse: once off, it stays off.
      if (wave == 0) {
        uint32_t e = (lane < a.n_buckets) ? qcount[lane] : 0u;
        e = (uint32_t)__builtin_amdgcn_readlane(wave_incl_scan_i32((int)e), 63);
        if (lane == 0) {
          const uint32_t n_nz_tile = comb_state[1];
          if (a.stats) {
            atomicAdd(a.stats + 4 * l + 0, 1u);
            atomicAdd(a.stats + 4 * l + 1, n_nz_tile);
            atomicAdd(a.stats + 4 * l + 2, e);
          }
          const int ln = (l + 1 < a.L) ? l + 1 : l;
          const float run_next = (float)comb_state[4] * (mul[0] / mul[ln]);
          const bool dense = run_next >= (float)kCombineMinRunQ8;
          const bool pays = (float)n_nz_tile - 1.6f * (float)e >= (float)kCombinePaysAt;
          // ... or while the tile's contributions pile onto few rows (eight or more per row: the
          // samples of rays that stop right in front of one camera; plain binning would overflow
          // its queues into same-address global atomics there, 80 us per tile and level)
          const bool hot = e > 0u && n_nz_tile >= 8u * e;
          if (!(dense || pays || hot)) comb_state[0] = 0u;
        }
      }
    } else {
      // ---- plain binning: one record per (point, corner).  When the queues cannot hold a whole
      // tile-level (F = 8 with 64 buckets) the tile's points take turns in `groups` rounds.
      //
      // One cell is special: points whose three scaled coordinates are all negative saturate to
      // cell (0,0,0) (quirk Q1) -- at the finest levels that is ~9 % of uniformly spread points, all on
      // the SAME eight rows, which then carry 20x the mean slice load: their records overflow every
      // queue downstream and end as same-address global atomics (config C5: 1.3-3 % of a fine level's
      // records, but most of the split pass's time).  Their contributions are summed exactly in LDS
      // instead and leave as a handful of f16 pieces per tile and level.  (Only for tables that take the
      // split pass: the extra barrier and checks cost the single-level bench workload 0.3 ms per chunk,
      // and there a hot slice merely fills its regions.)
      for (int g = 0; g < a.groups; g++) {
        if (g > 0) {
          __syncthreads();  // the previous round's flush has read the queues
          if (threadIdx.x < kMaxBuckets) qcount[threadIdx.x] = 0u;
          __syncthreads();
        }
        // (the tile's POINTS take turns; letting its corners take turns instead -- every thread busy
        // in every round, the hash computed once per round -- was measured and lost: 59.8 vs 58.1 ms
        // per 2^24 points of config C5)
        if (active && (int)((threadIdx.x * (unsigned)a.groups) / kBinBlock) == g) {
          const LevelParams lp = load_level(primes, bias, mul, l);
          corner_rows_and_weights<POW2>(x, y, z, lp, a.T, row, w);
          const bool sat3 = SAT && fmaf(x, lp.mul, lp.bx) < 0.f && fmaf(y, lp.mul, lp.by) < 0.f &&
                            fmaf(z, lp.mul, lp.bz) < 0.f;
#pragma unroll
          for (int d = 0; d < 8; d++) {
            uint32_t val[VW];
            corner_value(d, val);
            if (val_all_zero<F>(val)) continue;
            if constexpr (SAT) {
              bool finite = true;
#pragma unroll
              for (int k = 0; k < F; k++) finite &= !f16_bits_nonfinite(val_channel_bits<F>(val, k));
              if (sat3 && finite) {
#pragma unroll
                for (int k = 0; k < F; k++) {
                  const uint32_t hb = val_channel_bits<F>(val, k);
                  if (hb & 0x7fffu)
                    atomicAdd(&sat_acc[d * F + k], (unsigned long long)f16_bits_to_fixed(hb));
                }
                continue;
              }
            }
            enqueue(row[d], val, a.qcap);
          }
        }
        __syncthreads();
        if (SAT && g == a.groups - 1) {
          // corner d of cell (0,0,0): thread d re-expresses its F sums as f16 pieces
          if (threadIdx.x < 8) {
            const int d = (int)threadIdx.x;
            long long S[F];
            bool nz = false;
#pragma unroll
            for (int k = 0; k < F; k++) {
              S[k] = (long long)sat_acc[d * F + k];
              nz |= (S[k] != 0);
            }
            if (nz) {
              const LevelParams lp = load_level(primes, bias, mul, l);
              const uint32_t r = wrap_row<POW2>(
                ((d & 4) ? lp.pa : 0u) ^ ((d & 2) ? lp.pb : 0u) ^ ((d & 1) ? lp.pc : 0u), a.T);
              for (int piece = 0; piece < kMaxPieces && nz; piece++) {
                uint32_t val[VW];
#pragma unroll
                for (int j = 0; j < VW; j++) val[j] = 0u;
                nz = false;
#pragma unroll
                for (int k = 0; k < F; k++) {
                  const uint32_t hb = take_f16_piece(S[k]);
                  val[F >= 2 ? k / 2 : 0] |= (k & 1) ? (hb << 16) : hb;
                  nz |= (S[k] != 0);
                }
                enqueue(r, val, a.qcap);
              }
              if (nz) {
                if (a.overflow) atomicAdd(a.overflow, 1ull);
#pragma unroll
                for (int k = 0; k < F; k++)
                  if (S[k] != 0)
                    atomicAdd(
                      gbase + (int64_t)r * F + k,
                      (float)((double)S[k] * ((double)a.inv_scale * (1.0 / 16777216.0))));
              }
            }
          }
          __syncthreads();
        }
        if (g == 0) take_prefetch();
        flush(a.qcap, tile * a.groups + g, comb_allowed);
      }
    }
  }
}

// ------------------------------------------------------------------------------ pass B: split --

struct SplitArgs
{
  const uint32_t * a_records;
  const uint32_t * a_counts;
  uint32_t * b_records;
  uint32_t * b_counts;
  float * table_grad;
  int64_t level_stride;
  int64_t n_tiles_g;  // regions per (level, bucket) of pass A
  float inv_scale;
  int n_buckets, bshift, log2_sub, qcap, n_slices, tiles_per_part, n_parts, cap2;
  uint32_t * stats;  // optional [L][4] counters: [3] += records that overflowed a queue or a run
  unsigned long long * overflow;  // optional: += records applied with float atomics, else NULL
  Arena arena;
};

template <int F>
__global__ __launch_bounds__(kSplitBlock) void hash_bwd_split_kernel(const SplitArgs a)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  constexpr uint32_t kRows = kBinAcc / F;
  constexpr int kWaves = kSplitBlock / 64;
  __shared__ __attribute__((aligned(16))) uint32_t queue[kSplitQueueWords];
  __shared__ uint32_t qcount[64], cursor[64];
  // records that found their sub-slice queue full: {sub << 16 | row in slice, values}; they are
  // appended to their runs one by one after the queues (a slice that carries a few times the mean
  // load -- coarse levels, where a slice owns a dozen distinct rows -- overflows its queue by a
  // handful of records per round)
  __shared__ uint32_t spill[kSplitSpill * KW];
  __shared__ uint32_t spill_count;
  const int bucket = blockIdx.x, l = blockIdx.y, part = blockIdx.z;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int n_sub = 1 << a.log2_sub;
  const int Q = (kSplitQueueWords / (n_sub * KW)) & ~3;  // records per sub-slice queue
  const int64_t t_begin = (int64_t)part * a.tiles_per_part;
  const int64_t t_end = min(t_begin + (int64_t)a.tiles_per_part, a.n_tiles_g);
  const uint32_t * counts = a.a_counts + ((size_t)l * a.n_buckets + bucket) * a.n_tiles_g;
  const size_t tile_stride = (size_t)a.n_buckets * a.qcap * KW;
  const uint32_t * base = a.a_records + ((size_t)l * a.n_tiles_g * a.n_buckets + bucket) * (size_t)a.qcap * KW;
  float * gbase = a.table_grad + a.level_stride * l;
  const uint32_t row0 = (uint32_t)bucket << a.bshift;
  const int n_half = (a.qcap + 63) / 64;

  if (threadIdx.x < 64) {
    qcount[threadIdx.x] = 0u;
    cursor[threadIdx.x] = 0u;
  }
  if (threadIdx.x == 0) spill_count = 0u;
  __syncthreads();

  for (int64_t t0 = t_begin; t0 < t_end; t0 += (int64_t)kWaves * kSplitRegions) {
    // ingest: this wave's regions of the round, all loads issued before the first LDS push
    uint32_t cnt[kSplitRegions];
#pragma unroll
    for (int u = 0; u < kSplitRegions; u++) {
      const int64_t t = t0 + (int64_t)wave * kSplitRegions + u;
      cnt[u] = (t < t_end) ? counts[t] : 0u;
    }
    for (int h = 0; h < n_half; h++) {
      uint32_t r[kSplitRegions], v[kSplitRegions][VW];
#pragma unroll
      for (int u = 0; u < kSplitRegions; u++) {
        const int64_t t = t0 + (int64_t)wave * kSplitRegions + u;
        const uint32_t last = cnt[u] ? cnt[u] - 1u : 0u;
        const uint32_t i = min((uint32_t)(lane + 64 * h), last);  // branch-free: tail lanes re-read
        const uint32_t * region = base + (size_t)((t < t_end) ? t : t_begin) * tile_stride;
        load_record<F>(region, a.qcap, i, r[u], v[u]);
      }
#pragma unroll
      for (int u = 0; u < kSplitRegions; u++) {
        if ((uint32_t)(lane + 64 * h) < cnt[u]) {
          const uint32_t sub = r[u] / kRows, local = r[u] & (kRows - 1u);
          const uint32_t slot = atomicAdd(&qcount[sub], 1u);
          if (slot < (uint32_t)Q) {
            store_record<F>(queue + (size_t)sub * Q * KW, Q, slot, local, v[u]);
          } else {
            const uint32_t sp = atomicAdd(&spill_count, 1u);
            if (sp < (uint32_t)kSplitSpill) {
              spill[sp * KW] = (sub << 16) | local;
#pragma unroll
              for (int j = 0; j < VW; j++) spill[sp * KW + 1 + j] = v[u][j];
            } else {
              if (a.stats) atomicAdd(a.stats + 4 * l + 3, 1u);
              if (!arena_push<F>(a.arena, l, (uint32_t)bucket, sub, local, v[u])) {
                apply_record_atomic<F>(gbase, row0 + r[u], v[u], a.inv_scale);
                if (a.overflow) atomicAdd(a.overflow, 1ull);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    // append every queue to its slice's run of this part
    for (int sub = wave; sub < n_sub; sub += kWaves) {
      const uint32_t c = min(qcount[sub], (uint32_t)Q), cur = cursor[sub];
      const uint32_t fit = min(c, (uint32_t)a.cap2 - cur);
      const int slice = (bucket << a.log2_sub) + sub;
      if (slice < a.n_slices) {
        uint32_t * run = a.b_records + (((size_t)l * a.n_slices + slice) * a.n_parts + part) * (size_t)a.cap2 * KW;
        copy_records<F>(queue + (size_t)sub * Q * KW, Q, run, a.cap2, cur, fit, lane);
        for (uint32_t i = fit + lane; i < c; i += 64) {  // run full: apply directly
          uint32_t rr, vv[VW];
          load_record<F>(queue + (size_t)sub * Q * KW, Q, i, rr, vv);
          if (a.stats) atomicAdd(a.stats + 4 * l + 2, 1u);
          if (!arena_push<F>(a.arena, l, (uint32_t)bucket, (uint32_t)sub, rr, vv)) {
            apply_record_atomic<F>(gbase, row0 + (uint32_t)sub * kRows + rr, vv, a.inv_scale);
            if (a.overflow) atomicAdd(a.overflow, 1ull);
          }
        }
      }
      if (lane == 0) {
        cursor[sub] = cur + fit;
        qcount[sub] = 0u;
      }
    }
    __syncthreads();
    const uint32_t n_spill = min(spill_count, (uint32_t)kSplitSpill);  // (block-uniform)
    if (n_spill) {
      for (uint32_t i = threadIdx.x; i < n_spill; i += kSplitBlock) {
        const uint32_t head = spill[i * KW], sub = head >> 16, local = head & 0xffffu;
        uint32_t vv[VW];
#pragma unroll
        for (int j = 0; j < VW; j++) vv[j] = spill[i * KW + 1 + j];
        const int slice = (bucket << a.log2_sub) + (int)sub;
        const uint32_t pos = atomicAdd(&cursor[sub], 1u);
        if (slice < a.n_slices && pos < (uint32_t)a.cap2) {
          uint32_t * run = a.b_records + (((size_t)l * a.n_slices + slice) * a.n_parts + part) * (size_t)a.cap2 * KW;
          store_record<F>(run, a.cap2, pos, local, vv);
        } else {
          if (a.stats) atomicAdd(a.stats + 4 * l + 2, 1u);
          if (!arena_push<F>(a.arena, l, (uint32_t)bucket, sub, local, vv)) {
            apply_record_atomic<F>(gbase, row0 + sub * kRows + local, vv, a.inv_scale);
            if (a.overflow) atomicAdd(a.overflow, 1ull);
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) spill_count = 0u;
      // cursors that ran past the run's capacity through the adds above
      if (threadIdx.x < 64) cursor[threadIdx.x] = min(cursor[threadIdx.x], (uint32_t)a.cap2);
      __syncthreads();
    }
  }
  if ((int)threadIdx.x < n_sub) {
    const int slice = (bucket << a.log2_sub) + threadIdx.x;
    if (slice < a.n_slices)
      a.b_counts[((size_t)l * a.n_slices + slice) * a.n_parts + part] = cursor[threadIdx.x];
  }
}

// ------------------------------------------------------------------------------ pass C: reduce -


// Non-finite contributions (an incoming gradient beyond the f16 range) cannot enter the fixed-point
// sums; they are remembered per accumulator as three bit sets {+inf, -inf, NaN} and folded in when
// the slice is flushed: the row ends up inf / NaN exactly where the reference's f16 sums would.
constexpr int kPoisonWords = kBinAcc / 32;

// The sums are kept CHANNEL-major with an odd row pitch: sum[k * (rows + 1) + row].  Row-major
// (row * F + k) would put the 64 lanes of one ds_add_u64 -- random rows, one channel -- on only
// 64 / (2 F) distinct bank pairs (4 at F = 8: a 16-way bank conflict on every accumulate).
constexpr int kSumWords = kBinAcc + 8;

struct SliceAcc
{
  unsigned long long sum[kSumWords];
  uint32_t poison[3][kPoisonWords];
};

template <int F>
__device__ __forceinline__ uint32_t sum_index(uint32_t local_row, int k)
{
  return (uint32_t)k * (uint32_t)(kBinAcc / F + 1) + local_row;
}

__device__ __forceinline__ void zero_slice(SliceAcc & sa)
{
  for (int i = threadIdx.x; i < kSumWords; i += kBinBlock) sa.sum[i] = 0ull;
  for (int i = threadIdx.x; i < 3 * kPoisonWords; i += kBinBlock) (&sa.poison[0][0])[i] = 0u;
}

template <int F>
__device__ __forceinline__ void accumulate_record(SliceAcc & sa, uint32_t local_row, const uint32_t * val)
{
#pragma unroll
  for (int k = 0; k < F; k++) {
    const uint32_t hb = val_channel_bits<F>(val, k);
    const uint32_t i = local_row * F + k;
    if (f16_bits_nonfinite(hb)) {
      const int which = (hb & 0x3ffu) ? 2 : ((hb & 0x8000u) ? 1 : 0);
      atomicOr(&sa.poison[which][i >> 5], 1u << (i & 31u));
    } else if (hb & 0x7fffu) {
      atomicAdd(&sa.sum[sum_index<F>(local_row, k)], (unsigned long long)f16_bits_to_fixed(hb));
    }
  }
}

// Add the slice into the table gradient: contiguous float atomics in general (the reference's level
// windows overlap, quirk Q2, so two slices may own the same element), plain read-modify-write when
// the caller's level stride keeps the windows apart (DISJOINT).
template <int F, bool DISJOINT>
__device__ __forceinline__ void flush_slice(
  const SliceAcc & sa, float * gbase_slice, uint32_t row_lo, uint32_t T, float inv_scale)
{
  constexpr uint32_t kRows = kBinAcc / F;
  const uint32_t rows_here = (row_lo + kRows <= T) ? kRows : (T > row_lo ? T - row_lo : 0u);
  const int n_flush = (int)rows_here * F;
  const double unit = (double)inv_scale * (1.0 / 16777216.0);
  constexpr int kBatch = 8;  // DISJOINT: the batch's loads are all in flight before the first store
  for (int i0 = threadIdx.x; i0 < n_flush; i0 += kBinBlock * kBatch) {
    float add[kBatch], old[kBatch];
    bool live[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; u++) {
      const int i = i0 + u * kBinBlock;
      live[u] = false;
      add[u] = 0.f;
      old[u] = 0.f;
      if (i < n_flush) {
        const long long v = (long long)sa.sum[sum_index<F>((uint32_t)i / F, i % F)];
        const uint32_t bit = 1u << (i & 31);
        const bool pinf = sa.poison[0][i >> 5] & bit, ninf = sa.poison[1][i >> 5] & bit,
                   nan = sa.poison[2][i >> 5] & bit;
        add[u] = (float)((double)v * unit);
        if (nan || (pinf && ninf)) add[u] = __builtin_nanf("");
        else if (pinf) add[u] = __builtin_inff();
        else if (ninf) add[u] = -__builtin_inff();
        live[u] = (v != 0 || pinf || ninf || nan);
        if (DISJOINT && live[u]) old[u] = gbase_slice[i];
      }
    }
#pragma unroll
    for (int u = 0; u < kBatch; u++) {
      const int i = i0 + u * kBinBlock;
      if (live[u]) {
        if (DISJOINT) gbase_slice[i] = old[u] + add[u];
        else atomicAdd(gbase_slice + i, add[u]);
      }
    }
  }
}

// the records of this slice in the overflow arena of its (level, bucket)
template <int F>
__device__ __forceinline__ void accumulate_arena(
  SliceAcc & acc, const Arena & ar, int l, uint32_t bucket, uint32_t sub)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  if (!ar.counts) return;
  const size_t slot = (size_t)l * ar.n_buckets + bucket;
  const uint32_t n = min(ar.counts[slot], (uint32_t)ar.cap);  // (block-uniform)
  const uint32_t * recs = ar.records + slot * (size_t)ar.cap * KW;
  for (uint32_t i = threadIdx.x; i < n; i += kBinBlock) {
    const uint32_t head = recs[(size_t)i * KW];
    if ((head >> 16) != sub) continue;
    uint32_t vv[VW];
#pragma unroll
    for (int j = 0; j < VW; j++) vv[j] = recs[(size_t)i * KW + 1 + j];
    accumulate_record<F>(acc, head & 0xffffu, vv);
  }
}

// single-level: the regions pass A wrote for (level, slice = bucket), one per tile
template <int F, bool DISJOINT>
__global__ __launch_bounds__(kBinBlock) void hash_bwd_reduce_kernel(
  const uint32_t * __restrict__ ws_records, const uint32_t * __restrict__ ws_counts,
  float * __restrict__ table_grad, uint32_t T, int64_t level_stride, float inv_scale, int n_slices,
  int qcap, int64_t n_tiles, int level_first, int level_step)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  constexpr uint32_t kRows = kBinAcc / F;
  constexpr int kWaves = kBinBlock / 64;
  __shared__ SliceAcc acc;
  __shared__ uint32_t batch_off[kWaves][64 + 1];
  const int sidx = blockIdx.x, l = level_first + (int)blockIdx.y * level_step;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const uint32_t row_lo = (uint32_t)sidx * kRows;
  float * gbase_slice = table_grad + level_stride * l + (int64_t)row_lo * F;
  const uint32_t * counts = ws_counts + ((size_t)l * n_slices + sidx) * n_tiles;
  const size_t tile_stride = (size_t)n_slices * qcap * KW;  // words between tiles, same slice
  const uint32_t * base = ws_records + ((size_t)l * n_tiles * n_slices + sidx) * (size_t)qcap * KW;
  // (fewer tiles per batch when there are few tiles -- a 512-ray training batch has 512 -- so that
  // all sixteen waves have one)
  int bt = 64;
  while (bt > 8 && (int64_t)bt * kWaves > n_tiles) bt >>= 1;
  // the counts of this wave's first batch are requested before the sums are zeroed: one memory
  // latency less behind the barrier of each of the ~1000 workgroups
  const int64_t t_first = (int64_t)wave * bt;
  const uint32_t first_cnt = (lane < bt && t_first + lane < n_tiles) ? counts[t_first + lane] : 0u;
  zero_slice(acc);
  if (threadIdx.x < kWaves) batch_off[threadIdx.x][64] = 0xffffffffu;  // sentinel for the search
  __syncthreads();

  // A wave owns 64 consecutive tiles at a time.  One coalesced load fetches their counts; their
  // records are then walked as ONE flat list (lane f handles the f-th record of the batch, found by
  // a binary search over the scanned counts kept in LDS), so every load instruction has 64 busy
  // lanes and a whole batch is in flight at once -- regions of a few records each (sparse
  // gradients) cost no more than their records.
  uint32_t * woff = &batch_off[wave][0];
  constexpr int kFlat = 16;  // 64-record loads in flight per wave
  for (int64_t t0 = t_first; t0 < n_tiles; t0 += (int64_t)kWaves * bt) {
    const uint32_t asked =
      (t0 == t_first) ? first_cnt : ((lane < bt && t0 + lane < n_tiles) ? counts[t0 + lane] : 0u);
    const uint32_t my_cnt = min(asked, (uint32_t)qcap);
    const uint32_t incl = (uint32_t)wave_incl_scan_i32((int)my_cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total > 64u * 40u) {
      // well-filled regions (dense gradients): one 64-lane load per region and 64 records, several
      // regions in flight; the search below would cost more than the idle tail lanes
      constexpr int kRegions = 8;
      const int n_here = (int)min((int64_t)bt, n_tiles - t0);
      const int n_half = (qcap + 63) / 64;
      for (int j0 = 0; j0 < n_here; j0 += kRegions) {
        uint32_t cnt[kRegions];
        uint32_t cnt_max = 0u;
#pragma unroll
        for (int u = 0; u < kRegions; u++) {
          cnt[u] = (uint32_t)__builtin_amdgcn_readlane((int)my_cnt, (j0 + u) & 63);
          if ((j0 + u) >= n_here) cnt[u] = 0u;
          cnt_max = max(cnt_max, cnt[u]);
        }
        for (int h = 0; h < n_half && (uint32_t)(64 * h) < cnt_max; h++) {  // wave-uniform bound
          uint32_t r[kRegions], v[kRegions][VW];
#pragma unroll
          for (int u = 0; u < kRegions; u++) {
            const bool live = (j0 + u) < n_here;
            const uint32_t * region = base + (size_t)(t0 + (live ? j0 + u : 0)) * tile_stride;
            const uint32_t last = cnt[u] ? cnt[u] - 1u : 0u;  // tail lanes re-read the last record
            load_record<F>(region, qcap, min((uint32_t)(lane + 64 * h), last), r[u], v[u]);
          }
#pragma unroll
          for (int u = 0; u < kRegions; u++)
            if ((uint32_t)(lane + 64 * h) < cnt[u]) accumulate_record<F>(acc, r[u], v[u]);
        }
      }
      continue;
    }
    woff[lane] = incl - my_cnt;  // first flat index of tile `lane` (same-wave LDS accesses stay ordered)
    for (uint32_t f0 = 0; f0 < total; f0 += 64 * kFlat) {
      uint32_t r[kFlat], v[kFlat][VW];
#pragma unroll
      for (int u = 0; u < kFlat; u++) {
        const uint32_t f = min(f0 + 64u * u + lane, total - 1u);  // branch-free: tail lanes re-read
        uint32_t t = 0u;  // largest t with woff[t] <= f (empty tiles share an offset with their successor)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
          if (woff[t + step] <= f) t += step;
        const uint32_t * region = base + (size_t)(t0 + t) * tile_stride;
        load_record<F>(region, qcap, f - woff[t], r[u], v[u]);
      }
#pragma unroll
      for (int u = 0; u < kFlat; u++)
        if (f0 + 64u * u + lane < total) accumulate_record<F>(acc, r[u], v[u]);
    }
  }
  __syncthreads();
  flush_slice<F, DISJOINT>(acc, gbase_slice, row_lo, T, inv_scale);
}

// two-level: the runs pass B wrote for (level, slice), one per part
template <int F, bool DISJOINT>
__global__ __launch_bounds__(kBinBlock) void hash_bwd_reduce_runs_kernel(
  const uint32_t * __restrict__ b_records, const uint32_t * __restrict__ b_counts,
  float * __restrict__ table_grad, uint32_t T, int64_t level_stride, float inv_scale, int n_slices,
  int n_parts, int cap2, int log2_sub, const Arena arena, int level_first, int level_step)
{
  constexpr int KW = Rec<F>::kWords, VW = Rec<F>::kValWords;
  constexpr uint32_t kRows = kBinAcc / F;
  constexpr int kWaves = kBinBlock / 64;
  constexpr int kUnroll = 8;
  __shared__ SliceAcc acc;
  const int sidx = blockIdx.x, l = level_first + (int)blockIdx.y * level_step;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const uint32_t row_lo = (uint32_t)sidx * kRows;
  float * gbase_slice = table_grad + level_stride * l + (int64_t)row_lo * F;
  auto run_of = [&](int part, uint32_t & cnt) {
    const size_t ridx = ((size_t)l * n_slices + sidx) * n_parts + part;
    cnt = min(b_counts[ridx], (uint32_t)cap2);
    return b_records + ridx * (size_t)cap2 * KW;
  };
  auto load_chunk = [&](const uint32_t * run, uint32_t cnt, uint32_t i0, uint32_t (&r)[kUnroll],
                        uint32_t (&v)[kUnroll][VW]) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      load_record<F>(run, cap2, min(i0 + 64 * u + lane, cnt - 1u), r[u], v[u]);  // tail lanes re-read
  };
  auto add_chunk = [&](uint32_t cnt, uint32_t i0, const uint32_t (&r)[kUnroll],
                       const uint32_t (&v)[kUnroll][VW]) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
      if (i0 + 64 * u + lane < cnt) accumulate_record<F>(acc, r[u], v[u]);
  };
  // One workgroup per CU (137 KiB of sums) and ~130 of them in a row per CU: the count and the
  // first records of this wave's first run are requested BEFORE the sums are zeroed, so that the two
  // dependent memory latencies do not sit behind the barrier of every slice.
  uint32_t r0[kUnroll], v0[kUnroll][VW], cnt0 = 0u;
  const uint32_t * run0 = nullptr;
  if (wave < n_parts) {
    run0 = run_of(wave, cnt0);
    if (cnt0) load_chunk(run0, cnt0, 0u, r0, v0);
  }
  // (reading the slice's current gradient values here as well, so that the flush ends in plain
  // stores, was measured and bought nothing: 61.4 vs 61.2 ms per 2^24 points of config C5; sixteen
  // 64-record loads in flight per wave instead of eight: 67 ms)
  zero_slice(acc);
  __syncthreads();

  if (cnt0) {
    add_chunk(cnt0, 0u, r0, v0);
    for (uint32_t i0 = 64 * kUnroll; i0 < cnt0; i0 += 64 * kUnroll) {
      uint32_t r[kUnroll], v[kUnroll][VW];
      load_chunk(run0, cnt0, i0, r, v);
      add_chunk(cnt0, i0, r, v);
    }
  }
  for (int part = wave + kWaves; part < n_parts; part += kWaves) {
    uint32_t cnt;
    const uint32_t * run = run_of(part, cnt);
    for (uint32_t i0 = 0; i0 < cnt; i0 += 64 * kUnroll) {
      uint32_t r[kUnroll], v[kUnroll][VW];
      load_chunk(run, cnt, i0, r, v);
      add_chunk(cnt, i0, r, v);
    }
  }
  accumulate_arena<F>(
    acc, arena, l, (uint32_t)sidx >> log2_sub, (uint32_t)sidx & ((1u << log2_sub) - 1u));
  __syncthreads();
  flush_slice<F, DISJOINT>(acc, gbase_slice, row_lo, T, inv_scale);
}

// ------------------------------------------------------------------------------ planning -------

struct BinPlan
{
  bool ok = false;
  int n_slices = 0, n_buckets = 0, log2_sub = 0, bshift = 0, groups = 1, qcap = 0, qcap_comb = 0;
  int64_t chunk_tiles = 0;  // tiles processed per round of the three passes
  // per-chunk layout (for chunk_tiles tiles)
  int tiles_per_part = 0, n_parts = 0, cap2 = 0;
  int64_t a_counts_bytes = 0, a_records_bytes = 0, b_counts_bytes = 0, b_records_bytes = 0;
  int64_t arena_counts_bytes = 0, arena_bytes = 0;
  int arena_cap = 0;
  int64_t bytes = 0;
};

int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

int ilog2(uint32_t v)
{
  int r = 0;
  while ((1u << r) < v) r++;
  return r;
}

// layout of the workspace for `tiles` tiles under the capacities already chosen in pl
void layout_for(BinPlan & pl, int L, int F, int64_t tiles)
{
  const int kw = 1 + (F >= 2 ? F / 2 : 1);
  const int64_t tiles_g = tiles * pl.groups;
  pl.a_counts_bytes = align256((int64_t)L * pl.n_buckets * tiles_g * 4);
  pl.a_records_bytes = align256((int64_t)L * tiles_g * pl.n_buckets * pl.qcap * kw * 4);
  pl.b_counts_bytes = pl.b_records_bytes = 0;
  pl.tiles_per_part = pl.n_parts = pl.cap2 = 0;
  if (pl.log2_sub > 0) {
    // parts: about 32 per (level, bucket), whole multiples of the 8 x kSplitRegions regions a
    // round of pass B ingests
    const int round = (kSplitBlock / 64) * kSplitRegions;
    int64_t tpp = (tiles_g + 31) / 32;
    tpp = std::max<int64_t>((tpp + round - 1) / round * round, 2 * round);
    pl.tiles_per_part = (int)tpp;
    pl.n_parts = (int)((tiles_g + tpp - 1) / tpp);
    const double mean_region = 8.0 * kBinBlock / pl.groups / pl.n_buckets;
    const double mean_run = mean_region * (double)tpp / (double)(1 << pl.log2_sub);
    pl.cap2 = ((int)(mean_run * 1.5) + 128 + 3) & ~3;
    pl.b_counts_bytes = align256((int64_t)L * pl.n_slices * pl.n_parts * 4);
    pl.b_records_bytes = align256((int64_t)L * pl.n_slices * pl.n_parts * pl.cap2 * kw * 4);
  }
  // the arena (two-level tables only) grows with the batch: 64 records per tile round and
  // (level, bucket), 1024 .. kArenaRecords -- a percent of the record regions
  pl.arena_cap = pl.log2_sub > 0
                   ? (int)std::min<int64_t>(kArenaRecords, std::max<int64_t>(1024, 64 * tiles_g))
                   : 0;
  pl.arena_counts_bytes = pl.arena_cap ? align256((int64_t)L * pl.n_buckets * 4) : 0;
  pl.arena_bytes = align256((int64_t)L * pl.n_buckets * pl.arena_cap * kw * 4);
  pl.bytes = pl.a_counts_bytes + pl.a_records_bytes + pl.b_counts_bytes + pl.b_records_bytes +
             pl.arena_counts_bytes + pl.arena_bytes;
}

// Capacities of the binned backward for (n, L, F, T); !ok means "not applicable".  workspace_bytes
// > 0: shrink the number of tiles handled per round until the layout fits.
BinPlan bin_plan(int64_t n, int L, int F, uint32_t T, int64_t workspace_bytes)
{
  BinPlan pl;
  if (!(F == 1 || F == 2 || F == 4 || F == 8) || L < 1 || n < 65536) return pl;
  const int kw = 1 + (F >= 2 ? F / 2 : 1);
  const int64_t rows_per_slice = kBinAcc / F;
  const int64_t n_slices = ((int64_t)T + rows_per_slice - 1) / rows_per_slice;
  int s = 0;
  while (((n_slices + (1 << s) - 1) >> s) > kMaxBuckets) s++;
  if (s > kMaxLog2Sub) return pl;
  pl.n_slices = (int)n_slices;
  pl.log2_sub = s;
  pl.n_buckets = (int)((n_slices + (1 << s) - 1) >> s);
  pl.bshift = ilog2((uint32_t)rows_per_slice) + s;
  // corner groups: the queues of one round (8/groups corners of 1024 points) must fit the LDS stage
  // with 25 % slack over the mean region size
  for (int groups = 1; groups <= 8; groups *= 2) {
    const int avg = (8 * kBinBlock / groups + pl.n_buckets - 1) / pl.n_buckets;
    const int cap = std::min(2 * avg, kBinQueueWords / (pl.n_buckets * kw)) & ~3;
    if (cap >= avg + avg / 4) {
      pl.groups = groups;
      pl.qcap = cap;
      break;
    }
  }
  if (pl.qcap == 0) return pl;
  pl.qcap_comb = std::min(pl.qcap, kCombQueueWords / (pl.n_buckets * kw)) & ~3;
  const int64_t total_tiles = (n + kBinBlock - 1) / kBinBlock;
  int64_t tiles = total_tiles;
  layout_for(pl, L, F, tiles);
  if (workspace_bytes > 0 && pl.bytes > workspace_bytes) {
    // bytes grow (almost) linearly with the tile count: estimate, then step down until it fits
    tiles = std::max<int64_t>(1, (int64_t)((double)tiles * (double)workspace_bytes / (double)pl.bytes));
    for (;;) {
      layout_for(pl, L, F, tiles);
      if (pl.bytes <= workspace_bytes || tiles == 1) break;
      tiles = std::max<int64_t>(1, tiles - std::max<int64_t>(1, tiles / 16));
    }
    if (pl.bytes > workspace_bytes) return pl;  // not even one tile fits
  }
  pl.chunk_tiles = tiles;
  pl.ok = true;
  return pl;
}

constexpr int64_t kRecommendedWorkspaceCap = (int64_t)64 << 30;

std::atomic<uint32_t *> g_bin_stats{nullptr};
std::atomic<unsigned long long *> g_overflow_counter{nullptr};

inline bool is_pow2(uint32_t v) { return v && !(v & (v - 1u)); }

}  // namespace

// Measurement hook of tools/ab_hash_bwd.py (not part of the ABI header): a device buffer of
// [F2N_MAX_LEVELS][4] u32 that pass A fills with {tiles that combined the level, their non-zero
// contributions, the records they emitted, 0}; NULL switches it off.
extern "C" void f2n_debug_bin_stats(uint32_t * device_counters)
{
  g_bin_stats.store(device_counters, std::memory_order_relaxed);
}

extern "C" int f2n_hash_bwd_set_overflow_counter(uint64_t * device_counter)
{
  if (reinterpret_cast<uintptr_t>(device_counter) & 7u) return F2N_E_INVALID_ARG;
  g_overflow_counter.store(
    reinterpret_cast<unsigned long long *>(device_counter), std::memory_order_relaxed);
  return F2N_OK;
}

extern "C" int64_t f2n_hash_bwd_workspace_bytes(int64_t n, int L, int F, uint32_t T)
{
  BinPlan pl = bin_plan(n, L, F, T, 0);
  if (!pl.ok) return 0;
  if (pl.bytes <= kRecommendedWorkspaceCap) return pl.bytes;
  pl = bin_plan(n, L, F, T, kRecommendedWorkspaceCap);  // bigger batches run in several rounds
  return pl.ok ? pl.bytes : 0;
}

#define F2N_DISPATCH_F(F_, ...)      \
  switch (F_) {                      \
    case 1: { constexpr int FF = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int FF = 2; __VA_ARGS__; } break; \
    case 4: { constexpr int FF = 4; __VA_ARGS__; } break; \
    case 8: { constexpr int FF = 8; __VA_ARGS__; } break; \
    default: return F2N_E_UNSUPPORTED; \
  }

extern "C" int f2n_hash_bwd_binned(
  const float * pts, const int32_t * primes, const float * bias, const float * mul,
  const float * grad_out, int64_t g_ld_point, int64_t g_ld_chan, float * table_grad, int64_t n,
  int L, int F, uint32_t T, int64_t level_stride, float grad_scale, void * workspace,
  int64_t workspace_bytes, void * stream)
{
  if (!pts || !primes || !bias || !mul || !grad_out || !table_grad || !workspace)
    return F2N_E_INVALID_ARG;
  if (F != 1 && F != 2 && F != 4 && F != 8) return F2N_E_UNSUPPORTED;
  if (!f2n_hash_args_ok(n, L, F, T, level_stride)) return F2N_E_INVALID_ARG;
  int e = 0;
  const float m = frexpf(grad_scale, &e);
  if (!(grad_scale > 0.f) || m != 0.5f) return F2N_E_INVALID_ARG;
  if (reinterpret_cast<uintptr_t>(workspace) & 255u) return F2N_E_INVALID_ARG;
  if (workspace_bytes <= 0) return F2N_E_INVALID_ARG;
  BinPlan pl = bin_plan(n, L, F, T, workspace_bytes);
  if (!pl.ok) return F2N_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool p2 = is_pow2(T);
  const float inv = 1.f / grad_scale;
  const int combine = f2n_get_option(F2N_OPT_BWD_COMBINE) == 0 ? 1 : 0;
  // Level windows [stride*l, stride*l + T*F): where they do not overlap a slice owns its elements
  // alone and adds with a plain read-modify-write.  Where they do (the reference's layout: stride = T,
  // windows of 2 T elements at F = 2, quirk Q2) the slices of two levels share elements and add with
  // float atomics: into a zeroed gradient that is still order-independent (0 + a + b), into an
  // existing one (a view rendered in chunks, accumulated in place) the last bit of a shared element
  // depends on which level came first.  F2N_OPT_BWD_PHASES = 1 removes that: levels l, l + k, l + 2k,
  // ... do not overlap for k = ceil(T F / stride), so the reduce pass runs in k launches of plain
  // read-modify-writes -- bit-reproducible, 4 % slower on 8.4 M samples and 25 % on 1 M (half the
  // workgroups per launch).  More than 8 phases or a stride of 0: atomics either way.
  const bool disjoint = L == 1 || level_stride >= (int64_t)T * F;
  int phases = disjoint ? 1 : 0;  // 0: float atomics
  if (!disjoint && f2n_get_option(F2N_OPT_BWD_PHASES) == 1) {
    const int64_t k = level_stride > 0 ? ((int64_t)T * F + level_stride - 1) / level_stride : 0;
    if (k >= 2 && k <= 8) phases = (int)k;
  }
  const int64_t total_tiles = (n + kBinBlock - 1) / kBinBlock;

  for (int64_t tile0 = 0; tile0 < total_tiles; tile0 += pl.chunk_tiles) {
    const int64_t tiles = std::min(pl.chunk_tiles, total_tiles - tile0);
    const int64_t p0 = tile0 * kBinBlock;
    const int64_t n_c = std::min<int64_t>(n - p0, tiles * kBinBlock);
    if (tiles != pl.chunk_tiles || tile0 == 0) layout_for(pl, L, F, tiles);
    if (tiles * pl.groups > 0x7fffffff) return F2N_E_INVALID_ARG;
    char * w = (char *)workspace;
    uint32_t * a_counts = reinterpret_cast<uint32_t *>(w);
    uint32_t * a_records = reinterpret_cast<uint32_t *>(w + pl.a_counts_bytes);
    uint32_t * b_counts = reinterpret_cast<uint32_t *>(w + pl.a_counts_bytes + pl.a_records_bytes);
    uint32_t * b_records =
      reinterpret_cast<uint32_t *>(w + pl.a_counts_bytes + pl.a_records_bytes + pl.b_counts_bytes);
    Arena arena;
    arena.counts = reinterpret_cast<uint32_t *>(
      w + pl.a_counts_bytes + pl.a_records_bytes + pl.b_counts_bytes + pl.b_records_bytes);
    arena.records = reinterpret_cast<uint32_t *>(
      w + pl.a_counts_bytes + pl.a_records_bytes + pl.b_counts_bytes + pl.b_records_bytes +
      pl.arena_counts_bytes);
    arena.n_buckets = pl.n_buckets;
    arena.cap = pl.arena_cap;
    if (pl.arena_cap == 0) {
      arena.counts = nullptr;  // single-level table: no arena (see the top of the file)
    } else if (hipMemsetAsync(arena.counts, 0, (size_t)L * pl.n_buckets * 4, s) != hipSuccess) {
      return F2N_E_LAUNCH;
    }

    BinArgs ba;
    ba.g_ld_point = g_ld_point;
    ba.g_ld_chan = g_ld_chan;
    ba.n = n_c;
    ba.n_tiles = tiles;
    ba.level_stride = level_stride;
    ba.T = T;
    ba.L = L;
    ba.grad_scale = grad_scale;
    ba.inv_scale = inv;
    ba.n_buckets = pl.n_buckets;
    ba.bshift = pl.bshift;
    ba.groups = pl.groups;
    ba.qcap = pl.qcap;
    ba.qcap_comb = pl.qcap_comb;
    ba.combine = combine;
    ba.stats = g_bin_stats.load(std::memory_order_relaxed);
    ba.overflow = g_overflow_counter.load(std::memory_order_relaxed);
    const int64_t tiles_g = tiles * pl.groups;
#define F2N_BIN_LAUNCH(P2, SAT)                                                                     \
  hipLaunchKernelGGL(                                                                              \
    (hash_bwd_bin_kernel<FF, P2, SAT>), grid_a, block_a, 0, s, pts + 3 * p0, primes, bias, mul,    \
    grad_out + p0 * g_ld_point, table_grad, a_records, a_counts, ba)
    const dim3 grid_a((unsigned)tiles), block_a(kBinBlock);
    F2N_DISPATCH_F(F, {
      if (pl.log2_sub > 0) {
        if (p2) F2N_BIN_LAUNCH(true, true);
        else F2N_BIN_LAUNCH(false, true);
      } else {
        if (p2) F2N_BIN_LAUNCH(true, false);
        else F2N_BIN_LAUNCH(false, false);
      }
      if (pl.log2_sub == 0) {
        if (phases > 0) {
          for (int ph = 0; ph < phases && ph < L; ph++) {
            const dim3 grid_c((unsigned)pl.n_slices, (unsigned)((L - ph + phases - 1) / phases));
            hipLaunchKernelGGL(
              (hash_bwd_reduce_kernel<FF, true>), grid_c, dim3(kBinBlock), 0, s, a_records, a_counts,
              table_grad, T, level_stride, inv, pl.n_slices, pl.qcap, tiles_g, ph, phases);
          }
        } else {
          const dim3 grid_c((unsigned)pl.n_slices, (unsigned)L);
          hipLaunchKernelGGL(
            (hash_bwd_reduce_kernel<FF, false>), grid_c, dim3(kBinBlock), 0, s, a_records, a_counts,
            table_grad, T, level_stride, inv, pl.n_slices, pl.qcap, tiles_g, 0, 1);
        }
      } else {
        SplitArgs sa;
        sa.a_records = a_records;
        sa.a_counts = a_counts;
        sa.b_records = b_records;
        sa.b_counts = b_counts;
        sa.table_grad = table_grad;
        sa.level_stride = level_stride;
        sa.n_tiles_g = tiles_g;
        sa.inv_scale = inv;
        sa.n_buckets = pl.n_buckets;
        sa.bshift = pl.bshift;
        sa.log2_sub = pl.log2_sub;
        sa.qcap = pl.qcap;
        sa.n_slices = pl.n_slices;
        sa.tiles_per_part = pl.tiles_per_part;
        sa.n_parts = pl.n_parts;
        sa.cap2 = pl.cap2;
        sa.stats = g_bin_stats.load(std::memory_order_relaxed);
        sa.overflow = ba.overflow;
        sa.arena = arena;
        const dim3 grid_b((unsigned)pl.n_buckets, (unsigned)L, (unsigned)pl.n_parts);
        hipLaunchKernelGGL((hash_bwd_split_kernel<FF>), grid_b, dim3(kSplitBlock), 0, s, sa);
        if (phases > 0) {
          for (int ph = 0; ph < phases && ph < L; ph++) {
            const dim3 grid_c((unsigned)pl.n_slices, (unsigned)((L - ph + phases - 1) / phases));
            hipLaunchKernelGGL(
              (hash_bwd_reduce_runs_kernel<FF, true>), grid_c, dim3(kBinBlock), 0, s, b_records,
              b_counts, table_grad, T, level_stride, inv, pl.n_slices, pl.n_parts, pl.cap2,
              pl.log2_sub, arena, ph, phases);
          }
        } else {
          const dim3 grid_c((unsigned)pl.n_slices, (unsigned)L);
          hipLaunchKernelGGL(
            (hash_bwd_reduce_runs_kernel<FF, false>), grid_c, dim3(kBinBlock), 0, s, b_records,
            b_counts, table_grad, T, level_stride, inv, pl.n_slices, pl.n_parts, pl.cap2,
            pl.log2_sub, arena, 0, 1);
        }
      }
    })
#undef F2N_BIN_LAUNCH
    if (hipGetLastError() != hipSuccess) return F2N_E_LAUNCH;
  }
  return F2N_OK;
}
