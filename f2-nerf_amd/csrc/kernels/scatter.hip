// scatter.hip -- per-image appearance-embedding broadcast/add.
// Replaces ScatterIdxKernal, ScatterAddFuncForward and ScatterAddFuncBackwardBlock + torch::sum
// (reference src/CustomOps/Scatter.cu:11-132).
#include "common.hiph"

namespace
{

// One wavefront per ray: fill the ray's sample range with its image id (coalesced 256-B stores).
__global__ __launch_bounds__(F2N_BLOCK) void scatter_idx_kernel(
  const int32_t * __restrict__ idx, const int32_t * __restrict__ emb_idx,
  int32_t * __restrict__ all_emb_idx, int n_rays)
{
  const int r = (int)blockIdx.x * F2N_WAVES_PER_BLOCK + (int)(threadIdx.x >> 6);
  if (r >= n_rays) return;
  const int lane = lane_id();
  const int s = idx[2 * r], e = idx[2 * r + 1];
  const int fill = emb_idx[r];
  for (int i = s + lane; i < e; i += F2N_WAVE) all_emb_idx[i] = fill;
}

// One thread per output element; consecutive threads = consecutive channels of consecutive samples.
__global__ __launch_bounds__(F2N_BLOCK) void scatter_add_fwd_kernel(
  const float * __restrict__ emb, const int32_t * __restrict__ scatter_idx,
  const float * __restrict__ to_add, float * __restrict__ sum, int64_t n_elems, int C)
{
  const int64_t i = (int64_t)blockIdx.x * F2N_BLOCK + threadIdx.x;
  if (i >= n_elems) return;
  const int64_t p = i / C;
  const int c = (int)(i - p * C);
  sum[i] = to_add[i] + emb[(int64_t)scatter_idx[p] * C + c];
}

// Backward: demb[e, c] = sum over samples p with scatter_idx[p] == e of dsum[p, c].
// The reference compares every (sample block, image) pair: O(n * n_emb) index reads.  Samples of
// one ray are contiguous and share an image id, so here a thread (q, c) walks a span of samples,
// keeps a running sum while the id stays the same and flushes one atomic per run.
constexpr int kSpan = 64;

__global__ __launch_bounds__(F2N_BLOCK) void scatter_add_bwd_kernel(
  const int32_t * __restrict__ scatter_idx, const float * __restrict__ dsum,
  float * __restrict__ demb, int64_t n_all, int n_emb, int C, int cpad)
{
  const int c = (int)threadIdx.x % cpad;
  const int q = (int)threadIdx.x / cpad;
  const int spans_per_block = F2N_BLOCK / cpad;
  const int64_t span = (int64_t)blockIdx.x * spans_per_block + q;
  const int64_t lo = span * kSpan;
  if (c >= C || lo >= n_all) return;
  const int64_t hi = (lo + kSpan < n_all) ? lo + kSpan : n_all;
  int cur = scatter_idx[lo];
  float acc = 0.f;
  for (int64_t p = lo; p < hi; p++) {
    const int e = scatter_idx[p];
    if (e != cur) {
      if (cur >= 0 && cur < n_emb) atomicAdd(demb + (int64_t)cur * C + c, acc);
      cur = e;
      acc = 0.f;
    }
    acc += dsum[p * C + c];
  }
  if (cur >= 0 && cur < n_emb) atomicAdd(demb + (int64_t)cur * C + c, acc);
}

}  // namespace

extern "C" int f2n_scatter_idx(
  const int32_t * idx, const int32_t * emb_idx, int32_t * all_emb_idx, int n_rays, void * stream)
{
  if (!idx || !emb_idx || n_rays < 0) return F2N_E_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(
    scatter_idx_kernel, dim3(f2n_div_up(n_rays, F2N_WAVES_PER_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, idx, emb_idx, all_emb_idx, n_rays);
  return f2n_launch_status();
}

extern "C" int f2n_scatter_add_fwd(
  const float * emb, const int32_t * scatter_idx, const float * to_add, float * sum, int64_t n_all,
  int C, void * stream)
{
  if (!emb || n_all < 0 || C < 1) return F2N_E_INVALID_ARG;
  if (n_all == 0) return F2N_OK;
  if (!scatter_idx || !to_add || !sum) return F2N_E_INVALID_ARG;
  const int64_t n_elems = n_all * C;
  hipLaunchKernelGGL(
    scatter_add_fwd_kernel, dim3(f2n_div_up(n_elems, F2N_BLOCK)), dim3(F2N_BLOCK), 0,
    (hipStream_t)stream, emb, scatter_idx, to_add, sum, n_elems, C);
  return f2n_launch_status();
}

extern "C" int f2n_scatter_add_bwd(
  const int32_t * scatter_idx, const float * dsum, float * demb, int64_t n_all, int n_emb, int C,
  void * stream)
{
  if (!demb || n_all < 0 || n_emb < 0 || C < 1) return F2N_E_INVALID_ARG;
  if (C > F2N_BLOCK) return F2N_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n_emb > 0 &&
      hipMemsetAsync(demb, 0, sizeof(float) * (size_t)n_emb * C, s) != hipSuccess)
    return F2N_E_LAUNCH;
  if (n_all == 0 || n_emb == 0) return F2N_OK;
  if (!scatter_idx || !dsum) return F2N_E_INVALID_ARG;
  int cpad = 1;
  while (cpad < C) cpad <<= 1;
  const int spans_per_block = F2N_BLOCK / cpad;
  const int64_t n_spans = (n_all + kSpan - 1) / kSpan;
  hipLaunchKernelGGL(
    scatter_add_bwd_kernel, dim3(f2n_div_up(n_spans, spans_per_block)), dim3(F2N_BLOCK), 0, s,
    scatter_idx, dsum, demb, n_all, n_emb, C, cpad);
  return f2n_launch_status();
}
