// prims.hpp — device-wide exclusive scan and LSD radix sort (64-bit keys,
// 32-bit values) for gfx950.  Both are written for 64-lane wavefronts: the
// scan reduces with wave shuffles, the sort ranks a wave's 64 keys with eight
// ballots per digit.  They serve the one-off set-up work (Morton ordering of
// triangles and queries, CSR links, stream compaction), not the per-iteration
// hot loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpx
{

// ------------------------------------------------------------------ scan ----
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d)
      v += o;
  }
  return v;
}

// Exclusive scan of one tile per block; block totals go to bsum (may be null
// when the grid is a single block).
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile(
  const uint32_t* in, uint32_t* out, uint32_t* bsum, int64_t n)
{
  __shared__ uint32_t wsum[SCAN_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)tid * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t tsum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
  {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    tsum += v[i];
  }
  uint32_t incl = wave_incl_scan(tsum, lane);
  if (lane == 63)
    wsum[wid] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++)
  {
    uint32_t s = wsum[w];
    if (w < wid)
      woff += s;
    total += s;
  }
  uint32_t run = woff + incl - tsum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
  {
    if (base + i < n)
      out[base + i] = run;
    run += v[i];
  }
  if (bsum && tid == 0)
    bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(
  uint32_t* __restrict__ data, const uint32_t* __restrict__ boff, int64_t n)
{
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  const uint32_t o = boff[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n)
      data[base + i] += o;
}

// Scratch needed (in uint32 words) by exclusive_scan for n items.
inline int64_t scan_scratch_words(int64_t n)
{
  int64_t w = 0;
  while (n > SCAN_TILE)
  {
    n = (n + SCAN_TILE - 1) / SCAN_TILE;
    w += n;
  }
  return w + 1;
}

// out[i] = sum of in[0..i); in == out is allowed.
inline void exclusive_scan(
  const uint32_t* in, uint32_t* out, int64_t n, uint32_t* scratch, hipStream_t st)
{
  if (n <= 0)
    return;
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1)
  {
    k_scan_tile<<<1, SCAN_THREADS, 0, st>>>(in, out, nullptr, n);
    return;
  }
  k_scan_tile<<<(unsigned)nb, SCAN_THREADS, 0, st>>>(in, out, scratch, n);
  exclusive_scan(scratch, scratch, nb, scratch + nb, st);
  k_scan_add<<<(unsigned)nb, SCAN_THREADS, 0, st>>>(out, scratch, n);
}

// ------------------------------------------------------------ radix sort ----
constexpr int RS_BITS = 8;
constexpr int RS_BINS = 1 << RS_BITS;
constexpr int RS_TILE = 4096;      // keys per block in both passes
constexpr int RS_HTHREADS = 256;   // histogram kernel
constexpr int RS_BATCH = 16;       // keys per lane held in registers by the scatter wave

// Per-tile digit histogram, written bin-major: hist[bin * ntiles + tile].
__global__ __launch_bounds__(RS_HTHREADS) void k_rs_hist(
  const uint64_t* __restrict__ keys, uint32_t* __restrict__ hist, int64_t n, int shift, int ntiles)
{
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0; // RS_HTHREADS == RS_BINS
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  for (int i = threadIdx.x; i < RS_TILE; i += RS_HTHREADS)
  {
    int64_t p = base + i;
    if (p < n)
      atomicAdd(&h[(keys[p] >> shift) & (RS_BINS - 1)], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// One wavefront per tile.  The tile is walked in order, 64 keys at a time;
// eight ballots find the lanes that share a digit, the popcount of the lower
// lanes in that set is the key's stable rank, and a per-digit running offset
// in LDS (seeded from the scanned histogram) gives its destination.
__global__ __launch_bounds__(64) void k_rs_scatter(const uint64_t* __restrict__ kin,
  const uint32_t* __restrict__ vin, uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
  const uint32_t* __restrict__ offs, int64_t n, int shift, int ntiles)
{
  __shared__ uint32_t run[RS_BINS];
  const int lane = threadIdx.x;
#pragma unroll
  for (int b = lane; b < RS_BINS; b += 64)
    run[b] = offs[(int64_t)b * ntiles + blockIdx.x];
  __builtin_amdgcn_wave_barrier();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int b0 = 0; b0 < RS_TILE / 64; b0 += RS_BATCH)
  {
    uint64_t k[RS_BATCH];
    uint32_t v[RS_BATCH];
#pragma unroll
    for (int j = 0; j < RS_BATCH; j++)
    {
      int64_t p = base + (int64_t)(b0 + j) * 64 + lane;
      bool ok = p < n;
      k[j] = ok ? kin[p] : 0ull;
      v[j] = ok ? vin[p] : 0u;
    }
#pragma unroll
    for (int j = 0; j < RS_BATCH; j++)
    {
      int64_t p = base + (int64_t)(b0 + j) * 64 + lane;
      bool ok = p < n;
      uint32_t d = (uint32_t)(k[j] >> shift) & (RS_BINS - 1);
      uint64_t same = __ballot(ok);
#pragma unroll
      for (int bit = 0; bit < RS_BITS; bit++)
      {
        uint64_t m = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? m : ~m;
      }
      uint32_t rank = __popcll(same & lt);
      uint32_t cnt = __popcll(same);
      uint32_t off = 0;
      if (ok)
        off = run[d];
      __builtin_amdgcn_wave_barrier();
      if (ok && rank == 0)
        run[d] = off + cnt;
      __builtin_amdgcn_wave_barrier();
      if (ok)
      {
        kout[off + rank] = k[j];
        vout[off + rank] = v[j];
      }
    }
  }
}

// Scratch (uint32 words) for radix_sort_pairs of n keys.
inline int64_t rs_scratch_words(int64_t n)
{
  int64_t ntiles = (n + RS_TILE - 1) / RS_TILE;
  int64_t hw = ntiles * RS_BINS;
  return hw + scan_scratch_words(hw);
}

// Stable LSD sort of (key, value) pairs on bits [0, nbits).  Buffers ping-pong
// between (k0,v0) and (k1,v1); the return value says which pair holds the
// result (0 or 1).
inline int radix_sort_pairs(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, int64_t n,
  int nbits, uint32_t* scratch, hipStream_t st)
{
  if (n <= 1)
    return 0;
  const int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
  const int64_t hw = (int64_t)ntiles * RS_BINS;
  uint32_t* hist = scratch;
  uint32_t* sscr = scratch + hw;
  int cur = 0;
  for (int shift = 0; shift < nbits; shift += RS_BITS)
  {
    uint64_t* ki = cur ? k1 : k0;
    uint32_t* vi = cur ? v1 : v0;
    uint64_t* ko = cur ? k0 : k1;
    uint32_t* vo = cur ? v0 : v1;
    k_rs_hist<<<ntiles, RS_HTHREADS, 0, st>>>(ki, hist, n, shift, ntiles);
    exclusive_scan(hist, hist, hw, sscr, st);
    k_rs_scatter<<<ntiles, 64, 0, st>>>(ki, vi, ko, vo, hist, n, shift, ntiles);
    cur ^= 1;
  }
  return cur;
}

} // namespace icpx
