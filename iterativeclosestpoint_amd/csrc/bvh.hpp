// bvh.hpp — on-device construction of the search structure over the target
// triangles: 63-bit Morton codes of the centroids, a radix sort (prims.hpp),
// triangles gathered into that order, and a complete G-ary hierarchy of
// fp32 boxes over runs of G consecutive triangles (a linear BVH: every node
// covers a contiguous range of the Morton order).  The hierarchy is implicit:
// the children of node i of level L are nodes i*G .. i*G+G-1 of level L-1, so
// no child pointers are stored, and a node holds its children's boxes as six
// G-wide float rows (lo.x, lo.y, lo.z, hi.x, hi.y, hi.z) that G lanes read
// with one coalesced load per row.
//
// Replaces the vtkCellLocator built at ICP/vtkImplicitPolyDataDistance2.cxx:53-60.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace icpx
{

constexpr int MAXLEV = 16;

// Level 0 = leaves (runs of G sorted triangles, boxes held by level 1),
// levels 1..nlev = nodes; level nlev has exactly one node, the root.
struct TreeLevels
{
  int G;
  int nlev;
  int32_t cnt[MAXLEV + 1]; // cnt[0] leaves, cnt[L] nodes of level L
  int64_t off[MAXLEV + 1]; // float offset of level L's first node in the node array
  int64_t total_floats;
};

inline TreeLevels make_levels(int64_t nt, int G)
{
  TreeLevels t;
  t.G = G;
  int64_t c = (nt + G - 1) / G;
  if (c < 1)
    c = 1;
  t.cnt[0] = (int32_t)c;
  t.off[0] = 0;
  int64_t o = 0;
  int L = 0;
  do
  {
    L++;
    c = (c + G - 1) / G;
    t.cnt[L] = (int32_t)c;
    t.off[L] = o;
    o += c * 6 * G;
  } while (c > 1 && L < MAXLEV);
  t.nlev = L;
  t.total_floats = o;
  for (int i = L + 1; i <= MAXLEV; i++)
  {
    t.cnt[i] = 0;
    t.off[i] = 0;
  }
  return t;
}

__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__device__ __forceinline__ uint64_t morton63(float x, float y, float z, const float* lo,
  const float* inv)
{
  float fx = (x - lo[0]) * inv[0], fy = (y - lo[1]) * inv[1], fz = (z - lo[2]) * inv[2];
  fx = fminf(fmaxf(fx, 0.f), 1.f);
  fy = fminf(fmaxf(fy, 0.f), 1.f);
  fz = fminf(fmaxf(fz, 0.f), 1.f);
  uint32_t qx = min((uint32_t)(fx * 2097152.f), 2097151u);
  uint32_t qy = min((uint32_t)(fy * 2097152.f), 2097151u);
  uint32_t qz = min((uint32_t)(fz * 2097152.f), 2097151u);
  return (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
}

// Position along the Hilbert curve through the same 2^21 grid (Skilling's transform of the
// three coordinates, then the same interleave).  Consecutive positions are neighbouring grid
// cells without exception, where the Morton curve jumps across the box whenever a high bit
// flips: a run of consecutive points is one compact patch, never two far apart.
__device__ __forceinline__ uint64_t hilbert63(float x, float y, float z, const float* lo,
  const float* inv)
{
  float fx = (x - lo[0]) * inv[0], fy = (y - lo[1]) * inv[1], fz = (z - lo[2]) * inv[2];
  fx = fminf(fmaxf(fx, 0.f), 1.f);
  fy = fminf(fmaxf(fy, 0.f), 1.f);
  fz = fminf(fmaxf(fz, 0.f), 1.f);
  uint32_t X0 = min((uint32_t)(fx * 2097152.f), 2097151u);
  uint32_t X1 = min((uint32_t)(fy * 2097152.f), 2097151u);
  uint32_t X2 = min((uint32_t)(fz * 2097152.f), 2097151u);
  for (uint32_t Q = 1u << 20; Q > 1u; Q >>= 1)
  {
    const uint32_t P = Q - 1u;
    if (X0 & Q)
      X0 ^= P;
    if (X1 & Q)
      X0 ^= P;
    else
    {
      const uint32_t t = (X0 ^ X1) & P;
      X0 ^= t;
      X1 ^= t;
    }
    if (X2 & Q)
      X0 ^= P;
    else
    {
      const uint32_t t = (X0 ^ X2) & P;
      X0 ^= t;
      X2 ^= t;
    }
  }
  X1 ^= X0;
  X2 ^= X1;
  uint32_t t = 0u;
  for (uint32_t Q = 1u << 20; Q > 1u; Q >>= 1)
    if (X2 & Q)
      t ^= Q - 1u;
  X0 ^= t;
  X1 ^= t;
  X2 ^= t;
  return (spread21(X0) << 2) | (spread21(X1) << 1) | spread21(X2);
}

struct Box3
{
  float lo[3], inv[3];
};

// K1: Morton code of each triangle's centroid; value = triangle id.
__global__ void k_morton_tris(const float* __restrict__ xyz, const int32_t* __restrict__ tri,
  int64_t nt, Box3 b, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nt)
    return;
  const float* p0 = xyz + 3 * (int64_t)tri[3 * c];
  const float* p1 = xyz + 3 * (int64_t)tri[3 * c + 1];
  const float* p2 = xyz + 3 * (int64_t)tri[3 * c + 2];
  const float third = 1.f / 3.f;
  keys[c] = morton63((p0[0] + p1[0] + p2[0]) * third, (p0[1] + p1[1] + p2[1]) * third,
    (p0[2] + p1[2] + p2[2]) * third, b.lo, b.inv);
  vals[c] = (uint32_t)c;
}

// Place on the space-filling curve of the kept source points of a shard (point (k0+q)*step),
// value = q.  hilbert: the Hilbert curve (the queries' order: a wavefront's 64 consecutive
// queries are then always one patch); else the Morton curve.
__global__ void k_morton_points(const float* __restrict__ xyz, int64_t k0, int64_t step,
  int64_t nq, Box3 b, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int hilbert = 1)
{
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq)
    return;
  const float* p = xyz + 3 * ((k0 + q) * step);
  keys[q] = hilbert ? hilbert63(p[0], p[1], p[2], b.lo, b.inv) : morton63(p[0], p[1], p[2], b.lo, b.inv);
  vals[q] = (uint32_t)q;
}

// Triangles copied out in Morton order: 9 floats (P0,P1,P2 in cell order) and the cell id.
__global__ void k_gather_tris(const float* __restrict__ xyz, const int32_t* __restrict__ tri,
  const uint32_t* __restrict__ perm, int64_t nt, float* __restrict__ tverts,
  int32_t* __restrict__ tid)
{
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt)
    return;
  int64_t c = perm[s];
  tid[s] = (int32_t)c;
#pragma unroll
  for (int k = 0; k < 3; k++)
  {
    const float* p = xyz + 3 * (int64_t)tri[3 * c + k];
    tverts[9 * s + 3 * k + 0] = p[0];
    tverts[9 * s + 3 * k + 1] = p[1];
    tverts[9 * s + 3 * k + 2] = p[2];
  }
}

// Level 1: box of each leaf (run of G sorted triangles), widened by one float
// ulp outwards so that a closest point computed in fp64 on a triangle can
// never fall outside the box of its leaf.  Slots past the last leaf hold an
// empty box (+inf, -inf).
__global__ void k_build_level1(
  const float* __restrict__ tverts, int64_t nt, TreeLevels lv, float* __restrict__ nodes)
{
  const int G = lv.G;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)lv.cnt[1] * G)
    return;
  int64_t node = t / G;
  int j = (int)(t % G);
  int64_t leaf = node * G + j;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  if (leaf < lv.cnt[0])
  {
    int64_t s0 = leaf * G, s1 = min(nt, s0 + G);
    for (int64_t s = s0; s < s1; s++)
      for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++)
        {
          float v = tverts[9 * s + 3 * k + a];
          lo[a] = fminf(lo[a], v);
          hi[a] = fmaxf(hi[a], v);
        }
    if (s1 > s0)
      for (int a = 0; a < 3; a++)
      {
        lo[a] = nextafterf(lo[a], -INFINITY);
        hi[a] = nextafterf(hi[a], INFINITY);
      }
  }
  float* n = nodes + lv.off[1] + node * 6 * G;
  for (int a = 0; a < 3; a++)
  {
    n[a * G + j] = lo[a];
    n[(3 + a) * G + j] = hi[a];
  }
}

// Level L >= 2: the box of child c is the union of the boxes stored in node c of level L-1.
__global__ void k_build_level(int L, TreeLevels lv, float* __restrict__ nodes)
{
  const int G = lv.G;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)lv.cnt[L] * G)
    return;
  int64_t node = t / G;
  int j = (int)(t % G);
  int64_t child = node * G + j;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  if (child < lv.cnt[L - 1])
  {
    const float* c = nodes + lv.off[L - 1] + child * 6 * G;
    for (int a = 0; a < 3; a++)
      for (int k = 0; k < G; k++)
      {
        lo[a] = fminf(lo[a], c[a * G + k]);
        hi[a] = fmaxf(hi[a], c[(3 + a) * G + k]);
      }
  }
  float* n = nodes + lv.off[L] + node * 6 * G;
  for (int a = 0; a < 3; a++)
  {
    n[a * G + j] = lo[a];
    n[(3 + a) * G + j] = hi[a];
  }
}

} // namespace icpx
