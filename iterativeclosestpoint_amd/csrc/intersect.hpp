// intersect.hpp — first intersection of line segments with the target mesh, on the binary
// box tree the closest-point search uses.
//
// Replaces vtkImplicitPolyDataIntersection::IntersectWithLine
// (ICP/vtkImplicitPolyDataIntersection.cxx:110-337): vtkCellLocator::IntersectWithLine with
// tolerance 1e-12 (:148), then vtkTriangle::EvaluatePosition at the hit and the face / edge /
// angle-weighted vertex pseudo-normal there (:169-304), the code it shares with
// vtkImplicitPolyDataDistance2::ClosestPointAndNormal.
//
// One lane per segment.  The walk keeps a stack of far children in LDS and prunes with a
// slab test in fp64 against boxes grown by a margin that covers everything the exact test
// may accept (a hit point lies within rounding of its triangle, or within the tolerance of
// it).  Every triangle of a visited leaf gets the exact test, restated from VTK 9.3 in its
// order of operations: vtkTriangle::ComputeNormal, vtkPlane::IntersectWithLine (parallel
// when |n.(p2-p1)| <= 1e-6 |n.(pt1-p1)|), vtkTriangle::EvaluatePosition at the plane point;
// a hit when inside or within the tolerance.  The smallest parameter wins, equal parameters
// go to the smallest cell id (the reference keeps the first the locator meets).
#pragma once
#include "search3.hpp"
#include "tables.hpp"

namespace icpx
{

struct IntersectArgs
{
  const double* p1; // [n][3]
  const double* p2;
  int64_t n;
  const float* nodes;
  BinTree tree;
  const float* tverts;
  const int32_t* tid;
  int64_t nt;
  const int32_t* tri;
  const float* xyz;
  const double* fn64;
  const uint32_t* voff;
  const int32_t* vcell;
  float scene_max;
  double novalue; // NoValue: the length of the target's bounds (:64)
  int32_t* hit;
  double* ret;
  double* closest;
  double* normal;
  double* x;
  int32_t* cell;
  int32_t* kind;
};

constexpr int ISECT_THREADS = 128;
constexpr int ISECT_STACK = 40;

// vtkTriangle::IntersectWithLine for a non-degenerate triangle (P0,P1,P2 in cell order).
__device__ __forceinline__ bool tri_intersect_line(const d3& p1, const d3& p2, const d3& P0,
  const d3& P1, const d3& P2, double tol, double& t, d3& x)
{
  const d3 &pt1 = P1, &pt2 = P2, &pt3 = P0;
  d3 n;
  {
    // vtkTriangle::ComputeNormal(pt1, pt2, pt3, n): direction, then unit length
    const double ax = pt3.x - pt2.x, ay = pt3.y - pt2.y, az = pt3.z - pt2.z;
    const double bx = pt1.x - pt2.x, by = pt1.y - pt2.y, bz = pt1.z - pt2.z;
    n = mk(ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx);
    const double len = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
    if (len != 0.0)
      n = mk(n.x / len, n.y / len, n.z / len);
  }
  if (n.x == 0.0 && n.y == 0.0 && n.z == 0.0)
    return false;
  // vtkPlane::IntersectWithLine(p1, p2, n, pt1, t, x)
  const d3 p21 = mk(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
  const double num = dot3(n, pt1) - (n.x * p1.x + n.y * p1.y + n.z * p1.z);
  const double den = n.x * p21.x + n.y * p21.y + n.z * p21.z;
  const double fabsden = den < 0.0 ? -den : den;
  const double fabstol = num < 0.0 ? -num * 1.0e-06 : num * 1.0e-06;
  if (fabsden <= fabstol)
    return false;
  t = num / den;
  x = mk(p1.x + t * p21.x, p1.y + t * p21.y, p1.z + t * p21.z);
  if (!(t >= 0.0 && t <= 1.0))
    return false;
  d3 cp;
  double d2, w[3];
  if (!tri_eval(x, P0, P1, P2, cp, d2, w))
    return false;
  if (d2 <= tol * tol)
    return true;
  return w[0] >= 0.0 && w[0] <= 1.0 && w[1] >= 0.0 && w[1] <= 1.0 && w[2] >= 0.0 && w[2] <= 1.0;
}

// Does the segment p1 + t d, t in [0, tmax], pass through the box grown by m?  tnear: where
// it enters (never above the true entry parameter of the ungrown box).
__device__ __forceinline__ bool slab(const d3& p1, const d3& d, float lox, float loy, float loz,
  float hix, float hiy, float hiz, double m, double tmax, double& tnear)
{
  double t0 = 0.0, t1 = tmax;
  const double lo[3] = { (double)lox - m, (double)loy - m, (double)loz - m };
  const double hi[3] = { (double)hix + m, (double)hiy + m, (double)hiz + m };
  const double o[3] = { p1.x, p1.y, p1.z }, dd[3] = { d.x, d.y, d.z };
  // (the padding nodes of a tree that is not a power of two hold the empty box, lo = +inf,
  // hi = -inf: the quotients below would swap into (-inf, +inf) and let a segment in)
  if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2])
    return false;
#pragma unroll
  for (int a = 0; a < 3; a++)
  {
    if (dd[a] == 0.0)
    {
      if (o[a] < lo[a] || o[a] > hi[a])
        return false;
    }
    else
    {
      double ta = (lo[a] - o[a]) / dd[a], tb = (hi[a] - o[a]) / dd[a];
      if (ta > tb)
      {
        const double s = ta;
        ta = tb;
        tb = s;
      }
      // (a quotient is off by half a unit in the last place; the grown box covers that
      // wherever the quotient is of moderate size, and a huge one is far from [0, 1])
      t0 = fmax(t0, ta - 1e-12 * fabs(ta));
      t1 = fmin(t1, tb + 1e-12 * fabs(tb));
    }
  }
  tnear = t0;
  return t0 <= t1;
}

template <int L>
__global__ __launch_bounds__(ISECT_THREADS) void k_intersect_lines(IntersectArgs a)
{
  __shared__ uint32_t stk[ISECT_STACK][ISECT_THREADS];
  const int tidx = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * ISECT_THREADS + tidx;
  if (i >= a.n)
    return;
  const d3 p1 = mk(a.p1[3 * i], a.p1[3 * i + 1], a.p1[3 * i + 2]);
  const d3 p2 = mk(a.p2[3 * i], a.p2[3 * i + 1], a.p2[3 * i + 2]);
  const d3 dir = mk(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
  const int d = a.tree.depth;
  // what the exact test may accept beyond a triangle's own box: rounding of the plane
  // point (a few units in the last place of the coordinates) and the tolerance
  const double smax = fmax((double)a.scene_max,
    fmax(fmax(fabs(p1.x), fabs(p1.y)), fmax(fabs(p1.z), fmax(fabs(p2.x), fmax(fabs(p2.y), fabs(p2.z))))));
  const double margin = 1e-9 * smax + 1e-11;
  const bool finite = smax < 1e300;

  double tbest = 2.0; // (a hit has t in [0, 1])
  int32_t idbest = 0x7fffffff, sbest = -1;
  d3 xbest = mk(0.0, 0.0, 0.0);
  int sp = 0;
  uint32_t n = 1;
  bool go = finite;
  while (go)
  {
    const int lvl = 31 - __clz(n);
    if (lvl < d)
    {
      const float4* p = reinterpret_cast<const float4*>(a.nodes + (int64_t)n * NODE_FLOATS);
      const float4 u = p[0], v = p[1], w = p[2];
      const double tmax = fmin(1.0, tbest);
      double tn0, tn1;
      const bool ok0 = slab(p1, dir, u.x, u.y, u.z, u.w, v.x, v.y, margin, tmax, tn0);
      const bool ok1 = slab(p1, dir, v.z, v.w, w.x, w.y, w.z, w.w, margin, tmax, tn1);
      if (ok0 && ok1)
      {
        const uint32_t near = tn1 < tn0 ? 1u : 0u;
        if (sp < ISECT_STACK)
          stk[sp][tidx] = 2u * n + (near ^ 1u);
        sp++;
        n = 2u * n + near;
        continue;
      }
      if (ok0 || ok1)
      {
        n = 2u * n + (ok1 ? 1u : 0u);
        continue;
      }
    }
    else
    {
      const int64_t s0 = (int64_t)(n - (1u << d)) * L;
#pragma unroll 1
      for (int t_ = 0; t_ < L; t_++)
      {
        const int64_t s = s0 + t_;
        if (s >= a.nt || a.tid[s] < 0)
          continue;
        d3 P0, P1, P2, x;
        double t;
        load_tri(a.tverts, s, P0, P1, P2);
        if (!tri_intersect_line(p1, p2, P0, P1, P2, 1e-12, t, x))
          continue;
        const int32_t id = a.tid[s];
        if (t < tbest || (t == tbest && id < idbest))
        {
          tbest = t;
          idbest = id;
          sbest = (int32_t)s;
          xbest = x;
        }
      }
    }
    if (sp == 0)
      go = false;
    else
    {
      sp--;
      n = stk[sp][tidx]; // (depth <= 30 < ISECT_STACK: the stack cannot overflow)
    }
  }

  // outputs as the reference leaves them (:114-124 without a hit)
  int32_t hit = 0, cell = -1, kind = -1;
  double ret = a.novalue;
  d3 cp = mk(0.0, 0.0, 0.0), g = mk(0.0, 0.0, 1.0), xo = mk(0.0, 0.0, 0.0);
  if (sbest >= 0)
  {
    hit = 1;
    cell = idbest;
    xo = xbest;
    const double dx = p1.x - p2.x, dy = p1.y - p2.y, dz = p1.z - p2.z;
    ret = sqrt(dx * dx + dy * dy + dz * dz) * tbest; // :155
    d3 P0, P1, P2;
    double d2, w[3];
    load_tri(a.tverts, sbest, P0, P1, P2);
    tri_eval(xbest, P0, P1, P2, cp, d2, w); // :169
    const bool z0 = fabs(w[0]) < 1e-12, z1 = fabs(w[1]) < 1e-12, z2 = fabs(w[2]) < 1e-12;
    kind = (z0 ? 1 : 0) + (z1 ? 1 : 0) + (z2 ? 1 : 0);
    d3 nrm = mk(0.0, 0.0, 0.0);
    if (kind == 0)
      nrm = mk(a.fn64[3 * (int64_t)cell], a.fn64[3 * (int64_t)cell + 1], a.fn64[3 * (int64_t)cell + 2]);
    else if (kind == 1)
      nrm = edge_normal64(a.tri, a.voff, a.vcell, a.fn64, cell, z0 ? 0 : (z1 ? 1 : 2));
    else if (kind == 2)
      nrm = vertex_normal64(a.xyz, a.tri, a.voff, a.vcell, a.fn64,
        a.tri[3 * (int64_t)cell + (!z2 ? 2 : (!z1 ? 1 : 0))]);
    g = nrm; // :317-320
    // :324 as written: g has just been set to the pseudo-normal, so the product is its
    // squared length and the value is negated
    ret *= (dot3(g, nrm) < 0.) ? 1. : -1.;
  }
  a.hit[i] = hit;
  a.ret[i] = ret;
  a.cell[i] = cell;
  a.kind[i] = kind;
  a.closest[3 * i] = cp.x;
  a.closest[3 * i + 1] = cp.y;
  a.closest[3 * i + 2] = cp.z;
  a.normal[3 * i] = g.x;
  a.normal[3 * i + 1] = g.y;
  a.normal[3 * i + 2] = g.z;
  a.x[3 * i] = xo.x;
  a.x[3 * i + 1] = xo.y;
  a.x[3 * i + 2] = xo.z;
}

} // namespace icpx
