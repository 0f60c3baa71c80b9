// tri.hpp — fp64 point/triangle arithmetic on the device.
//
// The closest point on a triangle is computed the way VTK 9.3's
// vtkTriangle::EvaluatePosition does it (the routine behind
// vtkCellLocator::FindClosestPoint and the re-evaluation at
// ICP/vtkImplicitPolyDataDistance2.cxx:388): project onto the plane, solve the
// 2x2 barycentric system in the two axes orthogonal to the largest normal
// component, and outside the triangle fall back to the vertex / edge chosen by
// the signs of the weights, with vtkLine::DistanceToLine's parametrisation.
// The operation order is kept and the file is compiled with -ffp-contract=off,
// so every value is the IEEE result a scalar x86-64 build would produce.
//
// Where a reader with the VTK 9.3 source tree finds what each routine restates (VTK itself is
// not part of this repository or of the reference, so files and functions are named, not
// line numbers):
//   dist_to_line      Common/DataModel/vtkLine.cxx, vtkLine::DistanceToLine(x, p1, p2, t,
//                     closestPoint): num = p21.(x - p1), denom = p21.p21, tolerance
//                     |VTK_TOL * num| with VTK_TOL = 1e-5, p1 when denom < tolerance, then
//                     t = num / denom clamped to the end points
//   tri_eval          Common/DataModel/vtkTriangle.cxx, vtkTriangle::EvaluatePosition: points
//                     taken as pt1 = P1, pt2 = P2, pt3 = P0, vtkTriangle::ComputeNormalDirection,
//                     vtkPlane::GeneralizedProjectPoint (Common/DataModel/vtkPlane.h),
//                     the largest |n| component dropped, vtkMath::Determinant2x2 for det and
//                     the two parametric coordinates, weights (1 - r - s, r, s), the inside
//                     test 0 <= w <= 1, then the six outside branches in VTK's order (two
//                     negative weights: corner and two edges; one: one edge)
//   polygon_normal    Common/DataModel/vtkPolygon.cxx, vtkPolygon::ComputeNormal(vtkPoints*,
//                     double[3]): cross products accumulated corner by corner, then
//                     vtkMath::Normalize
//   tie rule          Common/DataModel/vtkCellLocator.cxx, FindClosestPoint keeps a candidate
//                     only when its dist2 is smaller than the best so far; which of several
//                     cells at the same distance it meets first depends on its buckets, so
//                     the smallest cell id is used here
#pragma once
#include <hip/hip_runtime.h>

namespace icpx
{

struct d3
{
  double x, y, z;
};

__device__ __forceinline__ d3 mk(double x, double y, double z)
{
  d3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
__device__ __forceinline__ d3 sel(bool c, const d3& a, const d3& b)
{
  return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z);
}
__device__ __forceinline__ double dot3(const d3& a, const d3& b)
{
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ double dist2(const d3& a, const d3& b)
{
  return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
}
__device__ __forceinline__ void normalize(d3& v)
{
  double den = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  if (den != 0.0)
  {
    v.x /= den;
    v.y /= den;
    v.z /= den;
  }
}

// vtkLine::DistanceToLine: squared distance from x to segment (p1,p2) and the closest point.
__device__ __forceinline__ double dist_to_line(const d3& x, const d3& p1, const d3& p2, d3& cp)
{
  d3 p21 = mk(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z);
  double num = p21.x * (x.x - p1.x) + p21.y * (x.y - p1.y) + p21.z * (x.z - p1.z);
  double denom = dot3(p21, p21);
  double tol = 1e-5 * num;
  if (tol < 0.0)
    tol = -tol;
  cp = p1;
  if (!(denom < tol || denom <= 0.0))
  {
    double t = num / denom;
    if (t < 0.0)
      cp = p1;
    else if (t > 1.0)
      cp = p2;
    else
      cp = mk(p1.x + t * p21.x, p1.y + t * p21.y, p1.z + t * p21.z);
  }
  return dist2(cp, x);
}

// vtkTriangle::EvaluatePosition(x) on cell points (P0,P1,P2).
// Returns false for a degenerate triangle (the locator skips those).
// w = (1-r-s, r, s).
__device__ __forceinline__ bool tri_eval(
  const d3& x, const d3& P0, const d3& P1, const d3& P2, d3& cp, double& d2, double w[3])
{
  const d3 &pt1 = P1, &pt2 = P2, &pt3 = P0;
  d3 n;
  {
    double ax = pt3.x - pt2.x, ay = pt3.y - pt2.y, az = pt3.z - pt2.z;
    double bx = pt1.x - pt2.x, by = pt1.y - pt2.y, bz = pt1.z - pt2.z;
    n = mk(ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx);
  }
  d3 proj;
  {
    d3 xo = mk(x.x - pt1.x, x.y - pt1.y, x.z - pt1.z);
    double t = dot3(n, xo);
    double n2 = dot3(n, n);
    if (n2 != 0.0)
      proj = mk(x.x - t * n.x / n2, x.y - t * n.y / n2, x.z - t * n.z / n2);
    else
      proj = x;
  }
  int idx = 0;
  {
    double maxc = 0.0;
    double f = n.x < 0 ? -n.x : n.x;
    if (f > maxc)
    {
      maxc = f;
      idx = 0;
    }
    f = n.y < 0 ? -n.y : n.y;
    if (f > maxc)
    {
      maxc = f;
      idx = 1;
    }
    f = n.z < 0 ? -n.z : n.z;
    if (f > maxc)
    {
      maxc = f;
      idx = 2;
    }
  }
  // the two kept axes: (y,z), (x,z) or (x,y)
  const bool i0 = (idx == 0), i2 = (idx == 2);
  double pu = i0 ? proj.y : proj.x, pv = i2 ? proj.y : proj.z;
  double au = i0 ? pt1.y : pt1.x, av = i2 ? pt1.y : pt1.z;
  double bu = i0 ? pt2.y : pt2.x, bv = i2 ? pt2.y : pt2.z;
  double cu = i0 ? pt3.y : pt3.x, cv = i2 ? pt3.y : pt3.z;
  double rhs0 = pu - cu, rhs1 = pv - cv;
  double c10 = au - cu, c11 = av - cv;
  double c20 = bu - cu, c21 = bv - cv;
  double det = c10 * c21 - c20 * c11;
  if (det == 0.0)
    return false;
  double pc0 = (rhs0 * c21 - c20 * rhs1) / det;
  double pc1 = (c10 * rhs1 - rhs0 * c11) / det;
  w[0] = 1 - (pc0 + pc1);
  w[1] = pc0;
  w[2] = pc1;
  const bool n0 = w[0] < 0.0, n1 = w[1] < 0.0, n2 = w[2] < 0.0;
  const bool inside = w[0] >= 0.0 && w[0] <= 1.0 && w[1] >= 0.0 && w[1] <= 1.0 && w[2] >= 0.0 &&
    w[2] <= 1.0;
  if (inside || !(n0 || n1 || n2))
  {
    // inside, or (rounding only) a weight above 1 with none below 0: the projection
    d2 = dist2(proj, x);
    cp = proj;
    return true;
  }
  // Corner regions (two negative weights) compare the corner and two edges;
  // edge regions (one negative weight) take one edge.  The same code serves
  // every case with selected end points, in VTK's order of evaluation.
  const bool k3 = n1 && n2;        // corner pt3: lines (pt1,pt3), (pt3,pt2)
  const bool k1 = !k3 && n2 && n0; // corner pt1: lines (pt1,pt3), (pt1,pt2)
  const bool k2 = !k3 && !k1 && n1 && n0; // corner pt2: lines (pt2,pt3), (pt1,pt2)
  const bool corner = k3 || k1 || k2;
  // first line: (pt1|pt2 , pt3|pt2)
  const bool a1_is_pt2 = corner ? k2 : (!n0 && n1);
  const bool a2_is_pt2 = !corner && n0;
  const d3 a1 = sel(a1_is_pt2, pt2, pt1);
  const d3 a2 = sel(a2_is_pt2, pt2, pt3);
  d3 cpa;
  double da = dist_to_line(x, a1, a2, cpa);
  d2 = da;
  cp = cpa;
  if (corner)
  {
    d3 b1 = sel(k3, pt3, pt1);
    d3 b2 = pt2;
    d3 cnr = sel(k3, pt3, sel(k1, pt1, pt2));
    d3 cpb;
    double db = dist_to_line(x, b1, b2, cpb);
    double dp = dist2(x, cnr);
    if (dp < da)
    {
      d2 = dp;
      cp = cnr;
    }
    if (db < d2)
    {
      d2 = db;
      cp = cpb;
    }
  }
  return true;
}

// vtkPolygon::ComputeNormal on a triangle's three points: sum of the corner
// cross products, normalised.
__device__ __forceinline__ d3 polygon_normal(const d3& q0, const d3& q1, const d3& q2)
{
  d3 n = mk(0.0, 0.0, 0.0);
  // i = 0: v0=q0 v1=q1 v2=q2 ; i = 1: v0=q1 v1=q2 v2=q0 ; i = 2: v0=q2 v1=q0 v2=q1
  const d3* v0s[3] = { &q0, &q1, &q2 };
  const d3* v1s[3] = { &q1, &q2, &q0 };
  const d3* v2s[3] = { &q2, &q0, &q1 };
#pragma unroll
  for (int i = 0; i < 3; i++)
  {
    const d3 &v0 = *v0s[i], &v1 = *v1s[i], &v2 = *v2s[i];
    double ax = v2.x - v1.x, ay = v2.y - v1.y, az = v2.z - v1.z;
    double bx = v0.x - v1.x, by = v0.y - v1.y, bz = v0.z - v1.z;
    n.x += (ay * bz - az * by);
    n.y += (az * bx - ax * bz);
    n.z += (ax * by - ay * bx);
  }
  normalize(n);
  return n;
}

} // namespace icpx
