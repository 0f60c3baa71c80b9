// tables.hpp — pseudo-normal tables of the target, built once per target.
//
// vtkImplicitPolyDataDistance2::ClosestPointAndNormal
// (ICP/vtkImplicitPolyDataDistance2.cxx:397-517) picks one of three normals
// for the closest point: the face normal, the normalised sum of the normals
// of the faces around an edge, or the angle-weighted sum over the faces around
// a vertex.  The last two are properties of the edge / vertex, not of the
// query, so they are computed here once (in fp64, in ascending cell order as
// the reference's link walks do, then rounded to the float the reference
// emits) instead of walking links per query.
#pragma once
#include "tri.hpp"
#include <stdint.h>

namespace icpx
{

__device__ __forceinline__ d3 load_vtx(const float* __restrict__ xyz, int32_t v)
{
  const float* p = xyz + 3 * (int64_t)v;
  return mk((double)p[0], (double)p[1], (double)p[2]);
}

// Face normals: computed as vtkPolygon::ComputeNormal does, or taken from the
// caller's per-cell normals.  fn64 feeds the edge / vertex sums, fn32 is the table.
__global__ void k_face_normals(const float* __restrict__ xyz, const int32_t* __restrict__ tri,
  const float* __restrict__ cell_normals, int64_t nt, double* __restrict__ fn64,
  float* __restrict__ fn32)
{
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nt)
    return;
  d3 n;
  if (cell_normals)
    n = mk((double)cell_normals[3 * c], (double)cell_normals[3 * c + 1],
      (double)cell_normals[3 * c + 2]);
  else
    n = polygon_normal(load_vtx(xyz, tri[3 * c]), load_vtx(xyz, tri[3 * c + 1]),
      load_vtx(xyz, tri[3 * c + 2]));
  fn64[3 * c] = n.x;
  fn64[3 * c + 1] = n.y;
  fn64[3 * c + 2] = n.z;
  fn32[3 * c] = (float)n.x;
  fn32[3 * c + 1] = (float)n.y;
  fn32[3 * c + 2] = (float)n.z;
}

// Links (vtkPolyData::BuildLinks): cells per point as CSR.
__global__ void k_valence(const int32_t* __restrict__ tri, int64_t n3, uint32_t* __restrict__ val)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3)
    atomicAdd(&val[tri[i]], 1u);
}

__global__ void k_fill_links(const int32_t* __restrict__ tri, int64_t n3,
  const uint32_t* __restrict__ voff, uint32_t* __restrict__ cursor, int32_t* __restrict__ vcell)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3)
    return;
  int32_t v = tri[i];
  uint32_t slot = atomicAdd(&cursor[v], 1u);
  vcell[voff[v] + slot] = (int32_t)(i / 3);
}

// Each point's cell list into ascending order (the order BuildLinks produces).
__global__ void k_sort_links(
  const uint32_t* __restrict__ voff, int64_t nv, int32_t* __restrict__ vcell)
{
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv)
    return;
  int32_t* a = vcell + voff[v];
  int n = (int)(voff[v + 1] - voff[v]);
  for (int i = 1; i < n; i++)
  {
    int32_t x = a[i];
    int j = i - 1;
    while (j >= 0 && a[j] > x)
    {
      a[j + 1] = a[j];
      j--;
    }
    a[j + 1] = x;
  }
}

// Point normals of a source mesh as the filter's own vtkPolyDataNormals makes them with
// splitting off (ICP/vtkPolyDataCorrespondenceFilter.cxx:356-359,457-464): the normalised sum
// of the unit normals of the faces at the point, one term per corner, in link order (fixed,
// so the result repeats run to run); a point without faces, or whose terms cancel, gets zero.
__global__ void k_point_normals(const uint32_t* __restrict__ voff,
  const int32_t* __restrict__ vcell, const double* __restrict__ fn64, int64_t nv,
  float* __restrict__ pn32)
{
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv)
    return;
  d3 s = mk(0.0, 0.0, 0.0);
  for (uint32_t q = voff[v]; q < voff[v + 1]; q++)
  {
    const int64_t c = vcell[q];
    s.x += fn64[3 * c];
    s.y += fn64[3 * c + 1];
    s.z += fn64[3 * c + 2];
  }
  normalize(s);
  pn32[3 * v] = (float)s.x;
  pn32[3 * v + 1] = (float)s.y;
  pn32[3 * v + 2] = (float)s.z;
}

// Angle-weighted pseudo-normal of every point (:475-516).
__device__ __forceinline__ d3 vertex_normal64(const float* __restrict__ xyz,
  const int32_t* __restrict__ tri, const uint32_t* __restrict__ voff,
  const int32_t* __restrict__ vcell, const double* __restrict__ fn64, int64_t v)
{
  const int32_t a = (int32_t)v;
  d3 pa = load_vtx(xyz, a);
  d3 aw = mk(0.0, 0.0, 0.0);
  for (uint32_t q = voff[v]; q < voff[v + 1]; q++)
  {
    int64_t c = vcell[q];
    int32_t t0 = tri[3 * c], t1 = tri[3 * c + 1], t2 = tri[3 * c + 2];
    int32_t b = t0, d = t1;
    if (a == b)
      b = t2;
    else if (a == d)
      d = t2;
    d3 pb = load_vtx(xyz, b), pc = load_vtx(xyz, d);
    pb = mk(pb.x - pa.x, pb.y - pa.y, pb.z - pa.z);
    pc = mk(pc.x - pa.x, pc.y - pa.y, pc.z - pa.z);
    normalize(pb);
    normalize(pc);
    double alpha = acos(dot3(pb, pc));
    aw.x += alpha * fn64[3 * c];
    aw.y += alpha * fn64[3 * c + 1];
    aw.z += alpha * fn64[3 * c + 2];
  }
  normalize(aw);
  return aw;
}

__global__ void k_vertex_normals(const float* __restrict__ xyz, const int32_t* __restrict__ tri,
  const uint32_t* __restrict__ voff, const int32_t* __restrict__ vcell,
  const double* __restrict__ fn64, int64_t nv, float* __restrict__ vn32)
{
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv)
    return;
  const d3 aw = vertex_normal64(xyz, tri, voff, vcell, fn64, v);
  vn32[3 * v] = (float)aw.x;
  vn32[3 * v + 1] = (float)aw.y;
  vn32[3 * v + 2] = (float)aw.z;
}

// Edge pseudo-normal for every (cell, edge): edge e joins points (e+1)%3 and
// (e+2)%3 of the cell; all cells of the first point that also use the second
// contribute, in link order (GetCellEdgeNeighbors with a bogus cell id, :433-450).
__device__ __forceinline__ d3 edge_normal64(const int32_t* __restrict__ tri,
  const uint32_t* __restrict__ voff, const int32_t* __restrict__ vcell,
  const double* __restrict__ fn64, int64_t c, int e)
{
  int32_t a = tri[3 * c + (e + 1) % 3];
  int32_t b = tri[3 * c + (e + 2) % 3];
  d3 aw = mk(0.0, 0.0, 0.0);
  int32_t prev = -1;
  for (uint32_t q = voff[a]; q < voff[a + 1]; q++)
  {
    int32_t o = vcell[q];
    if (o == prev)
      continue;
    prev = o;
    if (tri[3 * (int64_t)o] == b || tri[3 * (int64_t)o + 1] == b || tri[3 * (int64_t)o + 2] == b)
    {
      aw.x += fn64[3 * (int64_t)o];
      aw.y += fn64[3 * (int64_t)o + 1];
      aw.z += fn64[3 * (int64_t)o + 2];
    }
  }
  normalize(aw);
  return aw;
}

__global__ void k_edge_normals(const int32_t* __restrict__ tri,
  const uint32_t* __restrict__ voff, const int32_t* __restrict__ vcell,
  const double* __restrict__ fn64, int64_t nt, float* __restrict__ en32)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * nt)
    return;
  const d3 aw = edge_normal64(tri, voff, vcell, fn64, i / 3, (int)(i % 3));
  en32[3 * i] = (float)aw.x;
  en32[3 * i + 1] = (float)aw.y;
  en32[3 * i + 2] = (float)aw.z;
}

} // namespace icpx
