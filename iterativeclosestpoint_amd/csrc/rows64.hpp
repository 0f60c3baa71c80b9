// rows64.hpp — the correspondence filter's rows in double precision.
//
// With OutputPrecision = DOUBLE the reference runs Functor<vtkFloatArray, vtkDoubleArray,
// vtkDoubleArray> (ICP/vtkPolyDataCorrespondenceFilter.cxx:519-531): float input points,
// and the transformed point, the closest point and the locator's normal stored as the
// doubles they were computed in (no origin normals, no normal-compatibility test on that
// path).  The float tables of tables.hpp are roundings of fp64 values; here the face
// normal comes from the fp64 table they were rounded from and the rare edge / vertex
// normals are summed on the spot from the same links, as
// vtkImplicitPolyDataDistance2::ClosestPointAndNormal does per query
// (ICP/vtkImplicitPolyDataDistance2.cxx:410-517).
#pragma once
#include "query.hpp"
#include "tables.hpp"

namespace icpx
{

struct Rows64Args
{
  const float* qpos;
  const int32_t* qk;
  int64_t nq;
  XfRef T;
  const float* tverts;
  const int32_t* tid;
  const int32_t* tri;
  const float* xyz;
  const double* fn64;
  const uint32_t* voff;
  const int32_t* vcell;
  const int32_t* hint;
  double max_dist;
  // rows, indexed by kept index
  double *o_origin, *o_closest, *o_normal, *o_dist;
  int32_t* o_cell;
  uint32_t* o_flag;
};

__global__ __launch_bounds__(256) void k_rows64(Rows64Args a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.nq)
    return;
  const d3 x = xform(T, a.qpos + 3 * p);
  const int32_t hs = a.hint[p];
  const int64_t k = a.qk[p];
  bool accept = false;
  if (hs >= 0)
  {
    d3 P0, P1, P2, pcl, cp;
    double d2, w[3], d2b;
    load_tri(a.tverts, hs, P0, P1, P2);
    tri_eval(x, P0, P1, P2, pcl, d2, w);   // closest point p and vlen2 (:375)
    const double dist = sqrt(d2);          // :380
    tri_eval(pcl, P0, P1, P2, cp, d2b, w); // weights at p, closestPoint (:388)
    const int32_t cell = a.tid[hs];
    const bool z0 = fabs(w[0]) < 1e-12, z1 = fabs(w[1]) < 1e-12, z2 = fabs(w[2]) < 1e-12;
    const int count = (z0 ? 1 : 0) + (z1 ? 1 : 0) + (z2 ? 1 : 0);
    d3 n = mk(0.0, 0.0, 0.0);
    if (count == 0)
      n = mk(a.fn64[3 * (int64_t)cell], a.fn64[3 * (int64_t)cell + 1], a.fn64[3 * (int64_t)cell + 2]);
    else if (count == 1)
      n = edge_normal64(a.tri, a.voff, a.vcell, a.fn64, cell, z0 ? 0 : (z1 ? 1 : 2));
    else if (count == 2)
      n = vertex_normal64(a.xyz, a.tri, a.voff, a.vcell, a.fn64,
        a.tri[3 * (int64_t)cell + (!z2 ? 2 : (!z1 ? 1 : 0))]);
    accept = dist < a.max_dist;
    if (accept)
    {
      a.o_origin[3 * k] = x.x;
      a.o_origin[3 * k + 1] = x.y;
      a.o_origin[3 * k + 2] = x.z;
      a.o_closest[3 * k] = cp.x;
      a.o_closest[3 * k + 1] = cp.y;
      a.o_closest[3 * k + 2] = cp.z;
      a.o_normal[3 * k] = n.x;
      a.o_normal[3 * k + 1] = n.y;
      a.o_normal[3 * k + 2] = n.z;
      a.o_cell[k] = cell;
      a.o_dist[k] = dist;
    }
  }
  a.o_flag[k] = accept ? 1u : 0u;
}

struct Compact64Args
{
  int64_t nq, k0, step;
  const int32_t* kglobal; // slot -> kept index of a shard that is not a block of them (or null)
  const uint32_t* flag;
  const uint32_t* pos; // exclusive scan of flag
  const double *origin, *closest, *normal, *dist;
  const int32_t* cell;
  double *d_origin, *d_closest, *d_normal, *d_dist;
  int32_t* d_cell;
  int64_t* d_index;
};

__global__ void k_compact_rows64(Compact64Args a)
{
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.nq || !a.flag[k])
    return;
  int64_t r = a.pos[k];
  for (int c = 0; c < 3; c++)
  {
    a.d_origin[3 * r + c] = a.origin[3 * k + c];
    a.d_closest[3 * r + c] = a.closest[3 * k + c];
    a.d_normal[3 * r + c] = a.normal[3 * k + c];
  }
  a.d_cell[r] = a.cell[k];
  a.d_dist[r] = a.dist[k];
  a.d_index[r] = (a.kglobal ? (int64_t)a.kglobal[k] : a.k0 + k) * a.step;
}

} // namespace icpx
