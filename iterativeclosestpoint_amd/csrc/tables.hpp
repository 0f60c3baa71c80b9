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

// Per Morton position of a triangle: {face normal as emitted, cell id}, for the finalize
// kernels (tid: position -> cell id, -1 - id for a degenerate triangle, which never wins).
__global__ void k_face_by_position(const float* __restrict__ fn32, const int32_t* __restrict__ tid,
  int64_t nt, float4* __restrict__ fnpos)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt)
    return;
  const int32_t id = tid[s];
  const int64_t c = id >= 0 ? id : -1 - (int64_t)id;
  fnpos[s] = make_float4(fn32[3 * c], fn32[3 * c + 1], fn32[3 * c + 2], __int_as_float(id));
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

// ---- consistent winding (vtkPolyDataNormals with ConsistencyOn, which the filter's own
// normals generator runs with: ICP/vtkPolyDataCorrespondenceFilter.cxx:358).  VTK walks the
// mesh from the lowest unvisited cell across shared edges and reverses every neighbour that
// runs along the shared edge in the same direction; the outcome for a mesh of orientable
// pieces is that each edge-connected piece takes the orientation of its lowest cell.  Here:
// the neighbour across every manifold edge (exactly one other cell) with the relation
// "same direction = must differ in flip"; then the smallest cell id of the piece is spread
// over it together with the flip parity relative to that cell.  Edges with more than two
// cells are not walked across (VTK's walk over them depends on its queue order).
//
// nbr[3 c + e]: (neighbour << 1) | relation across edge e (points e, e+1 of cell c), or -1.
__global__ void k_edge_neighbours(const int32_t* __restrict__ tri, const uint32_t* __restrict__ voff,
  const int32_t* __restrict__ vcell, int64_t nt, int32_t* __restrict__ nbr,
  uint32_t* __restrict__ n_inconsistent)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * nt)
    return;
  const int64_t c = i / 3;
  const int e = (int)(i % 3);
  const int32_t a = tri[3 * c + e], b = tri[3 * c + (e + 1) % 3];
  int32_t found = -1, prev = -1;
  int count = 0, rel = 0;
  if (a != b)
    for (uint32_t q = voff[a]; q < voff[a + 1]; q++)
    {
      const int32_t o = vcell[q];
      if (o == prev)
        continue;
      prev = o;
      if (o == (int32_t)c)
        continue;
      const int32_t t0 = tri[3 * (int64_t)o], t1 = tri[3 * (int64_t)o + 1], t2 = tri[3 * (int64_t)o + 2];
      if (t0 != b && t1 != b && t2 != b)
        continue;
      count++;
      found = o;
      // consistent when the neighbour runs b -> a; the point after its first b is what VTK
      // looks at (vtkPolyDataNormals::TraverseAndOrder)
      const int l = (t0 == b) ? 0 : (t1 == b) ? 1 : 2;
      const int32_t next = (l == 0) ? t1 : (l == 1) ? t2 : t0;
      rel = (next != a) ? 1 : 0;
    }
  const bool manifold = count == 1;
  nbr[i] = manifold ? ((found << 1) | rel) : -1;
  if (manifold && rel)
    atomicAdd(n_inconsistent, 1u);
}

__global__ void k_orient_init(int64_t nt, uint32_t* __restrict__ lab)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nt)
    lab[c] = (uint32_t)c << 1; // (piece's lowest cell so far << 1) | flip relative to it
}

// One sweep: every cell takes the smallest label among its neighbours', with the parity the
// relation across the edge gives, and then looks where the cell its label names has got to
// meanwhile (that cell's own label, the two parities combined): labels travel along chains of
// such references as well as from neighbour to neighbour, so a long thin piece settles in a
// number of sweeps that grows with the logarithm of its length, not with the length.  Labels
// only ever decrease, so sweeps may overlap freely.  A neighbour that carries the cell's own
// label with the other parity belongs to a piece that cannot be wound consistently (a Moebius
// band): *conflict is raised; which of its cells end up reversed then depends on the order in
// which the sweeps happened to run.
__global__ void k_orient_sweep(const int32_t* __restrict__ nbr, int64_t nt,
  uint32_t* __restrict__ lab, uint32_t* __restrict__ changed, uint32_t* __restrict__ conflict)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nt)
    return;
  uint32_t best = lab[c];
  bool ch = false, clash = false;
#pragma unroll
  for (int e = 0; e < 3; e++)
  {
    const int32_t nb = nbr[3 * c + e];
    if (nb < 0)
      continue;
    const uint32_t lu = *reinterpret_cast<volatile uint32_t*>(&lab[nb >> 1]);
    const uint32_t cand = lu ^ (uint32_t)(nb & 1);
    if ((cand >> 1) < (best >> 1))
    {
      best = cand;
      ch = true;
    }
    else if ((cand >> 1) == (best >> 1) && cand != best)
      clash = true;
  }
  // (the label's cell relative to ITS label, and this cell relative to that one)
  const uint32_t up = *reinterpret_cast<volatile uint32_t*>(&lab[best >> 1]);
  if ((up >> 1) < (best >> 1))
  {
    best = (up & ~1u) | ((up ^ best) & 1u);
    ch = true;
  }
  if (ch)
  {
    lab[c] = best;
    *changed = 1u;
  }
  if (clash)
    *conflict = 1u;
}

// vtkPolyData::ReverseCell on the cells found flipped: (p0, p1, p2) -> (p2, p1, p0).
__global__ void k_orient_apply(const uint32_t* __restrict__ lab, int64_t nt, int32_t* __restrict__ tri,
  uint32_t* __restrict__ n_flipped)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nt || !(lab[c] & 1u))
    return;
  const int32_t t0 = tri[3 * c];
  tri[3 * c] = tri[3 * c + 2];
  tri[3 * c + 2] = t0;
  atomicAdd(n_flipped, 1u);
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
