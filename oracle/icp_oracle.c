/*
 * icp_oracle.c — CPU restatement of the reference ICP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library or in the Python
 * package may link, load or call this file; it exists so that tests, the
 * smoke check and the benchmark's cpu_baseline leg have an independent
 * answer to compare the HIP path with.
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path (its
 * only C++ test asserts nothing: ICP/Testing/Cxx/TestPolyDataCorrespondenceFilter.cxx:85)
 * and the arithmetic lives in VTK >= 9.3, which is not vendored and not
 * installed.  The VTK pieces below are restated from the published VTK 9.3
 * algorithms; the reference's own files are followed line by line where cited.
 *
 * Everything is plain fp64 with one rounding per operation (build with
 * -ffp-contract=off), in the operation order of the code it follows.
 *
 * Follows:
 *   ICP/vtkImplicitPolyDataDistance2.cxx:338-538  ClosestPointAndNormal
 *   ICP/vtkImplicitPolyDataDistance2.cxx:34-62    SetInput (links, locator)
 *   ICP/vtkPolyDataCorrespondenceFilter.cxx:87-137, 252-306, 473-477
 *   ICP/vtkICP.cxx:130-247                        InternalUpdate
 *   ICP/icp/point2point.cxx:25-66, point2plane.cxx:37-110, plane2plane.cxx:37-94
 *   VTK 9.3 (restated): vtkTriangle::EvaluatePosition, vtkLine::DistanceToLine,
 *   vtkPlane::GeneralizedProjectPoint, vtkPolygon::ComputeNormal,
 *   vtkCellLocator::FindClosestPoint (as: exact minimum over all triangles,
 *   first-smallest wins => smallest cell id on exact ties).
 *
 * For an audit against the VTK 9.3 source tree (not available here, so files and
 * functions are named rather than lines):
 *   dist_to_line       Common/DataModel/vtkLine.cxx       vtkLine::DistanceToLine (VTK_TOL 1e-5)
 *   orc_tri_eval       Common/DataModel/vtkTriangle.cxx   vtkTriangle::EvaluatePosition,
 *                                                         ComputeNormalDirection
 *                      Common/DataModel/vtkPlane.h        vtkPlane::GeneralizedProjectPoint
 *                      Common/Core/vtkMath.h              Determinant2x2, Distance2BetweenPoints
 *   polygon_normal     Common/DataModel/vtkPolygon.cxx    vtkPolygon::ComputeNormal(vtkPoints*, double*)
 *   tri_intersect_line Common/DataModel/vtkTriangle.cxx   vtkTriangle::IntersectWithLine, ComputeNormal
 *                      Common/DataModel/vtkPlane.cxx      vtkPlane::IntersectWithLine (VTK_PLANE_TOL 1e-6)
 *   links              Common/DataModel/vtkPolyData.cxx   BuildLinks, GetCellEdgeNeighbors, GetPointCells
 *   build_grid / orc_find_closest_grid
 *                      Common/DataModel/vtkCellLocator.cxx BuildLocator, FindClosestPoint (only
 *                                                         their shape: a uniform bucket grid
 *                                                         searched in growing shells)
 * Checks that share no arithmetic with this file: tests/test_pins.py (answers worked out by
 * hand, numpy's SVD / solve for the solvers) and, on the GPU, Ericson's closest point on a
 * triangle against the emitted distances.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_TOL 1e-12 /* vtkImplicitPolyDataDistance2.cxx:30 */

typedef struct orc_mesh
{
  int64_t nv, nt;
  float* xyz;      /* nv*3 */
  int32_t* tri;    /* nt*3 */
  double* fnorm;   /* nt*3 unit face normals (computed, or the caller's cell normals) */
  int64_t* voff;   /* nv+1: CSR of cells per point, ascending cell id (BuildLinks order) */
  int32_t* vcell;
  double bmin[3], bmax[3];
  /* uniform bucket grid, the shape of vtkCellLocator (10 cells per bucket, automatic) */
  int ndiv;
  double h[3];
  int64_t* goff;   /* ndiv^3+1 */
  int32_t* gcell;
  float* tbounds;  /* nt*6 cached cell bounds (CacheCellBoundsOn) */
} orc_mesh;

/* ------------------------------------------------------------------ vtkMath */
static inline double dot3(const double a[3], const double b[3])
{
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline double dist2pts(const double a[3], const double b[3])
{
  return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) +
    (a[2] - b[2]) * (a[2] - b[2]);
}
static inline double normalize3(double v[3])
{
  double den = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (den != 0.0)
  {
    v[0] /= den;
    v[1] /= den;
    v[2] /= den;
  }
  return den;
}

/* --------------------------------------------------- vtkLine::DistanceToLine */
static double dist_to_line(
  const double x[3], const double p1[3], const double p2[3], double cp[3])
{
  double p21[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
  double num = p21[0] * (x[0] - p1[0]) + p21[1] * (x[1] - p1[1]) + p21[2] * (x[2] - p1[2]);
  double denom = dot3(p21, p21);
  double tol = 1e-5 * num; /* VTK_TOL */
  double t;
  const double* closest;
  if (tol < 0.0)
    tol = -tol;
  if (denom < tol || denom <= 0.0)
    closest = p1;
  else if ((t = num / denom) < 0.0)
    closest = p1;
  else if (t > 1.0)
    closest = p2;
  else
  {
    p21[0] = p1[0] + t * p21[0];
    p21[1] = p1[1] + t * p21[1];
    p21[2] = p1[2] + t * p21[2];
    closest = p21;
  }
  cp[0] = closest[0];
  cp[1] = closest[1];
  cp[2] = closest[2];
  return dist2pts(cp, x);
}

/* ---------------------------------------------- vtkTriangle::EvaluatePosition
 * P0,P1,P2 in cell order.  Returns 1 inside, 0 outside, -1 degenerate.
 * weights = (1-r-s, r, s) for (P0,P1,P2). */
int orc_tri_eval(const double x[3], const double P0[3], const double P1[3], const double P2[3],
  double cp[3], double* dist2, double w[3])
{
  const double *pt1 = P1, *pt2 = P2, *pt3 = P0;
  double n[3], proj[3];
  /* vtkTriangle::ComputeNormalDirection(pt1, pt2, pt3, n) */
  {
    double ax = pt3[0] - pt2[0], ay = pt3[1] - pt2[1], az = pt3[2] - pt2[2];
    double bx = pt1[0] - pt2[0], by = pt1[1] - pt2[1], bz = pt1[2] - pt2[2];
    n[0] = ay * bz - az * by;
    n[1] = az * bx - ax * bz;
    n[2] = ax * by - ay * bx;
  }
  /* vtkPlane::GeneralizedProjectPoint(x, pt1, n, proj) */
  {
    double xo[3] = { x[0] - pt1[0], x[1] - pt1[1], x[2] - pt1[2] };
    double t = dot3(n, xo);
    double n2 = dot3(n, n);
    if (n2 != 0.0)
    {
      proj[0] = x[0] - t * n[0] / n2;
      proj[1] = x[1] - t * n[1] / n2;
      proj[2] = x[2] - t * n[2] / n2;
    }
    else
    {
      proj[0] = x[0];
      proj[1] = x[1];
      proj[2] = x[2];
    }
  }
  /* drop the axis of the largest normal component */
  int idx = 0, ind[2], i, j;
  double maxc = 0.0;
  for (i = 0; i < 3; i++)
  {
    double f = n[i] < 0 ? -n[i] : n[i];
    if (f > maxc)
    {
      maxc = f;
      idx = i;
    }
  }
  for (j = 0, i = 0; i < 3; i++)
    if (i != idx)
      ind[j++] = i;
  double rhs[2], c1[2], c2[2];
  for (i = 0; i < 2; i++)
  {
    rhs[i] = proj[ind[i]] - pt3[ind[i]];
    c1[i] = pt1[ind[i]] - pt3[ind[i]];
    c2[i] = pt2[ind[i]] - pt3[ind[i]];
  }
  double det = c1[0] * c2[1] - c2[0] * c1[1];
  if (det == 0.0)
    return -1;
  double pc0 = (rhs[0] * c2[1] - c2[0] * rhs[1]) / det;
  double pc1 = (c1[0] * rhs[1] - rhs[0] * c1[1]) / det;
  w[0] = 1 - (pc0 + pc1);
  w[1] = pc0;
  w[2] = pc1;
  if (w[0] >= 0.0 && w[0] <= 1.0 && w[1] >= 0.0 && w[1] <= 1.0 && w[2] >= 0.0 && w[2] <= 1.0)
  {
    *dist2 = dist2pts(proj, x);
    cp[0] = proj[0];
    cp[1] = proj[1];
    cp[2] = proj[2];
    return 1;
  }
  double c1p[3], c2p[3], d2p, d2l1, d2l2;
  const double* closest;
  if (w[1] < 0.0 && w[2] < 0.0)
  {
    d2p = dist2pts(x, pt3);
    d2l1 = dist_to_line(x, pt1, pt3, c1p);
    d2l2 = dist_to_line(x, pt3, pt2, c2p);
    if (d2p < d2l1)
    {
      *dist2 = d2p;
      closest = pt3;
    }
    else
    {
      *dist2 = d2l1;
      closest = c1p;
    }
    if (d2l2 < *dist2)
    {
      *dist2 = d2l2;
      closest = c2p;
    }
    cp[0] = closest[0], cp[1] = closest[1], cp[2] = closest[2];
  }
  else if (w[2] < 0.0 && w[0] < 0.0)
  {
    d2p = dist2pts(x, pt1);
    d2l1 = dist_to_line(x, pt1, pt3, c1p);
    d2l2 = dist_to_line(x, pt1, pt2, c2p);
    if (d2p < d2l1)
    {
      *dist2 = d2p;
      closest = pt1;
    }
    else
    {
      *dist2 = d2l1;
      closest = c1p;
    }
    if (d2l2 < *dist2)
    {
      *dist2 = d2l2;
      closest = c2p;
    }
    cp[0] = closest[0], cp[1] = closest[1], cp[2] = closest[2];
  }
  else if (w[1] < 0.0 && w[0] < 0.0)
  {
    d2p = dist2pts(x, pt2);
    d2l1 = dist_to_line(x, pt2, pt3, c1p);
    d2l2 = dist_to_line(x, pt1, pt2, c2p);
    if (d2p < d2l1)
    {
      *dist2 = d2p;
      closest = pt2;
    }
    else
    {
      *dist2 = d2l1;
      closest = c1p;
    }
    if (d2l2 < *dist2)
    {
      *dist2 = d2l2;
      closest = c2p;
    }
    cp[0] = closest[0], cp[1] = closest[1], cp[2] = closest[2];
  }
  else if (w[0] < 0.0)
    *dist2 = dist_to_line(x, pt1, pt2, cp);
  else if (w[1] < 0.0)
    *dist2 = dist_to_line(x, pt2, pt3, cp);
  else if (w[2] < 0.0)
    *dist2 = dist_to_line(x, pt1, pt3, cp);
  else
  {
    /* only reachable with a weight above 1 and none below 0 (rounding): keep the projection */
    *dist2 = dist2pts(proj, x);
    cp[0] = proj[0], cp[1] = proj[1], cp[2] = proj[2];
  }
  return 0;
}

static inline void vtx(const orc_mesh* m, int32_t v, double p[3])
{
  p[0] = m->xyz[3 * (int64_t)v + 0];
  p[1] = m->xyz[3 * (int64_t)v + 1];
  p[2] = m->xyz[3 * (int64_t)v + 2];
}

/* vtkPolygon::ComputeNormal(points, n) for a triangle: accumulate the cross
 * product at each of the three corners, then normalise. */
static void polygon_normal(const double q0[3], const double q1[3], const double q2[3], double n[3])
{
  const double* pts[3] = { q0, q1, q2 };
  const double *v0, *v1 = pts[0], *v2 = pts[1];
  n[0] = n[1] = n[2] = 0.0;
  for (int i = 0; i < 3; i++)
  {
    v0 = v1;
    v1 = v2;
    v2 = pts[(i + 2) % 3];
    double ax = v2[0] - v1[0], ay = v2[1] - v1[1], az = v2[2] - v1[2];
    double bx = v0[0] - v1[0], by = v0[1] - v1[1], bz = v0[2] - v1[2];
    n[0] += (ay * bz - az * by);
    n[1] += (az * bx - ax * bz);
    n[2] += (ax * by - ay * bx);
  }
  normalize3(n);
}

/* --------------------------------------------------------------- mesh set-up */
static void build_grid(orc_mesh* m);

orc_mesh* orc_mesh_create(
  const float* xyz, int64_t nv, const int32_t* tri, int64_t nt, const float* cell_normals)
{
  orc_mesh* m = (orc_mesh*)calloc(1, sizeof(orc_mesh));
  m->nv = nv;
  m->nt = nt;
  m->xyz = (float*)malloc(sizeof(float) * 3 * (size_t)nv);
  m->tri = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nt);
  memcpy(m->xyz, xyz, sizeof(float) * 3 * (size_t)nv);
  memcpy(m->tri, tri, sizeof(int32_t) * 3 * (size_t)nt);
  m->fnorm = (double*)malloc(sizeof(double) * 3 * (size_t)nt);
  for (int64_t c = 0; c < nt; c++)
  {
    if (cell_normals)
    {
      for (int k = 0; k < 3; k++)
        m->fnorm[3 * c + k] = cell_normals[3 * c + k];
    }
    else
    {
      double a[3], b[3], d[3];
      vtx(m, tri[3 * c], a);
      vtx(m, tri[3 * c + 1], b);
      vtx(m, tri[3 * c + 2], d);
      polygon_normal(a, b, d, &m->fnorm[3 * c]);
    }
  }
  /* vtkPolyData::BuildLinks: per point, the cells that use it, in ascending cell id */
  m->voff = (int64_t*)calloc((size_t)nv + 1, sizeof(int64_t));
  for (int64_t c = 0; c < nt; c++)
    for (int k = 0; k < 3; k++)
      m->voff[tri[3 * c + k] + 1]++;
  for (int64_t v = 0; v < nv; v++)
    m->voff[v + 1] += m->voff[v];
  m->vcell = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nt + 4);
  int64_t* fill = (int64_t*)calloc((size_t)nv, sizeof(int64_t));
  for (int64_t c = 0; c < nt; c++)
    for (int k = 0; k < 3; k++)
    {
      int32_t v = tri[3 * c + k];
      /* a cell that lists a point twice is linked twice by BuildLinks as well */
      m->vcell[m->voff[v] + fill[v]++] = (int32_t)c;
    }
  free(fill);
  for (int k = 0; k < 3; k++)
  {
    m->bmin[k] = INFINITY;
    m->bmax[k] = -INFINITY;
  }
  for (int64_t v = 0; v < nv; v++)
    for (int k = 0; k < 3; k++)
    {
      double x = xyz[3 * v + k];
      if (x < m->bmin[k])
        m->bmin[k] = x;
      if (x > m->bmax[k])
        m->bmax[k] = x;
    }
  build_grid(m);
  return m;
}

void orc_mesh_destroy(orc_mesh* m)
{
  if (!m)
    return;
  free(m->xyz);
  free(m->tri);
  free(m->fnorm);
  free(m->voff);
  free(m->vcell);
  free(m->goff);
  free(m->gcell);
  free(m->tbounds);
  free(m);
}

/* ------------------------------------------------------- closest cell search */
static inline int better(double d2, int64_t c, double bd2, int64_t bc)
{
  return d2 < bd2 || (d2 == bd2 && c < bc);
}

static inline int eval_cell(
  const orc_mesh* m, int64_t c, const double x[3], double cp[3], double* d2, double w[3])
{
  double a[3], b[3], d[3];
  vtx(m, m->tri[3 * c], a);
  vtx(m, m->tri[3 * c + 1], b);
  vtx(m, m->tri[3 * c + 2], d);
  return orc_tri_eval(x, a, b, d, cp, d2, w);
}

/* Exact minimum over every triangle; ties go to the smallest cell id. */
void orc_find_closest_brute(
  const orc_mesh* m, const double x[3], double p[3], int64_t* cell, double* dist2)
{
  double bd2 = INFINITY;
  int64_t bc = -1;
  for (int64_t c = 0; c < m->nt; c++)
  {
    double cp[3], d2, w[3];
    if (eval_cell(m, c, x, cp, &d2, w) == -1)
      continue;
    if (better(d2, c, bd2, bc))
    {
      bd2 = d2;
      bc = c;
      p[0] = cp[0], p[1] = cp[1], p[2] = cp[2];
    }
  }
  *cell = bc;
  *dist2 = bd2;
}

/* The bucket grid: 2^level divisions per axis with level chosen so that a
 * bucket holds about 10 triangles, each triangle listed in every bucket its
 * bounds overlap (vtkCellLocator with NumberOfCellsPerBucket=10, Automatic). */
static void build_grid(orc_mesh* m)
{
  int64_t nt = m->nt;
  int level = 0;
  if (nt > 10)
    level = (int)ceil(log((double)nt / 10.0) / log(8.0));
  if (level > 8)
    level = 8;
  int nd = 1 << level;
  m->ndiv = nd;
  for (int k = 0; k < 3; k++)
  {
    double len = m->bmax[k] - m->bmin[k];
    if (len <= 0.0)
      len = 1.0;
    m->h[k] = len / nd;
  }
  m->tbounds = (float*)malloc(sizeof(float) * 6 * (size_t)(nt ? nt : 1));
  int64_t nb = (int64_t)nd * nd * nd;
  m->goff = (int64_t*)calloc((size_t)nb + 1, sizeof(int64_t));
  int32_t* rng = (int32_t*)malloc(sizeof(int32_t) * 6 * (size_t)(nt ? nt : 1));
  for (int64_t c = 0; c < nt; c++)
  {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 3; k++)
    {
      const float* q = &m->xyz[3 * (int64_t)m->tri[3 * c + k]];
      for (int a = 0; a < 3; a++)
      {
        if (q[a] < lo[a])
          lo[a] = q[a];
        if (q[a] > hi[a])
          hi[a] = q[a];
      }
    }
    for (int a = 0; a < 3; a++)
    {
      m->tbounds[6 * c + a] = lo[a];
      m->tbounds[6 * c + 3 + a] = hi[a];
      int i0 = (int)floor((lo[a] - m->bmin[a]) / m->h[a]);
      int i1 = (int)floor((hi[a] - m->bmin[a]) / m->h[a]);
      if (i0 < 0)
        i0 = 0;
      if (i0 > nd - 1)
        i0 = nd - 1;
      if (i1 < 0)
        i1 = 0;
      if (i1 > nd - 1)
        i1 = nd - 1;
      rng[6 * c + a] = i0;
      rng[6 * c + 3 + a] = i1;
    }
    for (int i = rng[6 * c]; i <= rng[6 * c + 3]; i++)
      for (int j = rng[6 * c + 1]; j <= rng[6 * c + 4]; j++)
        for (int k = rng[6 * c + 2]; k <= rng[6 * c + 5]; k++)
          m->goff[((int64_t)i * nd + j) * nd + k + 1]++;
  }
  for (int64_t b = 0; b < nb; b++)
    m->goff[b + 1] += m->goff[b];
  m->gcell = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->goff[nb] + 1));
  int64_t* fill = (int64_t*)calloc((size_t)nb, sizeof(int64_t));
  for (int64_t c = 0; c < nt; c++)
    for (int i = rng[6 * c]; i <= rng[6 * c + 3]; i++)
      for (int j = rng[6 * c + 1]; j <= rng[6 * c + 4]; j++)
        for (int k = rng[6 * c + 2]; k <= rng[6 * c + 5]; k++)
        {
          int64_t b = ((int64_t)i * nd + j) * nd + k;
          m->gcell[m->goff[b] + fill[b]++] = (int32_t)c;
        }
  free(fill);
  free(rng);
}

static inline double dist2_to_bounds(const double x[3], const float* b)
{
  double d2 = 0.0;
  for (int a = 0; a < 3; a++)
  {
    double d = 0.0;
    if (x[a] < b[a])
      d = b[a] - x[a];
    else if (x[a] > b[3 + a])
      d = x[a] - b[3 + a];
    d2 += d * d;
  }
  return d2;
}

/* Bucket search in growing shells around the query's bucket; a triangle is
 * evaluated only if its cached bounds are not farther than the best so far.
 * The result is the same exact (dist2, cell id) minimum as the brute force. */
void orc_find_closest_grid(
  const orc_mesh* m, const double x[3], double p[3], int64_t* cell, double* dist2)
{
  int nd = m->ndiv;
  int ic[3];
  for (int a = 0; a < 3; a++)
  {
    int i = (int)floor((x[a] - m->bmin[a]) / m->h[a]);
    if (i < 0)
      i = 0;
    if (i > nd - 1)
      i = nd - 1;
    ic[a] = i;
  }
  double bd2 = INFINITY;
  int64_t bc = -1;
  double hmin = m->h[0] < m->h[1] ? m->h[0] : m->h[1];
  if (m->h[2] < hmin)
    hmin = m->h[2];
  /* distance from x to the clamped start bucket, for points outside the grid */
  for (int r = 0; r < 2 * nd; r++)
  {
    if (bc >= 0)
    {
      /* every bucket of shell r is at least (r-1)*hmin away from x along some axis,
       * measured from the start bucket that contains (or is nearest to) x */
      double reach = (r - 1) * hmin;
      if (reach > 0.0 && reach * reach > bd2)
        break;
    }
    int any = 0;
    for (int i = ic[0] - r; i <= ic[0] + r; i++)
    {
      if (i < 0 || i >= nd)
        continue;
      for (int j = ic[1] - r; j <= ic[1] + r; j++)
      {
        if (j < 0 || j >= nd)
          continue;
        int onij = (i == ic[0] - r || i == ic[0] + r || j == ic[1] - r || j == ic[1] + r);
        int kstep = onij ? 1 : 2 * r;
        if (kstep == 0)
          kstep = 1;
        for (int k = ic[2] - r; k <= ic[2] + r; k += kstep)
        {
          if (k < 0 || k >= nd)
            continue;
          any = 1;
          int64_t b = ((int64_t)i * nd + j) * nd + k;
          for (int64_t q = m->goff[b]; q < m->goff[b + 1]; q++)
          {
            int64_t c = m->gcell[q];
            if (dist2_to_bounds(x, &m->tbounds[6 * c]) > bd2)
              continue;
            double cp[3], d2, w[3];
            if (eval_cell(m, c, x, cp, &d2, w) == -1)
              continue;
            if (better(d2, c, bd2, bc))
            {
              bd2 = d2;
              bc = c;
              p[0] = cp[0], p[1] = cp[1], p[2] = cp[2];
            }
          }
        }
      }
    }
    if (!any && r > nd)
      break;
  }
  *cell = bc;
  *dist2 = bd2;
}

/* The pseudo-normal at a point of cell cellId whose EvaluatePosition weights are w: the
 * face normal, the normalised sum of the normals of the faces on the edge, or the
 * angle-weighted sum over the faces at the vertex (ICP/vtkImplicitPolyDataDistance2.cxx:391-517;
 * the same code in ICP/vtkImplicitPolyDataIntersection.cxx:172-304).  awnorm must come in as
 * zero.  Returns the number of weights below the tolerance (0 face, 1 edge, 2 vertex). */
static int pseudo_normal(const orc_mesh* m, int64_t cellId, const double w[3], double awnorm[3])
{
  const int32_t* ids = &m->tri[3 * cellId];
  int count = 0;
  for (int i = 0; i < 3; i++)
    count += (fabs(w[i]) < ORC_TOL ? 1 : 0);
  if (count == 0)
  {
    awnorm[0] = m->fnorm[3 * cellId];
    awnorm[1] = m->fnorm[3 * cellId + 1];
    awnorm[2] = m->fnorm[3 * cellId + 2];
  }
  else if (count == 1)
  {
    int32_t a = -1, b = -1;
    for (int e = 0; e < 3; e++)
      if (fabs(w[e]) < ORC_TOL)
      {
        a = ids[(e + 1) % 3];
        b = ids[(e + 2) % 3];
        break;
      }
    /* GetCellEdgeNeighbors(bogus, a, b): cells of a that also use b, in link order */
    for (int64_t q = m->voff[a]; q < m->voff[a + 1]; q++)
    {
      int64_t c = m->vcell[q];
      if (q > m->voff[a] && m->vcell[q - 1] == c)
        continue;
      const int32_t* t = &m->tri[3 * c];
      if (t[0] == b || t[1] == b || t[2] == b)
      {
        awnorm[0] += m->fnorm[3 * c];
        awnorm[1] += m->fnorm[3 * c + 1];
        awnorm[2] += m->fnorm[3 * c + 2];
      }
    }
    normalize3(awnorm);
  }
  else if (count == 2)
  {
    int32_t a = -1;
    for (int i = 0; i < 3; i++)
      if (fabs(w[i]) > ORC_TOL)
        a = ids[i];
    if (a == -1)
      return count;
    for (int64_t q = m->voff[a]; q < m->voff[a + 1]; q++)
    {
      int64_t c = m->vcell[q];
      const int32_t* t = &m->tri[3 * c];
      int32_t b = t[0], d = t[1];
      if (a == b)
        b = t[2];
      else if (a == d)
        d = t[2];
      double pa[3], pb[3], pc[3];
      vtx(m, a, pa);
      vtx(m, b, pb);
      vtx(m, d, pc);
      for (int j = 0; j < 3; j++)
      {
        pb[j] -= pa[j];
        pc[j] -= pa[j];
      }
      normalize3(pb);
      normalize3(pc);
      double alpha = acos(dot3(pb, pc));
      awnorm[0] += alpha * m->fnorm[3 * c];
      awnorm[1] += alpha * m->fnorm[3 * c + 1];
      awnorm[2] += alpha * m->fnorm[3 * c + 2];
    }
    normalize3(awnorm);
  }
  return count;
}

/* -------------------- vtkImplicitPolyDataDistance2::ClosestPointAndNormal
 * kind: 0 face, 1 edge, 2 vertex.  awnorm starts from zero in the edge and
 * vertex cases (as in SharedEvaluate, vtkImplicitPolyDataDistance2.cxx:180).
 * Returns the signed distance; cell = -1 and the NoValue defaults when the
 * mesh has no usable triangle. */
double orc_closest_point_and_normal(const orc_mesh* m, const double x[3], int use_grid,
  double awnorm[3], double closestPoint[3], int64_t* cell_out, int* kind_out)
{
  double ret = sqrt((m->bmax[0] - m->bmin[0]) * (m->bmax[0] - m->bmin[0]) +
    (m->bmax[1] - m->bmin[1]) * (m->bmax[1] - m->bmin[1]) +
    (m->bmax[2] - m->bmin[2]) * (m->bmax[2] - m->bmin[2])); /* NoValue = GetLength() */
  double g[3] = { 0.0, 0.0, 1.0 };
  closestPoint[0] = closestPoint[1] = closestPoint[2] = 0.0;
  awnorm[0] = awnorm[1] = awnorm[2] = 0.0;
  *cell_out = -1;
  *kind_out = -1;
  if (m->nt == 0)
    return ret;
  double p[3], vlen2;
  int64_t cellId;
  if (use_grid)
    orc_find_closest_grid(m, x, p, &cellId, &vlen2);
  else
    orc_find_closest_brute(m, x, p, &cellId, &vlen2);
  if (cellId == -1)
    return ret;
  *cell_out = cellId;
  ret = sqrt(vlen2);
  for (int i = 0; i < 3; i++)
    g[i] = (p[i] - x[i]) / (ret == 0. ? 1. : ret);

  double dist2, w[3];
  eval_cell(m, cellId, p, closestPoint, &dist2, w); /* :388 — evaluated at p, not at x */
  *kind_out = pseudo_normal(m, cellId, w, awnorm);
  if (ret == 0)
    for (int i = 0; i < 3; i++)
      g[i] = awnorm[i];
  ret *= (dot3(g, awnorm) < 0.) ? 1. : -1.;
  return ret;
}

/* ---------------------- vtkImplicitPolyDataIntersection::IntersectWithLine
 * (ICP/vtkImplicitPolyDataIntersection.cxx:110-337).  The first intersection of the segment
 * p1 -> p2 with the mesh, as vtkCellLocator::IntersectWithLine with tolerance 1e-12 finds it
 * (:148), restated as: over all triangles, vtkTriangle::IntersectWithLine (plane through
 * point 1 with the unit normal of vtkTriangle::ComputeNormal, vtkPlane::IntersectWithLine
 * with its 1e-6 parallelism test, then EvaluatePosition at the plane point: a hit when
 * inside, or when within the tolerance of the triangle), smallest parameter t first, equal
 * t to the smallest cell id.  Degenerate triangles (EvaluatePosition -1) never hit.
 * Outputs as the reference leaves them: ret = NoValue (the bounds' diagonal), g = (0,0,1),
 * closestPoint = 0 without a hit; with one, closestPoint = EvaluatePosition's point at the
 * hit, g = the pseudo-normal there, and ret = |p2 - p1| t with the sign the code gives it
 * (:324: g has just been set to awnorm, so dot(g, awnorm) >= 0 and ret is negated).
 * x: the plane point itself (the reference keeps it in a local).  Returns 1 on a hit. */
#define ORC_PLANE_TOL 1.0e-06 /* VTK_PLANE_TOL */
static int tri_intersect_line(const double p1[3], const double p2[3], const double P0[3],
  const double P1[3], const double P2[3], double tol, double* t, double x[3])
{
  const double *pt1 = P1, *pt2 = P2, *pt3 = P0;
  double n[3];
  {
    /* vtkTriangle::ComputeNormal(pt1, pt2, pt3, n): direction, then unit length */
    double ax = pt3[0] - pt2[0], ay = pt3[1] - pt2[1], az = pt3[2] - pt2[2];
    double bx = pt1[0] - pt2[0], by = pt1[1] - pt2[1], bz = pt1[2] - pt2[2];
    n[0] = ay * bz - az * by;
    n[1] = az * bx - ax * bz;
    n[2] = ax * by - ay * bx;
    double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len != 0.0)
    {
      n[0] /= len;
      n[1] /= len;
      n[2] /= len;
    }
  }
  if (n[0] == 0.0 && n[1] == 0.0 && n[2] == 0.0)
    return 0;
  /* vtkPlane::IntersectWithLine(p1, p2, n, pt1, t, x) */
  double p21[3] = { p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2] };
  double num = dot3(n, pt1) - (n[0] * p1[0] + n[1] * p1[1] + n[2] * p1[2]);
  double den = n[0] * p21[0] + n[1] * p21[1] + n[2] * p21[2];
  double fabsden = den < 0.0 ? -den : den;
  double fabstol = num < 0.0 ? -num * ORC_PLANE_TOL : num * ORC_PLANE_TOL;
  if (fabsden <= fabstol)
    return 0;
  *t = num / den;
  x[0] = p1[0] + (*t) * p21[0];
  x[1] = p1[1] + (*t) * p21[1];
  x[2] = p1[2] + (*t) * p21[2];
  if (!(*t >= 0.0 && *t <= 1.0))
    return 0;
  double cp[3], d2, w[3];
  int inside = orc_tri_eval(x, P0, P1, P2, cp, &d2, w);
  if (inside < 0)
    return 0;
  if (d2 <= tol * tol)
    return 1;
  return inside;
}

int orc_intersect_line(const orc_mesh* m, const double p1[3], const double p2[3], double* ret,
  double closestPoint[3], double g[3], double x_out[3], int64_t* cell_out, int* kind_out)
{
  *ret = sqrt((m->bmax[0] - m->bmin[0]) * (m->bmax[0] - m->bmin[0]) +
    (m->bmax[1] - m->bmin[1]) * (m->bmax[1] - m->bmin[1]) +
    (m->bmax[2] - m->bmin[2]) * (m->bmax[2] - m->bmin[2])); /* NoValue = GetLength() */
  g[0] = 0.0, g[1] = 0.0, g[2] = 1.0;
  closestPoint[0] = closestPoint[1] = closestPoint[2] = 0.0;
  x_out[0] = x_out[1] = x_out[2] = 0.0;
  *cell_out = -1;
  *kind_out = -1;
  double tbest = INFINITY, xb[3] = { 0.0, 0.0, 0.0 };
  int64_t cbest = -1;
  for (int64_t c = 0; c < m->nt; c++)
  {
    double a[3], b[3], d[3], t, x[3];
    vtx(m, m->tri[3 * c], a);
    vtx(m, m->tri[3 * c + 1], b);
    vtx(m, m->tri[3 * c + 2], d);
    if (!tri_intersect_line(p1, p2, a, b, d, ORC_TOL, &t, x))
      continue;
    if (t < tbest) /* ascending cell id: the first of equal t stays */
    {
      tbest = t;
      cbest = c;
      xb[0] = x[0], xb[1] = x[1], xb[2] = x[2];
    }
  }
  if (cbest < 0)
    return 0;
  *ret = sqrt(dist2pts(p1, p2)) * tbest; /* :155 */
  *cell_out = cbest;
  x_out[0] = xb[0], x_out[1] = xb[1], x_out[2] = xb[2];
  double dist2, w[3], awnorm[3] = { 0.0, 0.0, 0.0 };
  eval_cell(m, cbest, xb, closestPoint, &dist2, w); /* :169 */
  *kind_out = pseudo_normal(m, cbest, w, awnorm);
  for (int i = 0; i < 3; i++)
    g[i] = awnorm[i]; /* :317-320 */
  *ret *= (dot3(g, awnorm) < 0.) ? 1. : -1.; /* :324 */
  return 1;
}

/* n segments at once (thread-parallel); arrays of 3n doubles, hit / cell / kind per segment */
void orc_intersect_lines(const orc_mesh* m, const double* p1, const double* p2, int64_t n,
  int nthreads, int32_t* hit, double* ret, double* closest, double* g, double* x, int32_t* cell,
  int32_t* kind)
{
#ifdef _OPENMP
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
  for (int64_t i = 0; i < n; i++)
  {
    int64_t c;
    int k;
    hit[i] = orc_intersect_line(m, p1 + 3 * i, p2 + 3 * i, &ret[i], closest + 3 * i, g + 3 * i,
      x + 3 * i, &c, &k);
    cell[i] = (int32_t)c;
    kind[i] = k;
  }
}

/* ------------------------------- vtkPolyDataCorrespondenceFilter functors
 * Intended semantics where the code as written is undefined: the emitted
 * landmark normal is the locator's normal (.cxx:126-132 shadows it), and the
 * symmetric path's compatibility test dots the emitted landmark normal with
 * the source normal against MinNormalDot (.cxx:286-289).  Rows come out in
 * index order.  Outputs hold ceil(np/step) rows at most. */
static inline void xform(const double T[16], const double in[3], double out[3])
{
  double x = in[0], y = in[1], z = in[2];
  out[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
  out[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
  out[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

int64_t orc_correspond(const orc_mesh* m, const float* src, int64_t np, const float* src_normals,
  const double T[16], int64_t step, double max_dist, double min_dot, int use_grid, int nthreads,
  float* origin, float* origin_normal, float* closest, float* normal, int32_t* cell, double* dist,
  int64_t* src_index, int32_t* kind, int rotate_sn)
{
  if (step < 1)
    step = 1;
  int64_t nk = (np + step - 1) / step;
  char* ok = (char*)calloc((size_t)nk + 1, 1);
  float* tmp = (float*)malloc(sizeof(float) * 12 * (size_t)(nk + 1));
  int32_t* tc = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(nk + 1));
  double* td = (double*)malloc(sizeof(double) * (size_t)(nk + 1));
#ifdef _OPENMP
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
  for (int64_t k = 0; k < nk; k++)
  {
    int64_t i = k * step;
    double x[3] = { (double)src[3 * i], (double)src[3 * i + 1], (double)src[3 * i + 2] };
    double n[3], cp[3];
    int64_t c;
    int kd;
    xform(T, x, x);
    double sd = orc_closest_point_and_normal(m, x, use_grid, n, cp, &c, &kd);
    float nf[3] = { (float)n[0], (float)n[1], (float)n[2] };
    int accept = fabs(sd) < max_dist;
    float sf[3] = { 0.f, 0.f, 0.f };
    if (src_normals)
    {
      /* as written the reference uses the source normal unrotated (.cxx:284,299);
       * rotate_sn is the opt-in corrected mode: the normal of the moved point, T's 3x3
       * applied in fp64 and rounded to float */
      double sx = src_normals[3 * i], sy = src_normals[3 * i + 1], sz = src_normals[3 * i + 2];
      if (rotate_sn)
      {
        sf[0] = (float)(T[0] * sx + T[1] * sy + T[2] * sz);
        sf[1] = (float)(T[4] * sx + T[5] * sy + T[6] * sz);
        sf[2] = (float)(T[8] * sx + T[9] * sy + T[10] * sz);
      }
      else
      {
        sf[0] = (float)sx;
        sf[1] = (float)sy;
        sf[2] = (float)sz;
      }
    }
    if (accept && src_normals)
    {
      double dn = (double)nf[0] * (double)sf[0] + (double)nf[1] * (double)sf[1] +
        (double)nf[2] * (double)sf[2];
      accept = dn > min_dot;
    }
    if (c < 0)
      accept = 0;
    ok[k] = (char)accept;
    for (int a = 0; a < 3; a++)
    {
      tmp[12 * k + a] = (float)x[a];
      tmp[12 * k + 3 + a] = (float)cp[a];
      tmp[12 * k + 6 + a] = nf[a];
      tmp[12 * k + 9 + a] = sf[a];
    }
    tc[2 * k] = (int32_t)c;
    tc[2 * k + 1] = kd;
    td[k] = fabs(sd);
  }
  int64_t n = 0;
  for (int64_t k = 0; k < nk; k++)
  {
    if (!ok[k])
      continue;
    for (int a = 0; a < 3; a++)
    {
      origin[3 * n + a] = tmp[12 * k + a];
      closest[3 * n + a] = tmp[12 * k + 3 + a];
      normal[3 * n + a] = tmp[12 * k + 6 + a];
      if (origin_normal && src_normals)
        origin_normal[3 * n + a] = tmp[12 * k + 9 + a];
    }
    if (cell)
      cell[n] = tc[2 * k];
    if (kind)
      kind[n] = tc[2 * k + 1];
    if (dist)
      dist[n] = td[k];
    if (src_index)
      src_index[n] = k * step;
    n++;
  }
  free(ok);
  free(tmp);
  free(tc);
  free(td);
  return n;
}

/* The filter with OutputPrecision = DOUBLE_PRECISION
 * (ICP/vtkPolyDataCorrespondenceFilter.cxx:519-531): Functor with double outputs, i.e. the
 * same loop (:87-137) storing the transformed point, the closest point and the locator's
 * normal unrounded; no origin normals and no normal test on this path. */
int64_t orc_correspond_f64(const orc_mesh* m, const float* src, int64_t np, const double T[16],
  int64_t step, double max_dist, int use_grid, int nthreads, double* origin, double* closest,
  double* normal, int32_t* cell, double* dist, int64_t* src_index)
{
  if (step < 1)
    step = 1;
  int64_t nk = (np + step - 1) / step;
  char* ok = (char*)calloc((size_t)nk + 1, 1);
  double* tmp = (double*)malloc(sizeof(double) * 10 * (size_t)(nk + 1));
  int32_t* tc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nk + 1));
#ifdef _OPENMP
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
  for (int64_t k = 0; k < nk; k++)
  {
    int64_t i = k * step;
    double x[3] = { (double)src[3 * i], (double)src[3 * i + 1], (double)src[3 * i + 2] };
    double n[3], cp[3];
    int64_t c;
    int kd;
    xform(T, x, x);
    double sd = orc_closest_point_and_normal(m, x, use_grid, n, cp, &c, &kd);
    ok[k] = (char)(fabs(sd) < max_dist && c >= 0);
    for (int a = 0; a < 3; a++)
    {
      tmp[10 * k + a] = x[a];
      tmp[10 * k + 3 + a] = cp[a];
      tmp[10 * k + 6 + a] = n[a];
    }
    tmp[10 * k + 9] = fabs(sd);
    tc[k] = (int32_t)c;
  }
  int64_t n = 0;
  for (int64_t k = 0; k < nk; k++)
  {
    if (!ok[k])
      continue;
    for (int a = 0; a < 3; a++)
    {
      origin[3 * n + a] = tmp[10 * k + a];
      closest[3 * n + a] = tmp[10 * k + 3 + a];
      normal[3 * n + a] = tmp[10 * k + 6 + a];
    }
    if (cell)
      cell[n] = tc[k];
    if (dist)
      dist[n] = tmp[10 * k + 9];
    if (src_index)
      src_index[n] = k * step;
    n++;
  }
  free(ok);
  free(tmp);
  free(tc);
  return n;
}

/* ------------------------------------------------------------------ solvers */
#define REAL double
#define SUF(x) x##_f64
#include "solvers_impl.h"
#undef REAL
#undef SUF
#define REAL float
#define SUF(x) x##_f32
#include "solvers_impl.h"
#undef REAL
#undef SUF

/* metric: 0 point-to-point, 1 point-to-plane, 2 symmetric ("plane-to-plane").
 * precision: 64 = fp64 restatement (the parity gate), 32 = fp32 sequential
 * restatement of the reference's float solver.  M row-major 4x4.
 * Returns 0, or 1 when there are too few rows (M = identity). */
int orc_solve(int metric, int precision, int64_t n, const float* s, const float* t,
  const float* ns, const float* nt, double M[16])
{
  for (int i = 0; i < 16; i++)
    M[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (n < (metric == 0 ? 3 : 6))
    return 1;
  if (precision == 32)
  {
    float Mf[16];
    if (metric == 0)
      solve_p2p_f32(n, s, t, Mf);
    else if (metric == 1)
      solve_p2plane_f32(n, s, t, nt, Mf);
    else
      solve_sym_f32(n, s, t, ns, nt, Mf);
    for (int i = 0; i < 16; i++)
      M[i] = Mf[i];
  }
  else
  {
    if (metric == 0)
      solve_p2p_f64(n, s, t, M);
    else if (metric == 1)
      solve_p2plane_f64(n, s, t, nt, M);
    else
      solve_sym_f64(n, s, t, ns, nt, M);
  }
  return 0;
}

/* ------------------------------------------------- vtkICP::InternalUpdate
 * status: 0 ran to MaximumNumberOfIterations, 1 stopped on mean distance,
 * 2 stopped because too few correspondences were left (the reference has no
 * handling for that case; stopping with the current transform is this
 * restatement's choice). */
int orc_icp_run(const orc_mesh* m, const float* src, int64_t np, const float* src_normals,
  int metric, int max_iter, int64_t max_landmarks, double max_dist, int check_mean,
  double max_mean, double min_dot, int precision, int use_grid, int nthreads, double T[16],
  int* iters, double* mean_dist, int64_t* last_n, int rotate_sn)
{
  for (int i = 0; i < 16; i++)
    T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  int64_t step = 1;
  if (np > max_landmarks)
    step = np / max_landmarks; /* vtkPolyDataCorrespondenceFilter.cxx:473-477 */
  int64_t nk = (np + step - 1) / step;
  float* o = (float*)malloc(sizeof(float) * 3 * (size_t)(nk + 1));
  float* on = (float*)malloc(sizeof(float) * 3 * (size_t)(nk + 1));
  float* l = (float*)malloc(sizeof(float) * 3 * (size_t)(nk + 1));
  float* ln = (float*)malloc(sizeof(float) * 3 * (size_t)(nk + 1));
  int status = 0;
  *iters = 0;
  *mean_dist = 0.0;
  *last_n = 0;
  const float* sn = (metric == 2) ? src_normals : NULL;
  for (int it = 0; it < max_iter; it++)
  {
    int64_t n = orc_correspond(m, src, np, sn, T, step, max_dist, min_dot, use_grid, nthreads, o,
      on, l, ln, NULL, NULL, NULL, NULL, rotate_sn);
    *last_n = n;
    if (check_mean)
    {
      if (precision == 32)
      {
        float acc = 0.f;
        for (int64_t i = 0; i < n; i++)
        {
          float dx = o[3 * i] - l[3 * i], dy = o[3 * i + 1] - l[3 * i + 1],
                dz = o[3 * i + 2] - l[3 * i + 2];
          acc += sqrtf(dx * dx + dy * dy + dz * dz);
        }
        *mean_dist = n ? acc / (float)n : 0.0;
      }
      else
      {
        double acc = 0.0;
        for (int64_t i = 0; i < n; i++)
        {
          double dx = (double)o[3 * i] - l[3 * i], dy = (double)o[3 * i + 1] - l[3 * i + 1],
                 dz = (double)o[3 * i + 2] - l[3 * i + 2];
          acc += sqrt(dx * dx + dy * dy + dz * dz);
        }
        *mean_dist = n ? acc / (double)n : 0.0;
      }
      if (n > 0 && *mean_dist <= max_mean)
      {
        status = 1;
        break;
      }
    }
    double M[16];
    if (orc_solve(metric, precision, n, o, l, on, ln, M))
    {
      status = 2;
      break;
    }
    (*iters)++;
    /* T <- M * T  (vtkTransform PostMultiply + Concatenate, vtkICP.cxx:169,240) */
    double R[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++)
      {
        double acc = 0.0;
        for (int k = 0; k < 4; k++)
          acc += M[4 * r + k] * T[4 * k + c];
        R[4 * r + c] = acc;
      }
    memcpy(T, R, sizeof(R));
  }
  free(o);
  free(on);
  free(l);
  free(ln);
  return status;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
