// solve_core.hpp — one ICP iteration's O(1) arithmetic: from the (all-reduced) raw
// moments to the incremental 4x4, in fp64.  The same code runs on the host
// (icpx_solve, solve.cpp) and in a single GPU thread (k_solve, icpx.hip), so a loop
// whose transform never leaves the device takes the same steps as one solved on the host.
//
//   point-to-point   ICP/icp/point2point.cxx:25-66   (means, 3x3 covariance, SVD, reflection fix)
//   point-to-plane   ICP/icp/point2plane.cxx:37-110  (6x6 normal equations about the source
//                                                     mean, LDL^T, R = Rx*Ry*Rz)
//   symmetric        ICP/icp/plane2plane.cxx:37-94   (6x6 system about both means, LDL^T,
//                                                     Rodrigues half-rotation applied twice)
//
// The reference centres the clouds first and then loops over the rows; here
// the rows were summed once, un-centred, on the GPU (about a fixed origin), and
// the centred blocks are recovered algebraically: with a = u x n, c = (u-m) x n
// = a - [m]x n, so  sum c c^T = A - B [m]x^T - [m]x B^T + [m]x N [m]x^T  etc.
#pragma once
#include "../../include/icpx.h"
#include <math.h>

#if defined(__HIPCC__)
#define ICPX_HD __host__ __device__
#else
#define ICPX_HD
#endif
namespace icpx_core
{
ICPX_HD inline void swap_d(double& a, double& b)
{
  const double t = a;
  a = b;
  b = t;
}

struct M3
{
  double v[3][3];
};
struct V3
{
  double v[3];
};

ICPX_HD inline M3 zero3()
{
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.v[i][j] = 0.0;
  return r;
}
ICPX_HD inline M3 mul(const M3& a, const M3& b)
{
  M3 r = zero3();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++)
        r.v[i][j] += a.v[i][k] * b.v[k][j];
  return r;
}
ICPX_HD inline M3 tr(const M3& a)
{
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.v[i][j] = a.v[j][i];
  return r;
}
ICPX_HD inline V3 mulv(const M3& a, const V3& x)
{
  V3 r;
  for (int i = 0; i < 3; i++)
    r.v[i] = a.v[i][0] * x.v[0] + a.v[i][1] * x.v[1] + a.v[i][2] * x.v[2];
  return r;
}
ICPX_HD inline M3 skew(const V3& m)
{
  M3 r = zero3();
  r.v[0][1] = -m.v[2];
  r.v[0][2] = m.v[1];
  r.v[1][0] = m.v[2];
  r.v[1][2] = -m.v[0];
  r.v[2][0] = -m.v[1];
  r.v[2][1] = m.v[0];
  return r;
}
ICPX_HD inline M3 sym6(const double* s) // xx xy xz yy yz zz
{
  M3 r;
  r.v[0][0] = s[0];
  r.v[0][1] = r.v[1][0] = s[1];
  r.v[0][2] = r.v[2][0] = s[2];
  r.v[1][1] = s[3];
  r.v[1][2] = r.v[2][1] = s[4];
  r.v[2][2] = s[5];
  return r;
}
ICPX_HD inline M3 full9(const double* s)
{
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.v[i][j] = s[3 * i + j];
  return r;
}
ICPX_HD inline V3 cross(const V3& a, const V3& b)
{
  V3 r = { { a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2],
    a.v[0] * b.v[1] - a.v[1] * b.v[0] } };
  return r;
}
ICPX_HD inline double norm(const V3& a)
{
  return sqrt(a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2]);
}

// Rotation V * diag(1,1,det(U)det(V)) * U^T from S = U diag(s) V^T, by Hestenes
// (one-sided) Jacobi on the columns of S.  The third columns are rebuilt as the
// cross products of the first two, which folds the reflection fix of
// point2point.cxx:46-56 into the construction (u0 x u1 = det(U) u2).
ICPX_HD inline M3 kabsch_rotation(const M3& S)
{
  double B[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      B[i][j] = S.v[i][j];
  for (int sweep = 0; sweep < 64; sweep++)
  {
    double worst = 0.0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++)
      {
        double app = 0, aqq = 0, apq = 0;
        for (int r = 0; r < 3; r++)
        {
          app += B[r][p] * B[r][p];
          aqq += B[r][q] * B[r][q];
          apq += B[r][p] * B[r][q];
        }
        if (apq == 0.0 || app == 0.0 || aqq == 0.0)
          continue;
        worst = fmax(worst, fabs(apq) / sqrt(app * aqq));
        double tau = (aqq - app) / (2.0 * apq);
        double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; r++)
        {
          double bp = B[r][p], bq = B[r][q];
          B[r][p] = c * bp - s * bq;
          B[r][q] = s * bp + c * bq;
          double vp = V[r][p], vq = V[r][q];
          V[r][p] = c * vp - s * vq;
          V[r][q] = s * vp + c * vq;
        }
      }
    if (worst < 1e-15)
      break;
  }
  double len[3];
  int ord[3] = { 0, 1, 2 };
  for (int c = 0; c < 3; c++)
    len[c] = sqrt(B[0][c] * B[0][c] + B[1][c] * B[1][c] + B[2][c] * B[2][c]);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (len[ord[j]] < len[ord[j + 1]])
      {
        int t = ord[j];
        ord[j] = ord[j + 1];
        ord[j + 1] = t;
      }
  M3 I = zero3();
  I.v[0][0] = I.v[1][1] = I.v[2][2] = 1.0;
  if (len[ord[0]] <= 0.0)
    return I;
  V3 u[3], v[3];
  for (int c = 0; c < 2; c++)
    for (int r = 0; r < 3; r++)
    {
      v[c].v[r] = V[r][ord[c]];
      u[c].v[r] = len[ord[c]] > 0.0 ? B[r][ord[c]] / len[ord[c]] : 0.0;
    }
  if (len[ord[1]] <= len[ord[0]] * 1e-14)
  {
    // rank one: any unit vector orthogonal to u0
    int k = 0;
    for (int r = 1; r < 3; r++)
      if (fabs(u[0].v[r]) < fabs(u[0].v[k]))
        k = r;
    V3 e = { { 0, 0, 0 } };
    e.v[k] = 1.0;
    double d = u[0].v[k];
    for (int r = 0; r < 3; r++)
      u[1].v[r] = e.v[r] - d * u[0].v[r];
    double n = norm(u[1]);
    for (int r = 0; r < 3; r++)
      u[1].v[r] /= n;
  }
  u[2] = cross(u[0], u[1]);
  v[2] = cross(v[0], v[1]);
  M3 R = zero3();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 3; k++)
        R.v[r][c] += v[k].v[r] * u[k].v[c];
  return R;
}

// Symmetric 6x6 solve by LDL^T with diagonal pivoting; a zero pivot leaves
// that component at zero (the behaviour of Eigen::LDLT::solve the reference relies on).
ICPX_HD inline void solve6(const double Ain[6][6], const double bin[6], double x[6])
{
  // Every loop has a compile-time trip count and every index is a loop counter: unrolled,
  // the work arrays stay in registers on the device (which runs this in a single thread; a
  // run-time index, as pivoting by position would need, sends them to memory and makes
  // every access a round trip).  The pivot's row and column are exchanged by testing each
  // candidate position in turn.
  double A[6][6], y[6], col[6];
  int perm[6];
#pragma unroll
  for (int i = 0; i < 6; i++)
  {
#pragma unroll
    for (int j = 0; j < 6; j++)
      A[i][j] = Ain[i][j];
    y[i] = bin[i];
    perm[i] = i;
    col[i] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; k++)
  {
    int piv = k;
    double big = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; i++)
      if (fabs(A[i][i]) > big)
      {
        big = fabs(A[i][i]);
        piv = i;
      }
#pragma unroll
    for (int i = k + 1; i < 6; i++)
      if (piv == i)
      {
#pragma unroll
        for (int j = 0; j < 6; j++)
          swap_d(A[k][j], A[i][j]);
#pragma unroll
        for (int j = 0; j < 6; j++)
          swap_d(A[j][k], A[j][i]);
        swap_d(y[k], y[i]); // (the right-hand side goes along: y[i] = bin[perm[i]])
        const int tp = perm[k];
        perm[k] = perm[i];
        perm[i] = tp;
      }
    const double d = A[k][k];
    if (d == 0.0)
    {
#pragma unroll
      for (int i = k + 1; i < 6; i++)
        A[i][k] = A[k][i] = 0.0;
      continue;
    }
#pragma unroll
    for (int i = k + 1; i < 6; i++)
      col[i] = A[i][k];
#pragma unroll
    for (int i = k + 1; i < 6; i++)
    {
      const double li = col[i] / d; // one division per row
#pragma unroll
      for (int j = k + 1; j <= i; j++)
      {
        A[i][j] -= li * col[j];
        A[j][i] = A[i][j];
      }
      A[i][k] = li;
      A[k][i] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++)
  {
#pragma unroll
    for (int j = 0; j < i; j++)
      y[i] -= A[i][j] * y[j];
  }
#pragma unroll
  for (int i = 0; i < 6; i++)
    y[i] = (A[i][i] != 0.0) ? y[i] / A[i][i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; i--)
  {
#pragma unroll
    for (int j = i + 1; j < 6; j++)
      y[i] -= A[j][i] * y[j];
  }
  // x[perm[i]] = y[i]
#pragma unroll
  for (int o = 0; o < 6; o++)
  {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++)
      v = (perm[i] == o) ? y[i] : v;
    x[o] = v;
  }
}

// y = R x + t in the frame shifted by `origin`  ->  row-major 4x4 in the caller's frame.
ICPX_HD inline void emit(const M3& R, const V3& t, const double origin[3], double M[16])
{
  V3 o = { { origin[0], origin[1], origin[2] } };
  V3 Ro = mulv(R, o);
  for (int r = 0; r < 3; r++)
  {
    for (int c = 0; c < 3; c++)
      M[4 * r + c] = R.v[r][c];
    M[4 * r + 3] = t.v[r] + o.v[r] - Ro.v[r];
  }
  M[12] = M[13] = M[14] = 0.0;
  M[15] = 1.0;
}

// Centred 6x6 system from raw moments about mean vector m (see file header).
// A = sum a a^T, B = sum a n^T, N = sum n n^T, ra = sum r a, rn = sum r n;
// the residual is b = r - dbar.n (dbar = 0 for point-to-plane).
ICPX_HD inline void centred_system(const M3& A, const M3& B, const M3& N, const V3& ra, const V3& rn,
  const V3& m, const V3& dbar, double L[6][6], double rhs[6])
{
  const M3 K = skew(m);
  const M3 KN = mul(K, N);
  const M3 KB = mul(K, tr(B));
  const M3 KNK = mul(KN, tr(K));
  const V3 Krn = mulv(K, rn), Bd = mulv(B, dbar), KNd = mulv(KN, dbar), Nd = mulv(N, dbar);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
    {
      L[i][j] = A.v[i][j] - KB.v[j][i] - KB.v[i][j] + KNK.v[i][j]; // sum c c^T
      L[i][3 + j] = B.v[i][j] - KN.v[i][j];                        // sum c n^T
      L[3 + j][i] = L[i][3 + j];
      L[3 + i][3 + j] = N.v[i][j];
    }
  for (int i = 0; i < 3; i++)
  {
    rhs[i] = ra.v[i] - Krn.v[i] - Bd.v[i] + KNd.v[i]; // sum b c
    rhs[3 + i] = rn.v[i] - Nd.v[i];                   // sum b n
  }
}


// Whether a row-major 4x4 can be handed to the search kernels: every entry a number, the 3x3
// within 1e6 and the translation within 1e18 in magnitude (source coordinates are held to
// 1e12, so a moved point and the squares of its distances stay finite in single precision;
// the kernels' pruning bounds assume that).  A solve of sums that carry a NaN (normals that
// are not numbers) or of a system singular to within rounding does not pass.
ICPX_HD inline bool transform_usable(const double* T)
{
  bool ok = true;
  for (int r = 0; r < 3; r++)
  {
    for (int k = 0; k < 3; k++)
      ok = ok && fabs(T[4 * r + k]) <= 1e6;
    ok = ok && fabs(T[4 * r + 3]) <= 1e18;
  }
  return ok; // (a NaN fails every comparison)
}

// Returns ICPX_OK, ICPX_E_FEW (M = identity) or ICPX_E_INVALID; n and the mean distance
// |s - t| are reported in every case.
ICPX_HD inline int solve_moments(int metric, const double* sums, const double* origin, double* M,
  double* n_out, double* mean_dist_out)
{
  for (int i = 0; i < 16; i++)
    M[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (metric < 0 || metric > 2 || !sums || !origin)
    return ICPX_E_INVALID;
  const double n = sums[0];
  if (n_out)
    *n_out = n;
  if (mean_dist_out)
    *mean_dist_out = n > 0 ? sums[1] / n : 0.0;
  if (n < (metric == ICPX_METRIC_POINT_TO_POINT ? 3.0 : 6.0))
    return ICPX_E_FEW;

  if (metric == ICPX_METRIC_POINT_TO_POINT)
  {
    V3 sm, tm;
    for (int i = 0; i < 3; i++)
    {
      sm.v[i] = sums[2 + i] / n;
      tm.v[i] = sums[5 + i] / n;
    }
    M3 S;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        S.v[i][j] = (sums[8 + 3 * i + j] - sums[2 + i] * sums[5 + j] / n) / n;
    const M3 R = kabsch_rotation(S);
    const V3 Rs = mulv(R, sm);
    V3 t = { { tm.v[0] - Rs.v[0], tm.v[1] - Rs.v[1], tm.v[2] - Rs.v[2] } };
    emit(R, t, origin, M);
    return ICPX_OK;
  }

  double L[6][6], rhs[6], x[6];
  if (metric == ICPX_METRIC_POINT_TO_PLANE)
  {
    V3 sm = { { sums[2] / n, sums[3] / n, sums[4] / n } };
    V3 z = { { 0, 0, 0 } };
    V3 ra = { { sums[26], sums[27], sums[28] } }, rn = { { sums[29], sums[30], sums[31] } };
    centred_system(sym6(sums + 5), full9(sums + 11), sym6(sums + 20), ra, rn, sm, z, L, rhs);
    for (int i = 0; i < 6; i++)
      rhs[i] = -rhs[i]; // distToPlane = -((s - t).n), point2plane.cxx:80-83
    solve6(L, rhs, x);
    const double ca = cos(x[0]), sa = sin(x[0]);
    const double cb = cos(x[1]), sb = sin(x[1]);
    const double cg = cos(x[2]), sg = sin(x[2]);
    M3 R = { { { cb * cg, -cb * sg, sb },
      { ca * sg + sa * sb * cg, ca * cg - sa * sb * sg, -sa * cb },
      { sa * sg - ca * sb * cg, sa * cg + ca * sb * sg, ca * cb } } };
    const V3 Rs = mulv(R, sm);
    V3 t = { { x[3] + sm.v[0] - Rs.v[0], x[4] + sm.v[1] - Rs.v[1], x[5] + sm.v[2] - Rs.v[2] } };
    emit(R, t, origin, M);
    return ICPX_OK;
  }

  // symmetric
  V3 pm = { { sums[2] / n, sums[3] / n, sums[4] / n } };
  V3 qm = { { sums[5] / n, sums[6] / n, sums[7] / n } };
  V3 um = { { pm.v[0] + qm.v[0], pm.v[1] + qm.v[1], pm.v[2] + qm.v[2] } };
  V3 dm = { { pm.v[0] - qm.v[0], pm.v[1] - qm.v[1], pm.v[2] - qm.v[2] } };
  V3 ra = { { sums[29], sums[30], sums[31] } }, rn = { { sums[32], sums[33], sums[34] } };
  centred_system(sym6(sums + 8), full9(sums + 14), sym6(sums + 23), ra, rn, um, dm, L, rhs);
  for (int i = 0; i < 6; i++)
    rhs[i] = -rhs[i]; // plane2plane.cxx:70
  solve6(L, rhs, x);
  V3 at = { { x[0], x[1], x[2] } };
  const double an = norm(at);
  const double theta = atan(an);
  if (an > 0.0)
    for (int i = 0; i < 3; i++)
      at.v[i] /= an;
  const M3 W = skew(at);
  const M3 W2 = mul(W, W);
  const double st = sin(theta), ct = cos(theta);
  M3 Ri;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Ri.v[i][j] = (i == j ? 1.0 : 0.0) + st * W.v[i][j] + (1.0 - ct) * W2.v[i][j];
  const M3 R = mul(Ri, Ri);
  V3 t2 = { { ct * x[3], ct * x[4], ct * x[5] } };
  const V3 Rp = mulv(R, pm), Rt = mulv(Ri, t2);
  V3 t = { { -Rp.v[0] + Rt.v[0] + qm.v[0], -Rp.v[1] + Rt.v[1] + qm.v[1],
    -Rp.v[2] + Rt.v[2] + qm.v[2] } };
  emit(R, t, origin, M);
  return ICPX_OK;
}
} // namespace icpx_core
