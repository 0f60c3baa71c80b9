/*
 * solvers_impl.h — the three per-iteration solvers, written once over REAL and
 * included twice by icp_oracle.c (REAL=double: the fp64 restatement that the
 * parity gate uses; REAL=float: a sequential fp32 restatement of the
 * reference's float solver, used to measure the reference's own rounding noise).
 *
 * TEST INFRASTRUCTURE ONLY (see icp_oracle.c).
 *
 * Follows ICP/icp/point2point.cxx:25-66, ICP/icp/point2plane.cxx:37-110 and
 * ICP/icp/plane2plane.cxx:37-94.  Eigen's JacobiSVD and LDLT are not
 * available; a one-sided Jacobi SVD and a diagonally pivoted LDL^T stand in
 * for them (same factorisations, own code).
 */

/* A (row-major 3x3) = U diag(S) V^T, S descending, U and V orthogonal. */
static void SUF(svd3)(const REAL A[9], REAL U[9], REAL S[3], REAL V[9])
{
  REAL B[9];
  for (int i = 0; i < 9; i++)
  {
    B[i] = A[i];
    V[i] = (i % 4 == 0) ? (REAL)1 : (REAL)0;
  }
  for (int sweep = 0; sweep < 60; sweep++)
  {
    REAL off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++)
      {
        REAL al = 0, be = 0, ga = 0;
        for (int r = 0; r < 3; r++)
        {
          al += B[3 * r + p] * B[3 * r + p];
          be += B[3 * r + q] * B[3 * r + q];
          ga += B[3 * r + p] * B[3 * r + q];
        }
        if (ga == 0)
          continue;
        REAL lim = (REAL)sqrt((double)al * (double)be);
        if ((REAL)fabs((double)ga) <= (REAL)1e-300 || lim == 0)
          continue;
        REAL rel = (REAL)fabs((double)ga) / lim;
        if (rel > off)
          off = rel;
        REAL zeta = (be - al) / ((REAL)2 * ga);
        REAL tt = (REAL)((zeta >= 0 ? 1.0 : -1.0) /
          (fabs((double)zeta) + sqrt(1.0 + (double)zeta * (double)zeta)));
        REAL c = (REAL)(1.0 / sqrt(1.0 + (double)tt * (double)tt));
        REAL sn = c * tt;
        for (int r = 0; r < 3; r++)
        {
          REAL bp = B[3 * r + p], bq = B[3 * r + q];
          B[3 * r + p] = c * bp - sn * bq;
          B[3 * r + q] = sn * bp + c * bq;
          REAL vp = V[3 * r + p], vq = V[3 * r + q];
          V[3 * r + p] = c * vp - sn * vq;
          V[3 * r + q] = sn * vp + c * vq;
        }
      }
    if (off < (sizeof(REAL) == 8 ? (REAL)1e-15 : (REAL)1e-7))
      break;
  }
  REAL nrm[3];
  for (int c = 0; c < 3; c++)
    nrm[c] = (REAL)sqrt((double)(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]));
  int ord[3] = { 0, 1, 2 };
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (nrm[ord[j]] < nrm[ord[j + 1]])
      {
        int tmp = ord[j];
        ord[j] = ord[j + 1];
        ord[j + 1] = tmp;
      }
  REAL Vs[9];
  for (int c = 0; c < 3; c++)
  {
    S[c] = nrm[ord[c]];
    for (int r = 0; r < 3; r++)
    {
      Vs[3 * r + c] = V[3 * r + ord[c]];
      U[3 * r + c] = (S[c] > 0) ? B[3 * r + ord[c]] / S[c] : (REAL)0;
    }
  }
  for (int i = 0; i < 9; i++)
    V[i] = Vs[i];
  /* complete U where singular values vanish (relative to the largest) */
  REAL tiny = S[0] * (sizeof(REAL) == 8 ? (REAL)1e-14 : (REAL)1e-6);
  if (S[0] <= 0)
  {
    for (int i = 0; i < 9; i++)
      U[i] = (i % 4 == 0) ? (REAL)1 : (REAL)0;
    return;
  }
  if (S[1] <= tiny)
  {
    /* any unit vector orthogonal to column 0 */
    REAL u0[3] = { U[0], U[3], U[6] };
    int k = 0;
    if (fabs((double)u0[1]) < fabs((double)u0[k]))
      k = 1;
    if (fabs((double)u0[2]) < fabs((double)u0[k]))
      k = 2;
    REAL e[3] = { 0, 0, 0 };
    e[k] = 1;
    REAL d = u0[k];
    REAL v[3] = { e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2] };
    REAL nv = (REAL)sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    U[1] = v[0] / nv;
    U[4] = v[1] / nv;
    U[7] = v[2] / nv;
  }
  if (S[2] <= tiny)
  {
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}

static REAL SUF(det3)(const REAL A[9])
{
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
    A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* Solve the symmetric 6x6 system A x = b by LDL^T with diagonal pivoting;
 * a vanishing pivot zeroes that component (what Eigen's LDLT::solve does). */
static void SUF(ldlt6)(const REAL Ain[36], const REAL bin[6], REAL x[6])
{
  REAL A[36], b[6];
  int perm[6];
  for (int i = 0; i < 36; i++)
    A[i] = Ain[i];
  for (int i = 0; i < 6; i++)
  {
    b[i] = bin[i];
    perm[i] = i;
  }
  for (int k = 0; k < 6; k++)
  {
    int piv = k;
    REAL best = (REAL)fabs((double)A[7 * k]);
    for (int i = k + 1; i < 6; i++)
      if ((REAL)fabs((double)A[7 * i]) > best)
      {
        best = (REAL)fabs((double)A[7 * i]);
        piv = i;
      }
    if (piv != k)
    {
      for (int j = 0; j < 6; j++)
      {
        REAL t = A[6 * k + j];
        A[6 * k + j] = A[6 * piv + j];
        A[6 * piv + j] = t;
      }
      for (int j = 0; j < 6; j++)
      {
        REAL t = A[6 * j + k];
        A[6 * j + k] = A[6 * j + piv];
        A[6 * j + piv] = t;
      }
      int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
    }
    REAL d = A[7 * k];
    if (d == 0)
    {
      for (int i = k + 1; i < 6; i++)
        A[6 * i + k] = A[6 * k + i] = 0;
      continue;
    }
    REAL col[6];
    for (int i = k + 1; i < 6; i++)
      col[i] = A[6 * i + k];
    for (int i = k + 1; i < 6; i++)
    {
      REAL lik = col[i] / d;
      for (int j = k + 1; j <= i; j++)
      {
        A[6 * i + j] -= lik * col[j];
        A[6 * j + i] = A[6 * i + j];
      }
    }
    for (int i = k + 1; i < 6; i++)
    {
      A[6 * i + k] = col[i] / d;
      A[6 * k + i] = 0;
    }
  }
  /* now the strict lower part of A holds L, the diagonal holds D */
  REAL y[6];
  for (int i = 0; i < 6; i++)
    y[i] = b[perm[i]];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++)
      y[i] -= A[6 * i + j] * y[j];
  for (int i = 0; i < 6; i++)
    y[i] = (A[7 * i] != 0) ? y[i] / A[7 * i] : (REAL)0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++)
      y[i] -= A[6 * j + i] * y[j];
  for (int i = 0; i < 6; i++)
    x[perm[i]] = y[i];
}

static void SUF(set_affine)(const REAL R[9], const REAL t[3], REAL M[16])
{
  for (int r = 0; r < 3; r++)
  {
    for (int c = 0; c < 3; c++)
      M[4 * r + c] = R[3 * r + c];
    M[4 * r + 3] = t[r];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
}

/* point2point.cxx:25-66 */
static void SUF(solve_p2p)(int64_t n, const float* s, const float* t, REAL M[16])
{
  REAL w = (REAL)1 / (REAL)n;
  REAL sm[3] = { 0, 0, 0 }, tm[3] = { 0, 0, 0 };
  for (int a = 0; a < 3; a++)
    for (int64_t j = 0; j < n; j++)
    {
      sm[a] += (REAL)s[3 * j + a] * w;
      tm[a] += (REAL)t[3 * j + a] * w;
    }
  REAL sig[9] = { 0 };
  for (int64_t j = 0; j < n; j++)
  {
    REAL sc[3], tc[3];
    for (int a = 0; a < 3; a++)
    {
      sc[a] = (REAL)s[3 * j + a] - sm[a];
      tc[a] = (REAL)t[3 * j + a] - tm[a];
    }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        sig[3 * a + b] += (sc[a] * w) * tc[b];
  }
  REAL U[9], S[3], V[9], R[9];
  SUF(svd3)(sig, U, S, V);
  REAL flip = (SUF(det3)(U) * SUF(det3)(V) < 0) ? (REAL)-1 : (REAL)1;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] +
        flip * V[3 * r + 2] * U[3 * c + 2];
  REAL tr[3];
  for (int r = 0; r < 3; r++)
    tr[r] = tm[r] - (R[3 * r] * sm[0] + R[3 * r + 1] * sm[1] + R[3 * r + 2] * sm[2]);
  SUF(set_affine)(R, tr, M);
}

/* point2plane.cxx:37-110 */
static void SUF(solve_p2plane)(
  int64_t n, const float* s, const float* t, const float* nrm, REAL M[16])
{
  REAL wn = (REAL)1 / (REAL)n;
  REAL sm[3] = { 0, 0, 0 };
  for (int a = 0; a < 3; a++)
    for (int64_t j = 0; j < n; j++)
      sm[a] += (REAL)s[3 * j + a] * wn;
  REAL L[36] = { 0 }, rhs[6] = { 0 };
  for (int64_t j = 0; j < n; j++)
  {
    REAL sc[3], tc[3], nn[3], c[3];
    for (int a = 0; a < 3; a++)
    {
      sc[a] = (REAL)s[3 * j + a] - sm[a];
      tc[a] = (REAL)t[3 * j + a] - sm[a];
      nn[a] = (REAL)nrm[3 * j + a];
    }
    c[0] = sc[1] * nn[2] - sc[2] * nn[1];
    c[1] = sc[2] * nn[0] - sc[0] * nn[2];
    c[2] = sc[0] * nn[1] - sc[1] * nn[0];
    REAL x6[6] = { c[0], c[1], c[2], nn[0], nn[1], nn[2] };
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++)
        L[6 * a + b] += x6[a] * x6[b];
    REAL d = -((sc[0] - tc[0]) * nn[0] + (sc[1] - tc[1]) * nn[1] + (sc[2] - tc[2]) * nn[2]);
    for (int a = 0; a < 6; a++)
      rhs[a] += x6[a] * d;
  }
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < a; b++)
      L[6 * a + b] = L[6 * b + a];
  REAL x[6];
  SUF(ldlt6)(L, rhs, x);
  REAL ca = (REAL)cos((double)x[0]), sa = (REAL)sin((double)x[0]);
  REAL cb = (REAL)cos((double)x[1]), sb = (REAL)sin((double)x[1]);
  REAL cg = (REAL)cos((double)x[2]), sg = (REAL)sin((double)x[2]);
  /* R = Rx(alpha) * Ry(beta) * Rz(gamma) */
  REAL R[9] = { cb * cg, -cb * sg, sb, ca * sg + sa * sb * cg, ca * cg - sa * sb * sg, -sa * cb,
    sa * sg - ca * sb * cg, sa * cg + ca * sb * sg, ca * cb };
  REAL tr[3];
  for (int r = 0; r < 3; r++)
    tr[r] = x[3 + r] +
      (sm[r] - (R[3 * r] * sm[0] + R[3 * r + 1] * sm[1] + R[3 * r + 2] * sm[2]));
  SUF(set_affine)(R, tr, M);
}

/* plane2plane.cxx:37-94 */
static void SUF(solve_sym)(
  int64_t n, const float* P, const float* Q, const float* NP, const float* NQ, REAL M[16])
{
  REAL pm[3] = { 0, 0, 0 }, qm[3] = { 0, 0, 0 };
  for (int a = 0; a < 3; a++)
  {
    for (int64_t j = 0; j < n; j++)
    {
      pm[a] += (REAL)P[3 * j + a];
      qm[a] += (REAL)Q[3 * j + a];
    }
    pm[a] /= (REAL)n;
    qm[a] /= (REAL)n;
  }
  REAL L[36] = { 0 }, rhs[6] = { 0 };
  for (int64_t j = 0; j < n; j++)
  {
    REAL p[3], q[3], nn[3];
    for (int a = 0; a < 3; a++)
    {
      p[a] = (REAL)P[3 * j + a] - pm[a];
      q[a] = (REAL)Q[3 * j + a] - qm[a];
      nn[a] = (REAL)NP[3 * j + a] + (REAL)NQ[3 * j + a];
    }
    REAL b = (p[0] - q[0]) * nn[0] + (p[1] - q[1]) * nn[1] + (p[2] - q[2]) * nn[2];
    REAL u[3] = { p[0] + q[0], p[1] + q[1], p[2] + q[2] };
    REAL x6[6] = { u[1] * nn[2] - u[2] * nn[1], u[2] * nn[0] - u[0] * nn[2],
      u[0] * nn[1] - u[1] * nn[0], nn[0], nn[1], nn[2] };
    for (int a = 0; a < 6; a++)
      for (int c = 0; c < 6; c++)
        L[6 * a + c] += x6[a] * x6[c];
    for (int a = 0; a < 6; a++)
      rhs[a] += b * x6[a];
  }
  for (int a = 0; a < 6; a++)
  {
    for (int c = 0; c < a; c++)
      L[6 * a + c] = L[6 * c + a];
    rhs[a] = -rhs[a];
  }
  REAL y[6];
  SUF(ldlt6)(L, rhs, y);
  REAL an = (REAL)sqrt((double)(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]));
  REAL theta = (REAL)atan((double)an);
  REAL a[3] = { y[0], y[1], y[2] };
  if (an > 0)
  {
    a[0] /= an;
    a[1] /= an;
    a[2] /= an;
  }
  REAL W[9] = { 0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0 };
  REAL W2[9], Ri[9], R[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      W2[3 * r + c] = W[3 * r] * W[c] + W[3 * r + 1] * W[3 + c] + W[3 * r + 2] * W[6 + c];
  REAL st = (REAL)sin((double)theta), ct = (REAL)cos((double)theta);
  for (int i = 0; i < 9; i++)
    Ri[i] = ((i % 4 == 0) ? (REAL)1 : (REAL)0) + st * W[i] + ((REAL)1 - ct) * W2[i];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      R[3 * r + c] = Ri[3 * r] * Ri[c] + Ri[3 * r + 1] * Ri[3 + c] + Ri[3 * r + 2] * Ri[6 + c];
  REAL t2[3] = { ct * y[3], ct * y[4], ct * y[5] };
  REAL tr[3];
  for (int r = 0; r < 3; r++)
    tr[r] = -(R[3 * r] * pm[0] + R[3 * r + 1] * pm[1] + R[3 * r + 2] * pm[2]) +
      (Ri[3 * r] * t2[0] + Ri[3 * r + 1] * t2[1] + Ri[3 * r + 2] * t2[2]) + qm[r];
  SUF(set_affine)(R, tr, M);
}
