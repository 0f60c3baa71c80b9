// search2.hpp — the production closest-triangle search.
//
// Same answer as the exhaustive kernel in query.hpp (the exact minimum of
// (dist2, cell id) under the reference's triangle arithmetic), reached with far
// fewer exact evaluations:
//
//  * the hierarchy is a complete G-ary tree in heap order (children of node n
//    are G*n+1 .. G*n+G), so a stack entry is just a node number and a bound;
//  * boxes are tested in fp32 against a bound that is rounded so that it can
//    only under-estimate: the query's float rounding and the fp32 arithmetic
//    are covered by shrinking each axis distance by four float ulps of the
//    scene's largest coordinate;
//  * a leaf's G triangles are screened with a short fp64 closest-point routine
//    (fused multiply-adds, one division); it tracks the running minimum and a
//    small ring of "near ties": triangles whose screened distance is within
//    1e-9 x (largest coordinate) of the minimum;
//  * only the ring is evaluated with the exact routine of tri.hpp (VTK's
//    operation order, one rounding per operation), and the exact
//    (dist2, cell id) minimum over it is the winner.  Whenever the minimum
//    drops by more than the tolerance every older ring entry is provably out
//    of the race and the ring restarts, so it only ever holds genuine ties.
//
// Lanes of a group exchange values with DPP row operations (quad permutes and
// row mirrors) rather than through LDS; the stack lives in LDS, one slice per
// group.  Replaces vtkCellLocator::FindClosestPoint as called from
// ICP/vtkImplicitPolyDataDistance2.cxx:375.
#pragma once
#include "query.hpp"
#include "cert.hpp"
#include <float.h>

namespace icpx
{

// ------------------------------------------------------------ heap tree ----
struct HeapTree
{
  int G;
  int depth;          // leaves are level `depth`; levels 0..depth-1 are nodes
  int64_t first_leaf; // heap number of leaf 0 = number of nodes
  int64_t leaf_cap;   // G^depth
};

__host__ __device__ inline int64_t ipow(int64_t b, int e)
{
  int64_t r = 1;
  while (e-- > 0)
    r *= b;
  return r;
}
__host__ __device__ inline int64_t level_start(int G, int k)
{
  return (ipow(G, k) - 1) / (G - 1);
}
inline HeapTree make_heap_tree(int64_t nt, int G)
{
  HeapTree t;
  t.G = G;
  int64_t leaves = (nt + G - 1) / G;
  if (leaves < 1)
    leaves = 1;
  int d = 1;
  while (ipow(G, d) < leaves)
    d++;
  t.depth = d;
  t.first_leaf = level_start(G, d);
  t.leaf_cap = ipow(G, d);
  return t;
}

// Node n holds its G children's boxes as G records of six floats
// (lo.x lo.y lo.z hi.x hi.y hi.z).  Bottom node level: boxes of the leaves,
// widened by one float ulp outwards; empty slots hold (+inf, -inf).
__global__ void k_heap_bottom(const float* __restrict__ tverts, const int32_t* __restrict__ tid,
  int64_t nt, HeapTree t, float* __restrict__ nodes)
{
  const int G = t.G;
  int64_t leaf = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= t.leaf_cap)
    return;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  int64_t s0 = leaf * G, s1 = min(nt, s0 + G);
  bool any = false;
  for (int64_t s = s0; s < s1; s++)
  {
    if (tid[s] < 0)
      continue; // degenerate triangle: never a candidate
    any = true;
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++)
      {
        float v = tverts[9 * s + 3 * k + a];
        lo[a] = fminf(lo[a], v);
        hi[a] = fmaxf(hi[a], v);
      }
  }
  if (any)
    for (int a = 0; a < 3; a++)
    {
      lo[a] = nextafterf(lo[a], -INFINITY);
      hi[a] = nextafterf(hi[a], INFINITY);
    }
  const int64_t parent = level_start(G, t.depth - 1) + leaf / G;
  float* n = nodes + parent * 6 * G + 6 * (leaf % G);
  for (int a = 0; a < 3; a++)
  {
    n[a] = lo[a];
    n[3 + a] = hi[a];
  }
}

// Node level k < depth-1: the box of child c is the union of the boxes stored in node c.
__global__ void k_heap_level(int k, HeapTree t, float* __restrict__ nodes)
{
  const int G = t.G;
  const int64_t cnt = ipow(G, k + 1); // children of level k = nodes of level k+1
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt)
    return;
  const int64_t child = level_start(G, k + 1) + i;
  const float* c = nodes + child * 6 * G;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int j = 0; j < G; j++)
    for (int a = 0; a < 3; a++)
    {
      lo[a] = fminf(lo[a], c[6 * j + a]);
      hi[a] = fmaxf(hi[a], c[6 * j + 3 + a]);
    }
  const int64_t parent = level_start(G, k) + i / G;
  float* n = nodes + parent * 6 * G + 6 * (i % G);
  for (int a = 0; a < 3; a++)
  {
    n[a] = lo[a];
    n[3 + a] = hi[a];
  }
}

// Marks triangles that vtkTriangle::EvaluatePosition rejects (its 2x2
// determinant is exactly zero; that test does not depend on the query) by
// storing their cell id as -1-id.
__global__ void k_flag_degenerate(
  const float* __restrict__ tverts, int64_t nt, int32_t* __restrict__ tid)
{
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt)
    return;
  d3 P0, P1, P2, cp;
  double d2, w[3];
  load_tri(tverts, s, P0, P1, P2);
  if (!tri_eval(P0, P0, P1, P2, cp, d2, w))
    tid[s] = -1 - tid[s];
}

// ------------------------------------------------------ lane exchanges ----
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// value of lane (lane ^ M); M < 16 stays inside a 16-lane row and uses DPP
template <int M>
__device__ __forceinline__ int lane_xor_i(int v)
{
  if constexpr (M == 1)
    return dpp_mov<0xB1>(v); // quad_perm [1,0,3,2]
  else if constexpr (M == 2)
    return dpp_mov<0x4E>(v); // quad_perm [2,3,0,1]
  else if constexpr (M == 3)
    return dpp_mov<0x1B>(v); // quad_perm [3,2,1,0]
  else if constexpr (M == 7)
    return dpp_mov<0x141>(v); // row_half_mirror
  else if constexpr (M > 3 && M < 7)
    return lane_xor_i<(M ^ 7)>(dpp_mov<0x141>(v));
  else if constexpr (M == 15)
    return dpp_mov<0x140>(v); // row_mirror
  else if constexpr (M >= 8 && M < 15)
    return lane_xor_i<(M ^ 15)>(dpp_mov<0x140>(v));
  else
    return __shfl_xor(v, M, 64);
}
template <int M>
__device__ __forceinline__ double lane_xor_d(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(lane_xor_i<M>(hi), lane_xor_i<M>(lo));
}

// number of lanes of the group whose key is greater than this lane's (keys are distinct)
template <int G, int M = G - 1>
__device__ __forceinline__ int rank_above(int key)
{
  if constexpr (M == 0)
    return 0;
  else
    return (lane_xor_i<M>(key) > key ? 1 : 0) + rank_above<G, M - 1>(key);
}

// group minimum of a double (every lane gets it)
template <int G, int M = G / 2>
__device__ __forceinline__ double group_min(double v)
{
  if constexpr (M == 0)
    return v;
  else
    return group_min<G, M / 2>(fmin(v, lane_xor_d<M>(v)));
}

// group argmin of (d2, id) carrying s
template <int G, int M = G / 2>
__device__ __forceinline__ void group_argmin(double& d2, int32_t& id, int32_t& s)
{
  if constexpr (M != 0)
  {
    double od = lane_xor_d<M>(d2);
    int32_t oi = lane_xor_i<M>(id), os = lane_xor_i<M>(s);
    if (better(od, oi, d2, id))
    {
      d2 = od;
      id = oi;
      s = os;
    }
    group_argmin<G, M / 2>(d2, id, s);
  }
}

// ------------------------------------------- screening distance (fp64) ----
__device__ __forceinline__ double fdot(const d3& a, const d3& b)
{
  return __builtin_fma(a.x, b.x, __builtin_fma(a.y, b.y, a.z * b.z));
}

// Squared distance from p to triangle (a,b,c) by Voronoi-region classification
// of the barycentric coordinates; accurate to a few fp64 ulps of the
// coordinates, not bit-compatible with tri_eval (it only screens).
__device__ __forceinline__ double tri_d2_screen(const d3& p, const d3& a, const d3& b, const d3& c)
{
  const d3 ab = mk(b.x - a.x, b.y - a.y, b.z - a.z), ac = mk(c.x - a.x, c.y - a.y, c.z - a.z);
  const d3 ap = mk(p.x - a.x, p.y - a.y, p.z - a.z);
  const d3 bp = mk(p.x - b.x, p.y - b.y, p.z - b.z), cq = mk(p.x - c.x, p.y - c.y, p.z - c.z);
  const double d1 = fdot(ab, ap), d2 = fdot(ac, ap);
  const double d3_ = fdot(ab, bp), d4 = fdot(ac, bp);
  const double d5 = fdot(ab, cq), d6 = fdot(ac, cq);
  const double va = __builtin_fma(d3_, d6, -(d5 * d4));
  const double vb = __builtin_fma(d5, d2, -(d1 * d6));
  const double vc = __builtin_fma(d1, d4, -(d3_ * d2));
  const double e43 = d4 - d3_, e56 = d5 - d6;
  // cp = a + s*ab + t*ac with s = ns/den, t = nt/den; later selects have priority
  double ns = vb, nt = vc, den = va + vb + vc;             // interior
  if (va <= 0.0 && e43 >= 0.0 && e56 >= 0.0)               // edge bc
  {
    ns = e56;
    nt = e43;
    den = e43 + e56;
  }
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0)                 // edge ac
  {
    ns = 0.0;
    nt = d2;
    den = d2 - d6;
  }
  if (d6 >= 0.0 && d5 <= d6)                               // vertex c
  {
    ns = 0.0;
    nt = 1.0;
    den = 1.0;
  }
  if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0)                // edge ab
  {
    ns = d1;
    nt = 0.0;
    den = d1 - d3_;
  }
  if (d3_ >= 0.0 && d4 <= d3_)                             // vertex b
  {
    ns = 1.0;
    nt = 0.0;
    den = 1.0;
  }
  if (d1 <= 0.0 && d2 <= 0.0)                              // vertex a
  {
    ns = 0.0;
    nt = 0.0;
    den = 1.0;
  }
  const double r = (den != 0.0) ? 1.0 / den : 0.0;
  const double s = ns * r, t = nt * r;
  const d3 d = mk(__builtin_fma(-t, ac.x, __builtin_fma(-s, ab.x, ap.x)),
    __builtin_fma(-t, ac.y, __builtin_fma(-s, ab.y, ap.y)),
    __builtin_fma(-t, ac.z, __builtin_fma(-s, ab.z, ap.z)));
  return fdot(d, d);
}

// ------------------------------------------------------------- kernel ----
struct Search2Args
{
  const float* qpos;
  int64_t nq;
  XfRef T;
  const float* nodes;
  HeapTree tree;
  const float* tverts;
  const int32_t* tid; // cell id, or -1-id for a degenerate triangle
  int64_t nt;
  int32_t* hint;
  int use_hint;
  float scene_max; // largest |coordinate| of the target's bounds
  double cap2;     // square of the distance beyond which nothing needs to be found (DBL_MAX: none)
  float* cert;     // certificates (see local.hpp), or null.  A query that ends here with nothing
                   // within the cap gets a far certificate.  With one wavefront per query
                   // (G = 64) a single winner gets a near one (or, its nearest loser being a hair
                   // behind, the pair as a set), and two to RING3 tied contenders a set
                   // certificate; otherwise none (0)
  int32_t* cand;   // [nq][RING3] contender sets behind set certificates (or null: no sets)
  float widen;     // margin by which the walk looks beyond its bound, so that a far certificate
                   // is worth something (0 when certificates are not kept)
  const int32_t* cert_ok; // the certificates' standing in the run's state (cert.hpp: anchored passes), or null
  const double* cert_T;   // ... and the transform they refer to
  int pairs;       // keep a winner and a rival next to nothing behind as a set of two
};

template <int G>
struct Search2Cfg
{
  static constexpr int LEVELS = (G == 4) ? 14 : (G == 8) ? 9 : (G == 16) ? 7 : (G == 32) ? 6 : 5;
  static constexpr int STACK = (G - 1) * LEVELS + 2;
  static constexpr int RING = (G > 16) ? G : 16;
  static constexpr int GPB = 256 / G;
};

// The search of one query by its group of G lanes (group g of the block, lane j of the
// group).  Groups synchronise only within their wavefront, so wavefronts of a block may
// call this different numbers of times.
template <int G>
__device__ __forceinline__ void search2_body(const Search2Args& a, const Xf& T, int64_t q,
  bool active)
{
  using Cfg = Search2Cfg<G>;
  __shared__ unsigned long long stack[Cfg::GPB][Cfg::STACK];
  __shared__ int32_t ring[Cfg::GPB][Cfg::RING];
  const int tid = threadIdx.x;
  const int g = tid / G, j = tid % G;
  const int lane = tid & 63;
  const int gbase = lane & ~(G - 1);
  const uint32_t first_leaf = (uint32_t)a.tree.first_leaf;
  constexpr uint32_t GM = (G >= 32) ? 0xffffffffu : ((1u << (G & 31)) - 1u); // a group's lanes

  d3 x = mk(0.0, 0.0, 0.0);
  float xf = 0.f, yf = 0.f, zf = 0.f, eabs = 0.f;
  double tau = 0.0;
  double best = DBL_MAX;   // screened minimum
  float boundf = FLT_MAX;  // (sqrt(best) + tau)^2, rounded up
  double bound = DBL_MAX;
  int cnt = 0;             // ring entries
  // exact results folded so far (only when the ring had to be resolved early)
  double ed2 = INFINITY;
  int32_t eid = 0x7fffffff, es = -1;
  bool flushed = false;
  bool overflowed = false; // the ring had to be resolved early: more contenders than it holds
  int32_t hs_used = -1; // the hinted triangle already sits in the ring
  int top = 0;
  float minlb = INFINITY; // smallest lower bound (squared) of anything this lane turned away
  // for a near certificate: the boxes this lane turned away, the nearest triangle it screened
  // and turned away (by name), and the nearest after that
  float minbox = INFINITY, lose1 = INFINITY, lose2 = INFINITY;
  int32_t lose_id = -1;

  // how far behind the minimum a triangle still joins the ring: the tie tolerance; on a second
  // walk (below) 64 rounding allowances
  double ring_tol = 0.0;
  auto set_best = [&](double nb) {
    // older ring entries were all above the old minimum; if the new bound is
    // below it none of them can be within tolerance of the final minimum
    const double nbound = (sqrt(nb) + ring_tol) * (sqrt(nb) + ring_tol);
    if (nbound < best)
      cnt = 0;
    best = nb;
    bound = nbound;
    const double rw = sqrt(nb) + ring_tol + (double)a.widen; // (boxes are turned away beyond this)
    boundf = fminf(fmaxf(__double2float_ru(rw * rw), FLT_MIN), FLT_MAX);
  };

  // exact evaluation of the ring's entries, folded into (ed2, eid, es)
  auto resolve = [&]() {
    for (int b0 = 0; b0 < cnt; b0 += G)
    {
      double d2 = INFINITY;
      int32_t id = 0x7fffffff, s = -1;
      if (b0 + j < cnt)
      {
        s = ring[g][b0 + j];
        d3 P0, P1, P2, cp;
        double dd, w[3];
        load_tri(a.tverts, s, P0, P1, P2);
        if (tri_eval(x, P0, P1, P2, cp, dd, w))
        {
          d2 = dd;
          id = a.tid[s];
        }
      }
      group_argmin<G>(d2, id, s);
      if (better(d2, id, ed2, eid))
      {
        ed2 = d2;
        eid = id;
        es = s;
      }
    }
    cnt = 0;
    flushed = true;
  };

  // A second walk where the first one's certificate comes out at next to nothing (one wavefront
  // per query, sets kept): a noisy point's two or three nearest triangles can be within a
  // hair of each other, the certificate of the winner, or of winner and runner-up, is then worth
  // a pass or two, and the query is searched again for good (a settled pass of sixteen million
  // such points searched twenty thousand of them).  The second walk takes every triangle within
  // 64 rounding allowances of the nearest into the ring: if they are eight at most they are the
  // query's set, and its certificate is the distance to whatever comes after them.
  float first_cert = 0.f;
  int32_t first_win = -1;
  bool again = false;
#pragma unroll 1
  for (int attempt = 0; attempt < (G == 64 ? 4 : 1); attempt++)
  {
  if (attempt >= 1)
  {
    if (!again)
      break;
    __builtin_amdgcn_wave_barrier(); // (ring and stack are used anew)
    best = DBL_MAX;
    boundf = FLT_MAX;
    bound = DBL_MAX;
    cnt = 0;
    ed2 = INFINITY;
    eid = 0x7fffffff;
    es = -1;
    flushed = false;
    overflowed = false;
    hs_used = -1;
    top = 0;
    minlb = INFINITY;
    minbox = INFINITY;
    lose1 = INFINITY;
    lose2 = INFINITY;
    lose_id = -1;
  }
  if (active)
  {
    x = xform(T, a.qpos + 3 * q);
    xf = (float)x.x;
    yf = (float)x.y;
    zf = (float)x.z;
    const float smax = fmaxf(fmaxf(a.scene_max, fabsf(xf)), fmaxf(fabsf(yf), fabsf(zf)));
    eabs = smax * 2.3841858e-07f * 1.0000002f; // 2^-22 of the largest coordinate, rounded up
    tau = 1e-9 * (double)smax;
    // (a third and a fourth walk, each after one that met more triangles within its tolerance
    // than the ring holds: eight allowances, then one)
    ring_tol = attempt == 0 ? tau : fmax(tau, (attempt == 1 ? 64.0 : attempt == 2 ? 8.0 : 1.0) * (double)eabs);
    if (a.cap2 < 1e300)
      set_best(a.cap2);
    if (a.use_hint || attempt >= 1)
    {
      const int32_t hs = attempt == 0 ? a.hint[q] : first_win;
      if (hs >= 0 && a.tid[hs] >= 0)
      {
        d3 P0, P1, P2;
        load_tri(a.tverts, hs, P0, P1, P2);
        const double ds = tri_d2_screen(x, P0, P1, P2);
        if (ds < best)
        {
          set_best(ds);
          if (j == 0)
            ring[g][0] = hs;
          cnt = 1;
          hs_used = hs;
        }
      }
    }
    if (j == 0)
      stack[g][0] = 0ull; // root, bound 0
    top = 1;
  }
  __builtin_amdgcn_wave_barrier();

  int64_t leaf = -1;
  while (true)
  {
    while (leaf < 0 && top > 0)
    {
      const unsigned long long e = stack[g][--top];
      const float lbf = __uint_as_float((uint32_t)(e >> 32));
      if (lbf > boundf)
      {
        minlb = fminf(minlb, lbf);
        minbox = fminf(minbox, lbf);
        continue;
      }
      const uint32_t n = (uint32_t)e;
      if (n >= first_leaf)
      {
        leaf = (int64_t)(n - first_leaf);
        break;
      }
      const float2* bx = reinterpret_cast<const float2*>(a.nodes + (int64_t)n * (6 * G) + 6 * j);
      const float2 b0 = bx[0], b1 = bx[1], b2 = bx[2]; // lo.x lo.y | lo.z hi.x | hi.y hi.z
      const float dx = fmaxf(fmaxf(fmaxf(b0.x - xf, xf - b1.y), 0.f) - eabs, 0.f);
      const float dy = fmaxf(fmaxf(fmaxf(b0.y - yf, yf - b2.x), 0.f) - eabs, 0.f);
      const float dz = fmaxf(fmaxf(fmaxf(b1.x - zf, zf - b2.y), 0.f) - eabs, 0.f);
      const float lb = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * 0.99999952f;
      const bool push = lb <= boundf; // an empty box gives +inf and boundf is finite
      if (!push)
      {
        minlb = fminf(minlb, lb);
        minbox = fminf(minbox, lb);
      }
      // distinct integer keys: the bound's bits, lowered by G ulps, with the lane number in
      // the low bits (so the stored bound never exceeds the computed one)
      const uint32_t lbits = __float_as_uint(lb);
      const uint32_t low = lbits > (uint32_t)G ? lbits - (uint32_t)G : 0u;
      const int key = push ? (int)((low & ~(uint32_t)(G - 1)) | (uint32_t)j) : -1;
      const unsigned long long bal = __ballot(push);
      const int npush = (G == 64) ? __popcll(bal) : __popc((uint32_t)(bal >> gbase) & GM);
      const int pos = rank_above<G>(key); // farther children deeper, nearest on top
      if (push)
        stack[g][top + pos] =
          ((unsigned long long)(uint32_t)key << 32) | (uint32_t)((uint32_t)n * G + 1u + j);
      top += npush;
      __builtin_amdgcn_wave_barrier();
    }
    if (leaf < 0)
      break;

    // screen the leaf's triangles, one per lane
    const int64_t s = leaf * G + j;
    double d2 = INFINITY;
    if (s < a.nt && a.tid[s] >= 0)
    {
      d3 P0, P1, P2;
      load_tri(a.tverts, s, P0, P1, P2);
      d2 = tri_d2_screen(x, P0, P1, P2);
    }
    if (d2 < INFINITY)
      minlb = fminf(minlb, __double2float_rd(d2) * 0.999999f);
    const double m = group_min<G>(d2);
    if (m < best)
    {
      if (G == 64 && a.cert && cnt > 0 && j == 0)
      {
        // a new minimum that empties the ring leaves its members behind as losers, at least
        // the old minimum away (the first of them exactly that)
        const double r = sqrt(m) + tau;
        if (r * r < best)
        {
          const float v = __double2float_rd(best) * 0.999999f;
          note_loser(lose1, lose_id, lose2, v, ring[g][0]);
          if (cnt > 1)
            lose2 = fminf(lose2, v);
        }
      }
      set_best(m);
    }
    // the hinted triangle is met again during the walk: it is in the ring already, or out
    // of the race for good if the ring restarted since
    const bool cand = d2 <= bound && (int32_t)s != hs_used;
    if (G == 64 && a.cert && d2 < INFINITY && !(d2 <= bound) && (int32_t)s != hs_used)
      note_loser(lose1, lose_id, lose2, __double2float_rd(d2) * 0.999999f, (int32_t)s);
    const unsigned long long bal = __ballot(cand);
    const uint32_t gm = (G == 64) ? 0u : ((uint32_t)(bal >> gbase) & GM);
    const int k = (G == 64) ? __popcll(bal) : __popc(gm);
    if (k > 0)
    {
      if (cnt + k > Cfg::RING)
      {
        __builtin_amdgcn_wave_barrier();
        resolve(); // more ties than the ring holds: settle what is there exactly
        overflowed = true;
      }
      const int r = (G == 64) ? __popcll(bal & ((1ull << j) - 1ull))
                              : __popc(gm & ((1u << j) - 1u));
      if (cand)
        ring[g][cnt + r] = (int32_t)s;
      cnt += k;
      __builtin_amdgcn_wave_barrier();
    }
    leaf = -1;
  }

  if (active)
  {
    // a near certificate, from what the 64 lanes turned away between them (before the ring
    // is resolved: a set certificate keeps its members)
    float nearc = 0.f;
    bool have_near = false;
    if constexpr (G == 64)
    {
      if (a.cert && !flushed && cnt >= 1 && cnt <= RING3 && (cnt == 1 || a.cand))
      {
        const float m1 = (float)group_min<G>((double)lose1);
        const unsigned long long who = __ballot(lose1 == m1);
        const int leader = __ffsll((long long)who) - 1;
        const int32_t id1 = m1 < INFINITY ? __shfl(lose_id, leader, 64) : -1;
        const float m2 = (float)group_min<G>((double)(lane == leader ? lose2 : lose1));
        const float mb = (float)group_min<G>((double)minbox);
        // (these box bounds carry their rounding allowances already)
        const float dbox = __builtin_amdgcn_sqrtf(mb) * 0.999999f;
        have_near = true;
        if (cnt == 1)
        {
          bool pair;
          nearc = make_certificate_pair(1, best, m1, m2, a.cand ? id1 : -1, INFINITY, dbox, tau,
            eabs, a.cap2,
            a.pairs == 2 ? 2.f * eabs : a.pairs ? fmaxf(a.widen, 4.f * eabs) : -INFINITY, pair);
          if (pair && j == 0 && (attempt == 0 || nearc > first_cert))
          {
            a.cand[q * RING3 + 0] = ring[g][0];
            a.cand[q * RING3 + 1] = id1;
            a.cand[q * RING3 + 2] = -1;
          }
        }
        else
        {
          nearc = make_certificate(cnt, best, m1, INFINITY, dbox, tau, eabs, a.cap2);
          if (nearc != 0.f && j <= cnt && j < RING3 && (attempt == 0 || nearc > first_cert))
            a.cand[q * RING3 + j] = j < cnt ? ring[g][j] : -1;
        }
      }
    }
    if constexpr (G == 64)
    {
      // The second walk ended with more triangles within its tolerance than a set holds (a
      // point dozens of triangle widths off a patch of small triangles has some forty within
      // 64 rounding allowances of the nearest): the eight nearest of them are the set, and
      // what certifies it is the distance of the ninth (screened like the others, to the last
      // digits of a double: no rounding allowance comes into it), of the nearest triangle
      // screened and turned away, and of the nearest box turned away.  Without this such a
      // point has no certificate and is searched, thousands of boxes, on every pass.
      if (attempt >= 1 && a.cert && a.cand && !flushed && cnt > RING3 && cnt <= Cfg::RING)
      {
        double dj = INFINITY;
        int32_t sj = -1;
        if (j < cnt)
        {
          sj = ring[g][j];
          d3 P0, P1, P2;
          load_tri(a.tverts, sj, P0, P1, P2);
          dj = tri_d2_screen(x, P0, P1, P2);
        }
        int rank = 0;
#pragma unroll 1
        for (int k = 0; k < cnt; k++)
        {
          const double dk = __shfl(dj, k, 64);
          rank += (dk < dj || (dk == dj && k < j)) ? 1 : 0;
        }
        const bool member = j < cnt && rank < RING3;
        const double d9 = group_min<G>((j < cnt && !member) ? dj : (double)INFINITY);
        const float m1 = (float)group_min<G>((double)lose1);
        const float mb = (float)group_min<G>((double)minbox);
        const float dbox = __builtin_amdgcn_sqrtf(mb) * 0.999999f;
        nearc = make_certificate(RING3, best, fminf(m1, __double2float_rd(d9) * 0.999999f), INFINITY,
          dbox, tau, eabs, a.cap2);
        have_near = true;
        if (nearc != 0.f && nearc > first_cert && member)
          a.cand[q * RING3 + rank] = sj;
      }
    }
    int32_t win;
    if (cnt == 1 && !flushed)
      win = ring[g][0]; // a single contender needs no exact evaluation here
    else
    {
      if (cnt > 0)
        resolve();
      win = es;
    }
    // nothing within the cap: everything was turned away or screened beyond it, and the
    // nearest of those bounds says by how much
    float far = 0.f;
    if (a.cert && cnt == 0 && !flushed && a.cap2 < 1e300)
    {
      const float m2 = (float)group_min<G>((double)minlb);
      const float capd = __double2float_ru(sqrt(a.cap2));
      const float c = __builtin_amdgcn_sqrtf(m2) * 0.999999f - capd * 1.000001f - 2.1f * (float)tau -
        fminf(eabs, 0.2f * eabs * eabs / fmaxf(capd, 1e-30f));
      far = (c > 0.f && c < 1e30f) ? -c : 0.f;
    }
    if (attempt >= 1)
    {
      // (the winner is the first walk's: the exact minimum of a superset of its contenders)
      if (j == 0 && have_near && nearc > first_cert && win == first_win)
        a.cert[q] = cert_rebase_at(nearc, a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
      // (more triangles within the tolerance than the ring holds: once more, with an eighth of it)
      again = attempt < 3 && overflowed && a.cert && a.cand;
    }
    else
    {
      if (j == 0)
      {
        a.hint[q] = win;
        if (a.cert)
          a.cert[q] = cert_rebase_at(have_near ? nearc : far, a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
      }
      if constexpr (G == 64)
      {
        first_cert = have_near ? nearc : 0.f;
        first_win = win;
        again = a.pairs != 0 && a.cert && a.cand && have_near && win >= 0 && nearc >= 0.f &&
          nearc < 0.25f * eabs;
      }
    }
  }
  else
    again = false;
  }
}

template <int G>
__global__ __launch_bounds__(256) void k_search2(Search2Args a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  const int64_t q = (int64_t)blockIdx.x * Search2Cfg<G>::GPB + threadIdx.x / G;
  search2_body<G>(a, T, q, q < a.nq);
}

} // namespace icpx
