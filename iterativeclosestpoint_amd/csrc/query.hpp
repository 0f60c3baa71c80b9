// query.hpp — the per-iteration kernels.
//
//  k_search<G>    closest triangle for every kept source point: G lanes share
//                 one query, walk the G-ary box hierarchy with a stack in LDS
//                 (nearest child on top), and test the G triangles of a leaf
//                 in parallel with the exact fp64 point/triangle routine.
//                 G = 64 is one wavefront per query.  Replaces
//                 vtkCellLocator::FindClosestPoint as called from
//                 ICP/vtkImplicitPolyDataDistance2.cxx:375.
//  k_finalize<..> one thread per query: closest point, barycentric
//                 classification, pseudo-normal, thresholds, then either the
//                 metric's fp64 partial sums (wave shuffles -> LDS -> one
//                 partial per block) or the filter's output rows.  Replaces
//                 the rest of ClosestPointAndNormal (:379-527), the functor
//                 bodies of ICP/vtkPolyDataCorrespondenceFilter.cxx:110-136 /
//                 274-305 and the solvers' O(n) loops.
//
// The winner is the exact minimum of (dist2, cell id) over all triangles:
// a subtree is skipped only when the fp64 distance to its (outward-rounded)
// box exceeds the best dist2 so far, so the traversal order cannot change it.
#pragma once
#include "state.hpp"
#include "../../include/icpx.h"
#include "bvh.hpp"
#include "cert.hpp"
#include "tri.hpp"
#include <stdint.h>

namespace icpx
{

struct Xf
{
  double m[12]; // rows 0..2 of the 4x4
};

// How a kernel receives the transform: a row-major 4x4 in device memory (so that a loop
// whose solve runs on the device never brings it to the host), and the flag that tells a
// pass enqueued ahead of time that its registration has stopped in the meantime.
struct XfRef
{
  const double* T;
  const int32_t* stop;
};

__device__ __forceinline__ double uniform_f64(double v)
{
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Loads the transform into scalar registers; false when the pass must not run.
__device__ __forceinline__ bool load_xf(const XfRef& r, Xf& out)
{
  if (r.stop && __builtin_amdgcn_readfirstlane(*r.stop) != 0)
    return false;
#pragma unroll
  for (int i = 0; i < 12; i++)
    out.m[i] = uniform_f64(r.T[i]);
  return true;
}

// vtkTransform::InternalTransformPoint on a float point promoted to double
// (ICP/vtkPolyDataCorrespondenceFilter.cxx:116-119).
__device__ __forceinline__ d3 xform(const Xf& T, const float* __restrict__ p)
{
  double x = (double)p[0], y = (double)p[1], z = (double)p[2];
  return mk(T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3],
    T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7],
    T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11]);
}

// A certificate made during an anchored pass (cert.hpp), as it is stored: less twice the
// query's distance from where the certificates' transform has it (x: the query under this
// pass's transform, p: its point).  The transform is read here, where a search has just
// ended: a few reads per searched query, nothing for the others.
__device__ __forceinline__ float cert_rebase_at(float c, const int32_t* cert_ok,
  const double* __restrict__ cert_T, const d3& x, const float* __restrict__ p)
{
  if (c == 0.f || !cert_anchored(cert_ok))
    return c;
  Xf Tc;
#pragma unroll
  for (int i = 0; i < 12; i++)
    Tc.m[i] = uniform_f64(cert_T[i]);
  const d3 xp = xform(Tc, p);
  const double mx = x.x - xp.x, my = x.y - xp.y, mz = x.z - xp.z;
  const float mv = __builtin_amdgcn_sqrtf(__double2float_ru(mx * mx + my * my + mz * mz)) * 1.000001f;
  return cert_rebase(c, 2.000002f * mv + 1e-30f);
}

// A set's weighing has named its winner (cert.hpp): the winner's lead over the set's other
// members (l, from lead_of) goes on record for query q, stored as a certificate made in this
// pass is (in an anchored pass: less twice the query's distance from where the anchor has it),
// and the certificate's second bit says that there is one.  A tie (l = 0: most sets of a noisy
// cloud are points nearest to an edge, tied for good) leaves everything as it is and costs
// nothing: a lead that ran out was zeroed by the check that found it so (cert_lead_stands).
__device__ __forceinline__ void cert_note_lead(float* cert, float* lead, int64_t q, float l,
  const int32_t* cert_ok, const double* __restrict__ cert_T, const d3& x, const float* __restrict__ p)
{
  if (!(l > 0.f))
    return;
  const float g = cert_rebase_at(l, cert_ok, cert_T, x, p);
  const float c = cert[q];
  if (!(g > 0.f) || !cert_is_set(c))
    return;
  lead[q] = g;
  cert[q] = __uint_as_float(__float_as_uint(c) | 2u);
}
// The check of a lead on record (ld: the certificate says there is one; leadv: its value; mv:
// the query's move since the certificates' transform).  Stands: the set needs no weighing, and
// in a pass that is not anchored the lead is written back less twice the move, as the
// certificate is.  Ran out: it is zeroed (whatever the pass), so that it can never be held
// against a later, smaller move; the weighing that follows records a new one if there is one.
__device__ __forceinline__ bool cert_lead_stands(float* lead, int64_t q, float leadv, float mv,
  bool anchored)
{
  const float ll = leadv - 2.000001f * mv - 1e-30f;
  if (ll > 0.f)
  {
    if (!anchored)
      lead[q] = ll * 0.9999999f;
    return true;
  }
  if (leadv != 0.f)
    lead[q] = 0.f;
  return false;
}

__device__ __forceinline__ void load_tri(
  const float* __restrict__ tverts, int64_t s, d3& P0, d3& P1, d3& P2)
{
  const float* t = tverts + 9 * s;
  P0 = mk((double)t[0], (double)t[1], (double)t[2]);
  P1 = mk((double)t[3], (double)t[4], (double)t[5]);
  P2 = mk((double)t[6], (double)t[7], (double)t[8]);
}

__device__ __forceinline__ bool better(double d2, int32_t id, double bd2, int32_t bid)
{
  return d2 < bd2 || (d2 == bd2 && id < bid);
}

struct SearchArgs
{
  const float* qpos; // query points in query-Morton order
  int64_t nq;
  XfRef T;
  const float* nodes;
  TreeLevels lv;
  const float* tverts;
  const int32_t* tid;
  int64_t nt;
  int32_t* hint; // per query: sorted position of the winning triangle (in: last pass, out: this pass)
  int use_hint;
};

template <int G>
struct SearchCfg
{
  // deepest hierarchy the LDS stack is sized for (G^levels leaves)
  static constexpr int LEVELS = (G == 4) ? 14 : (G == 8) ? 10 : (G == 16) ? 7 : (G == 32) ? 6 : 5;
  static constexpr int STACK = (G - 1) * LEVELS + 2;
  static constexpr int GPB = 256 / G; // queries per 256-thread block
};

template <int G>
__global__ __launch_bounds__(256) void k_search(SearchArgs a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  using Cfg = SearchCfg<G>;
  __shared__ unsigned long long stack[Cfg::GPB][Cfg::STACK];
  __shared__ int64_t s_off[MAXLEV + 1];
  __shared__ int32_t s_cnt[MAXLEV + 1];
  const int tid = threadIdx.x;
  if (tid <= MAXLEV)
  {
    s_off[tid] = a.lv.off[tid];
    s_cnt[tid] = a.lv.cnt[tid];
  }
  __syncthreads();
  const int g = tid / G, j = tid % G;
  const int lane = tid & 63;
  const int gbase = lane & ~(G - 1);
  const int64_t q = (int64_t)blockIdx.x * Cfg::GPB + g;
  const bool active = q < a.nq;

  d3 x = mk(0.0, 0.0, 0.0);
  double best = INFINITY;
  int32_t bestId = 0x7fffffff, bestS = -1;
  int top = 0;
  if (active)
  {
    x = xform(T, a.qpos + 3 * q);
    if (a.use_hint)
    {
      int32_t hs = a.hint[q];
      if (hs >= 0)
      {
        d3 P0, P1, P2, cp;
        double d2, w[3];
        load_tri(a.tverts, hs, P0, P1, P2);
        if (tri_eval(x, P0, P1, P2, cp, d2, w))
        {
          best = d2;
          bestS = hs;
          bestId = a.tid[hs];
        }
      }
    }
    if (j == 0)
      stack[g][0] = (unsigned long long)((uint32_t)a.lv.nlev << 28); // root, bound 0.0f
    top = 1;
  }
  __builtin_amdgcn_wave_barrier();

  int32_t leaf = -1;
  while (true)
  {
    // descend until this group holds a leaf to test or its stack is empty
    while (leaf < 0 && top > 0)
    {
      unsigned long long e = stack[g][--top];
      float lbf = __uint_as_float((uint32_t)(e >> 32));
      if ((double)lbf > best)
        continue;
      uint32_t code = (uint32_t)e;
      int L = (int)(code >> 28);
      int32_t idx = (int32_t)(code & 0x0fffffffu);
      if (L == 0)
      {
        leaf = idx;
        break;
      }
      const int64_t child = (int64_t)idx * G + j;
      const float* n = a.nodes + s_off[L] + (int64_t)idx * (6 * G) + j;
      double lox = (double)n[0], loy = (double)n[G], loz = (double)n[2 * G];
      double hix = (double)n[3 * G], hiy = (double)n[4 * G], hiz = (double)n[5 * G];
      double dx = fmax(fmax(lox - x.x, x.x - hix), 0.0);
      double dy = fmax(fmax(loy - x.y, x.y - hiy), 0.0);
      double dz = fmax(fmax(loz - x.z, x.z - hiz), 0.0);
      double lb = dx * dx + dy * dy + dz * dz;
      const bool push = (child < s_cnt[L - 1]) && (lb <= best);
      const double key = push ? lb : -INFINITY;
      unsigned long long bal = __ballot(push);
      uint32_t gm = (G == 64) ? 0u : (uint32_t)((bal >> gbase) & ((1ull << (G & 63)) - 1ull));
      int npush = (G == 64) ? __popcll(bal) : __popc(gm);
      // farther children go deeper in the stack; the nearest ends on top
      int pos = 0;
#pragma unroll
      for (int i = 0; i < G; i++)
      {
        double o = __shfl(key, i, G);
        pos += (o > key || (o == key && i > j)) ? 1 : 0;
      }
      if (push)
      {
        uint32_t c = ((uint32_t)(L - 1) << 28) | (uint32_t)child;
        uint32_t f = __float_as_uint(__double2float_rd(lb));
        stack[g][top + pos] = ((unsigned long long)f << 32) | c;
      }
      top += npush;
      __builtin_amdgcn_wave_barrier();
    }
    if (leaf < 0)
      break;
    // exact test of the leaf's triangles, one per lane
    int64_t s = (int64_t)leaf * G + j;
    double d2 = INFINITY;
    int32_t id = 0x7fffffff;
    int32_t sw = (int32_t)s;
    if (s < a.nt)
    {
      d3 P0, P1, P2, cp;
      double dd, w[3];
      load_tri(a.tverts, s, P0, P1, P2);
      if (tri_eval(x, P0, P1, P2, cp, dd, w))
      {
        d2 = dd;
        id = a.tid[s];
      }
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1)
    {
      double od = __shfl_xor(d2, m, G);
      int32_t oi = __shfl_xor(id, m, G);
      int32_t os = __shfl_xor(sw, m, G);
      if (better(od, oi, d2, id))
      {
        d2 = od;
        id = oi;
        sw = os;
      }
    }
    if (better(d2, id, best, bestId))
    {
      best = d2;
      bestId = id;
      bestS = sw;
    }
    leaf = -1;
  }
  if (active && j == 0)
    a.hint[q] = bestS;
}

// ---------------------------------------------------------------- finalize ----
struct FinalArgs
{
  const float* qpos;   // sorted query points
  const float* qsn;    // sorted source normals (or null)
  const int32_t* qk;   // sorted position -> kept index local to the shard
  int64_t nq;
  XfRef T;
  double origin[3];
  const float* tverts;
  const int32_t* tid;
  const int32_t* tri;
  const float* fnorm;
  const float* enorm;
  const float* vnorm;
  // per Morton position of a triangle: {its face normal, its cell id (the int's bits)}: what
  // nearly every row needs of the tables, in one 16-byte read that goes out together with the
  // triangle's points instead of two reads that wait for one another behind the evaluation
  const float4* fnpos;
  const int32_t* hint;
  double max_dist, min_dot;
  int use_sn;
  int rotate_sn; // source normals are rotated by T's 3x3 first (the corrected mode)
  // rows, indexed by kept index (ROWS only)
  float* o_origin;
  float* o_onormal;
  float* o_closest;
  float* o_normal;
  int32_t* o_cell;
  double* o_dist;
  uint32_t* o_flag;
  // partial sums, [ICPX_NSUMS][gridDim.x] (sums only)
  double* partials;
  // a lean pass (k_finalize_mx<.., LEAN>): the certificates and what they refer to, the
  // contender sets (RING3 per query), the winners (rewritten for queries that stand on a
  // set), the squared distance threshold, and the flag raised when some query's certificate
  // does not hold
  float* cert;
  const double* cert_T;
  const int32_t* cert_ok;
  const double* cert_cap2;
  const int32_t* cand;
  int32_t* cand_rw;
  int32_t* hint_rw;
  double cap2;
  int32_t* lean_fail;
  // ... and for k_certify_rest (lean.hpp): the neighbour lists of local.hpp (a list longer
  // than nbr_cap was cut short), their radius, the scene's largest coordinate (tolerances)
  const uint32_t* noff;
  const int32_t* nbr;
  uint32_t nbr_cap;
  float eps;
  float scene_max;
  int lean_inject; // not zero: the pass fails whatever it finds (tests of the repeat)
  // not null (the matrix-core sums build only): one bit per query that this pass's certificate check let
  // through on a SET certificate; the wavefront that makes such a query's row first weighs the
  // set's members (cand[]) and writes the winner to hint[] (weigh_marked), which is otherwise
  // k_contested's work in a launch of its own
  const unsigned long long* setmask;
  // the leads of sets' winners (cert.hpp): read by a lean pass's check; written, with the
  // certificate's second bit, by weigh_marked (null: not kept)
  float* lead;
};

// (lean.hpp) the certificate loop of a lean pass for one wavefront (queries first + lane,
// first + stride + lane, ...): every query held against its certificate, the few winners
// that have to be worked out written to hint[].  Called by all 64 lanes.
__device__ __forceinline__ void lean_certify(const FinalArgs& a, const Xf& T, const Xf& Tc,
  bool far_on, int64_t first, int64_t stride, int lane, int32_t* list, bool& lean_bad);

// (lean.hpp) the queries among a wavefront's (first + lane, first + stride + lane, ...) that are
// marked in a.setmask: their sets weighed, the winners written to hint[].  list: room for `cap`
// (>= 64) entries in LDS, this wavefront's own.  Called by all 64 lanes.
__device__ __forceinline__ void weigh_marked(const FinalArgs& a, const Xf& T, int64_t first,
  int64_t stride, int lane, int32_t* list, int cap);

template <int METRIC>
struct SumCfg
{
  static constexpr int N = (METRIC == 0) ? 17 : (METRIC == 1) ? 32 : 35;
};

// One accepted row into the metric's raw moments; s, t, n, sn are the float
// rows promoted to double, o the moment origin.
template <int METRIC>
__device__ __forceinline__ void add_row(double* acc, const d3& s, const d3& t, const d3& n,
  const d3& sn, const double* o)
{
  const d3 sp = mk(s.x - o[0], s.y - o[1], s.z - o[2]);
  const d3 tp = mk(t.x - o[0], t.y - o[1], t.z - o[2]);
  const d3 df = mk(s.x - t.x, s.y - t.y, s.z - t.z);
  acc[0] += 1.0;
  acc[1] += sqrt(df.x * df.x + df.y * df.y + df.z * df.z);
  acc[2] += sp.x;
  acc[3] += sp.y;
  acc[4] += sp.z;
  if (METRIC == 0)
  {
    acc[5] += tp.x;
    acc[6] += tp.y;
    acc[7] += tp.z;
    acc[8] += sp.x * tp.x;
    acc[9] += sp.x * tp.y;
    acc[10] += sp.x * tp.z;
    acc[11] += sp.y * tp.x;
    acc[12] += sp.y * tp.y;
    acc[13] += sp.y * tp.z;
    acc[14] += sp.z * tp.x;
    acc[15] += sp.z * tp.y;
    acc[16] += sp.z * tp.z;
  }
  else
  {
    int b = 5;
    d3 nn = n, u = sp;
    if (METRIC == 2)
    {
      acc[5] += tp.x;
      acc[6] += tp.y;
      acc[7] += tp.z;
      b = 8;
      nn = mk(sn.x + n.x, sn.y + n.y, sn.z + n.z);
      u = mk(sp.x + tp.x, sp.y + tp.y, sp.z + tp.z);
    }
    const d3 a = mk(u.y * nn.z - u.z * nn.y, u.z * nn.x - u.x * nn.z, u.x * nn.y - u.y * nn.x);
    const double r = df.x * nn.x + df.y * nn.y + df.z * nn.z;
    acc[b + 0] += a.x * a.x;
    acc[b + 1] += a.x * a.y;
    acc[b + 2] += a.x * a.z;
    acc[b + 3] += a.y * a.y;
    acc[b + 4] += a.y * a.z;
    acc[b + 5] += a.z * a.z;
    acc[b + 6] += a.x * nn.x;
    acc[b + 7] += a.x * nn.y;
    acc[b + 8] += a.x * nn.z;
    acc[b + 9] += a.y * nn.x;
    acc[b + 10] += a.y * nn.y;
    acc[b + 11] += a.y * nn.z;
    acc[b + 12] += a.z * nn.x;
    acc[b + 13] += a.z * nn.y;
    acc[b + 14] += a.z * nn.z;
    acc[b + 15] += nn.x * nn.x;
    acc[b + 16] += nn.x * nn.y;
    acc[b + 17] += nn.x * nn.z;
    acc[b + 18] += nn.y * nn.y;
    acc[b + 19] += nn.y * nn.z;
    acc[b + 20] += nn.z * nn.z;
    acc[b + 21] += a.x * r;
    acc[b + 22] += a.y * r;
    acc[b + 23] += a.z * r;
    acc[b + 24] += nn.x * r;
    acc[b + 25] += nn.y * r;
    acc[b + 26] += nn.z * r;
  }
}

// The same moments as the Gram matrix of a 16-vector per accepted row,
//   f = [a(3), nn(3), r, 1, sp(3), tp(3), |s - t|, 0]
// (a = u x nn, r = (s - t).nn, with u and nn as add_row has them; a, nn, r zero for the
// point-to-point metric): sum over rows of f f^T holds every sum add_row keeps (count = the
// 1.1 entry, the plain sums in the row of the 1, the second moments where their factors
// meet), and a matrix core accumulates it in FOUR accumulator registers per lane where the
// per-lane sums take 35 (v_mfma_f64_16x16x4_f64: A[i][k] and B[k][j] one double per lane, i
// = j = lane & 15, k = lane >> 4; D[(lane >> 4) + 4 r][lane & 15] in register r).  A rejected
// row contributes the zero vector.
constexpr int FEAT = 16;
constexpr int FEAT_STRIDE = 65; // doubles per feature row in LDS: 64 queries + 1 (banks)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int METRIC>
__device__ __forceinline__ void row_features(double* f, const d3& s, const d3& t, const d3& n,
  const d3& sn, const double* o)
{
  const d3 sp = mk(s.x - o[0], s.y - o[1], s.z - o[2]);
  const d3 tp = mk(t.x - o[0], t.y - o[1], t.z - o[2]);
  const d3 df = mk(s.x - t.x, s.y - t.y, s.z - t.z);
#pragma unroll
  for (int i = 0; i < 7; i++)
    f[i] = 0.0;
  f[7] = 1.0;
  f[8] = sp.x;
  f[9] = sp.y;
  f[10] = sp.z;
  f[11] = tp.x;
  f[12] = tp.y;
  f[13] = tp.z;
  f[14] = sqrt(df.x * df.x + df.y * df.y + df.z * df.z);
  f[15] = 0.0;
  if (METRIC != 0)
  {
    d3 nn = n, u = sp;
    if (METRIC == 2)
    {
      nn = mk(sn.x + n.x, sn.y + n.y, sn.z + n.z);
      u = mk(sp.x + tp.x, sp.y + tp.y, sp.z + tp.z);
    }
    f[0] = u.y * nn.z - u.z * nn.y;
    f[1] = u.z * nn.x - u.x * nn.z;
    f[2] = u.x * nn.y - u.y * nn.x;
    f[3] = nn.x;
    f[4] = nn.y;
    f[5] = nn.z;
    f[6] = df.x * nn.x + df.y * nn.y + df.z * nn.z;
  }
}

// Where sum i of the metric's layout (add_row's) sits in the Gram matrix: row * 16 + column.
template <int METRIC>
__device__ __forceinline__ int gram_slot(int i)
{
  constexpr int ONE = 7, SP = 8, TP = 11, DIST = 14, A = 0, N = 3, R = 6;
  if (i == 0)
    return ONE * 16 + ONE;
  if (i == 1)
    return ONE * 16 + DIST;
  if (i < 5)
    return ONE * 16 + SP + (i - 2);
  if (METRIC == 0)
  {
    if (i < 8)
      return ONE * 16 + TP + (i - 5);
    return (SP + (i - 8) / 3) * 16 + TP + (i - 8) % 3;
  }
  int b = 5;
  if (METRIC == 2)
  {
    if (i < 8)
      return ONE * 16 + TP + (i - 5);
    b = 8;
  }
  const int k = i - b;
  // (xx, xy, xz, yy, yz, zz)
  const int up_r[6] = { 0, 0, 0, 1, 1, 2 }, up_c[6] = { 0, 1, 2, 1, 2, 2 };
  if (k < 6)
    return (A + up_r[k]) * 16 + A + up_c[k];
  if (k < 15)
    return (A + (k - 6) / 3) * 16 + N + (k - 6) % 3;
  if (k < 21)
    return (N + up_r[k - 15]) * 16 + N + up_c[k - 15];
  if (k < 24)
    return (A + (k - 21)) * 16 + R;
  return (N + (k - 24)) * 16 + R;
}

constexpr int FIN_THREADS = 256;

// The block a finalize block stands in for.  Blocks go to the chip's eight XCDs in turn (block b
// runs on XCD b mod 8, each with an L2 of its own), and a block's stretches of 256 queries
// are b, b + gridDim, ...: with the blocks taken as they come, neighbouring stretches of the
// curve order, whose winners are neighbouring triangles that share lines of the triangle
// records, are read through eight different L2s.  Dealt out this way the blocks of one XCD
// hold a run of consecutive stretches, and what a stretch shares with its neighbours is
// fetched from memory once.  Which queries make up which partial sum does not change (block
// vb's work is done by another block, not another work), so the sums keep their bits.
__device__ __forceinline__ unsigned fin_block()
{
  const unsigned g = gridDim.x, b = blockIdx.x;
  return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
}

// Two builds of the fused finalize.  k_finalize keeps the metric's sums in registers (35
// doubles per lane at most: 168 registers, three wavefronts per SIMD); k_finalize_mx hands
// the rows to the matrix core (four accumulator registers: 117, four wavefronts per SIMD).
// The matrix core works off 16 x 16 products per row where the sums need 45, on the same
// fp64 units: where every lane has a few rows to do (a million points over the whole chip)
// the kernel is bound by its chain of dependent reads and the fourth wavefront wins (36 ->
// 31 us); where it has dozens (sixteen million points) it is bound by arithmetic and the
// matrix core's surplus loses (358 -> 414 us).  The host picks by the size of the cloud.
template <int METRIC, bool ROWS>
__global__ __launch_bounds__(FIN_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_finalize(FinalArgs a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  const unsigned vb = fin_block();
  constexpr int NS = SumCfg<METRIC>::N;
  double acc[NS];
#pragma unroll
  for (int i = 0; i < NS; i++)
    acc[i] = 0.0;

  const int64_t stride = (int64_t)gridDim.x * FIN_THREADS;
  for (int64_t p = (int64_t)vb * FIN_THREADS + threadIdx.x; p < a.nq; p += stride)
  {
    const d3 x = xform(T, a.qpos + 3 * p);
    const int32_t hs = a.hint[p];
    const int64_t k = a.qk[p];
    bool accept = false;
    float of[3] = { (float)x.x, (float)x.y, (float)x.z };
    float cf[3] = { 0.f, 0.f, 0.f }, nf[3] = { 0.f, 0.f, 0.f }, sf[3] = { 0.f, 0.f, 0.f };
    int32_t cell = -1;
    double dist = 0.0;
    if (hs >= 0)
    {
      d3 P0, P1, P2, pcl, cp;
      double d2, w[3], d2b;
      load_tri(a.tverts, hs, P0, P1, P2);
      const float4 fr = a.fnpos[hs]; // (with the triangle: the read does not wait behind the evaluations)
      tri_eval(x, P0, P1, P2, pcl, d2, w);   // closest point p and vlen2 (:375)
      dist = sqrt(d2);                        // :380
      tri_eval(pcl, P0, P1, P2, cp, d2b, w);  // weights at p, closestPoint (:388)
      cell = __float_as_int(fr.w);
      const bool z0 = fabs(w[0]) < 1e-12, z1 = fabs(w[1]) < 1e-12, z2 = fabs(w[2]) < 1e-12;
      const int count = (z0 ? 1 : 0) + (z1 ? 1 : 0) + (z2 ? 1 : 0);
      const float* np = nullptr;
      if (count == 0)
      {
        nf[0] = fr.x;
        nf[1] = fr.y;
        nf[2] = fr.z;
      }
      else if (count == 1)
      {
        int e = z0 ? 0 : (z1 ? 1 : 2);
        np = a.enorm + 3 * (3 * (int64_t)cell + e);
      }
      else if (count == 2)
      {
        int i = !z2 ? 2 : (!z1 ? 1 : 0); // the last weight above the tolerance (:460-466)
        np = a.vnorm + 3 * (int64_t)a.tri[3 * (int64_t)cell + i];
      }
      if (np)
      {
        nf[0] = np[0];
        nf[1] = np[1];
        nf[2] = np[2];
      }
      cf[0] = (float)cp.x;
      cf[1] = (float)cp.y;
      cf[2] = (float)cp.z;
      accept = dist < a.max_dist;
      if (a.qsn)
      {
        sf[0] = a.qsn[3 * p];
        sf[1] = a.qsn[3 * p + 1];
        sf[2] = a.qsn[3 * p + 2];
        if (a.rotate_sn)
        {
          // the normal of the moved point: T's 3x3 applied in fp64, rounded to float as the
          // stored normals are (the reference leaves source normals unrotated)
          const double sx = sf[0], sy = sf[1], sz = sf[2];
          sf[0] = (float)(T.m[0] * sx + T.m[1] * sy + T.m[2] * sz);
          sf[1] = (float)(T.m[4] * sx + T.m[5] * sy + T.m[6] * sz);
          sf[2] = (float)(T.m[8] * sx + T.m[9] * sy + T.m[10] * sz);
        }
        if (a.use_sn)
        {
          double dn = (double)nf[0] * (double)sf[0] + (double)nf[1] * (double)sf[1] +
            (double)nf[2] * (double)sf[2];
          accept = accept && (dn > a.min_dot);
        }
      }
    }
    if (ROWS)
    {
      a.o_flag[k] = accept ? 1u : 0u;
      if (accept)
      {
        for (int c = 0; c < 3; c++)
        {
          a.o_origin[3 * k + c] = of[c];
          a.o_closest[3 * k + c] = cf[c];
          a.o_normal[3 * k + c] = nf[c];
          if (a.o_onormal)
            a.o_onormal[3 * k + c] = sf[c];
        }
        a.o_cell[k] = cell;
        a.o_dist[k] = dist;
      }
    }
    else if (accept)
    {
      add_row<METRIC>(acc, mk((double)of[0], (double)of[1], (double)of[2]),
        mk((double)cf[0], (double)cf[1], (double)cf[2]),
        mk((double)nf[0], (double)nf[1], (double)nf[2]),
        mk((double)sf[0], (double)sf[1], (double)sf[2]), a.origin);
    }
  }
  if (ROWS)
    return;

  // fixed-shape reduction: lanes by xor-shuffle, waves through LDS, one partial per block
  __shared__ double red[FIN_THREADS / 64][NS];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NS; i++)
  {
    double v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
      v += __shfl_xor(v, m, 64);
    if (lane == 0)
      red[wid][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < ICPX_NSUMS)
  {
    double v = 0.0;
    if (threadIdx.x < NS)
      for (int w = 0; w < FIN_THREADS / 64; w++)
        v += red[w][threadIdx.x];
    // (sum-major: the reduction's wavefront then reads 64 consecutive doubles, not one per line)
    a.partials[(int64_t)threadIdx.x * gridDim.x + vb] = v;
  }
}


//
// LEAN: the whole pass in this one kernel, for a registration that has settled: nearly every
// query is expected to stand on its certificate (lean.hpp::lean_settle says what happens to
// each).  A query that cannot be settled there raises *lean_fail: the pass's sums are then
// worth nothing and the host repeats the pass with the search kernels.  Rows, their order
// and the shape of the sums are those of the plain build, so a lean pass that succeeds
// leaves the sums the search kernels and the plain build would have left.
template <int METRIC, bool ROWS, bool LEAN = false>
__global__ __launch_bounds__(FIN_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k_finalize_mx(FinalArgs a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  const unsigned vb = fin_block();
  Xf Tc = T;
  bool far_on = false, lean_bad = false;
  if (LEAN)
  {
#pragma unroll
    for (int i = 0; i < 12; i++)
      Tc.m[i] = uniform_f64(a.cert_T[i]);
    lean_bad = __builtin_amdgcn_readfirstlane(*a.cert_ok) == 0 || a.lean_inject != 0; // (no certificate counts)
    far_on = a.cap2 <= uniform_f64(*a.cert_cap2) * (1.0 + 1e-12);
  }
  constexpr int NS = SumCfg<METRIC>::N;
  // (one region per wavefront: the rows' features on their way into the matrix core, and
  // the wavefront's Gram matrix at the end)
  __shared__ double feat[ROWS ? 1 : FIN_THREADS / 64][ROWS ? 1 : FEAT][ROWS ? 1 : FEAT_STRIDE];
  v4d gram = { 0.0, 0.0, 0.0, 0.0 }, gram2 = { 0.0, 0.0, 0.0, 0.0 }; // (two chains: one waits, the other issues)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;

  // (the trip count is the same for every lane of a wavefront: the matrix core takes all 64)
  const int64_t stride = (int64_t)gridDim.x * FIN_THREADS;
  if (LEAN)
  {
    // A loop of its own ahead of the rows': the wavefront's queries are held against their
    // certificates, and the few whose winner has to be worked out (lean.hpp) get it
    // written to hint[], where the rows' loop below finds it.  (In one loop with the rows
    // the rare paths cost every query registers: 55 us a pass instead of 37.)
    lean_certify(a, T, Tc, far_on, (int64_t)vb * FIN_THREADS + (threadIdx.x & ~63), stride,
      lane, reinterpret_cast<int32_t*>(&feat[wid][0][0]), lean_bad);
    if (__ballot(lean_bad) != 0ull && lane == 0)
      *a.lean_fail = 1;
  }
  else if (!ROWS && a.setmask)
    weigh_marked(a, T, (int64_t)vb * FIN_THREADS + (threadIdx.x & ~63), stride, lane,
      reinterpret_cast<int32_t*>(&feat[wid][0][0]), 1024);
  for (int64_t p0 = (int64_t)vb * FIN_THREADS + (threadIdx.x & ~63); p0 < a.nq; p0 += stride)
  {
    const int64_t p = min(p0 + lane, a.nq - 1);
    const bool valid = p0 + lane < a.nq;
    const d3 x = xform(T, a.qpos + 3 * p);
    const int32_t hs = valid ? a.hint[p] : -1;
    const int64_t k = a.qk[p];
    bool accept = false;
    float of[3] = { (float)x.x, (float)x.y, (float)x.z };
    float cf[3] = { 0.f, 0.f, 0.f }, nf[3] = { 0.f, 0.f, 0.f }, sf[3] = { 0.f, 0.f, 0.f };
    int32_t cell = -1;
    double dist = 0.0;
    if (hs >= 0)
    {
      d3 P0, P1, P2, pcl, cp;
      double d2, w[3], d2b;
      load_tri(a.tverts, hs, P0, P1, P2);
      // (with the triangle, so that the read does not wait behind the evaluations; the
      // point-to-point build has no register to spare for that, and a spilled register costs
      // a kernel more than the wait)
      float4 fr = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (METRIC != 0)
        fr = a.fnpos[hs];
      tri_eval(x, P0, P1, P2, pcl, d2, w);   // closest point p and vlen2 (:375)
      dist = sqrt(d2);                        // :380
      tri_eval(pcl, P0, P1, P2, cp, d2b, w);  // weights at p, closestPoint (:388)
      if constexpr (METRIC == 0)
        fr = a.fnpos[hs];
      cell = __float_as_int(fr.w);
      const bool z0 = fabs(w[0]) < 1e-12, z1 = fabs(w[1]) < 1e-12, z2 = fabs(w[2]) < 1e-12;
      const int count = (z0 ? 1 : 0) + (z1 ? 1 : 0) + (z2 ? 1 : 0);
      const float* np = nullptr;
      if (count == 0)
      {
        nf[0] = fr.x;
        nf[1] = fr.y;
        nf[2] = fr.z;
      }
      else if (count == 1)
      {
        int e = z0 ? 0 : (z1 ? 1 : 2);
        np = a.enorm + 3 * (3 * (int64_t)cell + e);
      }
      else if (count == 2)
      {
        int i = !z2 ? 2 : (!z1 ? 1 : 0); // the last weight above the tolerance (:460-466)
        np = a.vnorm + 3 * (int64_t)a.tri[3 * (int64_t)cell + i];
      }
      if (np)
      {
        nf[0] = np[0];
        nf[1] = np[1];
        nf[2] = np[2];
      }
      cf[0] = (float)cp.x;
      cf[1] = (float)cp.y;
      cf[2] = (float)cp.z;
      accept = dist < a.max_dist;
      if (a.qsn)
      {
        sf[0] = a.qsn[3 * p];
        sf[1] = a.qsn[3 * p + 1];
        sf[2] = a.qsn[3 * p + 2];
        if (a.rotate_sn)
        {
          // the normal of the moved point: T's 3x3 applied in fp64, rounded to float as the
          // stored normals are (the reference leaves source normals unrotated)
          const double sx = sf[0], sy = sf[1], sz = sf[2];
          sf[0] = (float)(T.m[0] * sx + T.m[1] * sy + T.m[2] * sz);
          sf[1] = (float)(T.m[4] * sx + T.m[5] * sy + T.m[6] * sz);
          sf[2] = (float)(T.m[8] * sx + T.m[9] * sy + T.m[10] * sz);
        }
        if (a.use_sn)
        {
          double dn = (double)nf[0] * (double)sf[0] + (double)nf[1] * (double)sf[1] +
            (double)nf[2] * (double)sf[2];
          accept = accept && (dn > a.min_dot);
        }
      }
    }
    if (ROWS)
    {
      if (valid)
        a.o_flag[k] = accept ? 1u : 0u;
      if (accept)
      {
        for (int c = 0; c < 3; c++)
        {
          a.o_origin[3 * k + c] = of[c];
          a.o_closest[3 * k + c] = cf[c];
          a.o_normal[3 * k + c] = nf[c];
          if (a.o_onormal)
            a.o_onormal[3 * k + c] = sf[c];
        }
        a.o_cell[k] = cell;
        a.o_dist[k] = dist;
      }
    }
    else
    {
      double f[FEAT];
#pragma unroll
      for (int i = 0; i < FEAT; i++)
        f[i] = 0.0;
      if (accept)
        row_features<METRIC>(f, mk((double)of[0], (double)of[1], (double)of[2]),
          mk((double)cf[0], (double)cf[1], (double)cf[2]),
          mk((double)nf[0], (double)nf[1], (double)nf[2]),
          mk((double)sf[0], (double)sf[1], (double)sf[2]), a.origin);
      // feature-major in LDS, then four rows at a time into the matrix core: lane l hands it
      // feature l & 15 of row 4 g + (l >> 4), as the A and as the B operand
#pragma unroll
      for (int i = 0; i < FEAT; i++)
        feat[wid][i][lane] = f[i];
      __builtin_amdgcn_wave_barrier(); // (a wavefront's LDS accesses complete in order)
#pragma unroll
      for (int g = 0; g < 16; g += 2)
      {
        const double v = feat[wid][lane & 15][4 * g + (lane >> 4)];
        const double v2 = feat[wid][lane & 15][4 * g + 4 + (lane >> 4)];
        gram = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, gram, 0, 0, 0);
        gram2 = __builtin_amdgcn_mfma_f64_16x16x4f64(v2, v2, gram2, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (ROWS)
    return;

  // fixed-shape reduction: the wavefronts' Gram matrices through LDS, the metric's sums
  // picked out of them, one partial per block
  double* const mine = &feat[wid][0][0];
#pragma unroll
  for (int r = 0; r < 4; r++)
    mine[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = gram[r] + gram2[r];
  __syncthreads();
  if (threadIdx.x < ICPX_NSUMS)
  {
    double v = 0.0;
    if (threadIdx.x < NS)
    {
      const int slot = gram_slot<METRIC>((int)threadIdx.x);
      for (int w = 0; w < FIN_THREADS / 64; w++)
        v += (&feat[w][0][0])[slot];
    }
    // (sum-major: the reduction's wavefront then reads 64 consecutive doubles, not one per line)
    a.partials[(int64_t)threadIdx.x * gridDim.x + vb] = v;
  }
}

// k_search3w's pool counters among the pass counters: one per run of batches, in the word
// after a segment count of k_certify's list (256 bytes apart; zero between passes)
constexpr int XPOOLS = 8;
constexpr int XPOOL_FIRST = 17;
constexpr int XPOOL_STRIDE = 64;
constexpr uint32_t XCHUNK = 128; // consecutive batches (of 64 queries) that belong to one run
// ... and, in the word after the first two pool counters, what the walk that can give up on a
// batch knows of the pass so far: the time its finished batches took (units of 256 shader
// cycles) and their number (zero between passes)
constexpr int WALK_TIME = XPOOL_FIRST + 1;
constexpr int WALK_DONE = XPOOL_FIRST + 1 + XPOOL_STRIDE;

// Block partials -> the ICPX_NSUMS totals.  One 64-lane block per sum: lane l adds
// partials l, l+64, ... in that order, then a fixed xor-shuffle tree; the shape
// depends only on the number of blocks, so results repeat run to run.
// cert_mode says what the pass did to the shortcut's certificates: 1 brought them up to
// date for the transform T (recorded in cert_T), 0 left them behind (cert_ok is cleared).
__device__ __forceinline__ bool reduce_partials_body(
  const double* __restrict__ partials, int nblocks, double* __restrict__ sums,
  const int32_t* __restrict__ stop, uint32_t* __restrict__ counters, double* __restrict__ stat,
  const double* __restrict__ T, double* __restrict__ cert_T, int32_t* __restrict__ cert_ok,
  int cert_mode, double cap2 = 0.0, double* __restrict__ cert_cap2 = nullptr, int share = 0,
  int64_t nq = 0, const int32_t* __restrict__ lean_fail = nullptr, bool skip = false)
{
  const int i = blockIdx.x, l = threadIdx.x;
  // (a lane's partials are requested sixteen at a time, ahead of the flags that say whether
  // the pass ran at all, and added in the order they always were: one after the other each
  // waits a trip to the other XCDs' memory side, 16 x 0.35 us for the 1024 blocks of a
  // million points, a third of the whole launch)
  const double* const mine = partials + (int64_t)i * nblocks;
  double t[16];
#pragma unroll
  for (int k = 0; k < 16; k++)
    t[k] = mine[min(l + 64 * k, nblocks - 1)]; // (no branch around a read: all sixteen go out together)
  const bool stopped = stop && *stop;
  // (block 0's counters with them: it has the most to do after the flags and is the one the
  // others wait for)
  uint32_t c_old = 0u, c_len = 0u;
  if (counters && i == 0)
  {
    c_old = counters[l & 3];
    c_len = counters[16 + (l & 31) * 64];
  }
  asm volatile("" : "+v"(c_old), "+v"(c_len));
  // (the values are asked for here, so that the compiler cannot move the reads behind the test
  // of the flags: reads and flags are then one trip to memory, not one after the other)
#pragma unroll
  for (int k = 0; k < 16; k++)
    asm volatile("" : "+v"(t[k]));
  if (stopped || skip)
    return false; // the pass did not run, or its sums are worth nothing
  // share: the loop of several ranks.  The spare slots of the vector carry this rank's policy
  // figures through the all-reduce (state.hpp: SLOT_*): the query count and the lean pass's
  // flag from their own blocks, the three counts from block 0, which moves the counters.
  if (share && i >= SLOT_NQ)
  {
    if (l == 0 && i == SLOT_NQ)
      sums[i] = (double)nq;
    if (l == 0 && i == SLOT_LEANFAIL)
      sums[i] = (lean_fail && *reinterpret_cast<const volatile int32_t*>(lean_fail) != 0) ? 1.0 : 0.0;
    return true;
  }
  if (cert_ok && i == 1)
  {
    // (an anchored pass left the certificates as they were: so stays their transform)
    if (cert_mode && l < 12 && cert_ok[2] == 0)
      cert_T[l] = T[l];
    if (l == 0)
    {
      *cert_ok = cert_mode ? 1 : 0;
      if (cert_cap2)
        *cert_cap2 = cap2;
    }
  }
  uint32_t old = 0u;
  if (counters && i == 0 && l < 4)
  {
    // the search kernels' four pass counters move up for whoever wants to look at them,
    // and are left at zero for the next pass
    old = c_old;
    counters[4 + l] = old;
    counters[l] = 0u;
    if (stat && l == 3)
      stat[2] = (double)old; // shortcut fallbacks of this pass
  }
  if (share && i == 0)
  {
    const uint32_t overflows = (uint32_t)__shfl((int)old, 1, 64);
    if (l == 0)
      sums[SLOT_CONTESTED] = (double)old + (double)overflows; // (ties + overflows)
    if (l == 3)
      sums[SLOT_FALLBACK] = (double)old;
  }
  if (counters && i == 0)
  {
    // the lengths of the segments of k_certify's list: their sum moves up beside the other
    // counts, and they are left at zero
    uint32_t len = l < 32 ? c_len : 0u;
    if (l < 32)
      counters[16 + l * 64] = 0u;
    if (l < XPOOLS)
      counters[XPOOL_FIRST + l * XPOOL_STRIDE] = 0u; // (k_search3w's pool counters)
    if (l < 2)
      counters[WALK_TIME + l * XPOOL_STRIDE] = 0u; // (... and its batches' time and number)
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1)
      len += __shfl_xor(len, m, 64);
    if (l == 0)
    {
      counters[8] = len;
      if (stat)
        stat[3] = (double)len;
      if (share)
        sums[SLOT_LISTED] = (double)len;
    }
  }
  if (share && !counters && i == 0 && l == 0)
    sums[SLOT_LISTED] = 0.0;
  if (stat && !counters && i == 0 && l == 0)
  {
    stat[2] = 0.0;
    stat[3] = 0.0;
  }
  double v = 0.0;
  for (int b0 = l; b0 < nblocks; b0 += 64 * 16)
  {
    if (b0 != l)
    {
#pragma unroll
      for (int k = 0; k < 16; k++)
        t[k] = mine[min(b0 + 64 * k, nblocks - 1)];
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
      const double w = v + t[k];
      v = (b0 + 64 * k < nblocks) ? w : v;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
    v += __shfl_xor(v, m, 64);
  if (l == 0)
  {
    sums[i] = v;
    // {n, sum |s-t|} (and, above, the shortcut's fallbacks) for the host's policy, in
    // pinned memory
    if (stat && i < 2)
      stat[i] = v;
    if (stat)
      __threadfence_system();
  }
  return true;
}

__global__ __launch_bounds__(64) void k_reduce_partials(
  const double* __restrict__ partials, int nblocks, double* __restrict__ sums,
  const int32_t* __restrict__ stop, uint32_t* __restrict__ counters, double* __restrict__ stat,
  const double* __restrict__ T, double* __restrict__ cert_T, int32_t* __restrict__ cert_ok,
  int cert_mode, double cap2, double* __restrict__ cert_cap2, int share = 0, int64_t nq = 0,
  const int32_t* __restrict__ lean_fail = nullptr)
{
  (void)reduce_partials_body(partials, nblocks, sums, stop, counters, stat, T, cert_T, cert_ok,
    cert_mode, cap2, cert_cap2, share, nq, lean_fail);
}

// A rank of a multi-rank loop that was to run a lean pass and cannot (it holds no certificates
// to check): its sums are zero and its flag says that the pass has failed, so that every rank
// drops the pass and repeats it in full.
__global__ void k_lean_unready(double* __restrict__ sums, const int32_t* __restrict__ stop, int64_t nq)
{
  if (*stop)
    return;
  const int l = threadIdx.x;
  if (l < ICPX_NSUMS)
    sums[l] = l == SLOT_LEANFAIL ? 1.0 : l == SLOT_NQ ? (double)nq : 0.0;
}

// Query set-up: gather the kept points (and normals) of the shard into Morton order.
__global__ void k_gather_queries(const float* __restrict__ xyz, const float* __restrict__ nrm,
  const uint32_t* __restrict__ order, int64_t k0, int64_t step, int64_t nq,
  float* __restrict__ qpos, float* __restrict__ qsn, int32_t* __restrict__ qk,
  int32_t* __restrict__ hint)
{
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nq)
    return;
  uint32_t k = order[p];
  int64_t i = (k0 + k) * step;
  qpos[3 * p] = xyz[3 * i];
  qpos[3 * p + 1] = xyz[3 * i + 1];
  qpos[3 * p + 2] = xyz[3 * i + 2];
  if (nrm)
  {
    qsn[3 * p] = nrm[3 * i];
    qsn[3 * p + 1] = nrm[3 * i + 1];
    qsn[3 * p + 2] = nrm[3 * i + 2];
  }
  qk[p] = (int32_t)k;
  hint[p] = -1;
}

// A shard cut from the curve order of all kept points (icpx_set_shard): `order` holds the
// kept indices of all of them in that order.  Rows come out by ascending source index, so
// every query also needs its rank among the shard's kept indices: they are sorted (key = kept
// index, value = position), and k_shard_slots turns the sorted pairs into slot -> kept index
// and position -> slot.
// Which positions of the curve order are a shard's: chunks of `chunk` consecutive positions
// are dealt out to the ranks in turn (rank r has chunks r, r + world, ...; chunk >= the
// number of kept points / world makes that one range per rank).  Query j of the shard:
struct ShardMap
{
  int64_t chunk;
  int rank, world;
  __device__ __host__ int64_t pos(int64_t j) const
  {
    return ((int64_t)rank + (j / chunk) * world) * chunk + j % chunk;
  }
};
__global__ void k_shard_keys(const uint32_t* __restrict__ order, ShardMap m, int64_t nq,
  uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nq)
    return;
  keys[p] = order[m.pos(p)];
  vals[p] = (uint32_t)p;
}
__global__ void k_shard_slots(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
  int64_t nq, int32_t* __restrict__ kglobal, int32_t* __restrict__ qk)
{
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nq)
    return;
  kglobal[r] = (int32_t)keys[r];
  qk[vals[r]] = (int32_t)r;
}
// ... and the gather for such a shard (qk is already filled in).
__global__ void k_gather_queries_curve(const float* __restrict__ xyz, const float* __restrict__ nrm,
  const uint32_t* __restrict__ order, ShardMap m, int64_t step, int64_t nq,
  float* __restrict__ qpos, float* __restrict__ qsn, int32_t* __restrict__ hint)
{
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nq)
    return;
  int64_t i = (int64_t)order[m.pos(p)] * step;
  qpos[3 * p] = xyz[3 * i];
  qpos[3 * p + 1] = xyz[3 * i + 1];
  qpos[3 * p + 2] = xyz[3 * i + 2];
  if (nrm)
  {
    qsn[3 * p] = nrm[3 * i];
    qsn[3 * p + 1] = nrm[3 * i + 1];
    qsn[3 * p + 2] = nrm[3 * i + 2];
  }
  hint[p] = -1;
}

// K8: accepted rows, already in ascending kept-index order, packed to the front.
struct CompactArgs
{
  int64_t nq, k0, step;
  const int32_t* kglobal; // slot -> kept index of a shard that is not a block of them (or null)
  const uint32_t* flag;
  const uint32_t* pos; // exclusive scan of flag
  const float *origin, *onormal, *closest, *normal;
  const int32_t* cell;
  const double* dist;
  float *d_origin, *d_onormal, *d_closest, *d_normal;
  int32_t* d_cell;
  double* d_dist;
  int64_t* d_index;
};

__global__ void k_compact_rows(CompactArgs a)
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
    if (a.d_onormal)
      a.d_onormal[3 * r + c] = a.onormal[3 * k + c];
  }
  a.d_cell[r] = a.cell[k];
  a.d_dist[r] = a.dist[k];
  a.d_index[r] = (a.kglobal ? (int64_t)a.kglobal[k] : a.k0 + k) * a.step;
}

} // namespace icpx
