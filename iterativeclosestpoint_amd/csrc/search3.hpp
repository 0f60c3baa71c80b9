// search3.hpp — closest-triangle search for large clouds: one lane per query.
//
// With millions of queries the scarce resource is wave-instructions per query,
// so here every lane walks its own query through a complete *binary* box
// hierarchy (heap order: children of n are 2n, 2n+1; a node record holds both
// children's boxes in one 64-byte line).  There is no stack: a 32-bit trail
// keeps one bit per level ("the other child of this ancestor is still owed a
// visit"), and coming back up re-tests that child's box from its parent's
// record against the bound as it stands by then.  Queries are in Morton order,
// so neighbouring lanes read the same lines and leave the loop together.
//
// Screening and exactness are as in search2.hpp: fp32 box distances that can
// only under-estimate, a short fp64 routine to screen triangles, and the exact
// routine of tri.hpp only for near-ties (within 1e-9 x the largest coordinate
// of the minimum).  A query whose minimum is not contested is finished by this
// kernel; contested ones (2..8 contenders) are listed and settled by
// k_contested (resolve_ties_body), eight lanes per query, one contender per lane; the rare query
// with more contenders than the ring holds is re-run by search3_body in its
// EXACT mode (k_contested's later blocks), which knows the final bound and evaluates every contender
// exactly as it meets it.
//
// Replaces vtkCellLocator::FindClosestPoint as called from
// ICP/vtkImplicitPolyDataDistance2.cxx:375.
#pragma once
#include "search2.hpp"

namespace icpx
{

struct BinTree
{
  int depth;     // leaves are heap numbers [2^depth, 2^(depth+1))
  int L;         // triangles per leaf
  int64_t nleaf; // leaves in use
};

inline BinTree make_bin_tree(int64_t nt, int L)
{
  BinTree t;
  t.L = L;
  t.nleaf = (nt + L - 1) / L;
  if (t.nleaf < 1)
    t.nleaf = 1;
  int d = 1;
  while ((1ll << d) < t.nleaf)
    d++;
  t.depth = d;
  return t;
}

constexpr int TPRE_PAD = 4; // bounds records past the last triangle (a leaf is read whole)
constexpr int NODE_FLOATS = 16; // lo0.xyz hi0.xyz lo1.xyz hi1.xyz + 4 pad

// Bottom node level: boxes of the two leaves under node n in [2^(d-1), 2^d).
__global__ void k_bin_bottom(const float* __restrict__ tverts, const int32_t* __restrict__ tid,
  int64_t nt, BinTree t, float* __restrict__ nodes)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // leaf slot
  if (i >= (1ll << t.depth))
    return;
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  const int64_t s0 = i * t.L, s1 = min(nt, s0 + t.L);
  bool any = false;
  for (int64_t s = s0; s < s1; s++)
  {
    if (tid[s] < 0)
      continue;
    any = true;
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++)
      {
        const float v = tverts[9 * s + 3 * k + a];
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
  const int64_t leaf_heap = (1ll << t.depth) + i;
  float* r = nodes + (leaf_heap >> 1) * NODE_FLOATS + 6 * (leaf_heap & 1);
  for (int a = 0; a < 3; a++)
  {
    r[a] = lo[a];
    r[3 + a] = hi[a];
  }
}

// Level k (nodes [2^k, 2^(k+1))), k < depth-1: child boxes = union of the child's record.
__global__ void k_bin_level(int k, float* __restrict__ nodes)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (2ll << k))
    return;
  const int64_t child = (2ll << k) + i; // a node of level k+1
  const float* c = nodes + child * NODE_FLOATS;
  float* r = nodes + (child >> 1) * NODE_FLOATS + 6 * (child & 1);
  for (int a = 0; a < 3; a++)
  {
    r[a] = fminf(c[a], c[6 + a]);
    r[3 + a] = fmaxf(c[3 + a], c[9 + a]);
  }
}

// Per triangle (Morton position), eight floats: unit normal and the allowance for the
// stored plane's offset, then centroid and the radius of the circle about it that holds
// the triangle; for a triangle too thin for its normal to be trusted the normal is zero.
// The allowance is positive for every triangle that can be a closest cell; records of
// degenerate triangles and the TPRE_PAD records past the end carry -1 there.
// From these a query gets a lower bound of its distance to the triangle that, unlike a
// box's, does not loosen when the triangle is slanted:
// (distance to the plane)^2 + (in-plane distance to the circle)^2.
// cell[c] = first position of the sorted codes whose code is not below c << (63 - bits), for
// every prefix c of `bits` bits (and cell[1 << bits] = nt).
__global__ void k_code_cells(const uint64_t* __restrict__ codes, int64_t nt, int bits,
  uint32_t* __restrict__ cell)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ((int64_t)1 << bits))
    return;
  const uint64_t key = (uint64_t)c << (63 - bits);
  int64_t lo = 0, hi = nt;
  while (lo < hi)
  {
    const int64_t mid = (lo + hi) >> 1;
    if (codes[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  cell[c] = (c == ((int64_t)1 << bits)) ? (uint32_t)nt : (uint32_t)lo;
}

__global__ void k_tri_bounds(const float* __restrict__ tverts, const int32_t* __restrict__ tid,
  int64_t nt, int64_t npad, float4* __restrict__ pre)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= npad)
    return;
  if (s >= nt || tid[s] < 0)
  {
    // not a candidate (padding, or a degenerate triangle): a negative allowance says so
    pre[2 * s] = make_float4(0.f, 0.f, 0.f, -1.f);
    pre[2 * s + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  d3 P0, P1, P2;
  load_tri(tverts, s, P0, P1, P2);
  const d3 c = mk((P0.x + P1.x + P2.x) / 3.0, (P0.y + P1.y + P2.y) / 3.0,
    (P0.z + P1.z + P2.z) / 3.0);
  const float cx = (float)c.x, cy = (float)c.y, cz = (float)c.z;
  const d3 cf = mk((double)cx, (double)cy, (double)cz); // the centre as stored
  const double r2 = fmax(dist2(P0, cf), fmax(dist2(P1, cf), dist2(P2, cf)));
  const float R = __double2float_ru(sqrt(r2) * (1.0 + 1e-12));
  const d3 e1 = mk(P1.x - P0.x, P1.y - P0.y, P1.z - P0.z);
  const d3 e2 = mk(P2.x - P0.x, P2.y - P0.y, P2.z - P0.z);
  const d3 e3 = mk(P2.x - P1.x, P2.y - P1.y, P2.z - P1.z);
  const d3 n = mk(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
  const double len = sqrt(dot3(n, n));
  const double emax2 = fmax(dot3(e1, e1), fmax(dot3(e2, e2), dot3(e3, e3)));
  // (a cross product this small against its edges has lost too many digits of direction)
  const bool ok = len > 1e-6 * emax2;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if (ok)
  {
    nx = (float)(n.x / len);
    ny = (float)(n.y / len);
    nz = (float)(n.z / len);
  }
  // the plane the search uses passes through the stored centre; its distance from the
  // true plane goes into the fourth float as an allowance
  const double off = ok ? fabs((double)nx * (cf.x - P0.x) + (double)ny * (cf.y - P0.y) +
                             (double)nz * (cf.z - P0.z)) : 0.0;
  pre[2 * s] = make_float4(nx, ny, nz, __double2float_ru(off * (1.0 + 1e-6) + 1e-300));
  pre[2 * s + 1] = make_float4(cx, cy, cz, R);
}

// Per leaf of a tree with L triangles per leaf, the same kind of record for the leaf's whole
// patch of surface: the unit normal of the patch (normalised sum of its triangles' area
// vectors; zero when that cannot be trusted), the half thickness of the slab about the
// stored centre, across that normal, that holds every point of its triangles, then the
// centre and the radius of the sphere about it that holds them.  An axis-aligned box
// around a slanted patch is thick, and a query many triangle widths off the surface then
// has dozens of leaf boxes within its bound; the slab of a smooth patch is a few percent of
// its width thick whatever its slant.  A leaf without a usable triangle carries -1 as its
// half thickness.
__global__ void k_leaf_bounds(const float* __restrict__ tverts, const int32_t* __restrict__ tid,
  int64_t nt, BinTree t, float4* __restrict__ lpre)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.nleaf)
    return;
  const int64_t s0 = i * t.L, s1 = min(nt, s0 + t.L);
  d3 c = mk(0.0, 0.0, 0.0), n = mk(0.0, 0.0, 0.0);
  int cnt = 0;
  double emax2 = 0.0;
  for (int64_t s = s0; s < s1; s++)
  {
    if (tid[s] < 0)
      continue;
    d3 P0, P1, P2;
    load_tri(tverts, s, P0, P1, P2);
    c = mk(c.x + P0.x + P1.x + P2.x, c.y + P0.y + P1.y + P2.y, c.z + P0.z + P1.z + P2.z);
    const d3 e1 = mk(P1.x - P0.x, P1.y - P0.y, P1.z - P0.z);
    const d3 e2 = mk(P2.x - P0.x, P2.y - P0.y, P2.z - P0.z);
    n = mk(n.x + (e1.y * e2.z - e1.z * e2.y), n.y + (e1.z * e2.x - e1.x * e2.z),
      n.z + (e1.x * e2.y - e1.y * e2.x));
    emax2 = fmax(emax2, fmax(dot3(e1, e1), dot3(e2, e2)));
    cnt++;
  }
  if (cnt == 0)
  {
    lpre[2 * i] = make_float4(0.f, 0.f, 0.f, -1.f);
    lpre[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float cx = (float)(c.x / (3.0 * cnt)), cy = (float)(c.y / (3.0 * cnt)),
              cz = (float)(c.z / (3.0 * cnt));
  const d3 cf = mk((double)cx, (double)cy, (double)cz); // the centre as stored
  const double len = sqrt(dot3(n, n));
  // (area vectors that cancel, or a patch of slivers: no direction to speak of)
  const bool ok = len > 1e-6 * emax2;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  if (ok)
  {
    nx = (float)(n.x / len);
    ny = (float)(n.y / len);
    nz = (float)(n.z / len);
  }
  double r2 = 0.0, half = 0.0;
  for (int64_t s = s0; s < s1; s++)
  {
    if (tid[s] < 0)
      continue;
    d3 P[3];
    load_tri(tverts, s, P[0], P[1], P[2]);
    for (int k = 0; k < 3; k++)
    {
      r2 = fmax(r2, dist2(P[k], cf));
      // every point of the patch is a convex combination of these: its offset along the
      // stored normal lies between the smallest and the largest of theirs
      half = fmax(half, fabs((double)nx * (P[k].x - cf.x) + (double)ny * (P[k].y - cf.y) +
                               (double)nz * (P[k].z - cf.z)));
    }
  }
  lpre[2 * i] = make_float4(nx, ny, nz, __double2float_ru(half * (1.0 + 1e-6) + 1e-300));
  lpre[2 * i + 1] = make_float4(cx, cy, cz, __double2float_ru(sqrt(r2) * (1.0 + 1e-12)));
}

struct Search3Args
{
  const float* qpos;
  int64_t nq;
  // a kernel given a list serves it only if its length is in [list_lo, list_hi): two
  // kernels can be enqueued over one list and the one that suits its length does the work
  int64_t list_lo, list_hi;
  XfRef T;
  const float* nodes;
  BinTree tree;
  const float* tverts;
  const int32_t* tid;
  const float4* tpre; // per-triangle bounds records (k_tri_bounds), used by k_search3w
  const float4* lpre; // per-leaf bounds records of this tree (k_leaf_bounds), used by k_search3w
  int64_t nt;
  int32_t* hint;
  int use_hint;
  float scene_max;
  const uint64_t* tcodes; // Morton codes of the triangles in sorted order (cold-start seed), or null
  // ... and, by the codes' leading tcell_bits bits, the first position whose code is not below
  // that prefix (k_code_cells): one read where the search by halves takes twenty, one after
  // the other, each lane its own; null: search
  const uint32_t* tcell;
  int tcell_bits;
  Box3 box;
  // ... and a triangle near each cell of a coarse grid over the bounds (k_seed_flood; null:
  // none).  seed_morton: the curve's seed is looked at as well (the better of the two wins).
  const int32_t* seed_grid;
  int seed_dim;
  int seed_morton;
  // contested queries
  int32_t* cand;        // [nq][RING3]
  uint2* tie_list;      // {query, contenders}
  uint32_t* counters;   // [0] ties, [1] overflows, [XPOOL_FIRST + r * XPOOL_STRIDE] next position of run r's pool
  uint32_t* ovf_list;   // queries with more contenders than the ring holds
  double* ovf_best;     // their screened minimum, indexed by query
  const uint32_t* qlist; // persistent kernel: serve only these queries (count in counters[3]), or null
  int refill;            // pooled kernels: idle lanes of a wavefront that trigger taking new queries
  // The filter drops a correspondence whose distance is not below MaximumDistance
  // (ICP/vtkPolyDataCorrespondenceFilter.cxx:122), so nothing farther than that needs to
  // be found: its square is every query's first bound (DBL_MAX: none).  A query with
  // nothing within it ends without a winner (hint -1), which the finalize kernels drop.
  double cap2;
  // Certificates (see local.hpp): per query, > 0: its winner stands while twice the query's
  // move stays below this; < 0: nothing comes within the cap while the move stays below
  // the magnitude; 0: none.  Written for every query a kernel settles (null: not kept).
  float* cert;
  const double* cert_T;    // the transform the certificates were last brought up to date for,
  const int32_t* cert_ok;  // whether they are up to date at all,
  const double* cert_cap2; // and the cap the far ones were made for (k_search3w checks them
                           // itself; elsewhere k_search_local does before the tree kernel runs)
  // A box turned away just beyond the bound says little about the triangles in it, so a
  // search that is to leave a useful certificate looks a little farther than it must: by
  // eight times the query's move since the last pass, at least two rounding allowances, at
  // most widen_max (a fraction of the mesh's edge length).  Triangles in that margin are
  // screened and counted among "the others" with their true distance.
  float widen_max;
  // Whether a single winner with a rival next to nothing behind may be kept with that rival
  // as a set of two (make_certificate_pair).  Every set costs k_contested an exact
  // evaluation per member and pass; searching the rival-ridden queries again instead costs
  // about 4 ns apiece per pass.  Sets win where those queries are many (a noisy cloud of
  // millions); for a thousand of them the search is lost in a launch that runs anyway.
  int pairs;
  // One bit per query: it stands on a set certificate that passed this pass's check, so its
  // contenders (cand[]) are to be weighed again by k_contested.  Every kernel that checks
  // certificates handles 64 consecutive queries per wavefront and stores their 64 bits as
  // one word, set or not: no counter, no list, nothing to clear.  Null in a pass without
  // such a kernel.
  unsigned long long* setmask;
  // Per query that stands on a set: the lead of the set's winner over its other members, as
  // the last weighing found it (cert.hpp; valid while the certificate's second bit says so).  A
  // check that finds the lead still standing does not mark the query: its winner in hint[]
  // stands without a weighing.  Null: not kept (every certified set is weighed every pass).
  float* lead;
  // A whole pass of the wavefront walk gives up on a batch that has been walking for
  // bail_ticks x 64 cycles of the shader clock (0: never): its queries go on this list (count in
  // counters[bail_slot]) with the best triangle found so far as their start, and the kernel behind the
  // walk (k_fallback) serves them one wavefront per query.  A few queries beside a patch of
  // triangles much smaller than their distance from it have thousands of boxes within their
  // bound; their batch of 64 then walks for as long as the rest of the pass takes together (a
  // mesh whose edge lengths range over a factor of 65: 3 500 node steps against a mean of
  // 100, a cold pass of 15.7 ms instead of 1).
  uint32_t* bail_list;
  uint32_t bail_ticks;
  int bail_mean; // 0: past the deadline is enough (tests); 1: ... and four times the pass's mean batch
  // Which of the pass counters holds the length of qlist (3: the shortcut's / k_certify's
  // leftovers; 2: the list of the queries given up on) and which one a walk that gives up
  // counts on (2: the list is one of its own, so that a walk over a list can give up as well).
  int count_slot, bail_slot;
  // k_fallback: lists up to this length go one wavefront per query, longer ones to the pooled
  // lanes (FALLBACK_SMALL; on a target whose triangles' sizes range widely twice that: a
  // pooled lane has a query with thousands of boxes in reach to itself, and the pass waits
  // for it, 5 ms for a list of 0.1 M where the wavefronts take 2; 0.3 M such queries cost the
  // wavefronts 25 ms and the pooled lanes, which hold that many at once, the same 5)
  uint32_t small_limit;
  // The pooled lanes give up on a query after this many steps of its walk (0: never) and put
  // it on bail_list as the wavefront walk does with a batch: a lane walks alone, a step of its
  // wavefront's loop per box, and a query far off the surface (a stray point in the hole of a
  // torus: tens of thousands of boxes within its bound) keeps its wavefront, and the pass,
  // for 10 ms, which one wavefront per query over the 64-ary tree serves in 0.5.
  uint32_t step_budget;
  // The wavefront walk sets a query aside before its batch's walk (onto bail_list, for the
  // kernel behind) when its first bound, the squared distance of its seed or hinted triangle,
  // is above this (0: never) and 64 times the geometric mean of its batch's: a stray point
  // among points on the surface, whose thousands of boxes the 63 others would sit through.
  float defer_far;
};

// the margin a query's search is widened by (see Search3Args::widen_max)
__device__ __forceinline__ float widen_of(const Search3Args& a, const Xf& T, const float* p,
  const d3& x, float eabs)
{
  if (!a.cert)
    return 0.f;
  float mv = 0.f;
  // (a cold pass has no previous pose to measure the move from)
  if (a.use_hint && __builtin_amdgcn_readfirstlane(*a.cert_ok) != 0)
  {
    Xf Tc;
#pragma unroll
    for (int i = 0; i < 12; i++)
      Tc.m[i] = uniform_f64(a.cert_T[i]);
    const d3 xp = xform(Tc, p);
    const double mx = x.x - xp.x, my = x.y - xp.y, mz = x.z - xp.z;
    mv = __builtin_amdgcn_sqrtf(__double2float_ru(mx * mx + my * my + mz * mz));
  }
  return fminf(fmaxf(8.f * mv, 2.f * eabs), fmaxf(a.widen_max, 2.f * eabs));
}

// The list of queries whose certificates failed (k_certify) is kept in this many segments,
// with counts this many words apart.
constexpr int CERTIFY_SEGS = 32;
// marked queries among a wavefront's 2048 up to which k_contested deals them out to groups
// of eight lanes (beyond: every lane weighs its own)
constexpr uint32_t SPARSE_SETS = 16;
constexpr int SETS_HALVES = 16; // 32-bit words of the set mask a wavefront of k_contested takes at a time
constexpr int CERTIFY_SEG_STRIDE = 64;

// fp32 lower bound of the squared distance to a box; never above the true value
__device__ __forceinline__ float box_lb(float lox, float loy, float loz, float hix, float hiy,
  float hiz, float xf, float yf, float zf, float eabs)
{
  const float dx = fmaxf(fmaxf(fmaxf(lox - xf, xf - hix), 0.f) - eabs, 0.f);
  const float dy = fmaxf(fmaxf(fmaxf(loy - yf, yf - hiy), 0.f) - eabs, 0.f);
  const float dz = fmaxf(fmaxf(fmaxf(loz - zf, zf - hiz), 0.f) - eabs, 0.f);
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) * 0.99999952f;
}

// The tree walk's box test without the per-axis allowance: the raw fp32 squared distance
// to the box, to be compared with a bound that carries the allowance instead.  If each
// axis distance is off by at most eabs, the raw distance exceeds the true one by at most
// sqrt(3)*eabs, so a box whose true distance is within sqrt(bound) has a raw squared
// distance within box_bound(bound, eabs).
__device__ __forceinline__ float box_d2(float lox, float loy, float loz, float hix, float hiy,
  float hiz, float xf, float yf, float zf)
{
  const float dx = fmaxf(fmaxf(lox - xf, xf - hix), 0.f);
  const float dy = fmaxf(fmaxf(loy - yf, yf - hiy), 0.f);
  const float dz = fmaxf(fmaxf(loz - zf, zf - hiz), 0.f);
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ float box_bound(float boundf, float eabs)
{
  // (every rounding below is covered by the two factors)
  const float r = __builtin_sqrtf(boundf) * 1.0000005f + 1.7320509f * eabs;
  return r * r * 1.000002f;
}

template <int L, bool EXACT>
__device__ __forceinline__ void search3_body(const Search3Args& a, unsigned block)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  __shared__ int32_t ring[RING3][256];
  const int tid = threadIdx.x;
  int64_t q = (int64_t)block * 256 + tid;
  bool active;
  if (EXACT)
  {
    const uint32_t novf = a.counters[1];
    active = q < (int64_t)novf;
    if (active)
      q = a.ovf_list[q];
  }
  else
    active = q < a.nq;
  const int d = a.tree.depth;

  d3 x = mk(0.0, 0.0, 0.0);
  float xf = 0.f, yf = 0.f, zf = 0.f, eabs = 0.f;
  double tau = 0.0, best = DBL_MAX, bound = DBL_MAX;
  float boundf = FLT_MAX, bboxf = INFINITY; // bound for triangles / for raw box distances
  int cnt = 0;
  int32_t hs_used = -1;  // the hinted triangle already sits in the ring
  double ed2 = INFINITY; // EXACT mode: exact minimum so far
  int32_t eid = 0x7fffffff, es = -1;

  auto set_best = [&](double nb) {
    const double r = sqrt(nb) + tau;
    const double nbound = r * r;
    if (nbound < best)
      cnt = 0; // every older contender was above the old minimum
    best = nb;
    bound = nbound;
    boundf = fminf(fmaxf(__double2float_ru(nbound), FLT_MIN), FLT_MAX);
    bboxf = box_bound(boundf, eabs);
  };

  if (active)
  {
    x = xform(T, a.qpos + 3 * q);
    xf = (float)x.x;
    yf = (float)x.y;
    zf = (float)x.z;
    const float smax = fmaxf(fmaxf(a.scene_max, fabsf(xf)), fmaxf(fabsf(yf), fabsf(zf)));
    eabs = smax * 2.3841858e-07f * 1.0000002f;
    tau = 1e-9 * (double)smax;
    if (EXACT)
      set_best(a.ovf_best[q]); // the final screened minimum is already known
    else
    {
      if (a.cap2 < 1e300)
        set_best(a.cap2);
      int32_t hs = a.use_hint ? a.hint[q] : -1;
      if (hs < 0 && a.tcodes)
      {
        // cold start: the triangle whose centroid follows the query along the Morton curve
        // gives a first bound, so the walk starts as tight as a hinted one
        const uint64_t code = morton63(xf, yf, zf, a.box.lo, a.box.inv);
        int64_t lo = 0, hi = a.nt; // first position with tcodes[pos] >= code
        while (lo < hi)
        {
          const int64_t mid = (lo + hi) >> 1;
          if (a.tcodes[mid] < code)
            lo = mid + 1;
          else
            hi = mid;
        }
        hs = (int32_t)min(lo, a.nt - 1);
      }
      if (hs >= 0 && a.tid[hs] >= 0)
      {
        d3 P0, P1, P2;
        load_tri(a.tverts, hs, P0, P1, P2);
        const double ds = tri_d2_screen(x, P0, P1, P2);
        if (ds < best)
        {
          set_best(ds);
          ring[0][tid] = hs;
          cnt = 1;
          hs_used = hs;
        }
      }
    }
  }

  // One step of the walk is a single code path for every lane: (1) a lane that has to
  // back up turns to the deepest ancestor still owed its other child (pure register
  // arithmetic; only for a pending *leaf* is the bound known, kept in lbleaf, because at
  // most one leaf-level entry is ever pending); (2) a lane standing on a node reads its
  // record and tests both children.  A far child that is an inner node is entered without
  // re-testing it: if it went stale its own visit rejects both children at once.
  uint32_t n = 1, trail = 0;
  float lbleaf = 0.f;
  bool done = !active, need_bt = false, atleaf = false;
  while (true)
  {
    while (!done && !atleaf)
    {
      if (need_bt)
      {
        if (trail == 0)
          done = true;
        else
        {
          const int k = 31 - __clz(trail);
          trail ^= 1u << k;
          const int lvl = 31 - __clz(n);
          n = (n >> (lvl - k - 1)) ^ 1u; // the child of the level-k ancestor not yet visited
          need_bt = false;
          if (k == d - 1)
          {
            if (lbleaf > bboxf)
              need_bt = true; // went stale meanwhile
            else
              atleaf = true;
          }
        }
      }
      if (!done && !need_bt && !atleaf)
      {
        const float4* p = reinterpret_cast<const float4*>(a.nodes + (int64_t)n * NODE_FLOATS);
        const float4 u = p[0], v = p[1], w = p[2];
        const float lb0 = box_d2(u.x, u.y, u.z, u.w, v.x, v.y, xf, yf, zf);
        const float lb1 = box_d2(v.z, v.w, w.x, w.y, w.z, w.w, xf, yf, zf);
        const bool ok0 = lb0 <= bboxf, ok1 = lb1 <= bboxf;
        if (ok0 || ok1)
        {
          const uint32_t near = (ok1 && (!ok0 || lb1 < lb0)) ? 1u : 0u;
          const int lvl = 31 - __clz(n);
          if (ok0 && ok1)
          {
            trail |= 1u << lvl;
            lbleaf = (lvl == d - 1) ? (near ? lb0 : lb1) : lbleaf;
          }
          n = 2u * n + near;
          atleaf = (lvl == d - 1);
        }
        else
          need_bt = true;
      }
    }
    if (done)
      break;

    // leaf: screen its triangles
    const int64_t s0 = (int64_t)(n - (1u << d)) * L;
#pragma unroll
    for (int t = 0; t < L; t++)
    {
      const int64_t s = s0 + t;
      if (s < a.nt && a.tid[s] >= 0)
      {
        d3 P0, P1, P2;
        load_tri(a.tverts, s, P0, P1, P2);
        const double d2 = tri_d2_screen(x, P0, P1, P2);
        if (EXACT)
        {
          if (d2 <= bound)
          {
            d3 cp;
            double dd, w3[3];
            const int32_t id = a.tid[s];
            const bool up = tri_eval(x, P0, P1, P2, cp, dd, w3) && better(dd, id, ed2, eid);
            ed2 = up ? dd : ed2;
            eid = up ? id : eid;
            es = up ? (int32_t)s : es;
          }
        }
        else
        {
          if (d2 < best)
            set_best(d2);
          // (the hinted triangle is met again during the walk; it is in the ring already,
          // or out of the race for good if the ring restarted since)
          if (d2 <= bound && (int32_t)s != hs_used)
          {
            if (cnt < RING3)
              ring[cnt][tid] = (int32_t)s;
            cnt++;
          }
        }
      }
    }
    need_bt = true;
    atleaf = false;
  }

  if (EXACT)
  {
    if (active)
      a.hint[q] = es;
    return;
  }

  // settle: a single contender wins outright; 2..RING3 go to the tie list; more overflow
  const bool tie = active && cnt >= 2 && cnt <= RING3;
  const bool ovf = active && cnt > RING3;
  if (active && !tie && !ovf)
    a.hint[q] = (cnt == 1) ? ring[0][tid] : -1;
  const int lane = tid & 63;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long mt = __ballot(tie);
  if (mt)
  {
    const int leader = __ffsll((long long)mt) - 1;
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(&a.counters[0], (uint32_t)__popcll(mt));
    base = __shfl(base, leader, 64);
    if (tie)
    {
      for (int i = 0; i < RING3; i++)
        if (i < cnt)
          a.cand[q * RING3 + i] = ring[i][tid];
      a.tie_list[base + __popcll(mt & lt)] = make_uint2((uint32_t)q, (uint32_t)cnt);
    }
  }
  const unsigned long long mo = __ballot(ovf);
  if (mo)
  {
    const int leader = __ffsll((long long)mo) - 1;
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(&a.counters[1], (uint32_t)__popcll(mo));
    base = __shfl(base, leader, 64);
    if (ovf)
    {
      a.ovf_best[q] = best;
      a.ovf_list[base + __popcll(mo & lt)] = (uint32_t)q;
    }
  }
}

// The same search with lanes that outlive their queries.  A fixed grid of wavefronts
// draws queries from a shared counter: whenever most of a wavefront's lanes (REFILL3 of
// 64; taking queries is itself a costly step, so it is done for many lanes at once) have
// finished, the finished ones write their results and take the next queries off the pool.  Query costs differ widely (a point ten triangle
// widths off the surface meets a hundred boxes, a point on it a dozen), and with
// one query per lane for the lifetime of a wavefront most lanes sit idle waiting
// for the slowest; refilling keeps them busy.  Results are per query and do not
// depend on which lane ran them.  (The pool is dealt out in runs, one per L2 cache: see the draw.)
#ifdef ICPX_STATS
// development build only: totals over all queries of {node visits, leaves screened, visits that rejected both children,
// triangle screens, loop iterations (per wavefront), refills}
__device__ unsigned long long g_stats[8];
// ... and of the wavefront walk's batches: {largest number of node steps of one batch, longest
// batch in ticks of the 100 MHz clock, ticks of all batches together, batches}
__device__ unsigned long long g_stats_batch[4 + 32 + 1]; // (+ where the longest batch starts) // (+ batches and their ticks by node steps / 128)
#define STAT(i, v) st_##i += (v)
#else
#define STAT(i, v)
#endif

#ifndef REFILL3
#define REFILL3 56
#endif
#ifndef LEAFGO3
#define LEAFGO3 24
#endif

template <int L>
__device__ __forceinline__ void search3p_body(const Search3Args& a, const Xf& T)
{
  __shared__ int32_t ring[RING3][256];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int d = a.tree.depth;

  int64_t q = -1;
  d3 x = mk(0.0, 0.0, 0.0);
  float xf = 0.f, yf = 0.f, zf = 0.f, eabs = 0.f;
  double tau = 0.0, best = DBL_MAX, bound = DBL_MAX;
  float boundf = FLT_MAX, bboxf = INFINITY; // bound for triangles / for raw box distances
  int cnt = 0;
  int32_t hs_used = -1;
  uint32_t n = 1, trail = 0;
  float lbleaf = 0.f;
  float sec2 = INFINITY, minraw = INFINITY; // for the certificate (make_certificate)
  float sec3 = INFINITY;                    // ... and the nearest loser by name (note_screened_named)
  int32_t sec_id = -1;
  float widen = 0.f;                        // margin by which the walk looks farther (widen_of)
  bool have = false, fin = false, need_bt = false, atleaf = false;
  bool gave = false;  // the query's walk was given up on (step_budget): it goes on bail_list
  uint32_t steps = 0; // boxes and leaves this query's walk has met
  const uint32_t budget = a.bail_list ? a.step_budget : 0u;
  bool dry = false; // the pools have run out (same for every lane of the wavefront)
  uint32_t tried = 0; // runs this wavefront has found used up
  const int64_t total = a.qlist ? (int64_t)a.counters[a.count_slot] : a.nq;
  if ((int64_t)blockIdx.x * 256 >= total)
    return; // a short list is served by the first blocks; the rest must not queue at the pool
  if (a.qlist && (total < a.list_lo || total >= a.list_hi))
    return; // (left to the other kernel enqueued over this list)
#ifdef ICPX_STATS
  unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0;
#endif

  auto set_best = [&](double nb) {
    const double r = sqrt(nb) + tau;
    const double nbound = r * r;
    if (nbound < best)
      cnt = 0; // every older contender was above the old minimum
    best = nb;
    bound = nbound;
    // boxes and triangles are turned away beyond the widened bound
    const double rw = r + (double)widen;
    boundf = fminf(fmaxf(__double2float_ru(rw * rw), FLT_MIN), FLT_MAX);
    bboxf = box_bound(boundf, eabs);
  };

  while (true)
  {
    const bool idle = !have || fin;
    const unsigned long long mi = __ballot(idle);
    if (__popcll(mi) >= a.refill || __ballot(have && !fin) == 0ull)
    {
      // results of the finished queries
      const bool given = have && fin && gave;
      const bool f = have && fin && !gave;
      const bool tie = f && cnt >= 2 && cnt <= RING3;
      const bool ovf = f && cnt > RING3;
      if (f && !tie && !ovf)
        a.hint[q] = (cnt == 1) ? ring[0][tid] : -1;
      const unsigned long long mg = budget ? __ballot(given) : 0ull;
      if (mg)
      {
        // (what is queued is left unscreened: the query starts again, one wavefront to itself,
        // from the best found so far)
        const int leader = __ffsll((long long)mg) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&a.counters[a.bail_slot], (uint32_t)__popcll(mg));
        base = __shfl(base, leader, 64);
        if (given)
        {
          a.hint[q] = cnt >= 1 ? ring[0][tid] : -1;
          a.bail_list[base + __popcll(mg & lt)] = (uint32_t)q;
        }
      }
      if (f && a.cert)
      {
        // (a certificate that the next few moves of this size would use up is not worth
        // more than none)
        bool pair;
        a.cert[q] = cert_rebase_at(
          make_certificate_pair(cnt, best, sec2, sec3, sec_id, minraw, INFINITY, tau, eabs, a.cap2,
            a.pairs == 2 ? 2.f * eabs : a.pairs ? fmaxf(widen, 4.f * eabs) : -INFINITY, pair),
          a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
        if (pair)
        {
          // the set of two, for later passes (this pass's winner is settled: the loser is
          // beyond the tie tolerance)
          a.cand[q * RING3 + 0] = ring[0][tid];
          a.cand[q * RING3 + 1] = sec_id;
          a.cand[q * RING3 + 2] = -1;
        }
      }
      const unsigned long long mt = __ballot(tie);
      if (mt)
      {
        const int leader = __ffsll((long long)mt) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&a.counters[0], (uint32_t)__popcll(mt));
        base = __shfl(base, leader, 64);
        if (tie)
        {
          for (int i = 0; i < RING3; i++)
            if (i <= cnt)
              a.cand[q * RING3 + i] = i < cnt ? ring[i][tid] : -1;
          a.tie_list[base + __popcll(mt & lt)] = make_uint2((uint32_t)q, (uint32_t)cnt);
        }
      }
      const unsigned long long mo = __ballot(ovf);
      if (mo)
      {
        const int leader = __ffsll((long long)mo) - 1;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&a.counters[1], (uint32_t)__popcll(mo));
        base = __shfl(base, leader, 64);
        if (ovf)
        {
          a.ovf_best[q] = best;
          a.ovf_list[base + __popcll(mo & lt)] = (uint32_t)q;
        }
      }
      have = have && !fin;
      fin = false;

      // next queries off the pool
      if (!dry)
      {
        const bool want = !have;
        const unsigned long long mw = __ballot(want);
        const int leader = __ffsll((long long)mw) - 1;
        // (the queries are dealt out as k_search3w's batches are, by the L2 cache the block's
        // number goes with: chunks of XCHUNK * 64 consecutive queries to XPOOLS runs in turn,
        // a pool counter per run, a wavefront whose run is used up goes on to the next one's)
        constexpr uint32_t CHQ = XCHUNK * 64u;
        const uint32_t seg = (blockIdx.x + tried) % (uint32_t)XPOOLS;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(&a.counters[XPOOL_FIRST + seg * XPOOL_STRIDE], (uint32_t)__popcll(mw));
        base = __shfl(base, leader, 64);
        const uint32_t t = base + (uint32_t)__popcll(mw & lt);
        const int64_t nq_ = ((int64_t)(t / CHQ) * XPOOLS + seg) * CHQ + t % CHQ;
        if (want && nq_ < total)
        {
          q = a.qlist ? (int64_t)a.qlist[nq_] : nq_;
          have = true;
          need_bt = false;
          atleaf = false;
          gave = false;
          steps = 0;
          n = 1;
          trail = 0;
          cnt = 0;
          hs_used = -1;
          sec2 = INFINITY;
          sec3 = INFINITY;
          sec_id = -1;
          minraw = INFINITY;
          best = DBL_MAX;
          bound = DBL_MAX;
          boundf = FLT_MAX;
          bboxf = INFINITY;
          x = xform(T, a.qpos + 3 * q);
          xf = (float)x.x;
          yf = (float)x.y;
          zf = (float)x.z;
          const float smax = fmaxf(fmaxf(a.scene_max, fabsf(xf)), fmaxf(fabsf(yf), fabsf(zf)));
          eabs = smax * 2.3841858e-07f * 1.0000002f;
          tau = 1e-9 * (double)smax;
          widen = widen_of(a, T, a.qpos + 3 * q, x, eabs);
          if (a.cap2 < 1e300)
            set_best(a.cap2);
          int32_t hs = a.use_hint ? a.hint[q] : -1;
          if (hs < 0 && a.tcodes)
          {
            // cold start: the triangle that follows the query along the Morton curve
            const uint64_t code = morton63(xf, yf, zf, a.box.lo, a.box.inv);
            int64_t lo = 0, hi = a.nt;
            while (lo < hi)
            {
              const int64_t mid = (lo + hi) >> 1;
              if (a.tcodes[mid] < code)
                lo = mid + 1;
              else
                hi = mid;
            }
            hs = (int32_t)min(lo, a.nt - 1);
          }
          if (hs >= 0 && a.tid[hs] >= 0)
          {
            d3 P0, P1, P2;
            load_tri(a.tverts, hs, P0, P1, P2);
            const double ds = tri_d2_screen(x, P0, P1, P2);
            if (ds < best)
            {
              set_best(ds);
              ring[0][tid] = hs;
              cnt = 1;
              hs_used = hs;
            }
          }
        }
        // (positions rise with the draws: a run that has handed out one past the end is used up)
        if (__ballot(want && nq_ >= total) != 0ull)
          dry = ++tried >= (uint32_t)XPOOLS;
      }
      STAT(5, lane == 0 ? 1 : 0);
      if (__ballot(have) == 0ull)
      {
        if (dry)
          break;
        continue; // (the next run's pool)
      }
    }
    STAT(4, lane == 0 ? 1 : 0);

    // one step of the walk for every lane that is between leaves (see k_search3)
    if (have && !fin && !atleaf)
    {
      if (need_bt)
      {
        if (trail == 0)
          fin = true;
        else
        {
          const int k = 31 - __clz(trail);
          trail ^= 1u << k;
          const int lvl = 31 - __clz(n);
          n = (n >> (lvl - k - 1)) ^ 1u;
          need_bt = false;
          if (k == d - 1)
          {
            if (lbleaf > bboxf)
            {
              minraw = fminf(minraw, lbleaf);
              need_bt = true;
            }
            else
              atleaf = true;
          }
        }
      }
      if (budget && ++steps > budget && !fin)
      {
        fin = true;
        gave = true;
      }
      if (!fin && !need_bt && !atleaf)
      {
        STAT(0, 1);
        const float4* p = reinterpret_cast<const float4*>(a.nodes + (int64_t)n * NODE_FLOATS);
        const float4 u = p[0], v = p[1], w = p[2];
        const float lb0 = box_d2(u.x, u.y, u.z, u.w, v.x, v.y, xf, yf, zf);
        const float lb1 = box_d2(v.z, v.w, w.x, w.y, w.z, w.w, xf, yf, zf);
        const bool ok0 = lb0 <= bboxf, ok1 = lb1 <= bboxf;
        minraw = fminf(minraw, fminf(ok0 ? INFINITY : lb0, ok1 ? INFINITY : lb1));
        if (ok0 || ok1)
        {
          const uint32_t near = (ok1 && (!ok0 || lb1 < lb0)) ? 1u : 0u;
          const int lvl = 31 - __clz(n);
          if (ok0 && ok1)
          {
            trail |= 1u << lvl;
            lbleaf = (lvl == d - 1) ? (near ? lb0 : lb1) : lbleaf;
          }
          n = 2u * n + near;
          atleaf = (lvl == d - 1);
        }
        else
        {
          STAT(2, 1);
          need_bt = true;
        }
      }
    }

    // leaves are screened once enough lanes stand on one (the screen is several times the
    // cost of a step, so it should not run for a handful of lanes), or when nobody can step
    const bool onleaf = have && !fin && atleaf;
    const int nleaf = __popcll(__ballot(onleaf));
    const bool anywalk = __ballot(have && !fin && !atleaf) != 0ull;
    if (onleaf && (nleaf >= LEAFGO3 || !anywalk))
    {
      STAT(1, 1);
      const int64_t s0 = (int64_t)(n - (1u << d)) * L;
#pragma unroll
      for (int t = 0; t < L; t++)
      {
        const int64_t s = s0 + t;
        if (s < a.nt && a.tid[s] >= 0)
        {
          STAT(3, 1);
          d3 P0, P1, P2;
          load_tri(a.tverts, s, P0, P1, P2);
          const double d2 = tri_d2_screen(x, P0, P1, P2);
          // (the start triangle, met again, is in the ring or has lost already)
          if ((int32_t)s != hs_used)
            note_screened_named(sec2, sec_id, sec3, d2, (int32_t)s, best, bound, tau, cnt,
              ring[0][tid]);
          if (d2 < best)
            set_best(d2);
          if (d2 <= bound && (int32_t)s != hs_used)
          {
            if (cnt < RING3)
              ring[cnt][tid] = (int32_t)s;
            cnt++;
          }
        }
      }
      need_bt = true;
      atleaf = false;
    }
  }
#ifdef ICPX_STATS
  atomicAdd(&g_stats[0], st_0);
  atomicAdd(&g_stats[1], st_1);
  atomicAdd(&g_stats[2], st_2);
  atomicAdd(&g_stats[3], st_3);
  atomicAdd(&g_stats[4], st_4);
  atomicAdd(&g_stats[5], st_5);
#endif
}

template <int L>
__global__ __launch_bounds__(256) void k_search3p(Search3Args a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  search3p_body<L>(a, T);
}

// What the shortcut left over, by the size of the list.  A lane that walks the tree alone
// takes ~150 us for one query whatever the number of queries, so a short list (a few
// stragglers of a converged pass) is served one WAVEFRONT per query instead, over the
// 64-ary tree of search2.hpp (a few steps of 64 box tests each); a long list by the
// pooled lanes.  Same results either way.
constexpr uint32_t FALLBACK_SMALL = 98304;

// (128 registers: four wavefronts per SIMD; the walks are latency-bound, and the one more
// register set the compiler would like to have costs a quarter of the walks in flight)
template <int L>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_fallback(Search3Args a, Search2Args w)
{
  const uint32_t total = a.counters[a.count_slot];
  if (total == 0u || (int64_t)total < a.list_lo || (int64_t)total >= a.list_hi)
    return; // (most converged passes: gone after one read; or the other kernel's list)
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  if (total > a.small_limit)
  {
    search3p_body<L>(a, T);
    return;
  }
  const int wave = threadIdx.x >> 6;
  for (uint32_t base = blockIdx.x * 4u; base < total; base += gridDim.x * 4u)
  {
    const uint32_t idx = base + (uint32_t)wave;
    const bool active = idx < total;
    search2_body<64>(w, T, active ? (int64_t)a.qlist[idx] : 0, active);
  }
}

// A first bound for a query that has no winner of a last pass to start from, wherever it is:
// a coarse grid over the target's bounds (dim^3 cells) whose every cell names a triangle
// (Morton position) with a centroid near the cell's centre.  Cells that hold centroids take
// one of them (the smallest position); the others are filled by flooding in halving strides
// (each pass a cell looks at its 26 neighbours at the stride and keeps the nearest centroid
// it has seen), which finds a near triangle, not always the nearest: good enough for a bound.
// The query's place on the triangles' Morton curve (the other seed) names a near triangle
// only when the query is on the surface; a point a few dozen edge lengths off it, as most
// are at the start of a registration, has curve neighbours as far away again.
__device__ __forceinline__ int64_t seed_cell_of(float x, float y, float z, const Box3& b, int dim)
{
  const float fd = (float)dim;
  const int ix = min((int)(fminf(fmaxf((x - b.lo[0]) * b.inv[0], 0.f), 1.f) * fd), dim - 1);
  const int iy = min((int)(fminf(fmaxf((y - b.lo[1]) * b.inv[1], 0.f), 1.f) * fd), dim - 1);
  const int iz = min((int)(fminf(fmaxf((z - b.lo[2]) * b.inv[2], 0.f), 1.f) * fd), dim - 1);
  return ((int64_t)iz * dim + iy) * dim + ix;
}

__global__ void k_seed_scatter(const float4* __restrict__ tpre, int64_t nt, Box3 b, int dim,
  uint32_t* __restrict__ grid)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt || tpre[2 * s].w < 0.f)
    return; // (a degenerate triangle is nobody's closest cell)
  const float4 c = tpre[2 * s + 1];
  atomicMin(&grid[seed_cell_of(c.x, c.y, c.z, b, dim)], (uint32_t)s); // (empty cells hold 0xffffffff)
}

__global__ void k_seed_flood(const int32_t* __restrict__ in, int32_t* __restrict__ out,
  const float4* __restrict__ tpre, Box3 b, int dim, int stride)
{
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)dim * dim * dim)
    return;
  const int ix = (int)(idx % dim), iy = (int)((idx / dim) % dim), iz = (int)(idx / ((int64_t)dim * dim));
  const float inv = 1.f / (float)dim;
  // the cell's centre (an axis the target has no extent along: inv = 0, every cell at lo)
  const float cx = b.lo[0] + (b.inv[0] > 0.f ? ((float)ix + 0.5f) * inv / b.inv[0] : 0.f);
  const float cy = b.lo[1] + (b.inv[1] > 0.f ? ((float)iy + 0.5f) * inv / b.inv[1] : 0.f);
  const float cz = b.lo[2] + (b.inv[2] > 0.f ? ((float)iz + 0.5f) * inv / b.inv[2] : 0.f);
  int32_t best = in[idx];
  float bd = INFINITY;
  if (best >= 0)
  {
    const float4 c = tpre[2 * (int64_t)best + 1];
    bd = (c.x - cx) * (c.x - cx) + (c.y - cy) * (c.y - cy) + (c.z - cz) * (c.z - cz);
  }
  for (int dz = -1; dz <= 1; dz++)
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++)
      {
        const int jx = ix + dx * stride, jy = iy + dy * stride, jz = iz + dz * stride;
        if ((dx | dy | dz) == 0 || jx < 0 || jy < 0 || jz < 0 || jx >= dim || jy >= dim || jz >= dim)
          continue;
        const int32_t t = in[((int64_t)jz * dim + jy) * dim + jx];
        if (t < 0 || t == best)
          continue;
        const float4 c = tpre[2 * (int64_t)t + 1];
        const float d = (c.x - cx) * (c.x - cx) + (c.y - cy) * (c.y - cy) + (c.z - cz) * (c.z - cz);
        if (d < bd || (d == bd && t < best))
        {
          bd = d;
          best = t;
        }
      }
  out[idx] = best;
}

// lower and upper fp32 bounds of the squared distance from the query to a triangle, from
// its record {unit normal, plane allowance} {centroid, radius}
// (RAW: the hardware's square root, within one unit in the last place, which the factors
// cover as well)
template <bool RAW = false>
__device__ __forceinline__ void tri_bounds_rec(float nx, float ny, float nz, float allow, float cx,
  float cy, float cz, float R, float xf, float yf, float zf, float eabs, float& lb, float& ub)
{
  const float vx = xf - cx, vy = yf - cy, vz = zf - cz;
  const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
  const float vlen = RAW ? __builtin_amdgcn_sqrtf(q2) : __builtin_sqrtf(q2);
  const float m = 4.f * eabs + allow + 6.0e-7f * vlen;
  const float pd = fabsf(__builtin_fmaf(nz, vz, __builtin_fmaf(ny, vy, nx * vx)));
  const float pd_lo = fmaxf(pd - m, 0.f), pd_hi = pd + m;
  const float vlo = fmaxf(vlen * 0.9999995f - 2.f * eabs, 0.f);
  const float in2 = fmaxf(__builtin_fmaf(vlo, vlo, -(pd_hi * pd_hi)), 0.f);
  const float ld =
    fmaxf((RAW ? __builtin_amdgcn_sqrtf(in2) : __builtin_sqrtf(in2)) * 0.9999995f - R - 3.f * eabs, 0.f);
  lb = __builtin_fmaf(ld, ld, pd_lo * pd_lo) * 0.999999f;
  // the stored centroid is within a rounding of a point of the triangle
  const float vhi = vlen * 1.000001f + 2.f * eabs;
  ub = vhi * vhi * 1.000001f;
}

// The lower bound of tri_bounds_rec with one square root instead of three, for the walk that
// takes it of every triangle of every leaf it enters.  v = query - centroid and q2 = |v|^2 as
// computed in fp32.  Where tri_bounds_rec uses |v| it is bounded from above by the sum of the
// components' magnitudes (in the allowance of the plane distance, and in the cross term of
// (|v| (1 - 5e-7) - 2 eabs)^2 >= 0.9999987 q2 - 4 eabs |v|, whose square term is dropped);
// 0.999998 leaves room for the roundings of the three operations that follow.
__device__ __forceinline__ float tri_lb_rec(float nx, float ny, float nz, float allow, float R,
  float vx, float vy, float vz, float q2, float eabs)
{
  const float l1 = (fabsf(vx) + fabsf(vy)) + fabsf(vz);
  const float m = __builtin_fmaf(6.1e-7f, l1, 4.f * eabs + allow);
  const float pd = fabsf(__builtin_fmaf(nz, vz, __builtin_fmaf(ny, vy, nx * vx)));
  const float pd_lo = fmaxf(pd - m, 0.f), pd_hi = pd + m;
  const float w = __builtin_fmaf(-4.00001f * eabs, l1, 0.999998f * q2);
  const float in2 = fmaxf(__builtin_fmaf(-pd_hi, pd_hi, w), 0.f);
  const float ld = fmaxf(__builtin_amdgcn_sqrtf(in2) * 0.9999995f - (R + 3.f * eabs), 0.f);
  return __builtin_fmaf(ld, ld, pd_lo * pd_lo) * 0.999999f;
}

// The search with one walk per wavefront.  Queries are in Morton order, so the 64 of a
// wavefront lie within a few triangle widths of each other and their walks visit nearly
// the same boxes; here the wavefront walks the tree ONCE for all of them.  The node, the
// stack and the order of descent are the same for every lane (node records and triangle
// records are read at wavefront-uniform addresses, on the scalar path: one request, not 64
// scattered ones), while every lane keeps its own query, minimum, bounds and contenders.
// A box is entered when any lane's bound reaches it.  Two levels are taken per step: the
// records of both children of a node share a 128-byte line and hold the four
// grandchildren's boxes, which halves the chain of dependent reads.  Order of descent
// follows the first interested lane.  At a leaf every lane takes the cheap lower bound of
// each triangle (tri_lb_rec) and the centroids' distances as upper bounds, tightens its bound
// with the latter and queues the triangle (per lane, in LDS) if the former is within it.
// When some lane's queue is full, and when the walk is over, the queues are weeded (entries
// the bound has overtaken go unscreened) and every lane screens its most promising entry in
// fp64: the bound that leaves behind turns most of the others away, and the screen runs with
// most lanes busy.  Wavefronts draw batches of 64 queries from the
// pools.  Results per query are those of k_search3p: bounds only ever remove
// triangles that cannot be contenders, and a triangle is a contender by the same per-lane
// test whatever the order of the walk.
typedef float vec4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const vec4f cfloat4;
#ifndef SEEDW
#define SEEDW 8
#endif
constexpr int PENDW = 16; // queue slots per lane (a power of two)

// A queued lower bound lb <= ref as 16 bits, and back (the value returned is never above lb).
__device__ __forceinline__ uint16_t pend_encode(float lb, float ref)
{
  const uint32_t b = __float_as_uint(fmaxf(ref - lb, 0.f)) + 0xffffu; // (ref - lb rounded up)
  return (uint16_t)(b >> 16);
}
__device__ __forceinline__ float pend_decode(uint16_t h, float ref)
{
  return fmaxf(ref - __uint_as_float((uint32_t)h << 16), 0.f) * 0.9999997f;
}

__device__ __forceinline__ uint32_t uni32(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// CERT: check and keep certificates (a pass right after a cold start does neither: its
// moves outrun any certificate, and the bookkeeping costs registers).
// BAIL: batches that walk for too long are given up on (Search3Args::bail_ticks).  A build of
// its own: the kernel is short of scalar registers, and the few the deadline takes cost
// every walk 3 % whether or not a batch is ever given up on; the host takes this build for
// targets whose triangles' sizes range widely, the only ones where it matters.
template <int L, bool CERT, bool BAIL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void k_search3w(Search3Args a)
{
  float* const cert = CERT ? a.cert : nullptr;
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  __shared__ int32_t ring[RING3][64];
  __shared__ int32_t pend_s[PENDW][64];
  // The entries' lower bounds, 16 bits each (LDS is what limits the wavefronts per CU): how far
  // the bound is BELOW the lane's reference `pref` (its pruning bound when the queue was last
  // weeded or empty), the float's upper half rounded up.  What decides whether an entry is
  // weeded is its distance from the pruning bound, and that is what these bits resolve: the
  // upper half of the bound itself is good to 1/256 of the bound, and a query dozens of
  // triangle widths off the surface has dozens of triangles that close to its bound, each of
  // which then cost an fp64 screen.
  __shared__ uint16_t pend_lb[PENDW][64];
  __shared__ uint32_t stk[64];
  const int lane = threadIdx.x;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int d = a.tree.depth;
  const int64_t total = a.qlist ? (int64_t)a.counters[a.count_slot] : a.nq;
  if ((int64_t)blockIdx.x * 64 >= total)
    return; // a short list is served by the first blocks; the rest must not queue at the pool
  if (a.qlist && (total < a.list_lo || total >= a.list_hi))
    return; // (left to the other kernel enqueued over this list)
  // (read-only for the whole launch and addressed uniformly: eligible for the scalar path)
  // certificates are checked here only on a whole pass (a list holds what failed the check)
  bool cert_on = false, far_on = false;
  // (cert.hpp: an anchored pass leaves standing certificates as they are)
  const bool anchored = cert && cert_anchored(a.cert_ok);
  if (cert && a.use_hint && !a.qlist && __builtin_amdgcn_readfirstlane(*a.cert_ok) != 0)
  {
    cert_on = true;
    far_on = a.cap2 <= uniform_f64(*a.cert_cap2) * (1.0 + 1e-12);
  }
  cfloat4* nodes4 = (cfloat4*)(uintptr_t)a.nodes;
  cfloat4* tpre4 = (cfloat4*)(uintptr_t)a.tpre;
  cfloat4* lpre4 = (cfloat4*)(uintptr_t)a.lpre;
#ifdef ICPX_STATS
  unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0, st_6 = 0, st_7 = 0;
#endif

  // The batches of 64 queries are dealt out in XPOOLS runs, each with a pool counter of its
  // own: chunk c of XCHUNK consecutive batches belongs to run c % XPOOLS, and a wavefront
  // draws from the run that goes with its block number modulo XPOOLS first.  Blocks b and
  // b + 8 share one of the chip's eight L2 caches and the queries are in Morton order, so
  // each cache sees the nodes and leaves of a few stretches of the cloud instead of a share
  // of everything the resident grid is walking, whose footprint is several times a cache's
  // size (memory-side reads of a cold pass over 16 M queries: 17 GB before, 5.5 GB so).
  // The runs advance through the cloud together, so they end together; a wavefront whose
  // run is used up goes on with the next one's.  (Placement is a matter of speed only: any
  // wavefront may take any batch.)
  // The first batch of a run's k-th block is the run's k-th, by its number; later ones come
  // off the run's pool behind those (thousands of wavefronts drawing at once would queue at
  // the counters).  Blocks beyond the number of batches have left above.
  const uint32_t nbatch = (uint32_t)((total + 63) >> 6);
  const uint32_t nlive = min(gridDim.x, nbatch);
  uint32_t tried = 0;
  bool own = true;
  while (true)
  {
    uint32_t base = 0u;
    bool got = false;
    while (tried < (uint32_t)XPOOLS)
    {
      const uint32_t seg = (blockIdx.x + tried) % (uint32_t)XPOOLS;
      // (how many batches the run has, and how many of them go to its blocks by their numbers)
      constexpr uint32_t CYCLE = (uint32_t)XPOOLS * XCHUNK;
      const uint32_t rest = nbatch % CYCLE;
      const uint32_t len = nbatch / CYCLE * XCHUNK +
        (rest > seg * XCHUNK ? min(rest - seg * XCHUNK, XCHUNK) : 0u);
      const uint32_t mine = min(len, seg < nlive ? (nlive - seg + (uint32_t)XPOOLS - 1u) / XPOOLS : 0u);
      uint32_t t = blockIdx.x / (uint32_t)XPOOLS;
      if (!own)
      {
        if (lane == 0)
          t = atomicAdd(&a.counters[XPOOL_FIRST + seg * XPOOL_STRIDE], 1u);
        t = uni32(t) + mine;
      }
      own = false;
      if (t < len)
      {
        base = ((t / XCHUNK * (uint32_t)XPOOLS + seg) * XCHUNK + t % XCHUNK) * 64u;
        got = true;
        break;
      }
      tried++;
    }
    if (!got)
      break;
#ifdef ICPX_STATS
    const unsigned long long bt0 = wall_clock64(), bs0 = st_0;
#endif
    // (when the batch's walk is to be given up on, by the shader clock counted in units of
    // 64 cycles; the constant 100 MHz clock takes several times as long to answer)
    // (0: never; a deadline that comes out as 0 is moved on by a tick)
    const uint32_t walk_t0 = BAIL ? (uint32_t)(__builtin_readcyclecounter() >> 6) : 0u;
    uint32_t walk_end = (BAIL && a.bail_ticks) ? max(walk_t0 + a.bail_ticks, 1u) : 0u;
    const int64_t idx = (int64_t)base + lane;
    bool active = idx < total;
    const int64_t q = active ? (a.qlist ? (int64_t)a.qlist[idx] : idx) : 0;
    const float slack = (cert_on && active) ? cert[q] : 0.f;
    const float leadv = (cert_on && active && a.lead) ? a.lead[q] : 0.f; // (with the certificate, not behind it)
    bool isset = false;
    if (__ballot(slack != 0.f) != 0ull)
    {
      // as in k_search_local: a query whose certificate covers its move keeps its result
      // (the certificates' transform is read here, not held across the walk)
      Xf Tc;
#pragma unroll
      for (int i = 0; i < 12; i++)
        Tc.m[i] = uniform_f64(a.cert_T[i]);
      if (slack > 0.f || (slack < 0.f && far_on))
      {
        const d3 xn = xform(T, a.qpos + 3 * q), xp = xform(Tc, a.qpos + 3 * q);
        const double mx = xn.x - xp.x, my = xn.y - xp.y, mz = xn.z - xp.z;
        const float mv =
          __builtin_amdgcn_sqrtf(__double2float_ru(mx * mx + my * my + mz * mz)) * 1.000001f;
        const float left = slack > 0.f ? slack - 2.000001f * mv - 1e-30f
                                       : -slack - 1.000001f * mv - 1e-30f;
        if (left > 0.f)
        {
          isset = cert_is_set(slack);
          const bool ld = a.lead && cert_has_lead(slack);
          const float keep = slack > 0.f ? cert_encode(left * 0.9999999f, isset, ld) : -left * 0.9999999f;
          if (!anchored)
            cert[q] = keep;
          // (a set certificate worn down to nothing: the search runs after all)
          isset = isset && keep > 0.f;
          active = !(keep != 0.f);
          // (the winner's lead over the set's other members still stands: no weighing)
          if (isset && ld && cert_lead_stands(a.lead, q, leadv, mv, anchored))
            isset = false;
        }
      }
    }
    if (a.setmask && !a.qlist)
    {
      // (a whole pass: these are queries base .. base + 63)
      const unsigned long long ms = __ballot(isset);
      if (lane == 0)
        a.setmask[base >> 6] = ms;
    }
    if (__ballot(active) == 0ull)
      continue; // every query of the block went through

    d3 x = mk(0.0, 0.0, 0.0);
    float xf = 0.f, yf = 0.f, zf = 0.f, eabs = 0.f, tauf = 0.f;
    double tau = 0.0, best = DBL_MAX, bound = DBL_MAX;
    // boundf: from fp64 screens; ubf: smallest cheap upper bound; prunef: what triangles are
    // tested against (the smaller of the two, the second widened by the tie tolerance);
    // bboxf: the same for raw box distances (an idle lane's is below every distance)
    // ubq: squared centroid distance behind ubf; rpf: square root of prunef, rounded up
    float boundf = FLT_MAX, ubf = FLT_MAX, prunef = FLT_MAX, bboxf = active ? INFINITY : -1.f;
    float ubq = FLT_MAX, rpf = 1.9e19f;
    float sec2 = INFINITY, minraw = INFINITY; // for the certificate
    float widen = 0.f; // margin by which the walk looks farther (widen_of)
    int cnt = 0, npend = 0; // queue: slots 0 .. npend-1 of this lane's column, in no order
    int pick = -1;          // after purge(): the slot of the queued triangle with the smallest bound
    int32_t hs_used = -1;
    float pref = 0.f; // reference of the queued bounds (pend_lb)

    auto retighten = [&]() {
      // (the hardware's square root is within one unit in the last place: the factors
      // cover it)
      const float r = __builtin_amdgcn_sqrtf(ubf) * 1.000001f + tauf + widen;
      prunef = fminf(boundf, r * r * 1.000001f);
      rpf = __builtin_amdgcn_sqrtf(prunef) * 1.000001f;
      const float rb_ = rpf + 1.7320509f * eabs; // as box_bound, from the root already at hand
      bboxf = rb_ * rb_ * 1.000002f;
    };
    auto set_best = [&](double nb) {
      const double r = sqrt(nb) + tau;
      const double nbound = r * r;
      if (nbound < best)
        cnt = 0; // every older contender was above the old minimum
      best = nb;
      bound = nbound;
      const double rw = r + (double)widen; // (turned away beyond the widened bound)
      boundf = fminf(fmaxf(__double2float_ru(rw * rw), FLT_MIN), FLT_MAX);
      retighten();
    };
    // The queue is weeded and its most promising entry found: entries whose bound has
    // fallen behind the lane's (they count for the certificate with that bound) go, the rest
    // close ranks.  A few cheap instructions per slot, against the few hundred of a screen.
    auto purge = [&]() {
      int w = 0;
      float mlb = INFINITY;
      pick = -1;
#pragma nounroll
      for (int i = 0; i < PENDW; i++)
      {
        if (__ballot(i < npend) == 0ull)
          break;
        if (i < npend)
        {
          const int32_t s = pend_s[i][lane];
          const float lbq = pend_decode(pend_lb[i][lane], pref);
          if (lbq > prunef)
            sec2 = fminf(sec2, lbq); // dropped against the tighter bound: at least that far
          else
          {
            pend_s[w][lane] = s;
            pend_lb[w][lane] = pend_encode(lbq, prunef); // (against the new reference)
            if (lbq < mlb)
            {
              mlb = lbq;
              pick = w;
            }
            w++;
          }
        }
      }
      npend = w;
      pref = prunef;
    };
    // The picked triangle is screened (straight after purge(): it is still in the race), the
    // last entry takes its slot.  Most promising first: the bound it leaves behind turns most
    // of the others away unscreened.
    auto screen_pick = [&]() {
      const int32_t s = pend_s[pick][lane];
      npend--;
      if (pick != npend)
      {
        pend_s[pick][lane] = pend_s[npend][lane];
        pend_lb[pick][lane] = pend_lb[npend][lane];
      }
      STAT(3, 1);
      d3 P0, P1, P2;
      load_tri(a.tverts, s, P0, P1, P2);
      const double d2 = tri_d2_screen(x, P0, P1, P2);
      sec2 = note_screened(sec2, d2, best, bound, tau, cnt);
      if (d2 < best)
        set_best(d2);
      if (d2 <= bound)
      {
        if (cnt < RING3)
          ring[cnt][lane] = s;
        cnt++;
      }
    };

    if (active)
    {
      x = xform(T, a.qpos + 3 * q);
      xf = (float)x.x;
      yf = (float)x.y;
      zf = (float)x.z;
      const float smax = fmaxf(fmaxf(a.scene_max, fabsf(xf)), fmaxf(fabsf(yf), fabsf(zf)));
      eabs = smax * 2.3841858e-07f * 1.0000002f;
      tau = 1e-9 * (double)smax;
      tauf = 1.0000001e-9f * smax;
      widen = CERT ? widen_of(a, T, a.qpos + 3 * q, x, eabs) : 0.f;
      if (a.cap2 < 1e300)
        set_best(a.cap2);
      int32_t hs = a.use_hint ? a.hint[q] : -1;
      // last pass's winner is a poor start when the pose has moved by many triangle widths
      // since (its centroid is then several of its radii away): such a query is seeded
      // like a cold one, if that is nearer
      float qbest = INFINITY;
      bool stale = false;
      if (hs >= 0)
      {
        const float4 rb = a.tpre[2 * (int64_t)hs + 1];
        const float vx = xf - rb.x, vy = yf - rb.y, vz = zf - rb.z;
        qbest = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
        stale = qbest > 16.f * rb.w * rb.w;
      }
      if ((hs < 0 || stale) && a.seed_grid)
      {
        // the triangles the query's grid cell and its six face neighbours name: the nearest
        // of seven centroids is worth what a grid of twice the resolution is (cold pass of a
        // million points 0.95 -> 0.90 ms either way) without its memory and build time; the 26
        // neighbours cost more reads than they save steps (0.94)
        const int dim = a.seed_dim;
        const int64_t cell = seed_cell_of(xf, yf, zf, a.box, dim), plane = (int64_t)dim * dim;
        const int ix = (int)(cell % dim), iy = (int)((cell / dim) % dim), iz = (int)(cell / plane);
        int32_t gsn[7];
        gsn[0] = a.seed_grid[cell];
        gsn[1] = ix > 0 ? a.seed_grid[cell - 1] : -1;
        gsn[2] = ix + 1 < dim ? a.seed_grid[cell + 1] : -1;
        gsn[3] = iy > 0 ? a.seed_grid[cell - dim] : -1;
        gsn[4] = iy + 1 < dim ? a.seed_grid[cell + dim] : -1;
        gsn[5] = iz > 0 ? a.seed_grid[cell - plane] : -1;
        gsn[6] = iz + 1 < dim ? a.seed_grid[cell + plane] : -1;
#pragma unroll
        for (int t = 0; t < 7; t++)
        {
          const int32_t gs = gsn[t];
          if (gs >= 0 && (t == 0 || gs != gsn[0]))
          {
            const float4 rb = a.tpre[2 * (int64_t)gs + 1];
            const float vx = xf - rb.x, vy = yf - rb.y, vz = zf - rb.z;
            const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
            if (q2 < qbest)
            {
              qbest = q2;
              hs = gs;
            }
          }
        }
      }
      if ((hs < 0 || stale) && a.tcodes && (a.seed_morton || !a.seed_grid))
      {
        // of the few triangles around the query's place on the Morton curve, the one with
        // the nearest centroid gives the first bound
        const uint64_t code = morton63(xf, yf, zf, a.box.lo, a.box.inv);
        int64_t lo = 0, hi = a.nt; // first position with tcodes[pos] >= code
        if (a.tcell)
        {
          // (the search by halves within the code's cell: a few triangles, not all of them)
          const uint64_t cl = code >> (63 - a.tcell_bits);
          lo = (int64_t)a.tcell[cl];
          hi = (int64_t)a.tcell[cl + 1];
        }
        while (lo < hi)
        {
          const int64_t mid = (lo + hi) >> 1;
          if (a.tcodes[mid] < code)
            lo = mid + 1;
          else
            hi = mid;
        }
#pragma unroll
        for (int t = -SEEDW / 2; t < SEEDW / 2; t++)
        {
          const int64_t sc = min(max(lo + t, (int64_t)0), a.nt - 1);
          const float4 ra = a.tpre[2 * sc], rb = a.tpre[2 * sc + 1];
          const float vx = xf - rb.x, vy = yf - rb.y, vz = zf - rb.z;
          const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
          if (ra.w >= 0.f && q2 < qbest)
          {
            qbest = q2;
            hs = (int32_t)sc;
          }
        }
      }
      if (hs >= 0 && a.tid[hs] >= 0)
      {
        d3 P0, P1, P2;
        load_tri(a.tverts, hs, P0, P1, P2);
        const double ds = tri_d2_screen(x, P0, P1, P2);
        if (ds < best)
        {
          set_best(ds);
          ring[0][lane] = hs;
          cnt = 1;
          hs_used = hs;
        }
      }
    }

    if (a.defer_far > 0.f && a.bail_list)
    {
      const bool seeded = active && cnt == 1;
      const float sd = seeded ? fmaxf((float)best, 1e-37f) : 1.f;
      const float lg = seeded ? __log2f(sd) : 0.f;
      float sum = lg;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1)
        sum += __shfl_xor(sum, m, 64);
      const int ns = __popcll(__ballot(seeded));
      const bool far = seeded && ns >= 16 && sd > a.defer_far && lg > sum / (float)ns + 6.f;
      const unsigned long long mf = __ballot(far);
      if (mf) [[unlikely]]
      {
        uint32_t tb = 0u;
        if (lane == 0)
          tb = atomicAdd(&a.counters[a.bail_slot], (uint32_t)__popcll(mf));
        tb = uni32(tb);
        if (far)
        {
          a.hint[q] = ring[0][lane];
          a.bail_list[tb + (uint32_t)__popcll(mf & lt)] = (uint32_t)q;
          // (an idle lane from here on: its bound is below every distance)
          active = false;
          bboxf = -1.f;
          cnt = 0;
        }
      }
    }

    // sort key of a box for the order of descent: the first interested lane's distance,
    // the box's number in the two low bits; ~0 for a box nobody wants
    auto key_of = [&](float lb, unsigned long long m, int rep, uint32_t k) -> uint32_t {
      const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lb), rep);
      return m ? ((bits & ~3u) | k) : 0xffffffffu;
    };

    int sp = 0;
    uint32_t n = 1;
    bool go = true;
    while (go)
    {
      STAT(0, lane == 0 ? 1 : 0);
      const int lvl = 31 - __clz(n);
      // the boxes one step down: the two children when they are leaves, else the four
      // grandchildren (the records of 2n and 2n+1 are one 128-byte line)
      const bool two = lvl == d - 1;
      const uint32_t cbase = two ? 2u * n : 4u * n; // heap number of box 0
      float lb0, lb1, lb2 = 0.f, lb3 = 0.f;
      unsigned long long m0, m1, m2 = 0ull, m3 = 0ull;
      if (two)
      {
        cfloat4* p = nodes4 + (int64_t)n * 4;
        const vec4f u = p[0], v = p[1], w = p[2];
        lb0 = box_d2(u.x, u.y, u.z, u.w, v.x, v.y, xf, yf, zf);
        lb1 = box_d2(v.z, v.w, w.x, w.y, w.z, w.w, xf, yf, zf);
        m0 = __ballot(lb0 <= bboxf);
        m1 = __ballot(lb1 <= bboxf);
        minraw = fminf(minraw, fminf(lb0 <= bboxf ? INFINITY : lb0, lb1 <= bboxf ? INFINITY : lb1));
      }
      else
      {
        cfloat4* p = nodes4 + (int64_t)n * 8;
        const vec4f u = p[0], v = p[1], w = p[2], u2 = p[4], v2 = p[5], w2 = p[6];
        lb0 = box_d2(u.x, u.y, u.z, u.w, v.x, v.y, xf, yf, zf);
        lb1 = box_d2(v.z, v.w, w.x, w.y, w.z, w.w, xf, yf, zf);
        lb2 = box_d2(u2.x, u2.y, u2.z, u2.w, v2.x, v2.y, xf, yf, zf);
        lb3 = box_d2(v2.z, v2.w, w2.x, w2.y, w2.z, w2.w, xf, yf, zf);
        m0 = __ballot(lb0 <= bboxf);
        m1 = __ballot(lb1 <= bboxf);
        m2 = __ballot(lb2 <= bboxf);
        m3 = __ballot(lb3 <= bboxf);
        minraw = fminf(minraw,
          fminf(fminf(lb0 <= bboxf ? INFINITY : lb0, lb1 <= bboxf ? INFINITY : lb1),
            fminf(lb2 <= bboxf ? INFINITY : lb2, lb3 <= bboxf ? INFINITY : lb3)));
      }
      const unsigned long long many = m0 | m1 | m2 | m3;
      uint32_t next = 0;
      if (many)
      {
        const int rep = __ffsll((long long)many) - 1;
        uint32_t k0 = key_of(lb0, m0, rep, 0u), k1 = key_of(lb1, m1, rep, 1u),
                 k2 = key_of(lb2, m2, rep, 2u), k3 = key_of(lb3, m3, rep, 3u);
        uint32_t t_;
#define CSWAP(A, B)                                                                               \
  t_ = min(A, B);                                                                                 \
  B = max(A, B);                                                                                  \
  A = t_
        CSWAP(k0, k1);
        CSWAP(k2, k3);
        CSWAP(k0, k2);
        CSWAP(k1, k3);
        CSWAP(k1, k2);
#undef CSWAP
        if (two || lvl + 2 == d)
        {
          // leaves, nearest first: cheap bounds of every triangle for the lanes whose bound
          // reaches the leaf's box
#pragma nounroll
          while (k0 != 0xffffffffu)
          {
            const uint32_t c = k0 & 3u;
            k0 = k1;
            k1 = k2;
            k2 = k3;
            k3 = 0xffffffffu;
            const float lbc = c == 0u ? lb0 : c == 1u ? lb1 : c == 2u ? lb2 : lb3;
            bool want = lbc <= bboxf;
            if (!want)
              minraw = fminf(minraw, lbc); // (the bound may have shrunk since the node's visit)
            if (__ballot(want) == 0ull)
              continue;
            STAT(1, lane == 0 ? 1 : 0);
            // room for this leaf in every lane's queue: weed, and screen while that is not
            // enough (every lane with something queued joins in: the instructions are spent
            // anyway)
            if (__ballot(npend > PENDW - L) != 0ull)
            {
              purge();
              while (__ballot(npend > PENDW - L) != 0ull)
              {
                STAT(4, lane == 0 ? 1 : 0);
                if (npend > 0)
                  screen_pick();
                purge();
              }
            }
            const int64_t leaf = (int64_t)(cbase + c - (1u << d));
            const int64_t s0 = leaf * L;
            if (s0 >= a.nt)
              continue; // a leaf past the last triangle
            // (the leaf's records are contiguous, padded past the end, and say themselves
            // whether the triangle counts; all are requested at once, the leaf's own with them)
            const vec4f la_ = lpre4[2 * leaf], lb_ = lpre4[2 * leaf + 1];
            vec4f rec[2 * L];
#pragma unroll
            for (int t = 0; t < 2 * L; t++)
              rec[t] = tpre4[2 * s0 + t];
            // The leaf's patch of surface between two planes and inside a sphere: a box
            // around a slanted patch is thick, and a query many triangle widths off the
            // surface has dozens of leaf boxes within its bound, but few slabs.
            if (la_.w >= 0.f)
            {
              const float ux = xf - lb_.x, uy = yf - lb_.y, uz = zf - lb_.z;
              const float u2 = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
              const float lbl = tri_lb_rec(la_.x, la_.y, la_.z, la_.w, lb_.w, ux, uy, uz, u2, eabs);
              if (want && lbl > prunef)
              {
                want = false;
                sec2 = fminf(sec2, lbl); // every triangle of the leaf is at least this far
              }
              if (__ballot(want) == 0ull)
                continue;
              STAT(7, lane == 0 ? 1 : 0);
            }
            // the centroids' distances are upper bounds of the triangles': the smallest met
            // so far bounds the query's result before anything has been screened
            float vx[L], vy[L], vz[L], q2[L];
            float ubn = ubq;
#pragma unroll
            for (int t = 0; t < L; t++)
            {
              const vec4f rb = rec[2 * t + 1];
              vx[t] = xf - rb.x;
              vy[t] = yf - rb.y;
              vz[t] = zf - rb.z;
              q2[t] = __builtin_fmaf(vz[t], vz[t], __builtin_fmaf(vy[t], vy[t], vx[t] * vx[t]));
              ubn = fminf(ubn, rec[2 * t].w < 0.f ? FLT_MAX : q2[t]);
            }
            if (__ballot(active && ubn < ubq) != 0ull)
            {
              if (active && ubn < ubq)
              {
                ubq = ubn;
                const float vhi = __builtin_amdgcn_sqrtf(ubn) * 1.000001f + 2.f * eabs;
                ubf = vhi * vhi * 1.000001f;
                retighten();
              }
            }
            // plane distance and in-plane distance to the circle about the centroid: the
            // lower bound every triangle of an entered leaf is held against (a cheaper
            // first test on the centroid alone let through three triangles of four: some
            // lane of 64 is near enough to most triangles of a leaf its wavefront enters)
#pragma unroll
            for (int t = 0; t < L; t++)
            {
              const vec4f ra = rec[2 * t], rb = rec[2 * t + 1];
              if (ra.w < 0.f)
                continue;
              STAT(6, lane == 0 ? 1 : 0);
              const float lb = tri_lb_rec(ra.x, ra.y, ra.z, ra.w, rb.w, vx[t], vy[t], vz[t], q2[t], eabs);
              if (want && lb > prunef)
                sec2 = fminf(sec2, lb); // turned away: at least this far
              if (want && lb <= prunef && (int32_t)(s0 + t) != hs_used)
              {
                if (npend == 0)
                  pref = prunef;
                pend_s[npend][lane] = (int32_t)(s0 + t);
                pend_lb[npend][lane] = pend_encode(lb, pref);
                npend++;
              }
            }
          }
        }
        else
        {
          // inner nodes: enter the nearest, owe the others a visit (farthest pushed first)
          if (lane == 0)
          {
            int s_ = sp;
            if (k3 != 0xffffffffu)
              stk[s_++] = cbase + (k3 & 3u);
            if (k2 != 0xffffffffu)
              stk[s_++] = cbase + (k2 & 3u);
            if (k1 != 0xffffffffu)
              stk[s_++] = cbase + (k1 & 3u);
          }
          sp += (k3 != 0xffffffffu ? 1 : 0) + (k2 != 0xffffffffu ? 1 : 0) +
            (k1 != 0xffffffffu ? 1 : 0);
          next = cbase + (k0 & 3u);
        }
      }
      else
        STAT(2, lane == 0 ? 1 : 0);
      if (next)
        n = next;
      else if (sp == 0)
        go = false;
      else
      {
        __builtin_amdgcn_wave_barrier();
        n = uni32(stk[--sp]);
        // Is the batch to be given up on?  By the clock, not by a count of steps (a step of a
        // batch in trouble, with its queues full and a round of screens per leaf, takes
        // several times an ordinary one), looked at when the walk comes back up to a stack
        // depth that is a multiple of four, and with visits still owed (a walk given up on
        // leaves the loop with a stack that is not empty: that is how the code below knows).
        if constexpr (BAIL)
        {
          if (walk_end != 0u && (sp & 3) == 0 && sp != 0) [[unlikely]]
          {
            const uint32_t now = (uint32_t)(__builtin_readcyclecounter() >> 6);
            if ((int32_t)(now - walk_end) > 0)
            {
              // Past the deadline: given up on if the batch has also taken four times what the
              // pass's finished batches took on average (where every batch is slow, a target of
              // slivers, none is what the pass waits for); else looked at again a quarter of
              // its time so far later.
              uint32_t tsum = 0u, tcnt = 0u;
              if (lane == 0)
              {
                tsum = atomicAdd(&a.counters[WALK_TIME], 0u);
                tcnt = atomicAdd(&a.counters[WALK_DONE], 0u);
              }
              tsum = uni32(tsum);
              tcnt = uni32(tcnt);
              const uint32_t spent = now - walk_t0;
              if (!a.bail_mean || (tcnt >= 64u && (uint64_t)(spent >> 2) * tcnt > 4ull * tsum))
                go = false;
              else
                walk_end = max(now + (spent >> 2), 1u);
            }
          }
        }
      }
    }
    if (BAIL && sp != 0) [[unlikely]]
    {
      // (what is queued is left unscreened: the queries start again, from the best so far)
      const unsigned long long mb = __ballot(active);
      uint32_t tb = 0u;
      if (lane == 0)
        tb = atomicAdd(&a.counters[a.bail_slot], (uint32_t)__popcll(mb));
      tb = uni32(tb);
      if (active)
      {
        a.hint[q] = cnt >= 1 ? ring[0][lane] : -1;
        a.bail_list[tb + (uint32_t)__popcll(mb & lt)] = (uint32_t)q;
      }
      continue;
    }
    if (BAIL && walk_end != 0u && lane == 0)
    {
      // (a finished walk's time goes into the pass's figures)
      atomicAdd(&a.counters[WALK_TIME], ((uint32_t)(__builtin_readcyclecounter() >> 6) - walk_t0) >> 2);
      atomicAdd(&a.counters[WALK_DONE], 1u);
    }
    // what is still queued, against the bounds as they stand at the end of the walk
    purge();
    while (__ballot(npend > 0) != 0ull)
    {
      STAT(5, lane == 0 ? 1 : 0);
      if (npend > 0)
        screen_pick();
      purge();
    }

    // settle: a single contender wins outright; 2..RING3 go to the tie list; more overflow
    const bool tie = active && cnt >= 2 && cnt <= RING3;
    const bool ovf = active && cnt > RING3;
    if (active && !tie && !ovf)
      a.hint[q] = (cnt == 1) ? ring[0][lane] : -1;
    if (active && cert)
      cert[q] = cert_rebase_at(make_certificate(cnt, best, sec2, minraw, INFINITY, tau, eabs, a.cap2),
        a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
    const unsigned long long mt = __ballot(tie);
    if (mt)
    {
      const int leader = __ffsll((long long)mt) - 1;
      uint32_t tb = 0;
      if (lane == leader)
        tb = atomicAdd(&a.counters[0], (uint32_t)__popcll(mt));
      tb = __shfl(tb, leader, 64);
      if (tie)
      {
        for (int i = 0; i < RING3; i++)
          if (i <= cnt)
            a.cand[q * RING3 + i] = i < cnt ? ring[i][lane] : -1;
        a.tie_list[tb + __popcll(mt & lt)] = make_uint2((uint32_t)q, (uint32_t)cnt);
      }
    }
    const unsigned long long mo = __ballot(ovf);
    if (mo)
    {
      const int leader = __ffsll((long long)mo) - 1;
      uint32_t ob = 0;
      if (lane == leader)
        ob = atomicAdd(&a.counters[1], (uint32_t)__popcll(mo));
      ob = __shfl(ob, leader, 64);
      if (ovf)
      {
        a.ovf_best[q] = best;
        a.ovf_list[ob + __popcll(mo & lt)] = (uint32_t)q;
      }
    }
#ifdef ICPX_STATS
    if (lane == 0)
    {
      const unsigned long long dt = wall_clock64() - bt0;
      if (atomicMax(&g_stats_batch[0], st_0 - bs0) < st_0 - bs0)
        g_stats_batch[36] = base; // (a race between two record holders is harmless)
      atomicMax(&g_stats_batch[1], dt);
      atomicAdd(&g_stats_batch[2], dt);
      atomicAdd(&g_stats_batch[3], 1ull);
      const int bucket = (int)min((st_0 - bs0) / 128ull, 15ull);
      atomicAdd(&g_stats_batch[4 + bucket], 1ull);
      atomicAdd(&g_stats_batch[20 + bucket], dt);
    }
#endif
  }
#ifdef ICPX_STATS
  atomicAdd(&g_stats[0], st_0);
  atomicAdd(&g_stats[1], st_1);
  atomicAdd(&g_stats[2], st_2);
  atomicAdd(&g_stats[3], st_3);
  atomicAdd(&g_stats[4], st_4);
  atomicAdd(&g_stats[5], st_5);
  atomicAdd(&g_stats[6], st_6);
  atomicAdd(&g_stats[7], st_7);
#endif
}

// Contested queries: eight lanes per query, one contender per lane, exact
// evaluation, (dist2, cell id) minimum by DPP exchanges.
__device__ __forceinline__ void resolve_ties_body(const Search3Args& a, unsigned block,
  unsigned nblocks)
{
  // (the list's length and the wavefront's first mask words are read side by side, before
  // the transform: in most passes of a clean cloud most wavefronts find nothing and leave
  // after this one round of reads)
  // A wavefront takes SETS_HALVES half words of the mask at a time (its first lanes one each):
  // 512 consecutive queries.  (It was 64 half words, 2048 queries: a cloud of a million points
  // then kept 500 wavefronts busy on a chip that holds 4000, each working through its marked
  // sets one after the other, two or three dependent reads apiece: 98 us a pass for 0.84 M
  // points of which a third stand on pairs, 71 us for 2 M noisy ones.)
  const int64_t halves = (a.nq + 31) / 32;
  const uint32_t* const half = reinterpret_cast<const uint32_t*>(a.setmask);
  const int lane = threadIdx.x & 63;
  const int64_t hfirst = ((int64_t)block * 4 + (threadIdx.x >> 6)) * SETS_HALVES;
  const int64_t hstride = (int64_t)nblocks * 4 * SETS_HALVES;
  const uint32_t total = a.counters[0];
  const uint32_t bits_first =
    (a.setmask && lane < SETS_HALVES && hfirst + lane < halves) ? half[hfirst + lane] : 0u;
  if (block * 32u >= total && hfirst + hstride >= halves && __ballot(bits_first != 0u) == 0ull)
    return;
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  __shared__ uint32_t setq[4][SETS_HALVES * 32 + 64];
  const int j = threadIdx.x & 7;
  const uint32_t ngroups = nblocks * 32u;
  for (uint32_t r = block * 32u + (threadIdx.x >> 3); r < total; r += ngroups)
  {
    const uint2 rec = a.tie_list[r];
    const int64_t q = rec.x;
    double d2 = INFINITY;
    int32_t id = 0x7fffffff, s = -1;
    if (j < (int)rec.y)
    {
      s = a.cand[q * RING3 + j];
      const d3 x = xform(T, a.qpos + 3 * q);
      d3 P0, P1, P2, cp;
      double dd, w[3];
      load_tri(a.tverts, s, P0, P1, P2);
      if (tri_eval(x, P0, P1, P2, cp, dd, w))
      {
        d2 = dd;
        id = a.tid[s];
      }
    }
    const double mine = d2;
    const int32_t smine = s;
    group_argmin<8>(d2, id, s);
    if (j == 0)
      a.hint[q] = s;
    if (a.lead && a.cert)
    {
      // (a search that ended with a set: its winner's lead goes on record with the first weighing)
      const double second = group_min<8>(smine == s ? (double)INFINITY : mine);
      if (j == 0 && s >= 0)
      {
        const d3 x = xform(T, a.qpos + 3 * q);
        const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)x.x)),
          fmaxf(fabsf((float)x.y), fabsf((float)x.z)));
        cert_note_lead(a.cert, a.lead, q, second < INFINITY ? lead_of(d2, second, 1e-9 * (double)smax) : 0.f,
          a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
      }
    }
  }
  // the queries that stand on a set certificate (one bit each in setmask): a wavefront takes
  // SETS_HALVES half words, 512 consecutive queries, at a time.  Where many of them are marked
  // (a fifteenth of a noisy cloud stands on sets) each is weighed by one lane, which takes the
  // set's members one after the other (cand[], -1 ends a set of fewer than RING3); where few
  // are (a thousand of a million), they are dealt out to groups of eight lanes, one member
  // per lane, so that a pass is one exact evaluation long instead of several in a row.
  if (!a.setmask)
    return;
  // The marked queries of a wavefront's chunks are listed in LDS (their numbers) and weighed
  // 64 at a time, one to a lane, whatever chunk they came from: a wavefront keeps reading
  // chunks until it has 64 listed (or has none left to read), so that every round of
  // evaluations but its last finds all lanes busy.  (One chunk's marked queries at a time
  // left half the lanes idle on a noisy cloud: 33 of 512 stand on sets there.)
  int64_t h0 = hfirst;
  uint32_t npend = 0u; // listed and not yet weighed
  for (;;)
  {
    while (npend < 64u && h0 < halves)
    {
      const int64_t h = h0 + lane;
      uint32_t bits = h0 == hfirst ? bits_first : ((lane < SETS_HALVES && h < halves) ? half[h] : 0u);
      const uint32_t mine = (uint32_t)__popc(bits);
      const uint32_t incl = wave_incl_scan(mine, lane);
      const uint32_t marked = (uint32_t)__shfl((int)incl, 63, 64);
      if (marked != 0u && marked <= SPARSE_SETS && npend == 0u && h0 + hstride >= halves)
      {
        // (few, and nothing else to do with them: dealt out to groups of eight lanes)
        for (uint32_t r0 = 0; r0 < marked; r0 += 8)
        {
          // the r-th marked query of the wavefront's 2048: the first lane whose running count
          // exceeds r holds it, as the (r - count before that lane)-th set bit of its word
          int64_t q = -1;
          for (uint32_t r = r0; r < marked && r < r0 + 8; r++)
          {
            const int src = __ffsll((long long)__ballot(incl > r)) - 1;
            const uint32_t isrc = (uint32_t)__shfl((int)incl, src, 64);
            const uint32_t msrc = (uint32_t)__shfl((int)mine, src, 64);
            uint32_t bb = (uint32_t)__shfl((int)bits, src, 64);
            for (uint32_t k = r - (isrc - msrc); k > 0; k--)
              bb &= bb - 1u;
            if ((uint32_t)(lane >> 3) == r - r0)
              q = (h0 + src) * 32 + (__ffs((int)bb) - 1);
          }
          int32_t s = q >= 0 ? a.cand[q * RING3 + j] : -1;
          // members up to the first -1
          const unsigned long long neg = __ballot(s < 0);
          const uint32_t negg = (uint32_t)(neg >> (lane & ~7)) & 0xffu;
          const int cnt = negg ? (__ffs((int)negg) - 1) : RING3;
          double d2 = INFINITY;
          int32_t id = 0x7fffffff;
          if (j < cnt)
          {
            const d3 x = xform(T, a.qpos + 3 * q);
            d3 P0, P1, P2, cp;
            double dd, w[3];
            load_tri(a.tverts, s, P0, P1, P2);
            if (tri_eval(x, P0, P1, P2, cp, dd, w))
            {
              d2 = dd;
              id = a.tid[s];
            }
            else
              s = -1;
          }
          else
            s = -1;
          const double mine = d2;
          const int32_t smine = s;
          group_argmin<8>(d2, id, s);
          if (j == 0 && q >= 0)
            a.hint[q] = s;
          if (a.lead)
          {
            const double second = group_min<8>((smine == s || smine < 0) ? (double)INFINITY : mine);
            if (j == 0 && q >= 0 && s >= 0)
            {
              const d3 x = xform(T, a.qpos + 3 * q);
              const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)x.x)),
                fmaxf(fabsf((float)x.y), fabsf((float)x.z)));
              cert_note_lead(a.cert, a.lead, q, second < INFINITY ? lead_of(d2, second, 1e-9 * (double)smax) : 0.f,
                a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
            }
          }
        }
      }
      else if (marked != 0u)
      {
        uint32_t pos = npend + incl - mine;
        while (bits)
        {
          const int b = __ffs((int)bits) - 1;
          bits &= bits - 1u;
          setq[threadIdx.x >> 6][pos++] = (uint32_t)(h * 32 + b);
        }
        npend += marked;
      }
      h0 += hstride;
    }
    if (npend == 0u)
      break;
    __builtin_amdgcn_wave_barrier(); // (a wavefront's LDS accesses complete in order)
    const uint32_t take = npend < 64u ? npend : 64u;
    if ((uint32_t)lane < take)
    {
      const int64_t q = (int64_t)setq[threadIdx.x >> 6][npend - take + (uint32_t)lane];
      const d3 x = xform(T, a.qpos + 3 * q);
      const int4 c0 = *reinterpret_cast<const int4*>(a.cand + q * RING3);
      double bd2 = INFINITY, sd2 = INFINITY; // (the best and the second best member's squared distance)
      int32_t bs = -1;
      const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)x.x)),
        fmaxf(fabsf((float)x.y), fabsf((float)x.z)));
      // A set of two (most are: a winner and the rival a hair behind it) is screened first: if
      // the screened distances are farther apart than the tie tolerance the nearer one wins, as
      // it does for a search that ends with one contender, at a third of the instructions of two
      // exact evaluations; only a real tie, or a larger set, is evaluated exactly.
      bool decided = false;
      if (c0.x >= 0 && c0.y >= 0 && c0.z < 0)
      {
        d3 A0, A1, A2, B0, B1, B2;
        load_tri(a.tverts, c0.x, A0, A1, A2);
        load_tri(a.tverts, c0.y, B0, B1, B2);
        const double da = tri_d2_screen(x, A0, A1, A2), db = tri_d2_screen(x, B0, B1, B2);
        const double rb = sqrt(fmin(da, db)) + 1e-9 * (double)smax;
        if (fmax(da, db) > rb * rb)
        {
          bs = da < db ? c0.x : c0.y;
          bd2 = fmin(da, db);
          sd2 = fmax(da, db);
          decided = true;
        }
      }
#pragma unroll 1
      for (int i = 0; i < RING3 && !decided; i++)
      {
        // (members read one by one: an array of them indexed by the loop's counter would live
        // in scratch memory)
        const int32_t mi = a.cand[q * RING3 + i];
        if (mi < 0)
          break;
        d3 P0, P1, P2, cp;
        double dd, w[3];
        load_tri(a.tverts, mi, P0, P1, P2);
        if (tri_eval(x, P0, P1, P2, cp, dd, w))
        {
          // (cell ids only break exact ties of the squared distance: they are read when one
          // occurs, not for every member; a scattered four-byte read is a 64-byte line from
          // memory, and these were a third of what the kernel fetched)
          bool wins = dd < bd2;
          if (dd == bd2)
            wins = a.tid[mi] < (bs >= 0 ? a.tid[bs] : 0x7fffffff);
          if (wins)
          {
            sd2 = bd2;
            bd2 = dd;
            bs = mi;
          }
          else
            sd2 = fmin(sd2, dd);
        }
      }
      a.hint[q] = bs;
      if (a.lead)
        cert_note_lead(a.cert, a.lead, q, (bs >= 0 && sd2 < INFINITY) ? lead_of(bd2, sd2, 1e-9 * (double)smax) : 0.f,
          a.cert_ok, a.cert_T, x, a.qpos + 3 * q);
    }
    npend -= take;
    __builtin_amdgcn_wave_barrier();
  }
}

// Both kinds of contested query in one launch: the first `rblocks` blocks resolve the near
// ties, the rest walk the tree again for the queries whose contenders overflowed the ring.
template <int L>
__global__ __launch_bounds__(256) void k_contested(Search3Args a, unsigned rblocks)
{
  // (the counts first: in most passes most of these blocks have nothing to do and should
  // be gone after one read)
  if (blockIdx.x < rblocks)
  {
    if (blockIdx.x * 32u < a.counters[0] || a.setmask)
      resolve_ties_body(a, blockIdx.x, rblocks);
  }
  else
  {
    const uint32_t novf = a.counters[1];
    const unsigned nover = gridDim.x - rblocks;
    for (unsigned b = blockIdx.x - rblocks; (uint64_t)b * 256u < novf; b += nover)
    {
      search3_body<L, true>(a, b);
      __syncthreads(); // (the ring in LDS is reused)
    }
  }
}

} // namespace icpx
