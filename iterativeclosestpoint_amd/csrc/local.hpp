// local.hpp — the shortcut for queries near the surface: no tree walk.
//
// If a query sits within r of a triangle h, every triangle within r of the query has
// its box within 2r of h's box (both closest points are within r of the query).  So
// for every triangle the list of triangles whose boxes come within eps of its own is
// built once per target (k_nbr: one lane per triangle, a fixed-radius walk of the
// one-triangle-per-leaf tree), and a query with 2r <= eps tests only h's list: a flat
// loop of box tests with similar trip counts for neighbouring lanes, then the usual
// screening of the few that pass.  The contenders found are the same as a tree walk's,
// so the winner is too.
//
// h is last pass's winner (or, on a cold start, the triangle next to the query on the
// Morton curve).  When h is too far for that, the query first walks along the surface,
// from list to list, towards itself; what still does not get within eps/2 of a triangle
// is appended to a list that the tree kernel then serves.
//
// eps is a property of the mesh (a fraction of its typical edge length, chosen by the
// host so that the lists stay short).
#pragma once
#include "search3.hpp"

namespace icpx
{

// fp32 lower bound of the squared distance between two boxes (records of six floats)
__device__ __forceinline__ float box_box_lb(const float* __restrict__ A, float b0, float b1,
  float b2, float b3, float b4, float b5)
{
  const float gx = fmaxf(fmaxf(A[0] - b3, b0 - A[3]), 0.f) * 0.99999976f;
  const float gy = fmaxf(fmaxf(A[1] - b4, b1 - A[4]), 0.f) * 0.99999976f;
  const float gz = fmaxf(fmaxf(A[2] - b5, b2 - A[5]), 0.f) * 0.99999976f;
  return __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx)) * 0.99999952f;
}

// Counting and filling both stop one past this many neighbours.  A leaf whose list is
// cut short (a box stretched across a jump of the Morton curve, or a radius that is too
// large for the mesh) is recognised by its length and its queries take the tree walk.
constexpr uint32_t NBR_CAP = 96;

// One lane per leaf: all leaves whose boxes come within sqrt(eps2) of it.
// FILL = false counts them, FILL = true writes them at off[leaf].  Lists are padded with
// -1 to a multiple of four entries, so that the search reads them as aligned int4.
template <bool FILL>
__global__ __launch_bounds__(256) void k_nbr(const float* __restrict__ nodes, BinTree t,
  float eps2, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
  int32_t* __restrict__ nbr)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= t.nleaf)
    return;
  const int d = t.depth;
  const uint32_t self = (1u << d) + (uint32_t)i;
  float A[6];
  {
    const float* r = nodes + (int64_t)(self >> 1) * NODE_FLOATS + 6 * (self & 1u);
#pragma unroll
    for (int k = 0; k < 6; k++)
      A[k] = r[k];
  }
  uint32_t count = 0;
  const uint32_t base = FILL ? off[i] : 0u;
  if (A[0] <= A[3]) // not an empty box
  {
    uint32_t n = 1, trail = 0;
    bool need_bt = false;
    while (true)
    {
      if (need_bt)
      {
        if (trail == 0)
          break;
        const int k = 31 - __clz(trail);
        trail ^= 1u << k;
        const int lvl = 31 - __clz(n);
        n = (n >> (lvl - k - 1)) ^ 1u;
        need_bt = false;
      }
      if (n >> d)
      {
        if (FILL)
          nbr[base + count] = (int32_t)(n - (1u << d));
        count++;
        if (count > NBR_CAP)
          break;
        need_bt = true;
        continue;
      }
      const float4* p = reinterpret_cast<const float4*>(nodes + (int64_t)n * NODE_FLOATS);
      const float4 u = p[0], v = p[1], w = p[2];
      const bool ok0 = box_box_lb(A, u.x, u.y, u.z, u.w, v.x, v.y) <= eps2;
      const bool ok1 = box_box_lb(A, v.z, v.w, w.x, w.y, w.z, w.w) <= eps2;
      if (ok0 || ok1)
      {
        if (ok0 && ok1)
          trail |= 1u << (31 - __clz(n));
        n = 2u * n + (ok0 ? 0u : 1u);
      }
      else
        need_bt = true;
    }
  }
  const uint32_t padded = (count + 3u) & ~3u;
  if (!FILL)
    cnt[i] = padded;
  else
    for (uint32_t j = count; j < padded; j++)
      nbr[base + j] = -1;
}

// Leaf boxes on a 16-bit grid over the target's bounds (one step for all three axes),
// rounded outwards by one extra step, six values packed into one 16-byte record: the
// search reads a neighbour's box with a single load and tests it in grid units, where a
// query is converted once.  The extra step covers that conversion and the coordinates'
// own rounding, so a grid-space distance never exceeds the true one divided by the step;
// a box test on the looser box can only admit more triangles.
struct QuantGrid
{
  float base[3];
  float step, inv_step;
};

__global__ void k_quant_boxes(const float* __restrict__ nodes, BinTree t, QuantGrid g,
  uint4* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.nleaf)
    return;
  const uint32_t self = (1u << t.depth) + (uint32_t)i;
  const float* r = nodes + (int64_t)(self >> 1) * NODE_FLOATS + 6 * (self & 1u);
  uint32_t q[6];
#pragma unroll
  for (int a = 0; a < 3; a++)
  {
    const double lo = floor(((double)r[a] - (double)g.base[a]) / (double)g.step) - 1.0;
    const double hi = ceil(((double)r[3 + a] - (double)g.base[a]) / (double)g.step) + 1.0;
    q[a] = (uint32_t)fmin(fmax(lo, 0.0), 65535.0);
    q[3 + a] = (uint32_t)fmin(fmax(hi, 0.0), 65535.0);
  }
  out[i] = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), 0u);
}

// squared distance, in grid steps, from the query (in grid coordinates) to a quantised box
__device__ __forceinline__ float qbox_g2(const uint4 w, float gx, float gy, float gz)
{
  const float dx = fmaxf(fmaxf((float)(w.x & 0xffffu) - gx, gx - (float)(w.y >> 16)), 0.f);
  const float dy = fmaxf(fmaxf((float)(w.x >> 16) - gy, gy - (float)(w.z & 0xffffu)), 0.f);
  const float dz = fmaxf(fmaxf((float)(w.y & 0xffffu) - gz, gz - (float)(w.z >> 16)), 0.f);
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// Per triangle (Morton position), three records {m.xyz, d}: the half-spaces m.y <= d that
// contain the triangle and are bounded by the planes through its edges perpendicular to
// it (m is the unit in-plane normal of the edge, pointing away from the third vertex).  A
// query outside one of them by delta is at least delta from the triangle: for a query on a
// smooth surface that rules out, with three dot products, the neighbours whose boxes it
// sits in but whose edges it is outside of.  A triangle too thin for its normal to be
// trusted, or an edge too short, gets a zero record (never rules anything out).
__global__ void k_tri_edges(const float* __restrict__ tverts, int64_t nt, float4* __restrict__ edges)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nt)
    return;
  d3 P[3];
  load_tri(tverts, s, P[0], P[1], P[2]);
  const d3 e1 = mk(P[1].x - P[0].x, P[1].y - P[0].y, P[1].z - P[0].z);
  const d3 e2 = mk(P[2].x - P[0].x, P[2].y - P[0].y, P[2].z - P[0].z);
  const d3 e3 = mk(P[2].x - P[1].x, P[2].y - P[1].y, P[2].z - P[1].z);
  const d3 n = mk(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
  const double nlen = sqrt(dot3(n, n));
  const double emax2 = fmax(dot3(e1, e1), fmax(dot3(e2, e2), dot3(e3, e3)));
  const bool flat_ok = nlen > 1e-6 * emax2;
#pragma unroll
  for (int i = 0; i < 3; i++)
  {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const d3 e = mk(P[j].x - P[i].x, P[j].y - P[i].y, P[j].z - P[i].z);
    d3 m = mk(e.y * n.z - e.z * n.y, e.z * n.x - e.x * n.z, e.x * n.y - e.y * n.x);
    const d3 toK = mk(P[k].x - P[i].x, P[k].y - P[i].y, P[k].z - P[i].z);
    if (dot3(m, toK) > 0.0)
      m = mk(-m.x, -m.y, -m.z);
    const double mlen = sqrt(dot3(m, m));
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (flat_ok && mlen > 1e-6 * nlen * sqrt(emax2) && mlen > 0.0)
    {
      const float mx = (float)(m.x / mlen), my = (float)(m.y / mlen), mz = (float)(m.z / mlen);
      // d for the normal as stored, at the edge's farther end (the half-space must hold
      // the whole triangle even after the normal's rounding)
      const double di = (double)mx * P[i].x + (double)my * P[i].y + (double)mz * P[i].z;
      const double dj = (double)mx * P[j].x + (double)my * P[j].y + (double)mz * P[j].z;
      const double dk = (double)mx * P[k].x + (double)my * P[k].y + (double)mz * P[k].z;
      rec = make_float4(mx, my, mz, __double2float_ru(fmax(fmax(di, dj), dk)));
    }
    edges[3 * s + i] = rec;
  }
}

// How far (at least) the query is outside one of the triangle's edge half-spaces; <= 0
// when it is outside none.  alw covers every rounding of the evaluation.
__device__ __forceinline__ float edge_excess(const float4* __restrict__ edges, int32_t s,
  float xf, float yf, float zf, float alw)
{
  const float4 a = edges[3 * (int64_t)s], b = edges[3 * (int64_t)s + 1],
               c = edges[3 * (int64_t)s + 2];
  const float da = __builtin_fmaf(a.z, zf, __builtin_fmaf(a.y, yf, a.x * xf)) - a.w;
  const float db = __builtin_fmaf(b.z, zf, __builtin_fmaf(b.y, yf, b.x * xf)) - b.w;
  const float dc = __builtin_fmaf(c.z, zf, __builtin_fmaf(c.y, yf, c.x * xf)) - c.w;
  return fmaxf(fmaxf(da, db), dc) - alw;
}

struct LocalArgs
{
  Search3Args s;        // the tree kernel's arguments (queries, tree, triangles, tie lists);
                        // nodes/tree are the one-triangle-per-leaf tree the lists index
  const uint32_t* noff; // neighbour lists, CSR over triangles (Morton positions); every
  const int32_t* nbr;   // list starts on a multiple of four entries and is padded with -1
  const uint4* qbox;    // quantised leaf boxes
  const float4* edges;  // edge half-spaces (k_tri_edges)
  QuantGrid grid;
  float eps;            // radius the lists were built for
  uint32_t* fallback;   // queries left to the tree kernel; their count is counters[3]
  // certificates: per query, how far it may still move before last pass's winner has to be
  // looked at again (0: no certificate); the transform they refer to; whether they count
  // (the certificates themselves are s.cert, shared with the tree kernels)
  const double* cert_T;
  const int32_t* cert_ok;
  const double* cert_cap2; // the cap the far certificates were made for
  // a pass whose certificates were already checked (k_certify) serves the listed queries
  // only (see CertifyArgs for the list's layout); null: every query
  const uint32_t* work;
  const uint32_t* work_count;
  int64_t segcap;
  // not zero: a query settled here with a single winner whose certificate comes out at next
  // to nothing (the runner-up is a hair behind: the triangle across an edge the foot point is
  // next to) is put on the tree kernel's list as well, which keeps winner and runner-up as a
  // set of two under a certificate worth having (make_certificate_pair); without this such
  // a query is searched here again on every pass, for good
  int pairs_to_tree;
};

constexpr int LOCAL_STEPS = 16; // moves along the surface before a query is given up
constexpr int LOCAL_PEND = 6;   // triangles a lane may hold for the screening loop

// One lane per query.  Start: last pass's winner, or (no winner) the triangle next to the
// query on the Morton curve.  While the start triangle is too far for its list to hold
// every contender, the query moves to the listed triangle whose box centre is nearest
// (a walk along the surface towards the query, strictly downhill in that distance, so it
// ends); a query that arrives within eps/2 of a triangle is settled from that triangle's
// list alone, the others are handed to the tree kernel with the triangle they reached as
// their starting bound.  Every query's slot in hint[] is rewritten.
//
// Certificates.  A query settled here with a single contender w also learns how far every
// other triangle is at least: the boxes and edge half-spaces that ruled listed triangles
// out are lower bounds of their distances, screened ones have their distance, and a
// triangle outside the list is farther than eps less the distance to the list's owner.
// The smallest of these less the distance to w, less the tie tolerance, is the query's
// slack: as long as the query has moved by less than half of it since (distances to a
// triangle change by at most the move), w is still the only contender, and the search
// would return it again.  Later passes only subtract twice the move from the slack of such
// a query and leave its hint alone; the search runs again when the slack is used up.  The
// tree kernels make certificates the same way from what they turned away
// (make_certificate in search3.hpp), including negative ones for queries with nothing
// within the distance cap (the nearest bound less the cap; such a query has no winner and
// stays so while its move is below that).  Every certificate, whoever made it, is checked
// here; what fails the check and cannot be settled from a list goes to the tree kernel.
template <bool LISTED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_search_local(LocalArgs la)
{
  const Search3Args& a = la.s;
  // everything that is the same for all lanes is requested at once, ahead of the test of
  // the stop flag: a pass that lets most queries through on their certificates is only as
  // long as its chain of dependent reads
  Xf T, Tc;
  const int tid = threadIdx.x, lane = tid & 63;
  // (a listed pass: block b serves segment b mod CERTIFY_SEGS, together with every
  // CERTIFY_SEGS-th other block)
  const int seg = LISTED ? (int)(blockIdx.x % CERTIFY_SEGS) : 0;
  const uint32_t totalv = LISTED ? la.work_count[seg * CERTIFY_SEG_STRIDE] : (uint32_t)a.nq;
  // (a whole pass: one round per block, and the query's own reads join the others)
  const int64_t q0 = (int64_t)blockIdx.x * 256 + tid;
  const int64_t qc0 = (!LISTED && q0 < a.nq) ? q0 : 0;
  float p30[3] = { 0.f, 0.f, 0.f }, slack00 = 0.f, lead00 = 0.f;
  if (!LISTED)
  {
    p30[0] = a.qpos[3 * qc0];
    p30[1] = a.qpos[3 * qc0 + 1];
    p30[2] = a.qpos[3 * qc0 + 2];
    slack00 = a.cert ? a.cert[qc0] : 0.f;
    lead00 = (a.cert && a.lead) ? a.lead[qc0] : 0.f; // (with the certificate, not behind it)
  }
  const int32_t stopv = a.T.stop ? *a.T.stop : 0;
  int32_t okv = 0;
#pragma unroll
  for (int i = 0; i < 12; i++)
    T.m[i] = a.T.T[i];
  double capv = 0.0;
  if (a.cert)
  {
    okv = *la.cert_ok;
    capv = *la.cert_cap2;
#pragma unroll
    for (int i = 0; i < 12; i++)
      Tc.m[i] = la.cert_T[i];
  }
  else
    Tc = T;
  if (__builtin_amdgcn_readfirstlane(stopv) != 0)
    return; // the registration this pass belongs to has already stopped
#pragma unroll
  for (int i = 0; i < 12; i++)
  {
    T.m[i] = uniform_f64(T.m[i]);
    Tc.m[i] = uniform_f64(Tc.m[i]);
  }
  __shared__ int32_t ring[RING3][256];
  __shared__ int32_t pendq[LOCAL_PEND][256];
  __shared__ uint32_t wcount[4];
  __shared__ uint32_t bbase;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t total = (int64_t)__builtin_amdgcn_readfirstlane(totalv);

  // (a listed pass runs a bounded grid over however many the list turned out to hold)
  int64_t first = (int64_t)(LISTED ? blockIdx.x / CERTIFY_SEGS : blockIdx.x) * 256;
  if (first >= total)
    return;
  const uint32_t* const work = LISTED ? la.work + seg * la.segcap : nullptr;
  do
  {
  const bool active = first + tid < total;
  const uint32_t entry = (LISTED && active) ? work[first + tid] : 0u;
  const int64_t q = !active ? 0 : LISTED ? (int64_t)(entry & 0x7fffffffu) : first + tid;
  float p3[3] = { p30[0], p30[1], p30[2] };
  if (LISTED)
  {
    p3[0] = a.qpos[3 * q];
    p3[1] = a.qpos[3 * q + 1];
    p3[2] = a.qpos[3 * q + 2];
  }
  const float slack0 = slack00;
  d3 x = mk(0.0, 0.0, 0.0);
  float xf = 0.f, yf = 0.f, zf = 0.f, eabs = 0.f;
  double tau = 0.0, best = DBL_MAX, bound = DBL_MAX;
  float boundf = FLT_MAX, boundg = INFINITY; // bound for triangles / in grid steps for boxes
  float gx = 0.f, gy = 0.f, gz = 0.f;        // the query in grid coordinates
  const float ginv2 = la.grid.inv_step * la.grid.inv_step * 1.00001f;
  int cnt = 0;
  int32_t hs = -1;
  // (a listed query with the top bit set passed the check and stands on a set certificate)
  bool local = false, certified = (entry >> 31) != 0u, certset = certified;
  float sec2 = 0.f; // lower bound of the squared distance of every triangle outside the ring
  // (the listed queries are the ones whose certificates failed)
  const bool cert_on = a.cert && !LISTED && a.use_hint && __builtin_amdgcn_readfirstlane(okv) != 0;
  // (cert.hpp: an anchored pass leaves standing certificates as they are)
  const bool anchored = a.cert && cert_anchored(la.cert_ok);
  // far certificates hold for the cap they were made for, or a smaller one
  const bool far_on = cert_on && a.cap2 <= uniform_f64(capv) * (1.0 + 1e-12);

  auto set_best = [&](double nb) {
    const double r = sqrt(nb) + tau;
    const double nbound = r * r;
    if (nbound < best)
      cnt = 0;
    best = nb;
    bound = nbound;
    boundf = fminf(fmaxf(__double2float_ru(nbound), FLT_MIN), FLT_MAX);
    boundg = boundf * ginv2;
  };

  uint32_t e = 0, eend = 0;
  if (active)
  {
    const float slack = cert_on ? slack0 : 0.f;
    x = xform(T, p3);
    xf = (float)x.x;
    yf = (float)x.y;
    zf = (float)x.z;
    const float smax = fmaxf(fmaxf(a.scene_max, fabsf(xf)), fmaxf(fabsf(yf), fabsf(zf)));
    eabs = smax * 2.3841858e-07f * 1.0000002f;
    tau = 1e-9 * (double)smax;
    if (cert_on && (slack > 0.f || (slack < 0.f && far_on)))
    {
      // the move since the certificate was last brought up to date
      const d3 xp = xform(Tc, p3);
      const double mx = x.x - xp.x, my = x.y - xp.y, mz = x.z - xp.z;
      const float m2 = __double2float_ru(mx * mx + my * my + mz * mz);
      const float mv = __builtin_amdgcn_sqrtf(m2) * 1.000001f;
      // a winner stands while the others stay farther (each side moves by mv); "nothing
      // within the cap" stands while the nearest stays beyond it (one side moves)
      const float left = slack > 0.f ? slack - 2.000001f * mv - 1e-30f
                                     : -slack - 1.000001f * mv - 1e-30f;
      if (left > 0.f)
      {
        const bool sc = cert_is_set(slack);
        const bool ld = a.lead && cert_has_lead(slack);
        const float keep = slack > 0.f ? cert_encode(left * 0.9999999f, sc, ld) : -left * 0.9999999f;
        if (!anchored)
          a.cert[q] = keep;
        certified = keep != 0.f;
        certset = sc && certified;
        // (the winner's lead over the set's other members still stands: no weighing)
        if (certset && ld && cert_lead_stands(a.lead, q, lead00, mv, anchored))
          certset = false;
        // (where lean passes are to follow, a single winner's certificate that is nearly
        // used up is made anew here, or as a pair by the tree kernel, while that costs a
        // search among hundreds: the lean passes hold every query against the certificates
        // as the last full pass leaves them)
        if (la.pairs_to_tree && slack > 0.f && !sc && keep < 2.f * eabs)
          certified = false;
      }
    }
    hs = (a.use_hint && !certified) ? a.hint[q] : -1;
  }
  if (active)
  {
    // (a query without a winner had nothing within the cap last time: the tree kernel looks
    // again, from the root)
    // a triangle can settle the query only if the query is within eps/2 of it; its box
    // gives a cheap lower bound of that distance
    gx = (xf - la.grid.base[0]) * la.grid.inv_step;
    gy = (yf - la.grid.base[1]) * la.grid.inv_step;
    gz = (zf - la.grid.base[2]) * la.grid.inv_step;
    const float half = 0.5f * la.eps;
    const float reach_g = half * half * ginv2;
    const float qx2 = 2.f * gx, qy2 = 2.f * gy, qz2 = 2.f * gz;
    bool moving = hs >= 0;
    for (int step = 0; moving; step++)
    {
      const uint4 wb = la.qbox[hs];
      e = la.noff[hs];
      eend = la.noff[hs + 1];
      if (eend - e > NBR_CAP)
        break; // its list was cut short: nothing can be concluded here
      if (qbox_g2(wb, gx, gy, gz) <= reach_g && a.tid[hs] >= 0)
      {
        d3 P0, P1, P2;
        load_tri(a.tverts, hs, P0, P1, P2);
        const double d2 = tri_d2_screen(x, P0, P1, P2);
        // every contender is within r of the query, hence within 2r of this triangle
        const double rr = sqrt(d2) + tau;
        if (2.0 * rr * 1.000001 + 4.0 * (double)eabs <= (double)la.eps)
        {
          best = DBL_MAX;
          set_best(d2);
          local = true;
          // a triangle outside this list is farther than eps from this triangle's box,
          // which the query is within sqrt(d2) of
          const float out = la.eps - __double2float_ru(sqrt(d2)) * 1.000001f - 4.f * eabs;
          sec2 = out > 0.f ? out * out * 0.999999f : 0.f;
          break;
        }
      }
      if (step == LOCAL_STEPS)
        break;
      // a query a few list radii from its start triangle, yet beyond reach, is off the
      // surface rather than displaced along it: no walk along the surface brings it within
      // reach, so it goes to the tree kernel at once
      if (qbox_g2(wb, gx, gy, gz) <= 64.f * reach_g)
        break;
      // move to the listed triangle whose box centre is nearest to the query
      float bestc = INFINITY;
      int32_t bestm = hs;
      for (uint32_t i = e; i < eend; i += 4)
      {
        const int4 m4 = *reinterpret_cast<const int4*>(la.nbr + i);
        const int32_t m[4] = { m4.x, m4.y, m4.z, m4.w };
        uint4 wq[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
          wq[k] = la.qbox[max(m[k], 0)];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          // twice the box centre, in grid steps, against twice the query
          const float cx = ((float)(wq[k].x & 0xffffu) + (float)(wq[k].y >> 16)) - qx2;
          const float cy = ((float)(wq[k].x >> 16) + (float)(wq[k].z & 0xffffu)) - qy2;
          const float cz = ((float)(wq[k].y & 0xffffu) + (float)(wq[k].z >> 16)) - qz2;
          const float dc = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, cx * cx));
          // (strict, and ties to the lower position, so the walk cannot cycle)
          const bool up = m[k] >= 0 && (dc < bestc || (dc == bestc && m[k] < bestm));
          bestc = up ? dc : bestc;
          bestm = up ? m[k] : bestm;
        }
      }
      if (bestm == hs)
        moving = false; // nothing listed is nearer: the surface is not within reach here
      hs = bestm;
    }
    if (local)
    {
      ring[0][tid] = hs;
      cnt = 1;
    }
    else if (hs >= 0 && a.tid[hs] < 0)
      hs = -1; // a degenerate triangle is no starting bound for the tree kernel
  }
  // queries this kernel cannot settle go to the tree kernel, with the triangle they
  // reached as their hint; one atomic per block
  {
    const bool fb = active && !local && !certified;
    const unsigned long long mf = __ballot(fb);
    if (lane == 0)
      wcount[tid >> 6] = (uint32_t)__popcll(mf);
    __syncthreads();
    if (tid == 0)
    {
      const uint32_t tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
      bbase = tot ? atomicAdd(&a.counters[3], tot) : 0u;
    }
    __syncthreads();
    if (fb)
    {
      uint32_t pos = bbase + (uint32_t)__popcll(mf & lt);
      for (int w = 0; w < (tid >> 6); w++)
        pos += wcount[w];
      la.fallback[pos] = (uint32_t)q;
      a.hint[q] = hs;
    }
  }

  // (a query that stands on a set certificate has its contenders weighed again by
  // k_contested: a whole pass marks the 64 queries of each wavefront in one word; a listed
  // pass has had that done by k_certify)
  if (!LISTED && a.setmask)
  {
    const unsigned long long ms = __ballot(certset);
    const int64_t w0 = first + (tid & ~63);
    if (lane == 0 && w0 < total)
      a.setmask[w0 >> 6] = ms;
  }

  // The list is taken eight entries at a time: the indices, then their boxes, are in
  // flight together (the loads of one entry depend on each other, those of different
  // entries do not).  Entries whose boxes pass meet the edge half-spaces; the few that
  // survive are held back and screened after the scan, so that the fp64 screen runs once
  // or twice per wavefront instead of once per "some lane's box passed".
  const float alw = 6.f * eabs;
  const float step2lo = la.grid.step * la.grid.step * 0.99999f; // grid steps^2 -> length^2, rounded down
  int npend = 0;
  auto screen = [&](int32_t s) {
    d3 P0, P1, P2;
    load_tri(a.tverts, s, P0, P1, P2);
    const double d2 = tri_d2_screen(x, P0, P1, P2);
    sec2 = note_screened(sec2, d2, best, bound, tau, cnt);
    if (d2 < best)
      set_best(d2);
    if (d2 <= bound)
    {
      if (cnt < RING3)
        ring[cnt][tid] = s;
      cnt++;
    }
  };
  while (local && e < eend)
  {
    // eight entries (two aligned groups) per round
    const int4 ma = *reinterpret_cast<const int4*>(la.nbr + e);
    int4 mb = make_int4(-1, -1, -1, -1);
    if (e + 4 < eend)
      mb = *reinterpret_cast<const int4*>(la.nbr + e + 4);
    const int32_t m[8] = { ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w };
    uint4 wq[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
      wq[k] = la.qbox[max(m[k], 0)];
    e += 8;
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      const int32_t s = m[k];
      if (s < 0 || s == hs)
        continue;
      const float g2 = qbox_g2(wq[k], gx, gy, gz);
      if (g2 > boundg)
      {
        sec2 = fminf(sec2, g2 * step2lo); // ruled out by its box: at least that far
        continue;
      }
      const float ex = edge_excess(la.edges, s, xf, yf, zf, alw);
      if (a.tid[s] < 0)
        continue; // degenerate: never a closest cell
      if (ex > 0.f && ex * ex * 0.999999f > boundf)
      {
        sec2 = fminf(sec2, ex * ex * 0.999999f); // outside an edge by more than the bound
        continue;
      }
      if (npend < LOCAL_PEND)
        pendq[npend++][tid] = s;
      else
        screen(s);
    }
  }
  for (int i = 0; i < npend; i++)
    screen(pendq[i][tid]);

  // the certificate of a query settled with a single contender; none for any other query
  // that was looked at
  bool to_tree = false;
  if (a.cert && active && !certified && local)
  {
    float slack = 0.f;
    if (local && cnt >= 1 && cnt <= RING3)
    {
      const float dw = __double2float_ru(sqrt(best));
      slack = __builtin_amdgcn_sqrtf(sec2) * 0.999999f - dw * 1.000001f - 2.1f * (float)tau -
        fminf(eabs, 0.2f * eabs * eabs / fmaxf(dw, 1e-30f)); // (see make_certificate)
    }
    a.cert[q] = cert_rebase_at(cert_encode(slack, cnt >= 2), la.cert_ok, la.cert_T, x, p3);
    to_tree = la.pairs_to_tree && cnt == 1 && !(slack >= 2.f * eabs);
  }
  if (la.pairs_to_tree)
  {
    // (the winner found here is the tree kernel's starting bound: hint[] below)
    const unsigned long long mp = __ballot(to_tree);
    if (mp)
    {
      const int leader = __ffsll((long long)mp) - 1;
      uint32_t base = 0;
      if (lane == leader)
        base = atomicAdd(&a.counters[3], (uint32_t)__popcll(mp));
      base = __shfl(base, leader, 64);
      if (to_tree)
        la.fallback[base + __popcll(mp & lt)] = (uint32_t)q;
    }
  }

#ifdef ICPX_STATS
  {
    // development build only: queries let through on their certificate / given a new one
    const unsigned long long mc = __ballot(certified);
    const unsigned long long mn = __ballot(a.cert && active && !certified && a.cert[q] != 0.f);
    if (lane == 0)
    {
      atomicAdd(&g_stats[6], (unsigned long long)__popcll(mc));
      atomicAdd(&g_stats[7], (unsigned long long)__popcll(mn));
    }
  }
#endif

  // settle (as the tree kernel does)
  const bool tie = local && cnt >= 2 && cnt <= RING3;
  const bool ovf = local && cnt > RING3;
  if (local && !tie && !ovf)
    a.hint[q] = (cnt == 1) ? ring[0][tid] : -1;
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
  if (!LISTED)
    break;
  __syncthreads(); // (the rings and counts are reused by the next round)
  first += (int64_t)(gridDim.x / CERTIFY_SEGS) * 256;
  } while (first < total);
}

// The certificate check of k_search_local on its own, one light lane per query: a pass in
// which nearly every query stands on its certificate is then a short streaming kernel, and
// the search kernel (whose registers allow few wavefronts per CU) runs over the list of
// those that do not.
struct CertifyArgs
{
  const float* qpos;
  int64_t nq;
  float* cert;
  const double* T;
  const int32_t* stop;
  const double* cert_T;
  const int32_t* cert_ok;
  const double* cert_cap2;
  double cap2;
  // The list of queries the search kernel has to look at comes in CERTIFY_SEGS segments of
  // segcap entries, each with a count of its own (work_count[s * CERTIFY_SEG_STRIDE]; zero on
  // entry, the pass's reduction leaves them so): block b appends to segment b mod
  // CERTIFY_SEGS.  Atomics on one address go through at some 10 ns apiece, which a single
  // count would make the length of this kernel.  An entry with the top bit set passed the
  // check but stands on a set certificate: the search kernel puts it on the tie list.
  uint32_t* work;
  uint32_t* work_count;
  int64_t segcap;
  int rounds; // queries per block / 256 (1 .. CERTIFY_ROUNDS)
  unsigned long long* setmask; // one bit per query: passed, and stands on a set certificate
                               // whose members have to be weighed again
  float* lead;                 // the leads of sets' winners (Search3Args::lead), or null
  // not null: the failed queries go on this one dense list (length *direct_count), straight
  // for the tree kernels, instead of the segmented one for the shortcut
  uint32_t* direct;
  uint32_t* direct_count;
};

constexpr int CERTIFY_ROUNDS = 16;

__global__ __launch_bounds__(256) void k_certify(CertifyArgs c)
{
  Xf T, Tc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t stopv = c.stop ? *c.stop : 0;
  const int32_t okv = *c.cert_ok;
  const double capv = *c.cert_cap2;
#pragma unroll
  for (int i = 0; i < 12; i++)
  {
    T.m[i] = c.T[i];
    Tc.m[i] = c.cert_T[i];
  }
  if (__builtin_amdgcn_readfirstlane(stopv) != 0)
    return;
#pragma unroll
  for (int i = 0; i < 12; i++)
  {
    T.m[i] = uniform_f64(T.m[i]);
    Tc.m[i] = uniform_f64(Tc.m[i]);
  }
  // per round and wavefront: queries for the work list / for the tie list; then their
  // exclusive prefix sums in the same order
  __shared__ uint32_t wcount[CERTIFY_ROUNDS * 4];
  __shared__ uint32_t bbase;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const bool cert_on = __builtin_amdgcn_readfirstlane(okv) != 0;
  const bool far_on = cert_on && c.cap2 <= uniform_f64(capv) * (1.0 + 1e-12);
  const bool anchored = cert_anchored(c.cert_ok); // (cert.hpp: standing certificates stay as they are)
  const int rounds = c.rounds;
  const int64_t first = (int64_t)blockIdx.x * (256 * rounds) + tid;
  uint32_t todo_bits = 0;
  for (int k = 0; k < rounds; k++)
  {
    const int64_t q = first + (int64_t)k * 256;
    const bool active = q < c.nq;
    const int64_t qc = active ? q : 0;
    const float p3[3] = { c.qpos[3 * qc], c.qpos[3 * qc + 1], c.qpos[3 * qc + 2] };
    const float slack = c.cert[qc];
    const float leadv = c.lead ? c.lead[qc] : 0.f; // (with the certificate, not behind it)
    bool certified = false, certset = false;
    if (active && cert_on && (slack > 0.f || (slack < 0.f && far_on)))
    {
      // (the same arithmetic as in k_search_local)
      const d3 x = xform(T, p3);
      const d3 xp = xform(Tc, p3);
      const double mx = x.x - xp.x, my = x.y - xp.y, mz = x.z - xp.z;
      const float m2 = __double2float_ru(mx * mx + my * my + mz * mz);
      const float mv = __builtin_amdgcn_sqrtf(m2) * 1.000001f;
      const float left = slack > 0.f ? slack - 2.000001f * mv - 1e-30f
                                     : -slack - 1.000001f * mv - 1e-30f;
      if (left > 0.f)
      {
        const bool sc = cert_is_set(slack);
        const bool ld = c.lead && cert_has_lead(slack);
        const float keep = slack > 0.f ? cert_encode(left * 0.9999999f, sc, ld) : -left * 0.9999999f;
        if (!anchored)
          c.cert[q] = keep;
        certified = keep != 0.f;
        certset = sc && certified;
        // (the winner's lead over the set's other members still stands: no weighing)
        if (certset && ld && cert_lead_stands(c.lead, q, leadv, mv, anchored))
          certset = false;
      }
    }
    const bool todo = active && !certified;
    todo_bits |= (todo ? 1u : 0u) << k;
    // (the wavefront's 64 consecutive queries: their marks are one word)
    const unsigned long long ms = __ballot(certset);
    const int64_t w0 = (int64_t)blockIdx.x * (256 * rounds) + (int64_t)k * 256 + wave * 64;
    if (lane == 0 && w0 < c.nq)
      c.setmask[w0 >> 6] = ms;
    const unsigned long long mf = __ballot(todo);
    if (lane == 0)
      wcount[k * 4 + wave] = (uint32_t)__popcll(mf);
  }
  __syncthreads();
  const int seg = (int)(blockIdx.x % CERTIFY_SEGS);
  if (tid == 0)
  {
    // (the counts to their exclusive prefix sums, one atomic)
    uint32_t run = 0;
    for (int i = 0; i < rounds * 4; i++)
    {
      const uint32_t v = wcount[i];
      wcount[i] = run;
      run += v;
    }
    bbase = !run ? 0u
      : c.direct ? atomicAdd(c.direct_count, run)
                 : atomicAdd(&c.work_count[seg * CERTIFY_SEG_STRIDE], run);
  }
  __syncthreads();
  uint32_t* const work = c.direct ? c.direct : c.work + seg * c.segcap;
  for (int k = 0; k < rounds; k++)
  {
    const int64_t q = first + (int64_t)k * 256;
    const bool listed = (todo_bits >> k) & 1u;
    const unsigned long long mf = __ballot(listed);
    if (listed)
      work[bbase + wcount[k * 4 + wave] + (uint32_t)__popcll(mf & lt)] = (uint32_t)q;
  }
}

} // namespace icpx
