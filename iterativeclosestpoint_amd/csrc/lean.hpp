// lean.hpp — what a lean pass (k_finalize_mx<.., LEAN> in query.hpp) does with a query before
// its row is made.
//
// A registration that has settled moves by a hundred-millionth of the scene per iteration,
// back and forth about its fixed point.  The certificates left by the last pass that ran the
// search kernels say, per query, how far it may be from where it then was before its winner
// (or its small set of contenders) has to be looked at again; a lean pass holds every query
// against that and changes nothing: the certificates and the transform they refer to stay
// those of that last full pass (their anchor), so the jitter does not wear them down pass
// after pass as the search kernels' write-back would.
//   - the certificate stands, a single winner: the winner is last pass's (hint[]);
//   - it stands, a set of contenders (cand[]): they are weighed again (lean_weigh_list: a set
//     of two by its members' screened distances, a tie or a larger set by exact evaluation as
//     k_contested does it), the winner goes to hint[];
//   - "nothing within the distance threshold" stands: no row;
//   - no certificate, or it does not reach: the pass has failed (lean_bad): the host repeats
//     it with the search kernels, which give the query a new certificate.  The full passes of
//     a run that may go lean end with k_certify_rest (below), which certifies what the search
//     kernels could not, so that a settled cloud has no such query.
#pragma once
#include "query.hpp"
#include "search2.hpp"

namespace icpx
{

// The winner among the members of the set a query stands on (c0: the first four).
__device__ __forceinline__ int32_t lean_weigh(const FinalArgs& a, int64_t p, const int4& c0,
  const d3& x, bool record = false)
{
  // The winner is the one k_contested's exact evaluation of all members would name.  A set of
  // two is screened first: if the screened distances are farther apart than the tie tolerance
  // the nearer one is the winner, as it is for a search that ends with one contender; only a
  // real tie, or a larger set, is evaluated exactly.
  double bd2 = INFINITY, sd2 = INFINITY; // (the best and the second best member's squared distance)
  int32_t bid = 0x7fffffff, bs = -1;
  bool decided = false;
  const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)x.x)),
    fmaxf(fabsf((float)x.y), fabsf((float)x.z)));
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
  if (!decided)
  {
    // (members read one by one: an array of them indexed by the loop's counter would live in
    // scratch memory)
#pragma unroll 1
    for (int i = 0; i < RING3; i++)
    {
      const int32_t mi = a.cand[p * RING3 + i];
      if (mi < 0)
        break;
      d3 P0, P1, P2, cp;
      double dd, w[3];
      load_tri(a.tverts, mi, P0, P1, P2);
      const int32_t id = a.tid[mi];
      if (tri_eval(x, P0, P1, P2, cp, dd, w))
      {
        if (better(dd, id, bd2, bid))
        {
          sd2 = bd2;
          bd2 = dd;
          bid = id;
          bs = mi;
        }
        else
          sd2 = fmin(sd2, dd);
      }
    }
  }
  a.hint_rw[p] = bs;
  if (record)
    cert_note_lead(a.cert, a.lead, p, (bs >= 0 && sd2 < INFINITY) ? lead_of(bd2, sd2, 1e-9 * (double)smax) : 0.f,
      a.cert_ok, a.cert_T, x, a.qpos + 3 * p);
  return bs;
}

// Certificates for the queries the search kernels left without one (k_certify_rest, run at
// the end of a full pass's searches where lean passes are to follow; T is the pass's
// transform, which its certificates refer to).  A winner with several triangles next to
// nothing behind it (a foot point a hair from a vertex: the fan around it) gets no
// certificate from any search kernel, single or pair, and one such query would fail every
// lean pass.  Here the wavefront takes its queries without a certificate one after the other,
// all 64 lanes on one: if the query is within half the neighbour lists' radius of its winner,
// every triangle that matters is on the winner's list (as in k_search_local), one listed
// triangle per lane is screened, the winner is found again (screened minimum, exact
// evaluation within the tie tolerance: it must be the one the pass found), and the
// certificate is made from all those distances: the members are the triangles within 64
// rounding allowances of the nearest, eight at most, and everything else is at least
// `rest` or the lists' radius away.
__device__ __forceinline__ void certify_rest_wave(const FinalArgs& a, unsigned long long need,
  int64_t p, int lane, const d3& x, int32_t hs)
{
  while (need)
  {
    const int src = __ffsll((long long)need) - 1;
    need &= need - 1ull;
    const d3 xq = mk(__shfl(x.x, src, 64), __shfl(x.y, src, 64), __shfl(x.z, src, 64));
    const int32_t h = __shfl(hs, src, 64);
    const int64_t pq = p - lane + src;
    int32_t win = -1;
    float fresh = 0.f; // the query's new certificate
    if (h >= 0 && a.tid[h] >= 0)
    {
      const uint32_t e0 = a.noff[h], e1 = a.noff[h + 1];
      d3 P0, P1, P2;
      load_tri(a.tverts, h, P0, P1, P2);
      const double dh = tri_d2_screen(xq, P0, P1, P2);
      const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)xq.x)),
        fmaxf(fabsf((float)xq.y), fabsf((float)xq.z)));
      const float eabs = smax * 2.3841858e-07f * 1.0000002f;
      const double tau = 1e-9 * (double)smax;
      if (e1 - e0 <= a.nbr_cap && e1 - e0 <= 128u &&
        2.0 * (sqrt(dh) + tau) * 1.000001 + 4.0 * (double)eabs <= (double)a.eps)
      {
        // lane l: entries l and l + 64 of the list; lane 0 the list's owner as well
        // (scalars, not arrays: an array indexed by a loop counter lives in scratch memory)
        int32_t s0 = -1, s1 = -1;
        double d0 = INFINITY, d1 = INFINITY;
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const uint32_t i = e0 + (uint32_t)(64 * r + lane);
          if (i < e1)
          {
            const int32_t t = a.nbr[i];
            if (t >= 0 && t != h && a.tid[t] >= 0)
            {
              load_tri(a.tverts, t, P0, P1, P2);
              const double dt = tri_d2_screen(xq, P0, P1, P2);
              if (r == 0)
              {
                s0 = t;
                d0 = dt;
              }
              else
              {
                s1 = t;
                d1 = dt;
              }
            }
          }
        }
        const double best = fmin(dh, group_min<64>(fmin(d0, d1)));
        const double rb = sqrt(best) + tau, bound = rb * rb;
        double bd2 = INFINITY;
        int32_t bid = 0x7fffffff, bs = -1;
#pragma unroll 1
        for (int r = 0; r < 3; r++)
        {
          const int32_t t = r == 0 ? s0 : r == 1 ? s1 : (lane == 0 ? h : -1);
          const double dt = r == 0 ? d0 : r == 1 ? d1 : dh;
          if (t < 0 || !(dt <= bound))
            continue;
          d3 cp;
          double dd, w[3];
          load_tri(a.tverts, t, P0, P1, P2);
          if (tri_eval(xq, P0, P1, P2, cp, dd, w))
          {
            const int32_t id = a.tid[t];
            if (better(dd, id, bd2, bid))
            {
              bd2 = dd;
              bid = id;
              bs = t;
            }
          }
        }
        group_argmin<64>(bd2, bid, bs);
        if (bs >= 0 && best < a.cap2)
          win = bs;
        // the certificate: make_certificate's for that set of members
        const double rm = sqrt(best) + tau + 64.0 * (double)eabs, mb = rm * rm;
        const bool in0 = s0 >= 0 && d0 <= mb, in1 = s1 >= 0 && d1 <= mb, own = dh <= mb;
        const unsigned long long b0 = __ballot(in0), b1 = __ballot(in1);
        const int cnt = __popcll(b0) + __popcll(b1) + (own ? 1 : 0);
        double rest = group_min<64>(fmin(in0 ? (double)INFINITY : d0, in1 ? (double)INFINITY : d1));
        if (!own)
          rest = fmin(rest, dh);
        const float out = a.eps - __double2float_ru(sqrt(dh)) * 1.000001f - 4.f * eabs;
        if (win >= 0 && cnt >= 1 && cnt <= RING3 && out > 0.f)
        {
          const float sec2 = rest < (double)INFINITY ? __double2float_rd(rest) * 0.999999f : INFINITY;
          const float c = make_certificate(cnt, best, sec2, INFINITY, out, tau, eabs, a.cap2);
          if (c > 0.f && win == h)
            fresh = c;
          if (fresh != 0.f && cnt >= 2)
          {
            const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int o = own ? 1 : 0;
            if (own && lane == 0)
              a.cand_rw[pq * RING3] = h;
            if (in0)
              a.cand_rw[pq * RING3 + o + __popcll(b0 & lt)] = s0;
            if (in1)
              a.cand_rw[pq * RING3 + o + __popcll(b0) + __popcll(b1 & lt)] = s1;
            if (cnt < RING3 && lane == 0)
              a.cand_rw[pq * RING3 + cnt] = -1;
          }
        }
      }
    }
    if (lane == src && fresh != 0.f)
      a.cert[p] = cert_rebase_at(fresh, a.cert_ok, a.cert_T, x, a.qpos + 3 * p);
  }

}

__global__ __launch_bounds__(256) void k_certify_rest(FinalArgs a)
{
  Xf T;
  if (!load_xf(a.T, T))
    return; // the registration this pass belongs to has already stopped
  const int lane = threadIdx.x & 63;
  const int64_t p0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);
  if (p0 >= a.nq)
    return;
  const int64_t p = min(p0 + lane, a.nq - 1);
  const bool valid = p0 + lane < a.nq;
  const float slack = valid ? a.cert[p] : 1.f;
  const unsigned long long need0 = __ballot(slack == 0.f);
  if (need0 == 0ull)
    return;
  const int32_t hs = slack == 0.f ? a.hint[p] : -1;
  const unsigned long long need = __ballot(hs >= 0);
  if (need == 0ull)
    return;
  const d3 x = xform(T, a.qpos + 3 * p);
  certify_rest_wave(a, need, p, lane, x, hs);
}

// The queries of a wavefront that stand on sets, weighed together once the certificate loop is
// through (list: the queries, in LDS).  Weighed where they are met, a lane at a time, each costs its wavefront two dependent
// reads and two screens while the other 63 lanes wait, and the wavefront with three or four of
// them among its queries is what the whole kernel waits for (44 us a pass instead of 31 for
// the rows alone); here every member of every listed set has a lane of its own, so that all of
// them together are one read and one screen long.  A set of two: the members' screened
// distances; farther apart than the tie tolerance, the nearer one wins, as it does for a
// search that ends with one contender; else both are evaluated exactly, as k_contested does.
// A larger set is left to one lane (lean_weigh).
// record (a full pass's weighing, weigh_marked): the winners' leads go on record (cert.hpp); a
// lean pass leaves certificates and leads as the last full pass left them.
__device__ __forceinline__ void lean_weigh_list(const FinalArgs& a, const Xf& T, const int32_t* list,
  int nset, int lane, bool record = false)
{
  if (nset > 32)
  {
    // many sets (a noisy cloud: a sixteenth of its points are nearest to an edge or stand on
    // pairs): a lane per set, its members one after the other, so that a wavefront's few dozen
    // are one round of reads long, not two
    for (int e0 = 0; e0 < nset; e0 += 64)
    {
      const int e = e0 + lane;
      if (e < nset)
      {
        const int64_t p = list[e];
        (void)lean_weigh(a, p, *reinterpret_cast<const int4*>(a.cand + p * RING3), xform(T, a.qpos + 3 * p),
          record);
      }
    }
    return;
  }
  const int k = lane & 1;
  for (int e0 = 0; e0 < nset; e0 += 32)
  {
    const int e = e0 + (lane >> 1);
    const bool on = e < nset;
    const int64_t p = on ? list[e] : 0;
    // (the set's first members are read here, by the few queries that stand on sets: read
    // with the certificates, 16 bytes of every query's, they cost the pass a third of its time)
    const int4 c0 = on ? *reinterpret_cast<const int4*>(a.cand + p * RING3) : make_int4(-1, -1, -1, -1);
    const int4 ent = make_int4(0, c0.x, c0.y, c0.z >= 0 ? 1 : 0);
    const bool two = on && ent.w == 0;
    const int32_t m = k ? ent.z : ent.y;
    d3 x = mk(0.0, 0.0, 0.0), P0 = x, P1 = x, P2 = x;
    double d = INFINITY;
    if (two)
    {
      x = xform(T, a.qpos + 3 * p);
      load_tri(a.tverts, m, P0, P1, P2);
      d = tri_d2_screen(x, P0, P1, P2);
    }
    const double dp = __shfl_xor(d, 1, 64); // (the partner lane's member)
    bool tie = false;
    if (two)
    {
      const float smax = fmaxf(fmaxf(a.scene_max, fabsf((float)x.x)),
        fmaxf(fabsf((float)x.y), fabsf((float)x.z)));
      const double rb = sqrt(fmin(d, dp)) + 1e-9 * (double)smax;
      tie = !(fmax(d, dp) > rb * rb);
      if (!tie && k == 0)
      {
        a.hint_rw[p] = d < dp ? ent.y : ent.z;
        if (record)
          cert_note_lead(a.cert, a.lead, p, lead_of(fmin(d, dp), fmax(d, dp), 1e-9 * (double)smax),
            a.cert_ok, a.cert_T, x, a.qpos + 3 * p);
      }
    }
    if (__ballot(tie) != 0ull)
    {
      double dd = INFINITY;
      int32_t id = 0x7fffffff;
      if (tie)
      {
        d3 cp;
        double w[3], de;
        if (tri_eval(x, P0, P1, P2, cp, de, w))
        {
          dd = de;
          id = a.tid[m];
        }
      }
      const double ddp = __shfl_xor(dd, 1, 64);
      const int32_t idp = __shfl_xor(id, 1, 64);
      if (tie && k == 0)
      {
        // (members in their order in the set: the first, then the second if it is better)
        int32_t bs = dd < INFINITY ? ent.y : -1;
        if (ddp < INFINITY && (bs < 0 || better(ddp, idp, dd, id)))
          bs = ent.z;
        a.hint_rw[p] = bs;
        if (record) // (within the tie tolerance of each other: no lead worth having)
          cert_note_lead(a.cert, a.lead, p, 0.f, a.cert_ok, a.cert_T, x, a.qpos + 3 * p);
      }
    }
    if (on && ent.w != 0 && k == 0)
      (void)lean_weigh(a, p, c0, xform(T, a.qpos + 3 * p), record);
  }
}

// The marked queries (a.setmask: let through on a set certificate by this pass's check) among
// a wavefront's first + lane, first + stride + lane, ...: listed 64 trips of the rows' loop at a
// time (every lane reads one trip's mask word), weighed by lean_weigh_list whenever the list is
// full and at the end.  first and stride are multiples of 64: a trip's queries are one word.
__device__ __forceinline__ void weigh_marked(const FinalArgs& a, const Xf& T, int64_t first,
  int64_t stride, int lane, int32_t* list, int cap)
{
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t words = (a.nq + 63) >> 6, wstride = stride >> 6;
  int nset = 0;
  for (int64_t w0 = first >> 6; w0 < words; w0 += 64 * wstride)
  {
    const int64_t wi = w0 + (int64_t)lane * wstride;
    const unsigned long long mine = wi < words ? a.setmask[wi] : 0ull;
    unsigned long long any = __ballot(mine != 0ull);
    while (any)
    {
      const int trip = __ffsll((long long)any) - 1;
      any &= any - 1ull;
      const unsigned long long word = __shfl(mine, trip, 64);
      const int64_t p = ((w0 + (int64_t)trip * wstride) << 6) + lane;
      const bool marked = ((word >> lane) & 1ull) != 0ull && p < a.nq;
      const unsigned long long ms = __ballot(marked);
      const int cnt = __popcll(ms);
      if (nset + cnt > cap)
      {
        __builtin_amdgcn_wave_barrier(); // (a wavefront's LDS accesses complete in order)
        lean_weigh_list(a, T, list, nset, lane, a.lead != nullptr);
        __builtin_amdgcn_wave_barrier();
        nset = 0;
      }
      if (marked)
        list[nset + __popcll(ms & lt)] = (int32_t)p;
      nset += cnt;
    }
  }
  __builtin_amdgcn_wave_barrier();
  lean_weigh_list(a, T, list, nset, lane, a.lead != nullptr);
  __builtin_amdgcn_wave_barrier();
}

// The certificate loop of a lean pass for one wavefront: its queries first + lane, first +
// stride + lane, ...: each held against its certificate (k_search_local's arithmetic, the
// anchor's transform Tc); the ones that stand on sets listed for lean_weigh_list (list: room
// for LEAN_LIST entries in LDS, this wavefront's own).
constexpr int LEAN_LIST = 1024;
__device__ __forceinline__ void lean_certify(const FinalArgs& a, const Xf& T, const Xf& Tc,
  bool far_on, int64_t first, int64_t stride, int lane, int32_t* list, bool& lean_bad)
{
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int nset = 0;
  for (int64_t p0 = first; p0 < a.nq; p0 += stride)
  {
    const bool valid = p0 + lane < a.nq;
    const int64_t p = min(p0 + lane, a.nq - 1);
    const float q3[3] = { a.qpos[3 * p], a.qpos[3 * p + 1], a.qpos[3 * p + 2] };
    const float slack = valid ? a.cert[p] : 0.f;
    const float leadv = (valid && a.lead) ? a.lead[p] : 0.f; // (with the certificate, not behind it)
    bool stands = false, led = false;
    if (slack > 0.f || (slack < 0.f && far_on))
    {
      // the move since the anchor
      const d3 x = xform(T, q3), xp = xform(Tc, q3);
      const double mx = x.x - xp.x, my = x.y - xp.y, mz = x.z - xp.z;
      const float m2 = __double2float_ru(mx * mx + my * my + mz * mz);
      const float mv = __builtin_amdgcn_sqrtf(m2) * 1.000001f;
      const float left = slack > 0.f ? slack - 2.000001f * mv - 1e-30f
                                     : -slack - 1.000001f * mv - 1e-30f;
      stands = left > 0.f;
      // (a set whose winner's lead still stands needs no weighing: hint[] has the winner)
      led = a.lead && cert_has_lead(slack) && leadv - 2.000001f * mv - 1e-30f > 0.f;
    }
    lean_bad = lean_bad || (valid && !stands);
    const bool isset = stands && cert_is_set(slack) && !led;
    const unsigned long long ms = __ballot(isset);
    if (ms)
    {
      const int slot = nset + __popcll(ms & lt);
      if (isset)
      {
        if (slot < LEAN_LIST)
          list[slot] = (int32_t)p;
        else // (no room: weighed on the spot)
          (void)lean_weigh(a, p, *reinterpret_cast<const int4*>(a.cand + p * RING3), xform(T, q3));
      }
      nset = min(nset + __popcll(ms), LEAN_LIST);
    }
  }
  __builtin_amdgcn_wave_barrier(); // (a wavefront's LDS accesses complete in order)
  lean_weigh_list(a, T, list, nset, lane);
  __builtin_amdgcn_wave_barrier();
}

} // namespace icpx
