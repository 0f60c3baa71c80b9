// cert.hpp — the certificates that let a query keep its result from pass to pass without a
// search: how they are encoded, made from what a search turned away, and widened to a pair
// when the nearest loser is a hair behind the winner.  Shared by every search kernel that
// keeps certificates (search2.hpp, search3.hpp, local.hpp).
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpx
{

constexpr int RING3 = 8;

// A positive certificate's last two mantissa bits say what stands while it lasts.  Bit 0: 0, the
// single winner; 1, the set of contenders kept in cand[] (a query on an edge or a vertex shared
// by several triangles has them at the same distance for good; a winner with a rival a hair
// behind it is kept as a set of two): no search is needed to know that nobody else is in the
// race, and the winner among them is worked out by exact evaluation.  Bit 1 (sets only): the
// last such weighing found the winner ahead of every other member by the lead on record in
// lead[] (a distance, worn down by the passes' moves exactly as the certificate is): while that
// lasts too, not even the weighing is needed, the winner in hint[] stands.  Every maker of a
// certificate goes through cert_encode without the third argument, so a new set never has a
// lead on record; only the weighing kernels raise the bit (cert_note_lead).  (The value gives
// way by up to four units in the last place.)
__device__ __forceinline__ float cert_encode(float c, bool set, bool lead = false)
{
  const uint32_t b = __float_as_uint(c);
  if (!(c > 0.f) || b < 16u)
    return 0.f;
  return __uint_as_float(((b - 4u) & ~3u) | (set ? 1u : 0u) | ((set && lead) ? 2u : 0u));
}
__device__ __forceinline__ bool cert_is_set(float c)
{
  return c > 0.f && (__float_as_uint(c) & 1u) != 0u;
}
__device__ __forceinline__ bool cert_has_lead(float c)
{
  return c > 0.f && (__float_as_uint(c) & 3u) == 3u;
}
// By how much (a distance) the nearest member of a set is ahead of the next one, for lead[]:
// d1 <= d2 their squared distances (exact or screened), tau the tie tolerance.  While twice the
// query's move stays below it the exact comparison names the same winner, whatever the ids.
__device__ __forceinline__ float lead_of(double d1, double d2, double tau)
{
  const float l = __double2float_rd(sqrt(d2)) * 0.999999f - __double2float_ru(sqrt(d1)) * 1.000001f -
    2.1f * (float)tau;
  return (l > 0.f && l < 1e30f) ? l : 0.f;
}

// Anchored passes.  While a registration only jitters about its fixed point, the certificates
// are held against the transform of the last pass that brought them all up to date (their
// anchor) and left as they are, instead of being worn down by every pass's move (a cloud that
// moves by 1e-9 a pass, back and forth, uses up 4e-9 of every slack per pass: of sixteen
// million slacks spread over a few 1e-6 that is sixteen thousand searches a pass, for good).
// The state's words behind cert_ok say whether the coming pass is such a one ([2]).  A
// certificate made during such a pass is good for the query's place NOW; stored less twice the
// query's distance from where the anchor has it, it is good measured from the anchor, which is
// how it will be checked.
__device__ __forceinline__ bool cert_anchored(const int32_t* cert_ok)
{
  return cert_ok && __builtin_amdgcn_readfirstlane(cert_ok[2]) != 0;
}
// c: a certificate as the makers return it; adj: twice the query's distance from the anchor
__device__ __forceinline__ float cert_rebase(float c, float adj)
{
  if (c == 0.f)
    return c;
  if (c > 0.f)
    return cert_encode(c - adj, cert_is_set(c));
  const float l = -c - 0.5f * adj; // (a far certificate stands while ONE move stays below it)
  return l > 0.f ? -l : 0.f;
}

// The certificate of a finished query.  cnt: contenders (1: a single winner; 2..RING3: a set,
// all in the ring); best: the winner's screened squared distance; sec2: lower bound of the
// squared distance of every screened triangle that is not (or no longer) in the ring;
// minraw: smallest raw squared box distance (box_d2) among the boxes turned away;
// minlin: smallest lower bound, as a distance, of triangles turned away by cheap bounds.
__device__ __forceinline__ float make_certificate(int cnt, double best, float sec2, float minraw,
  float minlin, double tau, float eabs, double cap2)
{
  const bool near = cnt >= 1 && cnt <= RING3;
  // (a raw box distance exceeds the true one by at most sqrt(3) eabs)
  const float dbox = __builtin_amdgcn_sqrtf(minraw) * 0.999999f - 1.7320509f * eabs;
  const float others = fminf(fminf(dbox, minlin), __builtin_amdgcn_sqrtf(sec2) * 0.999999f);
  // what the screened distance of the winner may be off by: a few units in the last place
  // of the squared coordinates, i.e. ~0.02 eabs^2 / distance, never more than eabs
  const float dw = __double2float_ru(sqrt(near ? best : cap2));
  const float slop = fminf(eabs, 0.2f * eabs * eabs / fmaxf(dw, 1e-30f)) + 2.1f * (float)tau;
  float c = 0.f;
  if (near)
    c = others - dw * 1.000001f - slop;
  else if (cnt == 0 && cap2 < 1e300)
    c = -(others - dw * 1.000001f - slop);
  // (a near certificate must be positive, a far one negative; infinities mean nothing was
  // ever turned away, which only a mesh without triangles does)
  if (near)
    return (c > 0.f && c < 1e30f) ? cert_encode(c, cnt >= 2) : 0.f;
  return (c < 0.f && c > -1e30f) ? c : 0.f;
}

// A screened triangle joins the bookkeeping of the certificate: one that is turned away is
// at least its distance away; a new minimum that empties the ring leaves contenders behind
// that were at least the old minimum away.  (Call before set_best / the ring's update.)
__device__ __forceinline__ float note_screened(float sec2, double d2, double best, double bound,
  double tau, int cnt)
{
  if (d2 < best)
  {
    const double r = sqrt(d2) + tau;
    return (r * r < best && cnt > 0) ? fminf(sec2, __double2float_rd(best) * 0.999999f) : sec2;
  }
  return d2 <= bound ? sec2 : fminf(sec2, __double2float_rd(d2) * 0.999999f);
}

// The same bookkeeping when the nearest loser is to be known by name as well (a walk that
// screens every triangle it meets): r1 / r1id, the smallest such lower bound and the triangle
// it belongs to; r2, the smallest among all the others.  ring0: the ring's first member.
__device__ __forceinline__ void note_loser(float& r1, int32_t& r1id, float& r2, float v, int32_t id)
{
  if (v < r1)
  {
    r2 = r1;
    r1 = v;
    r1id = id;
  }
  else
    r2 = fminf(r2, v);
}
__device__ __forceinline__ void note_screened_named(float& r1, int32_t& r1id, float& r2, double d2,
  int32_t s, double best, double bound, double tau, int cnt, int32_t ring0)
{
  if (d2 < best)
  {
    const double r = sqrt(d2) + tau;
    if (r * r < best && cnt > 0)
    {
      // the ring is about to be emptied: its members were at least the old minimum away,
      // the first of them exactly that
      const float v = __double2float_rd(best) * 0.999999f;
      note_loser(r1, r1id, r2, v, ring0);
      if (cnt > 1)
        r2 = fminf(r2, v);
    }
  }
  else if (!(d2 <= bound))
    note_loser(r1, r1id, r2, __double2float_rd(d2) * 0.999999f, s);
}

// A single winner whose nearest loser is next to nothing farther away (a query whose foot
// point lies a hair inside an edge: the triangle across that edge) gets no certificate worth
// having from make_certificate, pass after pass.  If the loser after that one is well
// away, the winner and its nearest loser are kept as a set of two instead (see cert_encode):
// the returned certificate then covers every triangle outside that pair.  pair: the set was
// chosen; `floor`: a single-winner certificate at least this large is kept as it is.
__device__ __forceinline__ float make_certificate_pair(int cnt, double best, float r1, float r2,
  int32_t r1id, float minraw, float minlin, double tau, float eabs, double cap2, float floor_,
  bool& pair)
{
  pair = false;
  const float single = make_certificate(cnt, best, r1, minraw, minlin, tau, eabs, cap2);
  if (cnt != 1 || r1id < 0 || single >= floor_)
    return single;
  // (cnt = 2: the encoding of a set; the arithmetic is the single winner's with the nearest
  // loser left out of "the others")
  const float both = make_certificate(2, best, r2, minraw, minlin, tau, eabs, cap2);
  if (both > single && both > 0.f)
  {
    pair = true;
    return both;
  }
  return single;
}

} // namespace icpx
