// state.hpp — what a registration keeps on the device between iterations, and the step
// that advances it.
#pragma once
#include "solve_core.hpp"
#include <stdint.h>

namespace icpx
{

// State of a registration that stays on the device between iterations.
struct RunState
{
  double T[16]; // running transform, row-major
  // ---- the 32 bytes from here on are what the host polls (RunRecord)
  double mean, n;  // of the last solved iteration
  int32_t iters;   // solved iterations (vtkICP's NumberOfIterations)
  int32_t status;  // ICPX_RUN_*
  int32_t done;    // 1: converged / too few rows; 2: a lean pass found a query it could not
                   // answer and the host has to repeat it in full; later passes return at once
  uint32_t fallback; // queries the shortcut left to the tree walk in the last pass
  // ---- the shortcut's certificates (local.hpp): the transform they were last brought up
  // to date for (rows 0..2), and whether they are up to date at all
  double cert_T[12];
  double cert_cap2; // the cap (squared) the far certificates were made for
  int32_t cert_ok;
  int32_t lean_fail; // raised by a lean pass (k_finalize_mx<.., LEAN>) that met such a query
  // (cert.hpp: anchored passes) whether the coming pass leaves the certificates and cert_T as
  // they are
  int32_t anchored;
  int32_t pad_;
};
struct RunRecord
{
  double mean, n;
  int32_t iters, status, done;
  uint32_t fallback;
  uint32_t seq; // the solve's serial number, stored after the rest: the host waits for it
  uint32_t listed; // queries on k_certify's list of failed certificates in that pass (0: no such list)
  uint32_t contested; // queries that pass left to k_contested through the tie and overflow lists
  uint32_t nq_total;  // queries those three counts are of: this rank's, or (a loop whose sums are
                      // all-reduced with the figures riding in the spare slots) every rank's
};
// The spare slots of the ICPX_NSUMS vector (no metric has more than 35 sums).  In the loop of
// several ranks (icpx_run_advance) the reduction puts this rank's policy figures there, the
// all-reduce adds them up with the sums, and every rank's solve reads the job's totals: all
// ranks see the same records, decide alike and enqueue the same sequence of passes.
constexpr int SLOT_NQ = 35;        // queries searched
constexpr int SLOT_FALLBACK = 36;  // ... left to the tree walk by the shortcut
constexpr int SLOT_LISTED = 37;    // ... on k_certify's list
constexpr int SLOT_CONTESTED = 38; // ... left to k_contested's lists
constexpr int SLOT_LEANFAIL = 39;  // ranks whose lean pass met a query it could not answer
__device__ __host__ inline uint32_t slot_count(double v)
{
  return v > 0.0 ? (v < 4294967295.0 ? (uint32_t)v : 0xffffffffu) : 0u;
}
// One iteration's solve and transform update in a single GPU thread: the body of the
// loop in vtkICP::InternalUpdate after the correspondence filter (ICP/vtkICP.cxx:218-245),
// and the iteration's record for the host, written straight to pinned host memory, where
// the host finds it two iterations later by its serial number (no event, hence no marker
// in the queue between iterations).  run: this pass ran and the registration is not over;
// fallback_now: the pass's count of queries the shortcut left to the tree walk; listed_now:
// the length of k_certify's list in that pass; contested_now: its tie and overflow lists.
// (Everything the step reads is requested at once and the record is written from
// registers: one lane's chain of dependent reads is what this costs.)
#ifdef ICPX_STATS
__device__ unsigned long long g_rs_stamps[8];
#define RS_STAMP(i) g_rs_stamps[i] = wall_clock64()
#else
#define RS_STAMP(i)
#endif
__device__ inline void solve_and_publish(RunState* s, const double* __restrict__ sums, int metric,
  const double* origin, int check_mean, double max_mean, bool ran, uint32_t fallback_now,
  uint32_t listed_now, uint32_t contested_now, RunRecord* rec, uint32_t seq, float src_rmax = 0.f,
  float anchor_reach = 0.f, uint32_t nq_total = 0u)
{
  double Tm[16], Tc[12];
#pragma unroll
  for (int i = 0; i < 16; i++)
    Tm[i] = s->T[i];
  // (the certificates' transform with the rest: read where the anchor test needs it, the read
  // is a trip to memory behind the solve)
#pragma unroll
  for (int i = 0; i < 12; i++)
    Tc[i] = s->cert_T[i];
  const int32_t cert_ok = s->cert_ok;
  double mean = s->mean, n = s->n;
  int32_t iters = s->iters, status = s->status, done = s->done;
  uint32_t fallback = s->fallback;
  if (ran && !done)
  {
    fallback = fallback_now;
    s->fallback = fallback;
    double M[16];
    RS_STAMP(4);
    const int rc = icpx_core::solve_moments(metric, sums, origin, M, &n, &mean);
    RS_STAMP(5);
    s->mean = mean;
    s->n = n;
    if (check_mean && n > 0.0 && mean <= max_mean)
    {
      status = ICPX_RUN_CONVERGED;
      done = 1;
    }
    else if (rc != ICPX_OK)
    {
      status = ICPX_RUN_TOO_FEW;
      done = 1;
    }
    else
    {
      double Tn[16];
      for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++)
        {
          double acc = 0.0;
          for (int j = 0; j < 4; j++)
            acc += M[4 * r + j] * Tm[4 * j + k];
          Tn[4 * r + k] = acc;
        }
      // (a transform the search kernels cannot take, from sums that carry a NaN or a system
      // singular to within rounding: the registration ends here with the last good one)
      if (!icpx_core::transform_usable(Tn))
      {
        status = ICPX_RUN_DIVERGED;
        done = 1;
      }
      else
      {
      iters++;
      s->iters = iters;
      for (int i = 0; i < 16; i++)
        s->T[i] = Tn[i];
      // Is the coming pass an anchored one (cert.hpp)?  It is while no source point (within
      // src_rmax of the origin) is farther than anchor_reach from where the certificates'
      // transform has it: |dR p + dt| <= |dR|_F |p| + |dt|.
      double r2 = 0.0, t2 = 0.0;
      for (int r = 0; r < 3; r++)
      {
        for (int k = 0; k < 3; k++)
        {
          const double dlt = Tn[4 * r + k] - Tc[4 * r + k];
          r2 += dlt * dlt;
        }
        const double dlt = Tn[4 * r + 3] - Tc[4 * r + 3];
        t2 += dlt * dlt;
      }
      const double reach = (sqrt(r2) * (double)src_rmax + sqrt(t2)) * 1.000001;
      s->anchored = (cert_ok != 0 && anchor_reach > 0.f && reach <= (double)anchor_reach) ? 1 : 0;
      }
    }
    if (done)
    {
      s->status = status;
      s->done = 1;
      s->anchored = 0;
    }
  }
  RS_STAMP(6);
  rec->mean = mean;
  rec->n = n;
  rec->iters = iters;
  rec->status = status;
  rec->done = done;
  rec->fallback = fallback;
  rec->listed = ran ? listed_now : 0u;
  rec->contested = ran ? contested_now : 0u;
  rec->nq_total = nq_total;
  __threadfence_system();
  *reinterpret_cast<volatile uint32_t*>(&rec->seq) = seq;
  RS_STAMP(7);
}

} // namespace icpx
