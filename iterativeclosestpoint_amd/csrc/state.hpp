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
  int32_t done;    // 1: converged / too few rows; later passes return at once
  uint32_t fallback; // queries the shortcut left to the tree walk in the last pass
  // ---- the shortcut's certificates (local.hpp): the transform they were last brought up
  // to date for (rows 0..2), and whether they are up to date at all
  double cert_T[12];
  double cert_cap2; // the cap (squared) the far certificates were made for
  int32_t cert_ok;
  int32_t pad_;
};
struct RunRecord
{
  double mean, n;
  int32_t iters, status, done;
  uint32_t fallback;
  uint32_t seq; // the solve's serial number, stored after the rest: the host waits for it
  uint32_t pad_;
};
// One iteration's solve and transform update in a single GPU thread: the body of the
// loop in vtkICP::InternalUpdate after the correspondence filter (ICP/vtkICP.cxx:218-245).
__device__ inline void solve_step(RunState* s, const double* __restrict__ sums, int metric,
  const double* origin, int check_mean, double max_mean)
{
  double M[16], n = 0.0, md = 0.0;
  const int rc = icpx_core::solve_moments(metric, sums, origin, M, &n, &md);
  s->mean = md;
  s->n = n;
  if (check_mean && n > 0.0 && md <= max_mean)
  {
    s->status = ICPX_RUN_CONVERGED;
    s->done = 1;
    return;
  }
  if (rc != ICPX_OK)
  {
    s->status = ICPX_RUN_TOO_FEW;
    s->done = 1;
    return;
  }
  s->iters = s->iters + 1;
  double R[16];
  for (int r = 0; r < 4; r++)
    for (int k = 0; k < 4; k++)
    {
      double acc = 0.0;
      for (int j = 0; j < 4; j++)
        acc += M[4 * r + j] * s->T[4 * j + k];
      R[4 * r + k] = acc;
    }
  for (int i = 0; i < 16; i++)
    s->T[i] = R[i];
}

// The iteration's record for the host: written straight to pinned host memory, where the
// host finds it two iterations later by its serial number (no event, hence no marker in
// the queue between iterations).
__device__ inline void publish_record(const RunState* s, RunRecord* rec, uint32_t seq)
{
  rec->mean = s->mean;
  rec->n = s->n;
  rec->iters = s->iters;
  rec->status = s->status;
  rec->done = s->done;
  rec->fallback = s->fallback;
  __threadfence_system();
  *reinterpret_cast<volatile uint32_t*>(&rec->seq) = seq;
  __threadfence_system();
}

} // namespace icpx
