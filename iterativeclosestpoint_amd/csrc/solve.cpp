// solve.cpp — host entry point of the O(1) solve (the arithmetic is in solve_core.hpp).
#include "solve_core.hpp"
#include <cmath>

extern "C" int icpx_solve(int metric, const double sums[ICPX_NSUMS], const double origin[3],
  double M[16], int64_t* n_out, double* mean_dist_out)
{
  if (!M)
    return ICPX_E_INVALID;
  double n = 0.0;
  const int rc = icpx_core::solve_moments(metric, sums, origin, M, &n, mean_dist_out);
  if (n_out)
    *n_out = (int64_t)std::llround(n);
  return rc;
}
