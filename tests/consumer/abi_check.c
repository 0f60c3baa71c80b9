/* A plain C99 consumer of include/icpx.h and include/icpx_dev.h: proves that the headers
 * compile as C, that the library links from C, and that the host-only entry points behave
 * as declared (no GPU is needed: icpx_create on a device that does not exist must fail
 * cleanly, icpx_solve runs on the host). */
#include "icpx.h"
#include "icpx_dev.h"

#include <stdio.h>
#include <string.h>

int main(void)
{
  icpx_params p;
  icpx_ctx* ctx = (icpx_ctx*)0;
  double sums[ICPX_NSUMS], origin[3] = { 0.0, 0.0, 0.0 }, M[16], mean = -1.0;
  int64_t n = -1;
  int i, rc;

  icpx_default_params(&p);
  /* vtkICP's constructor defaults (ICP/vtkICP.cxx:27-40), landmarks as documented */
  if (p.metric != ICPX_METRIC_POINT_TO_PLANE || p.max_iterations != 50 || p.max_landmarks != 200 ||
    p.max_distance != 10.0 || p.check_mean_distance != 0 || p.max_mean_distance != 0.01 ||
    p.min_normal_dot != 0.0 || p.rotate_source_normals != 0)
  {
    fprintf(stderr, "unexpected defaults\n");
    return 1;
  }
  rc = icpx_create(&ctx, 9999);
  if (rc != ICPX_E_INVALID || ctx != (icpx_ctx*)0 || strlen(icpx_last_error()) == 0)
  {
    fprintf(stderr, "icpx_create(9999) = %d (%s)\n", rc, icpx_last_error());
    return 2;
  }
  for (i = 0; i < ICPX_NSUMS; i++)
    sums[i] = 0.0;
  rc = icpx_solve(ICPX_METRIC_POINT_TO_PLANE, sums, origin, M, &n, &mean);
  if (rc != ICPX_E_FEW || n != 0)
  {
    fprintf(stderr, "icpx_solve on empty sums = %d, n = %ld\n", rc, (long)n);
    return 3;
  }
  for (i = 0; i < 16; i++)
    if (M[i] != ((i % 5 == 0) ? 1.0 : 0.0))
      return 4;
  if (icpx_last_build_seconds((icpx_ctx*)0) != 0.0) /* (a development entry point links too) */
    return 5;
  printf("abi ok, %d device(s)\n", icpx_device_count());
  return 0;
}
