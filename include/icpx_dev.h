/*
 * icpx_dev.h — development and measurement entry points of libicpx.
 *
 * Nothing here replaces a reference interface: these calls select between kernel variants
 * that return the same results, read internal arrays back for tests, and time kernels for
 * the benchmark.  A binding of the reference's classes needs include/icpx.h only.
 */
#ifndef ICPX_DEV_H
#define ICPX_DEV_H

#include "icpx.h"

#ifdef __cplusplus
extern "C"
{
#endif

  /* Wall-clock seconds the last icpx_set_target spent building the BVH and
   * tables on the device (for reporting). */
  double icpx_last_build_seconds(icpx_ctx* ctx);

  /* Search-kernel variant: lanes that cooperate on one query.  1 = one lane per
   * query over a binary box tree (large clouds); 4, 8, 16, 32 or 64 = that many
   * lanes per query over a tree of the same fan-out (64 = one wavefront per
   * query, for a few hundred landmarks); 0 = choose from the number of queries.
   * Trees are built on first use.  Results do not depend on the choice. */
  int icpx_set_group_width(icpx_ctx* ctx, int lanes);

  /* Triangles per leaf of the lane-per-query kernel's tree: 1, 2 or 4 (default). */
  int icpx_set_leaf_size(icpx_ctx* ctx, int triangles);

  /* Hinted shortcut of the lane-per-query path: 1 (default) = once the mean residual is a
   * fraction of the target's edge length, queries close to their previous winner test only
   * the triangles near it (per-triangle neighbour lists, built once per target) instead of
   * walking the tree; the rest still walk it.  A query settled that way with a single
   * contender also keeps a certificate: a lower bound of how much farther every other
   * triangle is, less the tie tolerance; while twice its accumulated move stays below that,
   * later passes keep its winner without searching (a search would return it again).  A
   * query with several contenders at the same distance (on an edge or a vertex) keeps the
   * set of them the same way: while the certificate lasts only the exact evaluation of
   * those few is repeated.
   * 2 = the shortcut without certificates.  0 = always walk the tree.  3 = as 1, with the
   * certificates checked by a light kernel of their own and the shortcut run over the
   * queries that failed (what 1 does by itself for clouds of four million points and more,
   * where the extra launch pays).  Same results. */
  int icpx_set_local(icpx_ctx* ctx, int enable);

  /* Tree kernel of the lane-per-query path.  4 (default) = by regime: a whole pass over a
   * cloud about as dense as the mesh runs one walk per wavefront (the 64 neighbouring
   * queries of a wavefront share the walk; node and triangle records are read once per
   * wavefront), what the shortcut leaves over and sparse clouds run one pooled lane per
   * query.  3 = always one walk per wavefront; 1 = always pooled lanes (a resident grid
   * whose lanes draw queries from a pool as they finish).  Same results. */
  int icpx_set_persistent(icpx_ctx* ctx, int mode);

  /* Pooled kernels: how many of a wavefront's 64 lanes must be idle before it takes new
   * queries off the pool.  0 (default) = by regime (56 on a cold pass, 40 on a hinted one).
   * Same results. */
  int icpx_set_refill(icpx_ctx* ctx, int lanes);

  /* Whole passes of the wavefront walk: after how many microseconds of walking a batch of 64
   * queries is given up on and its queries are searched one wavefront each instead (a few
   * queries beside a patch of triangles much smaller than their distance from it keep their
   * whole batch walking for as long as the rest of the pass takes).  1 (default) = on targets
   * whose triangles' sizes range widely (median longest edge over the shortest edge of
   * the smallest thousandth above 8), a time that goes by the cloud's size (800 us at least); 0 = never; any
   * other value = on every target, that many, and whatever the other batches of the pass took
   * (by default a batch must also have taken four times their mean).  Same results. */
  int icpx_set_bail(icpx_ctx* ctx, int microseconds);

  /* Steps (boxes and leaves met) after which the pooled lanes give up on a query and leave it
   * to the kernel behind them, one wavefront per query; 0: never.  Default 512. */
  int icpx_set_step_budget(icpx_ctx* ctx, int steps);

  /* The wavefront walk sets a query aside, for the kernel behind it (one wavefront per query),
   * whose first bound is more than `edges` of the mesh's edge lengths and eight times the
   * geometric mean of its batch's; 0: never (the default).  For clouds with stray points
   * and no distance threshold. */
  int icpx_set_defer(icpx_ctx* ctx, int edges);

  /* The fused finalize: 0 (default) = by the size of the cloud; 1 = the build that keeps the
   * metric's sums in registers; 2 = the build that accumulates them on the matrix core
   * (v_mfma_f64_16x16x4_f64 over a 16-vector per row).  The two add the rows up in different
   * orders: the sums agree to rounding (1e-13 relative), not to the last bit. */
  int icpx_set_finalize(icpx_ctx* ctx, int build);

  /* Anchored passes of the device-resident loop (default 1 = on).  While a registration only
   * jitters about its fixed point (no source point farther than one rounding allowance of
   * the scene from where the certificates' transform has it), the passes hold the
   * certificates against that transform and leave them as they are, instead of wearing
   * them down by every pass's move; certificates made meanwhile are stored less twice the
   * query's distance from there.  0 = every pass brings the certificates up to date for its
   * own transform.  Same results. */
  int icpx_set_anchor(icpx_ctx* ctx, int enable);

  /* Lean passes of icpx_run's loop on one rank (default 1 = on).  Once a registration of a
   * cloud below four million points has settled, every query stands on a certificate that
   * its winner (or its small set of contenders) cannot have changed, and a pass is two
   * launches: the fused finalize with the certificate check inside it, and the reduction
   * with the solve.  Such a pass is tried when the last iteration record seen was of a pass
   * that left nothing to the tree walk; if it meets a query whose certificate fails, its
   * sums are dropped on the device and the loop repeats that pass with the search kernels.
   * 0 = every pass runs the search kernels; 2 = a lean pass is tried whenever the last pass
   * kept every certificate up to date, whatever the records say (for tests: lean passes
   * then run while the cloud still moves and search many queries themselves); 3 = as 2, and
   * every third lean pass is made to fail on the device, so that the repeat is exercised.
   * Same results to the last bit. */
  int icpx_set_lean(icpx_ctx* ctx, int enable);

  /* How icpx_set_shard cuts the kept points into shards: 2 (default) = the kept points are
   * ordered along the space-filling curve through the target's bounds and chunks of that order
   * (8192 points; fewer for a small cloud, so that every rank has at least eight) are dealt
   * out to the ranks in turn; 1 = `world` equal ranges of that order, one stretch of the
   * surface per rank (the ranks' first passes then differ by a factor of two: some stretches
   * are farther from their place than others); 0 = blocks of consecutive kept indices (each
   * rank's points lie all over the surface, `world` times sparser).  1 and 0 are kept for
   * measurements.  Sums over all shards are the same to rounding. */
  int icpx_set_shard_order(icpx_ctx* ctx, int order);

  /* Where a query without a winner of a last pass gets its first bound from, in the
   * one-walk-per-wavefront kernel: 1 (default) = the better of the triangle its cell of a
   * coarse grid over the target's bounds names and the nearest of the eight around its place
   * on the triangles' Morton curve; 0 = the curve's alone; 2 = the grid's alone.  Same results. */
  int icpx_set_seed(icpx_ctx* ctx, int mode);

  /* How the queries that a pass's certificate check lets through on SET certificates get their
   * winners.  Bit 0 (1): in a pass that ends in the matrix-core build of the fused sums (clouds
   * up to four million points) their sets are weighed by that kernel's wavefronts ahead of
   * their rows; else by the contested-queries kernel in a launch of its own, as passes that
   * emit rows always do.  Bit 1 (2): a weighing also records by how much the winner is ahead of
   * the set's other members, and while that lead outlasts the query's moves the set is not
   * weighed at all.  Default 3; 0 = every certified set weighed every pass by the kernel of its
   * own.  Same winners in every mode.  For measurements. */
  int icpx_set_weigh(icpx_ctx* ctx, int mode);

  /* 1 (default): a cloud of a quarter of a million points or more per rank whose mean residual
   * stays beyond the reach of the shortcut's neighbour lists (a noisy scan) goes by the large
   * cloud's policy whatever its size; 0: the policy goes by size alone.  For measurements. */
  int icpx_set_noisy_policy(icpx_ctx* ctx, int enable);

  /* Number of queries per rank from which a pass's certificates are checked by the light
   * kernel of their own and the host's policy treats the cloud as a large one (default four
   * million; negative restores it).  For measurements of where that boundary belongs. */
  int icpx_set_certify_min(icpx_ctx* ctx, int64_t queries);

  /* Resident single-wavefront blocks per CU of the one-walk-per-wavefront kernel (default 19:
   * what its registers and LDS allow).  For measurements. */
  int icpx_set_wave_blocks(icpx_ctx* ctx, int blocks_per_cu);

  /* The curve the kept source points are ordered along (for the wavefronts' batches of 64
   * consecutive queries, and for the shards): 1 (default) = Hilbert, 0 = Morton.  Same
   * results; the sums' additions come in another order. */
  int icpx_set_query_curve(icpx_ctx* ctx, int hilbert);

  /* Search kernel selection.  0 (default): the production kernel, which screens
   * triangles with a short fp64 routine and evaluates only near-ties (within
   * 1e-9 x the largest coordinate of the minimum) with the reference's exact
   * arithmetic.  1: the exhaustive kernel, which evaluates every visited
   * triangle exactly.  Both return the same winners; the second exists as a
   * cross-check. */
  int icpx_set_exhaustive(icpx_ctx* ctx, int enable);

  /* Diagnostic: copy an internal array to the host.  what = 0: winners of the last search
   * pass (Morton positions of triangles, int32 per query of this shard); 1: Morton
   * position -> cell id (int32 per triangle, -1-id for a degenerate one); 2: {ties,
   * overflows, pool position, shortcut fallbacks} of the last lane-per-query pass (4 x
   * uint32); 3: screened minima of the queries whose contenders overflowed (double per
   * query); 4: {shortcut passes, tree-only passes, neighbour-list builds, list radius x
   * 1e9, list entries, passes whose certificates were checked by the light kernel, passes
   * that enqueued both tree kernels over the leftovers, and of a run's such passes how many
   * the wavefront walk served and how many the pooled lanes, cells the last
   * icpx_point_normals reversed for consistent winding, passes that sent the light
   * kernel's list straight to the tree kernels, lean passes enqueued, lean passes that
   * failed and were repeated in full, whether the last icpx_point_normals met a piece that
   * cannot be wound consistently} (14 x int64); 5: the shortcut's
   * certificates (float per query of
   * this shard: how far the query may still move before its winner is looked at again,
   * 0 = none, negative = how far before something may come within the distance threshold;
   * a positive value whose last mantissa bit is set stands for a set of contenders rather
   * than a single winner, and the bit above it says that the winner's lead is on record,
   * array 9); 6: the library's query order (int32 per query: position -> kept
   * index within the shard; arrays 0, 5 and 7 are in this order); 7: the contender sets
   * (8 x int32 per query, Morton positions, -1 ends a shorter set; meaningful for queries
   * whose certificate stands for a set); 8: the last pass's list of queries left to the
   * tree kernels (uint32 each); 9: per query on a set, by how much its winner was ahead of
   * the set's other members at the last weighing, less the moves since (float; meaningful
   * where the certificate says so). */
  int icpx_debug_dump(icpx_ctx* ctx, int what, void* out, int64_t count);

  /* Forget the winners of the previous search pass.  A pass normally starts each
   * query from the triangle that won last time (only as an upper bound: results
   * do not depend on it); after this call the next pass starts cold, as the
   * first iteration of a new vtkICP::Update() does. */
  int icpx_reset_hints(icpx_ctx* ctx);

  /* Measurement aid: record HIP events on the context's stream around every launch of the
   * search kernels and of the finalize/reduction kernel.  icpx_get_timing synchronises and
   * returns the accumulated kernel milliseconds and launch counts since timing was
   * enabled.  enable = 2 times the search kernels only (every event is a marker in the
   * queue that costs a few microseconds between launches). */
  int icpx_set_timing(icpx_ctx* ctx, int enable);
  int icpx_get_timing(icpx_ctx* ctx, double* search_ms, int64_t* search_launches,
    double* finalize_ms, int64_t* finalize_launches);

  /* The same measurement pass by pass: the search kernels' milliseconds of the passes
   * timed since icpx_set_timing, oldest first (at most `cap` of them are copied);
   * *n_out = how many there are.  kind_out (optional, int32 per pass): 0 = a whole pass of
   * the tree kernel (a cold pass when hinted_out is 0), 1 = the hinted shortcut first, 2 = a
   * lean pass (icpx_set_lean): its one kernel checks the certificates and makes the rows and
   * sums, so its time is the whole pass but the reduction.  A lean pass that failed and the
   * full pass that repeated it are two entries. */
  int icpx_get_pass_timings(icpx_ctx* ctx, double* ms_out, int32_t* kind_out,
    int32_t* hinted_out, int64_t cap, int64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* ICPX_DEV_H */
