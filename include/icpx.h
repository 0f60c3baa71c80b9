/*
 * icpx.h — C ABI of libicpx, the MI355X (gfx950) implementation of the ICP
 * correspondence-and-alignment hot path.
 *
 * The reference has no FFI layer; its hot path is reached through the VTK
 * class surface of vtkICP / vtkPolyDataCorrespondenceFilter.  Each entry point
 * below names the reference code it replaces (paths relative to the reference
 * tree).  Conventions:
 *   - every function returns 0 on success or a negative ICPX_E_* code;
 *     icpx_last_error() gives the message of the calling thread's last failure;
 *   - no C++ exception crosses this boundary;
 *   - matrices are row-major 4x4 doubles that act on column vectors
 *     (vtkMatrix4x4 element order);
 *   - the caller owns every host buffer it passes in; a context owns its
 *     device memory; one context is used by one host thread at a time;
 *   - point, triangle and landmark counts are below 2^31.
 */
#ifndef ICPX_H
#define ICPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

#define ICPX_OK 0
#define ICPX_E_INVALID (-1)  /* bad argument, or a call out of order */
#define ICPX_E_HIP (-2)      /* a HIP runtime call failed */
#define ICPX_E_NOMEM (-3)
#define ICPX_E_FEW (-4)      /* too few correspondences to solve */

#define ICPX_METRIC_POINT_TO_POINT 0 /* vtkICP::MetricPointToPoint, ICP/vtkICP.h:36 */
#define ICPX_METRIC_POINT_TO_PLANE 1 /* vtkICP::MetricPointToPlane, ICP/vtkICP.h:37 */
#define ICPX_METRIC_PLANE_TO_PLANE 2 /* vtkICP::MetricPlaneToPlane (symmetric), ICP/vtkICP.h:38 */

/* Length of the partial-sum vector one search pass produces.  Slot 0 is the
 * number of accepted correspondences, slot 1 the sum of |s - t| over them; the
 * rest are the metric's raw moments (layout private to icpx_solve).  Partial
 * sums of several contexts (GPUs) over disjoint source shards add
 * element-wise. */
#define ICPX_NSUMS 40

  typedef struct icpx_ctx icpx_ctx;

  /* Message of the calling thread's most recent failure ("" if none). */
  const char* icpx_last_error(void);

  /* Number of HIP devices visible to the process (0 if none / no driver). */
  int icpx_device_count(void);

  /* Create / destroy a context bound to one GPU.  A process that shards the
   * source over several GPUs makes one context per device (or one process per
   * GPU with one context each).  Replaces the construction of
   * vtkPolyDataCorrespondenceFilter + its two vtkImplicitPolyDataDistance2
   * locators (ICP/vtkPolyDataCorrespondenceFilter.cxx:349-366). */
  int icpx_create(icpx_ctx** out, int device);
  void icpx_destroy(icpx_ctx* ctx);

  /* Run the context's work on the caller's HIP stream (hipStream_t passed as
   * void*; NULL is HIP's null stream).  Until this is called the context uses a
   * non-blocking stream of its own. */
  int icpx_set_stream(icpx_ctx* ctx, void* hip_stream);

  /* Wait for everything queued on the context's stream. */
  int icpx_synchronize(icpx_ctx* ctx);

  /* Target mesh: nv float points, nt triangles (vertex indices), optional
   * per-triangle unit normals (used verbatim instead of computed ones).
   * Builds, on the device, the Morton-ordered wide BVH and the face / edge /
   * vertex pseudo-normal tables.  Replaces
   * vtkImplicitPolyDataDistance2::SetInput (ICP/vtkImplicitPolyDataDistance2.cxx:34-62:
   * triangle filter, BuildLinks, vtkCellLocator::BuildLocator) and the run-time
   * GetCellEdgeNeighbors / GetPointCells walks of ClosestPointAndNormal
   * (same file, :433, :475).  Inputs must already be triangles; cell ids
   * returned later are indices into `tri`. */
  int icpx_set_target(icpx_ctx* ctx, const float* xyz, int64_t nv, const int32_t* tri, int64_t nt,
    const float* cell_normals_or_null);

  /* Per-point normals of a triangle mesh, computed on the device: the normalised sum of the
   * unit normals of the faces at each point (zero for a point without faces).  Replaces the
   * filter's own vtkPolyDataNormals with splitting off and consistency on
   * (ICP/vtkPolyDataCorrespondenceFilter.cxx:356-359, run at :457-464), whose output the
   * symmetric metric takes as source normals; pass the result to icpx_set_source.
   * Consistency: before the normals are made, every edge-connected piece of the mesh takes
   * the winding of its lowest-numbered cell (cells that run along a shared edge in the same
   * direction as their neighbour are reversed), which is what VTK's walk from the lowest
   * unvisited cell yields for orientable pieces; edges shared by more than two cells are not
   * walked across.  A piece that cannot be wound consistently (a Moebius band) keeps a seam
   * of cells that disagree with their neighbours, as it does in VTK; where that seam runs is
   * not defined (icpx_dev.h: icpx_debug_dump reports that such a piece was met).
   * normals_out: 3 * nv floats.  The context only supplies the device and the stream. */
  int icpx_point_normals(icpx_ctx* ctx, const float* xyz, int64_t nv, const int32_t* tri,
    int64_t nt, float* normals_out);

  /* Source points (float, as vtkICP requires: ICP/vtkICP.cxx:145-150) and
   * optional per-point normals (needed by the symmetric metric; the reference
   * takes them from vtkPolyDataNormals, ICP/vtkPolyDataCorrespondenceFilter.cxx:457-471).
   * A coordinate that is not a number, or beyond 1e12 in magnitude, is ICPX_E_INVALID and
   * leaves the context without a source (the reference's locator derives a bucket index from
   * such a point, which is undefined; here the kernels' pruning bounds assume that a moved
   * point's squared distances are finite in single precision).  Transforms handed in are held
   * to 1e6 in the 3x3 and 1e18 in the translation for the same reason. */
  int icpx_set_source(
    icpx_ctx* ctx, const float* xyz, int64_t np, const float* normals_or_null);

  /* Restrict this context to shard `rank` of `world` of the kept source points
   * {0, step, 2*step, ...}.  Default (0, 1).  The kept points are ordered along a
   * space-filling curve through the target's bounds, and chunks of that order (8192
   * points; fewer for a small cloud) are dealt out to the ranks in turn: a rank's
   * points come in runs of neighbours on the surface, which its searches share work
   * over, and every rank has its share of every part of the surface.  Every rank
   * works the order out for itself from the same source and target, with the same
   * result; shards differ in size by a chunk at most.  Sums of the shards add up to
   * the sums of the whole (in another order of additions: to rounding); rows of a
   * shard (icpx_correspond) come out by ascending source index and the shards' rows
   * together are the rows of the whole.  The reference has one process; the points
   * are independent of one another there too
   * (ICP/vtkPolyDataCorrespondenceFilter.cxx:110-136). */
  int icpx_set_shard(icpx_ctx* ctx, int rank, int world);

  /* Point about which the raw moments are taken (centre of the target's
   * bounds); identical on every context that holds the same target. */
  int icpx_get_origin(icpx_ctx* ctx, double origin[3]);

  /* One pass of the correspondence filter fused with the solver's
   * accumulation: for every kept source index i (i % step == 0, in this
   * context's shard) transform the point by T in fp64, find the closest point,
   * cell and pseudo-normal on the target, apply the |d| < max_dist test (and
   * dot(n_target, n_source) > min_dot when use_source_normals != 0), and add
   * the accepted row, rounded to float as the reference's outputs are, into
   * the metric's fp64 sums.  No rows are written.
   *   use_source_normals: 0 = no test; 1 = the source normals as given (the reference
   *   does not rotate them with the running transform, ICP/vtkPolyDataCorrespondenceFilter.cxx:284,299);
   *   2 = the corrected mode: every source normal is rotated by T's 3x3 (fp64, rounded to
   *   float) before the test, the sums and the emitted origin normals, so that it stays
   *   the normal of the moved point.
   *   sums_host   != NULL: the ICPX_NSUMS sums are copied there (synchronises);
   *   sums_device != NULL: they are left in that device buffer, stream-ordered,
   *                        ready for an all-reduce across GPUs.
   * Replaces Functor/NormalsFunctor::operator() + Reduce
   * (ICP/vtkPolyDataCorrespondenceFilter.cxx:87-168, 252-341), the three Eigen
   * copies of vtkICP::InternalUpdate (ICP/vtkICP.cxx:193-216) and the O(n)
   * loops of the solvers (ICP/icp/point2point.cxx:33-44,
   * point2plane.cxx:43-84, plane2plane.cxx:41-67). */
  int icpx_accumulate(icpx_ctx* ctx, int metric, const double T[16], int64_t step,
    double max_dist, double min_dot, int use_source_normals, double* sums_host,
    double* sums_device);

  /* icpx_accumulate for callers that cannot allocate device memory: code under HIP graph
   * capture, tight serving loops that re-register a few hundred landmarks per frame,
   * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The call
   * allocates nothing: the kept points are ordered, searched and summed in a static device
   * array inside the library (about 2.1 MiB, one copy per GPU, held until the process
   * ends).  The meshes are the context's (icpx_set_target / icpx_set_source, which do
   * allocate, come first); the ICPX_NSUMS sums are left in the caller's device buffer
   * sums_device, stream-ordered; arguments otherwise as icpx_accumulate.  Same sums, bit
   * for bit, as icpx_accumulate gives for the same arguments with the default kernels (the
   * development switches of icpx_dev.h that pick other search kernels do not reach here).
   * Limits: at most ICPX_STATIC_MAX_QUERIES kept points in this context's shard (in the whole
   * job when the context is one of several shards: all of them are ordered to cut the shard
   * out; more is ICPX_E_INVALID: use icpx_accumulate or a larger step), and the search tree of that size
   * class must exist: one earlier icpx_accumulate call on this target with a step of the
   * same size class (up to 4096 kept points, or more) builds it, otherwise ICPX_E_INVALID.
   * Every call on one GPU, from any context, uses that GPU's one array: calls that overlap
   * in time on one GPU race and must be serialised by the caller. */
#define ICPX_STATIC_MAX_QUERIES 32768
  int icpx_accumulate_static(icpx_ctx* ctx, int metric, const double T[16], int64_t step,
    double max_dist, double min_dot, int use_source_normals, double* sums_device);

  /* The same search, returning the filter's outputs instead of sums: accepted
   * rows in ascending source-index order, compacted on the device.  Any output
   * pointer may be NULL.  Buffers must hold ceil(np/step) rows (this shard's
   * share of them when sharded).  Replaces
   * vtkPolyDataCorrespondenceFilter::RequestData's two output ports
   * (ICP/vtkPolyDataCorrespondenceFilter.cxx:397-547); cell, dist and
   * src_index are extra (the reference keeps them local, :363,:375). */
  int icpx_correspond(icpx_ctx* ctx, const double T[16], int64_t step, double max_dist,
    double min_dot, int use_source_normals, float* origin, float* origin_normal, float* closest,
    float* normal, int32_t* cell, double* dist, int64_t* src_index, int64_t* n_out);

  /* The same rows in double precision: what the filter produces with OutputPrecision =
   * DOUBLE_PRECISION, where the reference runs its functor with double output arrays
   * (ICP/vtkPolyDataCorrespondenceFilter.cxx:481-483, 519-531): the transformed point,
   * the closest point and the locator's normal as the doubles they were computed in.
   * That path has no origin normals and no normal-compatibility test
   * (Functor, not NormalsFunctor, :525).  The normal is the fp64 face normal, or the
   * edge / angle-weighted vertex sum made from the fp64 face normals
   * (ICP/vtkImplicitPolyDataDistance2.cxx:397-517). */
  int icpx_correspond_f64(icpx_ctx* ctx, const double T[16], int64_t step, double max_dist,
    double* origin, double* closest, double* normal, int32_t* cell, double* dist,
    int64_t* src_index, int64_t* n_out);

  /* First intersection of each of n line segments p1 -> p2 (3n doubles each) with the target
   * mesh, and the pseudo-normal there.  Replaces
   * vtkImplicitPolyDataIntersection::IntersectWithLine
   * (ICP/vtkImplicitPolyDataIntersection.cxx:110-337) over the target of icpx_set_target:
   * vtkCellLocator::IntersectWithLine with tolerance 1e-12 (:148; the smallest parameter t
   * wins, equal t goes to the smallest cell id), EvaluatePosition at the hit (:169) and the
   * face / edge-sum / angle-weighted vertex normal of the place it falls on (:172-304).
   * Outputs per segment, any of which may be NULL, with the values the reference leaves in
   * its arguments:
   *   hit     1 / 0 (the function's return value);
   *   ret     |p2 - p1| * t with the sign line :324 gives it: g has just been set to the
   *           pseudo-normal there, so the value is always negated; NoValue (the length of the
   *           target's bounds, :64) without a hit;
   *   closest EvaluatePosition's closest point at the hit (0,0,0 without one);
   *   normal  g: the pseudo-normal ((0,0,1) without a hit);
   *   point   the intersection point itself (a local in the reference; 0 without a hit);
   *   cell    index of the triangle hit in the array given to icpx_set_target (-1);
   *   kind    0 face, 1 edge, 2 vertex: vtkDataObject::CELL / EDGE / VERTEX of the six-
   *           argument overload (-1).
   * ICPX_E_INVALID with "No polygons to intersect!" when no target is set. */
  int icpx_intersect_lines(icpx_ctx* ctx, const double* p1, const double* p2, int64_t n,
    int32_t* hit, double* ret, double* closest, double* normal, double* point, int32_t* cell,
    int32_t* kind);

  /* Host-side solve of one iteration from (all-reduced) sums: 3x3 SVD for
   * point-to-point (ICP/icp/point2point.cxx:45-58), 6x6 LDL^T + Euler XYZ for
   * point-to-plane (ICP/icp/point2plane.cxx:88-106), 6x6 LDL^T + Rodrigues
   * half-rotation for the symmetric metric (ICP/icp/plane2plane.cxx:68-92).
   * M is the incremental transform; n_out / mean_dist_out (optional) return
   * the accepted count and the mean |s - t|.  ICPX_E_FEW (M = identity) when
   * fewer than 3 (point-to-point) or 6 rows were accepted. */
  int icpx_solve(int metric, const double sums[ICPX_NSUMS], const double origin[3], double M[16],
    int64_t* n_out, double* mean_dist_out);

  /* Parameters of a whole registration; defaults as vtkICP's constructor
   * (ICP/vtkICP.cxx:27-40) with MaximumNumberOfLandmarks = 200 as documented
   * in ICP/vtkICP.h:52-59 (the reference leaves it uninitialised). */
  typedef struct icpx_params
  {
    int metric;
    int max_iterations;
    int64_t max_landmarks;
    double max_distance;
    int check_mean_distance;
    double max_mean_distance;
    double min_normal_dot;
    /* 0 (default): source normals are used as given in every iteration, as the reference
     * does (ICP/vtkPolyDataCorrespondenceFilter.cxx:284,299: points are transformed,
     * normals are not).  1: opt-in corrected mode, the running rotation is applied to them
     * (symmetric metric only; no counterpart in the reference). */
    int rotate_source_normals;
  } icpx_params;
  void icpx_default_params(icpx_params* p);

  /* Optional hook called once per iteration on the host copy of the sums,
   * between the search and the solve; a multi-process driver sums them over
   * ranks there (in place).  Return non-zero to abort the run. */
  typedef int (*icpx_allreduce_fn)(double* sums, int n, void* user);
  int icpx_set_allreduce(icpx_ctx* ctx, icpx_allreduce_fn fn, void* user);

  /* status values of icpx_run */
#define ICPX_RUN_MAX_ITERATIONS 0 /* ran MaximumNumberOfIterations */
#define ICPX_RUN_CONVERGED 1      /* mean distance <= MaximumMeanDistance */
#define ICPX_RUN_TOO_FEW 2        /* too few correspondences; transform kept */
#define ICPX_RUN_DIVERGED 3       /* a solve gave a transform that is not finite (sums that carry a NaN, \
                                     from normals that are not numbers) or beyond the limits of        \
                                     icpx_set_source (a system singular to within rounding); the last  \
                                     good transform kept */

  /* The whole loop of vtkICP::InternalUpdate (ICP/vtkICP.cxx:189-246):
   * search, optional mean-distance early-out, solve, T <- M*T, repeated.
   * T_out is the accumulated transform (what vtkICP copies into this->Matrix). */
  int icpx_run(icpx_ctx* ctx, const icpx_params* p, double T_out[16], int* iterations_out,
    double* mean_distance_out, int* status_out);

  /* The same loop, one iteration at a time and without a host round trip, for a driver
   * that reduces the sums over GPUs with a stream-ordered collective (RCCL through
   * torch.distributed, or ncclAllReduce on the context's stream).  The transform, the
   * iteration count and the stop condition live on the device: icpx_run_update solves
   * there (one GPU thread running the code of icpx_solve) and, once the run has converged
   * or run out of rows, every pass enqueued afterwards returns at once, so T_out,
   * iterations and status are exactly those of icpx_run.  Nothing blocks except that the
   * host stays at most two iterations ahead of the device.
   *
   *   icpx_run_begin(ctx, p);
   *   for (i = 0; i < p->max_iterations; i++) {
   *     icpx_run_enqueue(ctx, sums_dev, &done);   // search + fused sums -> sums_dev
   *     if (done) break;                          // (known two iterations late; harmless)
   *     all-reduce sums_dev in place on the same stream;
   *     icpx_run_update(ctx, sums_dev);           // solve, T <- M*T, stop test
   *   }
   *   icpx_run_end(ctx, T, &iters, &mean, &status);   // waits for the stream
   *
   * sums_device = NULL uses an internal buffer (single GPU; icpx_run does exactly that).
   * Replaces ICP/vtkICP.cxx:189-246 like icpx_run; the reference has no multi-GPU loop.
   *
   * icpx_run_advance is icpx_run_enqueue for a loop that lets the library keep the
   * position, which is what the ranks of a multi-GPU job should run:
   *
   *   icpx_run_begin(ctx, p);
   *   for (;;) {
   *     icpx_run_advance(ctx, sums_dev, &more);   // the next pass, if there is one
   *     if (!more) break;
   *     all-reduce (sum) all ICPX_NSUMS doubles of sums_dev in place on the same stream;
   *     icpx_run_update(ctx, sums_dev);
   *   }
   *   icpx_run_end(ctx, T, &iters, &mean, &status);
   *
   * The spare slots of the sum vector then carry every rank's pass figures (queries
   * searched, queries the shortcut left to the tree walk, a flag) through the all-reduce,
   * so that all ranks read the same iteration records, decide alike and enqueue the same
   * sequence of passes and collectives.  That allows what icpx_run does on one rank: once
   * the registration has settled, a pass is tried as two launches (the check that no
   * query's winner can have changed inside the kernel that makes the sums, then the
   * reduction); if any rank meets a query it cannot answer that way, the flag makes every
   * rank drop that pass, the position goes back to it and it is repeated with the search
   * kernels.  Same transform, iteration count and status as the counted loop above.  A
   * run is driven by one of icpx_run_enqueue and icpx_run_advance, not both; every rank
   * of a job must use the same one, the same parameters and the same development
   * switches (icpx_dev.h). */
  int icpx_run_begin(icpx_ctx* ctx, const icpx_params* p);
  int icpx_run_enqueue(icpx_ctx* ctx, double* sums_device, int* done_out);
  int icpx_run_advance(icpx_ctx* ctx, double* sums_device, int* more_out);
  int icpx_run_update(icpx_ctx* ctx, const double* sums_device);
  int icpx_run_end(icpx_ctx* ctx, double T_out[16], int* iterations_out,
    double* mean_distance_out, int* status_out);

#ifdef __cplusplus
}
#endif
#endif /* ICPX_H */
