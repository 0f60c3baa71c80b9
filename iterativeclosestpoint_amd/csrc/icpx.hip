// icpx.hip — the C ABI of libicpx (include/icpx.h) over the gfx950 kernels in
// bvh.hpp / tables.hpp / query.hpp / prims.hpp.  Host code here only validates
// shapes, owns device memory and orders launches on one HIP stream; all
// per-point arithmetic runs on the GPU and the only host arithmetic is the O(1)
// solve in solve.cpp.
#include "../../include/icpx.h"
#include "../../include/icpx_dev.h"
#include "bvh.hpp"
#include "prims.hpp"
#include "query.hpp"
#include "search2.hpp"
#include "search3.hpp"
#include "local.hpp"
#include "lean.hpp"
#include "state.hpp"
#include "tables.hpp"
#include "rows64.hpp"
#include "intersect.hpp"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

using namespace icpx;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(expr)                                                                              \
  do                                                                                              \
  {                                                                                               \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(ICPX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
        __LINE__);                                                                                \
  } while (0)

#define LAUNCHCHK() HIPCHK(hipGetLastError())

static const int kWidths[5] = { 4, 8, 16, 32, 64 };

struct Tree
{
  TreeLevels lv;
  float* nodes = nullptr;
  bool built = false;
};
struct Heap
{
  HeapTree t;
  float* nodes = nullptr;
  bool built = false;
};
struct Bin
{
  BinTree t;
  float* nodes = nullptr;
  float4* lpre = nullptr; // per-leaf bounds records (k_leaf_bounds)
  bool built = false;
};

struct Mat16
{
  double m[16];
};

__global__ void k_set_state(RunState* s, Mat16 T, int reset)
{
  const int i = threadIdx.x;
  if (i < 16)
    s->T[i] = T.m[i];
  if (i == 0)
  {
    s->done = 0;
    s->lean_fail = 0;
    s->anchored = 0; // (a transform from outside: the certificates are brought up to date for it)
    if (reset)
    {
      s->mean = 0.0;
      s->n = 0.0;
      s->iters = 0;
      s->status = ICPX_RUN_MAX_ITERATIONS;
      s->fallback = 0;
      s->cert_ok = 0;
    }
  }
}

// What a pass without the fused sums did to the shortcut's certificates (the fused pass
// says it in k_reduce_partials).
__global__ void k_cert_mark(RunState* s, int mode, double cap2)
{
  if (s->done)
    return;
  if (mode && threadIdx.x < 12)
    s->cert_T[threadIdx.x] = s->T[threadIdx.x];
  if (threadIdx.x == 0)
  {
    s->cert_ok = mode ? 1 : 0;
    s->cert_cap2 = cap2;
  }
}

// (one build per metric: with the metric a run-time value all three solvers are part of the
// kernel, 41 KB of code of which a launch runs a quarter, straight through, once)
// share: the sums were all-reduced over the ranks of a job with every rank's policy figures
// in their spare slots (state.hpp: SLOT_*): the record carries the job's totals, and a lean
// pass that failed on ANY rank is dropped here on EVERY rank (nothing is solved, the
// registration is suspended, done = 2, until the hosts, which all read the same in their
// records, repeat the pass in full).
template <int METRIC>
__global__ void k_solve(RunState* s, const double* __restrict__ sums, double ox,
  double oy, double oz, int check_mean, double max_mean, const uint32_t* __restrict__ counters,
  RunRecord* rec, uint32_t seq, float src_rmax, float anchor_reach, int share)
{
  if (threadIdx.x != 0 || blockIdx.x != 0)
    return;
  const double origin[3] = { ox, oy, oz };
  uint32_t fallback = counters ? counters[7] : 0u, listed = counters ? counters[8] : 0u,
           contested = counters ? counters[4] + counters[5] : 0u, nq_total = 0u;
  bool ran = true;
  if (share)
  {
    fallback = slot_count(sums[SLOT_FALLBACK]);
    listed = slot_count(sums[SLOT_LISTED]);
    contested = slot_count(sums[SLOT_CONTESTED]);
    nq_total = slot_count(sums[SLOT_NQ]);
    if (!(sums[SLOT_LEANFAIL] == 0.0)) // (some rank's flag, or not a number: the pass counts for nothing)
    {
      ran = false;
      if (s->done == 0)
        s->done = 2;
    }
    s->lean_fail = 0;
  }
  solve_and_publish(s, sums, METRIC, origin, check_mean, max_mean, ran, fallback, listed, contested,
    rec, seq, src_rmax, anchor_reach, nq_total);
}

// Largest |v[i]| as a float's bits (*out starts at zero; a NaN or infinity ends up on top).
__global__ void k_abs_max(const float* __restrict__ v, int64_t n, uint32_t* __restrict__ out)
{
  uint32_t m = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(v[i]) & 0x7fffffffu);
  for (int k = 32; k >= 1; k >>= 1)
    m = max(m, (uint32_t)__shfl_xor((int)m, k, 64));
  if ((threadIdx.x & 63) == 0 && m)
    atomicMax(out, m);
}

// A lean pass that failed left the registration suspended (done == 2); the host enqueues this
// ahead of the full pass that repeats it.
__global__ void k_resume(RunState* s)
{
  if (threadIdx.x == 0 && s->done == 2)
    s->done = 0;
}

// The reduction and the solve in one launch, for a loop with nothing between them (one
// rank): the block that finishes its sum last takes the ICPX_NSUMS totals and advances the
// registration as k_solve does.  *ticket counts finished blocks and is left at zero.
// lean: the pass was a lean one (k_finalize_mx<.., LEAN> alone); if it raised lean_fail its
// sums are worth nothing: nothing is solved or recorded as brought up to date, the
// registration is suspended (done = 2: passes enqueued behind this one return at once) and
// the record tells the host, which repeats the pass in full.
template <int METRIC>
__global__ __launch_bounds__(64) void k_reduce_solve(const double* __restrict__ partials,
  int nblocks, double* sums, uint32_t* __restrict__ counters, RunState* s, int cert_mode,
  double ox, double oy, double oz, int check_mean, double max_mean, RunRecord* rec,
  uint32_t seq, uint32_t* ticket, double cap2, int lean, float src_rmax, float anchor_reach)
{
  __shared__ double tot[ICPX_NSUMS];
  __shared__ uint32_t last;
#ifdef ICPX_STATS
  const unsigned long long rs_t0 = wall_clock64();
#endif
  // (every block reads the flag before it draws its ticket; the block that draws the last
  // one clears it)
  // (read whether the pass was a lean one or not: a branch around the read would wait for it
  // before the reduction's reads go out)
  const int32_t lean_flag = s->lean_fail;
  const bool failed = lean && lean_flag != 0;
  // (a lean pass leaves the certificates and the transform they refer to as they are)
  const bool ran = reduce_partials_body(partials, nblocks, sums, &s->done, counters, nullptr, s->T,
    s->cert_T, lean ? nullptr : &s->cert_ok, cert_mode, cap2, &s->cert_cap2, 0, 0, nullptr, failed);
#ifdef ICPX_STATS
  const unsigned long long rs_t1 = wall_clock64();
#endif
  if (threadIdx.x == 0)
  {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!last)
    return;
  __threadfence();
  if (threadIdx.x < ICPX_NSUMS)
    tot[threadIdx.x] = *reinterpret_cast<volatile double*>(sums + threadIdx.x);
  __syncthreads();
  if (threadIdx.x != 0)
    return;
#ifdef ICPX_STATS
  g_rs_stamps[0] = rs_t0;
  g_rs_stamps[1] = rs_t1;
  RS_STAMP(2);
#endif
  *ticket = 0u;
  if (failed)
  {
    if (s->done == 0)
      s->done = 2;
    s->lean_fail = 0;
    if (counters)
      counters[3] = 0u; // (the failed pass's count of queries it searched itself)
  }
  const double origin[3] = { ox, oy, oz };
  solve_and_publish(s, tot, METRIC, origin, check_mean, max_mean, ran, counters ? counters[7] : 0u,
    counters ? counters[8] : 0u, counters ? counters[4] + counters[5] : 0u, rec, seq, src_rmax,
    anchor_reach);
}

// The pools the tree kernels draw their work from, back at their start: for a second such
// kernel in one pass (the one that serves the queries a walk gave up on; the pass's reduction
// leaves them at zero for the next pass).
__global__ void k_reset_pools(uint32_t* __restrict__ counters)
{
  if (threadIdx.x < XPOOLS)
    counters[XPOOL_FIRST + threadIdx.x * XPOOL_STRIDE] = 0u;
}

// ---- icpx_accumulate_static: scratch that lives in the code object, not on the heap ----
// Everything that entry point needs beyond the context's meshes and the caller's output: the
// query order (keys, values, sort histograms), the ordered queries and the block partials.
// One copy per GPU the library runs on, held until the process ends; every call on a GPU
// uses that GPU's copy, so calls that overlap on one GPU race.
constexpr int64_t STATIC_MAX_QUERIES = ICPX_STATIC_MAX_QUERIES; // kept source points per call
constexpr int STATIC_MAX_BLOCKS = 1024;       // finalize blocks (partials)
constexpr size_t st_align(size_t b)
{
  return (b + 255) & ~(size_t)255;
}
constexpr size_t ST_KEYS = st_align(sizeof(uint64_t) * STATIC_MAX_QUERIES);
constexpr size_t ST_VALS = st_align(sizeof(uint32_t) * STATIC_MAX_QUERIES);
constexpr size_t ST_SORT_WORDS = 16384; // histograms + scan levels of the radix sort
constexpr size_t ST_SORT = st_align(sizeof(uint32_t) * ST_SORT_WORDS);
constexpr size_t ST_VEC3 = st_align(sizeof(float) * 3 * STATIC_MAX_QUERIES);
constexpr size_t ST_I32 = st_align(sizeof(int32_t) * STATIC_MAX_QUERIES);
constexpr size_t ST_PART = st_align(sizeof(double) * STATIC_MAX_BLOCKS * ICPX_NSUMS);
constexpr size_t ST_OFF_KA = 0, ST_OFF_KB = ST_OFF_KA + ST_KEYS, ST_OFF_VA = ST_OFF_KB + ST_KEYS,
                 ST_OFF_VB = ST_OFF_VA + ST_VALS, ST_OFF_SORT = ST_OFF_VB + ST_VALS,
                 ST_OFF_QPOS = ST_OFF_SORT + ST_SORT, ST_OFF_QSN = ST_OFF_QPOS + ST_VEC3,
                 ST_OFF_QK = ST_OFF_QSN + ST_VEC3, ST_OFF_HINT = ST_OFF_QK + ST_I32,
                 ST_OFF_PART = ST_OFF_HINT + ST_I32, ST_BYTES = ST_OFF_PART + ST_PART;
static __device__ __align__(256) unsigned char scratch[ST_BYTES]; // (about 2.1 MiB)

// The array's address on each device (the name above is only a host-side handle: host code
// passes these pointers, never the name).  Filled by icpx_create with that device current.
constexpr int ST_MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[ST_MAX_DEVICES];
static std::mutex g_scratch_mutex;

static int scratch_for_device(int device)
{
  if (device < 0 || device >= ST_MAX_DEVICES)
    return ICPX_OK; // (beyond the table: the static entry point reports it)
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device])
    return ICPX_OK;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !p)
    return fail(ICPX_E_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device,
      hipGetErrorString(e));
  g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  return ICPX_OK;
}

// From this many queries on, a pass that finds certificates checks them in a kernel of its
// own (k_certify) and runs the shortcut over the rest; below, the extra launch costs more
// than the lighter shortcut pass saves (measured: +6 % at a million queries, -8 % of the
// search time at sixteen million).
constexpr int64_t CERTIFY_MIN = 4000000;
// pass counters: {ties, overflows, pool position, shortcut fallbacks}, the last pass's copy
// of the four, then (from word 16) the segment counts of k_certify's list
constexpr int PASS_COUNTERS = 16 + CERTIFY_SEGS * CERTIFY_SEG_STRIDE;
static_assert(XPOOL_FIRST + (XPOOLS - 1) * XPOOL_STRIDE < PASS_COUNTERS && XPOOL_FIRST > 16 &&
    (XPOOL_FIRST - 16) % CERTIFY_SEG_STRIDE != 0 && XPOOL_STRIDE == CERTIFY_SEG_STRIDE,
  "the tree kernels' pool counters lie among the pass counters, beside the list's segment counts");
static_assert(WALK_DONE < PASS_COUNTERS && (WALK_TIME - 16) % CERTIFY_SEG_STRIDE > 1,
  "the walk's figures lie beside the first two pool counters");

struct icpx_ctx
{
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;

  // target
  int64_t nv = 0, nt = 0;
  float* d_xyz = nullptr;
  int32_t* d_tri = nullptr;
  float* d_tverts = nullptr;
  int32_t* d_tid = nullptr;
  uint64_t* d_tcodes = nullptr; // sorted Morton codes of the triangles
  uint32_t* d_tcell = nullptr;  // first position per leading tcell_bits bits of a code (k_code_cells)
  int tcell_bits = 0;
  int32_t* d_seed = nullptr;    // a near triangle per cell of a coarse grid over the bounds (k_seed_flood)
  int seed_dim = 0;
  int seed_mode = 1;            // cold seeds: 0 the Morton curve's, 1 the grid's and the curve's, 2 the grid's
  float4* d_tpre = nullptr;     // per-triangle plane bounds
  float* d_cert = nullptr;      // per query: the shortcut's certificate (see local.hpp)
  float* d_lead = nullptr;      // per query on a set: its winner's lead over the other members (cert.hpp)
  uint32_t* d_ticket = nullptr; // k_reduce_solve's count of finished blocks (zero between launches)
  float *d_fn = nullptr, *d_en = nullptr, *d_vn = nullptr;
  float4* d_fnpos = nullptr; // {face normal, cell id} by Morton position (the finalize kernels' read)
  // kept for the double-precision rows: fp64 face normals and the links (cells per point)
  double* d_fn64 = nullptr;
  uint32_t* d_voff = nullptr;
  int32_t* d_vcell = nullptr;
  Box3 box;
  double bounds_length = 0.0; // vtkDataSet::GetLength() of the target: the diagonal of its bounds
  double origin[3] = { 0, 0, 0 };
  Tree trees[5];  // exhaustive-exact kernel's level-array trees
  Heap heaps[5];  // group kernel's heap-ordered trees
  Bin bins[3];    // lane-per-query kernel's binary trees (1, 2, 4 triangles per leaf)
  int leaf_size = 4;
  int persistent = 4; // tree kernel: 1 pooled lanes, 3 one walk per wavefront, 4 by regime
  int wave_blocks = 19; // blocks per CU of the one-walk-per-wavefront kernel
  int noisy_policy = 1; // a cloud that stays beyond the lists' reach goes by the large cloud's policy (heavy_cloud)
  // The queries a pass's certificate check lets through on SET certificates have their sets
  // weighed by the fused finalize itself (query.hpp: weigh_marked) when the pass ends in the
  // matrix-core build of the sums, whose wavefronts have a few dozen such queries each at most:
  // a launch less, and k_contested's was as long as its chain of reads (30-45 us a pass on
  // clouds of one or two million points).  By k_contested otherwise: rows, and the register
  // build of large clouds, where a wavefront has hundreds of sets to weigh before its first row
  // and a kernel of its own with more wavefronts in flight does it faster (sixteen million
  // noisy points: registration 92.4 -> 96.6 ms with the finalize weighing).  0 here: always
  // k_contested, for measurements.
  int weigh_in_finalize = 3; // bit 0: as said; bit 1: the winners' leads are kept (cert.hpp), so that a
                             // set whose winner is clear of its other members is not weighed at all
  bool weigh_later = false; // this pass's searches leave the marked queries to the finalize
  bool pass_marked = false; // this pass's searches wrote the set mask
  int query_curve = 1;  // the queries' order: 1 along the Hilbert curve, 0 along the Morton curve
  int num_cus = 256;
  // hinted shortcut: per-leaf neighbour lists for one radius, and the residual statistics
  // of the previous pass that choose the radius
  int use_local = 1;
  int refill_override = 0; // pooled kernels' refill threshold (0 = by regime)
  int defer_edges = 0;     // the wavefront walk sets a query aside that is this many edge lengths out and
                           // 64 times (squared) beyond its batch's geometric mean (0: never, the default:
                           // the test and the launch behind cost a clean million-point cloud's cold
                           // pass 0.07 ms of 0.92)
  int step_budget = 512;   // steps after which the pooled lanes give up on a query (0: never)
  int bail_mode = 1;       // whole passes of the wavefront walk give up on long batches: 0 never,
                           // 1 on a target whose triangles' sizes range widely, after a time that goes by
                           // the cloud's size, else always, after that many microseconds
  int finalize_build = 0;  // fused finalize: 0 by size, 1 register sums, 2 matrix core
  int nbr_state = 0;    // 0 not tried, 1 built, -1 not worth it for this mesh
  float nbr_eps = 0.f;  // radius the lists cover
  float mesh_edge = 0.f; // typical edge length of the target (sampled)
  float mesh_grading = 1.f; // median longest edge over the shortest edges' first thousandth (sampled)
  int64_t bail_passes = 0;  // whole passes of the wavefront walk run by the build that can give up on batches
  uint32_t* d_noff = nullptr;
  int32_t* d_nbr = nullptr;
  uint4* d_qbox = nullptr; // quantised leaf boxes of the one-triangle-per-leaf tree
  float4* d_edges = nullptr; // edge half-spaces of the triangles
  QuantGrid qgrid;
  uint32_t* d_fallback = nullptr;
  uint32_t* d_bail = nullptr; // queries of the batches a walk gave up on (count in counters[2])
  double* h_stat = nullptr; // pinned {n, sum |s-t|, shortcut fallbacks} of a finished pass
  hipEvent_t ev_stat = nullptr;
  bool stat_pending = false, stat_valid = false;
  double last_mean = 0.0;
  int local_holdoff = 0; // passes to go without the shortcut (it served too few queries)
  int bad_streak = 0;    // consecutive passes it served too few queries while the residual was large
  int hopeless_streak = 0; // consecutive listed passes in which it settled less than half of its list
  bool direct = false;     // k_certify's list goes straight to the tree kernels (see note_pass)
  int direct_passes = 0;
  int use_cert = 1;      // the shortcut keeps certificates (icpx_set_local(2) turns them off)
  int holdoff_len = 8;   // length of the next suspension: doubles each time one is called for
  bool after_cold = false; // the last pass was a cold one
  int since_cold = 0;      // passes since the last cold one
  uint32_t* d_work = nullptr;  // queries whose certificates failed (k_certify's list)
  unsigned long long* d_setmask = nullptr; // one bit per query: to be weighed again by k_contested
  bool certs_live = false;     // the last pass kept certificates for this one to check
  int64_t certify_min = CERTIFY_MIN; // clouds from this size have their certificates checked by k_certify
  bool pass_certs = false; // the pass being enqueued keeps every query's certificate up to date
  double cap2 = DBL_MAX;   // this pass's distance threshold, squared (what the searches stop at)
  bool probation = false; // back from a suspension: one more bad pass suspends it again
  bool last_local = false;   // the pass enqueued last tried the shortcut
  bool stat_local = false;   // ... and so did the pass whose figures are pending in h_stat
  bool pass_local[4] = { false, false, false, false }; // ... and the run's passes, by record slot
  bool last_listed = false;  // the pass just enqueued ran the shortcut over k_certify's list
  bool stat_listed = false;
  bool pass_listed[4] = { false, false, false, false };
  // device-resident loop
  RunState* d_state = nullptr;
  RunRecord* h_rec = nullptr; // pinned ring of the last four iterations' records
  RunRecord* d_rec = nullptr; // the same memory as the device sees it
  double* d_stat = nullptr;   // h_stat as the device sees it
  bool counters_clean = false; // the pass counters are zero (the reduction left them so)
  uint32_t rec_seq = 0;                      // serial number of the last solve enqueued
  uint32_t slot_seq[4] = { 0, 0, 0, 0 };     // ... and of the solve that writes each record slot
  bool run_active = false, run_done_known = false;
  int64_t run_noted = 0; // iteration records of this run the host has looked at
  bool noisy = false;    // the last record's mean residual is beyond the neighbour lists' reach
  double record_wait_limit = 120.0; // seconds the host waits for one iteration's record
  int64_t run_it = 0, run_step = 1;
  icpx_params run_p;
  int64_t local_passes = 0, tree_passes = 0, nbr_builds = 0, nbr_total = 0;
  int64_t certify_passes = 0; // passes whose certificates were checked by k_certify
  int64_t split_passes = 0;   // passes that enqueued both tree kernels over the leftovers
  int64_t split_wave = 0, split_pool = 0; // ... of a run, by the kernel the list's length picked
  bool last_split = false;                // the pass enqueued last was such a pass
  bool pass_split[4] = { false, false, false, false }; // ... and the run's passes, by record slot
  // Lean passes (one rank's fused loop, clouds below certify_min): once a registration has
  // settled, every query stands on its certificate and a pass is the fused finalize with the
  // certificate check in it (k_finalize_mx<.., LEAN>) plus the reduction and solve: two
  // launches instead of five.  The host cannot know that a pass will be like that (records
  // arrive two passes late), so it is tried when the last record seen was of a pass that left
  // nothing to the tree walk or to k_contested's lists; a lean pass that meets a query
  // whose certificate fails suspends the registration on the device, the host sees that in
  // the pass's record, takes the loop back to that pass and repeats it in full.
  int use_lean = 1;
  bool run_fused = false;     // this run is one rank's fused loop
  bool run_shared = false;    // this run is a rank's share of a multi-rank loop (icpx_run_advance)
  bool last_lean = false;     // the pass enqueued last was a lean one (its solve is still to come)
  bool lean_pairs_pass = false; // the pass being enqueued prepares certificates for lean passes
  bool lean_base = false;     // the pass enqueued last left every certificate up to date the shortcut's way
  bool clean_noted = false;   // the last record looked at was of a pass that left nothing over
  int lean_unnoted = 0;       // lean passes enqueued whose records have not been looked at
  int lean_holdoff = 0;       // full passes to go before the next lean one is tried
  int lean_backoff = 4;       // ... after the next failure (doubles while failures follow one another)
  bool pass_lean[4] = { false, false, false, false };
  int64_t lean_passes = 0, lean_failures = 0;
  // lane-per-query kernel's tie bookkeeping, sized for q3_cap queries
  int64_t q3_cap = 0;
  int32_t* d_cand = nullptr;
  uint2* d_tie_list = nullptr;
  uint32_t* d_counters = nullptr;
  uint32_t* d_ovf_list = nullptr;
  double* d_ovf_best = nullptr;
  float scene_max = 0.f;
  float src_max = 0.f;  // largest |coordinate| of the source points
  int use_anchor = 1;   // anchored passes (cert.hpp)
  int exhaustive = 0;
  double build_seconds = 0.0;
  int64_t last_flips = 0; // cells the last icpx_point_normals reversed for consistent winding
  bool last_unorientable = false; // ... and whether it met a piece that cannot be wound consistently

  // source
  int64_t np = 0;
  float* d_src = nullptr;
  float* d_srcn = nullptr;
  uint64_t src_version = 0;
  int rank = 0, world = 1;
  // How the kept points are cut into shards.  They are ordered along the space-filling curve
  // through the target's bounds; 2 (default): chunks of that order (8192 points, fewer for a
  // small cloud) are dealt out to the ranks in turn, so that a rank's wavefronts hold
  // neighbours on the surface, as one rank's do, and every rank has its share of every part
  // of the surface, the parts that are far from their place at the start included; 1: `world`
  // equal ranges of the order (one stretch of the surface per rank); 0: blocks of consecutive
  // kept indices, whatever the points' places.
  int shard_order = 2;

  // queries prepared for (step, shard, source version)
  int64_t prep_step = -1;
  int prep_rank = -1, prep_world = -1, prep_order = -1;
  uint64_t prep_version = 0;
  int64_t nq = 0, k0 = 0, q_cap = 0;
  int64_t nk = 0;               // kept points of the whole job (all shards)
  bool curve_shard = false;     // the prepared shard is a range of the curve order
  int32_t* d_kglobal = nullptr; // ... then: slot (rank among the shard's kept indices) -> kept index
  float* d_qpos = nullptr;
  float* d_qsn = nullptr;
  int32_t* d_qk = nullptr;
  int32_t* d_hint = nullptr;
  bool hint_valid = false;

  // per-pass scratch
  double* d_partials = nullptr;
  double* d_sums = nullptr;
  double* h_sums = nullptr; // pinned
  int max_blocks = 1024;

  // row outputs (icpx_correspond)
  int64_t rows_cap = 0;
  float *r_origin = nullptr, *r_onormal = nullptr, *r_closest = nullptr, *r_normal = nullptr;
  int32_t* r_cell = nullptr;
  double* r_dist = nullptr;
  uint32_t *r_flag = nullptr, *r_pos = nullptr, *r_scan = nullptr;
  // the same in double precision (icpx_correspond_f64), allocated on first use
  double *r64_origin = nullptr, *r64_closest = nullptr, *r64_normal = nullptr, *r64_dist = nullptr;
  double *c64_origin = nullptr, *c64_closest = nullptr, *c64_normal = nullptr, *c64_dist = nullptr;
  int64_t rows64_cap = 0;
  float *c_origin = nullptr, *c_onormal = nullptr, *c_closest = nullptr, *c_normal = nullptr;
  int32_t* c_cell = nullptr;
  double* c_dist = nullptr;
  int64_t* c_index = nullptr;

  int group_width = 0;
  icpx_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;

  // optional HIP-event timing of the two hot kernels
  int timing = 0; // 0 off, 1 both hot kernels, 2 the search kernels only
  struct Pending
  {
    hipEvent_t a, b;
    int which; // 0 search, 1 finalize
    int kind;  // search: 0 a whole pass of the tree kernel, 1 the hinted shortcut first
    int hinted; // search: the pass started from last pass's winners
  };
  struct PassTime
  {
    double ms;
    int kind, hinted;
  };
  std::vector<PassTime> pass_times; // the search kernels' time pass by pass (icpx_get_pass_timings)
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  // the search kernels' pair rides on the first and the last kernel of the pass themselves
  // (their own start and end times) instead of two markers in the queue, each of which
  // holds the next kernel back by ~6 us
  hipEvent_t ev_first = nullptr, ev_last = nullptr;
  double t_ms[2] = { 0.0, 0.0 };
  int64_t t_n[2] = { 0, 0 };
};

static hipEvent_t take_event(icpx_ctx* c)
{
  hipEvent_t e = nullptr;
  if (!c->free_events.empty())
  {
    e = c->free_events.back();
    c->free_events.pop_back();
  }
  // (a device-scope release: these events only time kernels, nobody on the host waits for
  // memory behind them, and the default system-scope release holds the next kernel back)
  else if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess)
    e = nullptr;
  return e;
}

// Fold finished event pairs into the totals (the stream must be idle).
static void resolve_timing(icpx_ctx* c)
{
  for (auto& p : c->pending)
  {
    float ms = 0.f;
    if (p.a && p.b && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
    {
      c->t_ms[p.which] += ms;
      c->t_n[p.which]++;
      if (p.which == 0 && c->pass_times.size() < (1u << 20))
        c->pass_times.push_back({ (double)ms, p.kind, p.hinted });
    }
    if (p.a)
      c->free_events.push_back(p.a);
    if (p.b)
      c->free_events.push_back(p.b);
  }
  c->pending.clear();
}

struct ScopedTimer
{
  icpx_ctx* c;
  hipEvent_t a = nullptr, b = nullptr;
  int which;
  int kind = 0, hinted = 0;
  bool attach; // the events ride on the pass's first and last kernel (launch_ev)
  ScopedTimer(icpx_ctx* ctx, int w, bool att = false)
    : c(ctx)
    , which(w)
    , attach(att)
  {
    if (!c->timing || (c->timing == 2 && which != 0))
      return;
    if (c->pending.size() >= 16384)
    {
      (void)hipStreamSynchronize(c->stream);
      resolve_timing(c);
    }
    a = take_event(c);
    b = take_event(c);
    if (attach && a && b)
    {
      c->ev_first = a; // (taken by the launches: launch_ev)
      c->ev_last = b;
    }
    else if (a)
      (void)hipEventRecord(a, c->stream);
  }
  ~ScopedTimer()
  {
    if (!a && !b)
      return;
    if (attach && a && b)
    {
      // (a pass that ended before its launches leaves a pair that was never recorded;
      // resolve_timing skips it)
      c->ev_first = nullptr;
      c->ev_last = nullptr;
    }
    else if (b)
      (void)hipEventRecord(b, c->stream);
    c->pending.push_back({ a, b, which, kind, hinted });
  }
};

// A launch that may carry the pass's timing events: the first kernel of the search takes
// the start event, the last one (last == true) the stop event.
template <typename... KArgs, typename... Args>
static void launch_ev(icpx_ctx* c, bool last, void (*kernel)(KArgs...), unsigned grid,
  unsigned block, hipStream_t st, Args... args)
{
  hipEvent_t e0 = c->ev_first, e1 = last ? c->ev_last : nullptr;
  c->ev_first = nullptr;
  if (e0 || e1)
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, e0, e1, 0, args...);
  else
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, st, args...);
}

template <typename T>
static int dev_alloc(T** p, int64_t count)
{
  if (*p)
  {
    (void)hipFree(*p);
    *p = nullptr;
  }
  if (count <= 0)
    count = 1;
  hipError_t e = hipMalloc((void**)p, sizeof(T) * (size_t)count);
  if (e != hipSuccess)
  {
    *p = nullptr;
    return fail(ICPX_E_NOMEM, "hipMalloc of %lld bytes failed: %s",
      (long long)(sizeof(T) * (size_t)count), hipGetErrorString(e));
  }
  return ICPX_OK;
}
template <typename T>
static void dev_free(T** p)
{
  if (*p)
    (void)hipFree(*p);
  *p = nullptr;
}
#define ALLOC(ptr, count)                                                                         \
  do                                                                                              \
  {                                                                                               \
    int rc_ = dev_alloc(&(ptr), (count));                                                         \
    if (rc_)                                                                                      \
      return rc_;                                                                                 \
  } while (0)

static inline unsigned nblk(int64_t n, int threads)
{
  return (unsigned)((n + threads - 1) / threads);
}

static void free_target(icpx_ctx* c)
{
  dev_free(&c->d_xyz);
  dev_free(&c->d_tri);
  dev_free(&c->d_tverts);
  dev_free(&c->d_tid);
  dev_free(&c->d_tcodes);
  dev_free(&c->d_tcell);
  dev_free(&c->d_seed);
  dev_free(&c->d_tpre);
  dev_free(&c->d_fn);
  dev_free(&c->d_fnpos);
  dev_free(&c->d_fn64);
  dev_free(&c->d_voff);
  dev_free(&c->d_vcell);
  dev_free(&c->d_en);
  dev_free(&c->d_vn);
  for (auto& t : c->trees)
  {
    dev_free(&t.nodes);
    t.built = false;
  }
  for (auto& t : c->heaps)
  {
    dev_free(&t.nodes);
    t.built = false;
  }
  for (auto& t : c->bins)
  {
    dev_free(&t.nodes);
    dev_free(&t.lpre);
    t.built = false;
  }
  dev_free(&c->d_noff);
  dev_free(&c->d_nbr);
  dev_free(&c->d_qbox);
  dev_free(&c->d_edges);
  c->nbr_eps = 0.f;
  c->nbr_state = 0;
  c->stat_valid = false;
  c->nv = c->nt = 0;
}
static void free_queries(icpx_ctx* c)
{
  dev_free(&c->d_qpos);
  dev_free(&c->d_qsn);
  dev_free(&c->d_qk);
  dev_free(&c->d_hint);
  dev_free(&c->d_kglobal);
  dev_free(&c->d_cand);
  dev_free(&c->d_tie_list);
  dev_free(&c->d_counters);
  dev_free(&c->d_ovf_list);
  dev_free(&c->d_ovf_best);
  dev_free(&c->d_fallback);
  dev_free(&c->d_bail);
  dev_free(&c->d_cert);
  dev_free(&c->d_lead);
  dev_free(&c->d_setmask);
  dev_free(&c->d_work);
  c->q_cap = 0;
  c->q3_cap = 0;
  c->prep_step = -1;
}
static void free_rows(icpx_ctx* c)
{
  dev_free(&c->r_origin);
  dev_free(&c->r_onormal);
  dev_free(&c->r_closest);
  dev_free(&c->r_normal);
  dev_free(&c->r_cell);
  dev_free(&c->r_dist);
  dev_free(&c->r_flag);
  dev_free(&c->r_pos);
  dev_free(&c->r_scan);
  dev_free(&c->c_origin);
  dev_free(&c->c_onormal);
  dev_free(&c->c_closest);
  dev_free(&c->c_normal);
  dev_free(&c->c_cell);
  dev_free(&c->c_dist);
  dev_free(&c->c_index);
  c->rows_cap = 0;
  dev_free(&c->r64_origin);
  dev_free(&c->r64_closest);
  dev_free(&c->r64_normal);
  dev_free(&c->r64_dist);
  dev_free(&c->c64_origin);
  dev_free(&c->c64_closest);
  dev_free(&c->c64_normal);
  dev_free(&c->c64_dist);
  c->rows64_cap = 0;
}

extern "C" const char* icpx_last_error(void)
{
  return g_err;
}

extern "C" int icpx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

extern "C" int icpx_create(icpx_ctx** out, int device)
{
  if (!out)
    return fail(ICPX_E_INVALID, "icpx_create: out is null");
  *out = nullptr;
  int n = icpx_device_count();
  if (device < 0 || device >= n)
    return fail(ICPX_E_INVALID, "icpx_create: device %d not available (%d visible)", device, n);
  HIPCHK(hipSetDevice(device));
  icpx_ctx* c = new (std::nothrow) icpx_ctx();
  if (!c)
    return fail(ICPX_E_NOMEM, "icpx_create: out of host memory");
  c->device = device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      c->num_cus = prop.multiProcessorCount;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess)
  {
    delete c;
    return fail(ICPX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  int rc = scratch_for_device(device);
  if (!rc)
    rc = dev_alloc(&c->d_partials, (int64_t)c->max_blocks * ICPX_NSUMS);
  if (!rc)
    rc = dev_alloc(&c->d_sums, ICPX_NSUMS);
  if (!rc && hipHostMalloc((void**)&c->h_sums, sizeof(double) * ICPX_NSUMS) != hipSuccess)
    rc = fail(ICPX_E_NOMEM, "hipHostMalloc failed");
  if (!rc &&
    (hipHostMalloc((void**)&c->h_stat, sizeof(double) * 4,
       hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_stat, c->h_stat, 0) != hipSuccess))
    rc = fail(ICPX_E_NOMEM, "hipHostMalloc failed");
  if (!rc && hipEventCreateWithFlags(&c->ev_stat, hipEventDisableTiming) != hipSuccess)
    rc = fail(ICPX_E_HIP, "hipEventCreate failed");
  if (!rc)
    rc = dev_alloc(&c->d_state, 1);
  if (!rc && hipMemset(c->d_state, 0, sizeof(RunState)) != hipSuccess)
    rc = fail(ICPX_E_HIP, "hipMemset failed");
  if (!rc)
    rc = dev_alloc(&c->d_ticket, 1);
  if (!rc && hipMemset(c->d_ticket, 0, sizeof(uint32_t)) != hipSuccess)
    rc = fail(ICPX_E_HIP, "hipMemset failed");

  if (!rc &&
    (hipHostMalloc((void**)&c->h_rec, sizeof(RunRecord) * 4,
       hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_rec, c->h_rec, 0) != hipSuccess))
    rc = fail(ICPX_E_NOMEM, "hipHostMalloc failed");
  if (!rc)
    std::memset(c->h_rec, 0, sizeof(RunRecord) * 4);
  if (rc)
  {
    icpx_destroy(c);
    return rc;
  }
  *out = c;
  return ICPX_OK;
}

extern "C" void icpx_destroy(icpx_ctx* c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  free_target(c);
  free_queries(c);
  free_rows(c);
  dev_free(&c->d_src);
  dev_free(&c->d_srcn);
  dev_free(&c->d_partials);
  dev_free(&c->d_sums);
  if (c->h_sums)
    (void)hipHostFree(c->h_sums);
  if (c->h_stat)
    (void)hipHostFree(c->h_stat);
  if (c->ev_stat)
    (void)hipEventDestroy(c->ev_stat);
  dev_free(&c->d_state);
  dev_free(&c->d_ticket);
  if (c->h_rec)
    (void)hipHostFree(c->h_rec);
  resolve_timing(c);
  for (auto e : c->free_events)
    (void)hipEventDestroy(e);
  if (c->own_stream)
    (void)hipStreamDestroy(c->own_stream);
  delete c;
}

extern "C" int icpx_set_stream(icpx_ctx* c, void* s)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = (hipStream_t)s; // NULL is HIP's null stream
  return ICPX_OK;
}

extern "C" int icpx_synchronize(icpx_ctx* c)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ICPX_OK;
}

extern "C" int icpx_set_group_width(icpx_ctx* c, int lanes)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (lanes != 0 && lanes != 1 && lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 &&
    lanes != 64)
    return fail(ICPX_E_INVALID, "group width must be 0, 1, 4, 8, 16, 32 or 64 (got %d)", lanes);
  c->group_width = lanes;
  return ICPX_OK;
}

extern "C" int icpx_set_timing(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  resolve_timing(c);
  c->timing = enable < 0 ? 0 : enable > 2 ? 2 : enable;
  c->t_ms[0] = c->t_ms[1] = 0.0;
  c->t_n[0] = c->t_n[1] = 0;
  c->pass_times.clear();
  return ICPX_OK;
}

extern "C" int icpx_get_pass_timings(icpx_ctx* c, double* ms_out, int32_t* kind_out,
  int32_t* hinted_out, int64_t cap, int64_t* n_out)
{
  if (!c || !n_out)
    return fail(ICPX_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  resolve_timing(c);
  const int64_t n = (int64_t)c->pass_times.size();
  *n_out = n;
  for (int64_t i = 0; i < std::min(n, cap); i++)
  {
    if (ms_out)
      ms_out[i] = c->pass_times[(size_t)i].ms;
    if (kind_out)
      kind_out[i] = c->pass_times[(size_t)i].kind;
    if (hinted_out)
      hinted_out[i] = c->pass_times[(size_t)i].hinted;
  }
  return ICPX_OK;
}

extern "C" int icpx_get_timing(icpx_ctx* c, double* search_ms, int64_t* search_launches,
  double* finalize_ms, int64_t* finalize_launches)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  resolve_timing(c);
  if (search_ms)
    *search_ms = c->t_ms[0];
  if (search_launches)
    *search_launches = c->t_n[0];
  if (finalize_ms)
    *finalize_ms = c->t_ms[1];
  if (finalize_launches)
    *finalize_launches = c->t_n[1];
  return ICPX_OK;
}

// Diagnostic: copy internal arrays to the host. what: 0 = winners of the last search pass
// (Morton positions, int32 x nq), 1 = Morton position -> cell id (int32 x nt, -1-id for
// degenerate), 2 = {ties, overflows} of the last lane-per-query pass (uint32 x 2).
extern "C" int icpx_debug_dump(icpx_ctx* c, int what, void* out, int64_t count)
{
  if (!c || !out)
    return fail(ICPX_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const void* src = nullptr;
  size_t bytes = 0;
  if (what == 0)
  {
    src = c->d_hint;
    bytes = sizeof(int32_t) * (size_t)std::min<int64_t>(count, c->nq);
  }
  else if (what == 1)
  {
    src = c->d_tid;
    bytes = sizeof(int32_t) * (size_t)std::min<int64_t>(count, c->nt);
  }
  else if (what == 2 && c->d_counters)
  {
    src = c->d_counters + (c->counters_clean ? 4 : 0); // the reduction moves them up
    bytes = sizeof(uint32_t) * (size_t)std::min<int64_t>(count, 4);
  }
  else if (what == 4)
  {
    int64_t v[16] = { c->local_passes, c->tree_passes, c->nbr_builds, (int64_t)(c->nbr_eps * 1e9f),
      c->nbr_total, c->certify_passes, c->split_passes, c->split_wave, c->split_pool,
      c->last_flips, c->direct_passes, c->lean_passes, c->lean_failures,
      c->last_unorientable ? 1 : 0, (int64_t)(c->mesh_grading * 1e3f), c->bail_passes };
    std::memcpy(out, v, sizeof(int64_t) * (size_t)std::min<int64_t>(count, 16));
    return ICPX_OK;
  }
  else if (what == 3 && c->d_ovf_best)
  {
    src = c->d_ovf_best;
    bytes = sizeof(double) * (size_t)std::min<int64_t>(count, c->nq);
  }
  else if (what == 5 && c->d_cert)
  {
    src = c->d_cert;
    bytes = sizeof(float) * (size_t)std::min<int64_t>(count, c->nq);
  }
  else if (what == 6 && c->d_qk)
  {
    src = c->d_qk;
    bytes = sizeof(int32_t) * (size_t)std::min<int64_t>(count, c->nq);
  }
  else if (what == 7 && c->d_cand)
  {
    src = c->d_cand; // RING3 entries per query (count = number of entries)
    bytes = sizeof(int32_t) * (size_t)std::min<int64_t>(count, c->nq * RING3);
  }
  else if (what == 8 && c->d_fallback)
  {
    src = c->d_fallback; // the last pass's list of queries left to the tree kernel
    bytes = sizeof(uint32_t) * (size_t)std::min<int64_t>(count, c->nq);
  }
  else if (what == 9 && c->d_lead)
  {
    src = c->d_lead; // (meaningful where the certificate's low bits are both set)
    bytes = sizeof(float) * (size_t)std::min<int64_t>(count, c->nq);
  }
  if (!src)
    return fail(ICPX_E_INVALID, "nothing to dump");
  HIPCHK(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return ICPX_OK;
}

#ifdef ICPX_STATS
// development build only: read and clear the traversal statistics
extern "C" int icpx_stats_read(icpx_ctx* c, unsigned long long out[8])
{
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(icpx::g_stats), sizeof(unsigned long long) * 8));
  unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(icpx::g_stats), z, sizeof(z)));
  return ICPX_OK;
}
extern "C" int icpx_stats_read_batches(icpx_ctx* c, unsigned long long out[37])
{
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  {
    unsigned long long rs[8];
    HIPCHK(hipMemcpyFromSymbol(rs, HIP_SYMBOL(icpx::g_rs_stamps), sizeof(rs)));
    fprintf(stderr, "k_reduce_solve stamps (10 ns ticks from entry): reduce %llu ticket+totals %llu state reads %llu solve %llu compose %llu publish %llu\n",
      rs[1] - rs[0], rs[2] - rs[1], rs[4] - rs[2], rs[5] - rs[4], rs[6] - rs[5], rs[7] - rs[6]);
  }
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(icpx::g_stats_batch), sizeof(unsigned long long) * 37));
  unsigned long long z[37] = { 0 };
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(icpx::g_stats_batch), z, sizeof(z)));
  return ICPX_OK;
}
#endif

extern "C" int icpx_set_leaf_size(icpx_ctx* c, int triangles)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (triangles != 1 && triangles != 2 && triangles != 4)
    return fail(ICPX_E_INVALID, "leaf size must be 1, 2 or 4 (got %d)", triangles);
  c->leaf_size = triangles;
  return ICPX_OK;
}

extern "C" int icpx_set_local(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->use_local = enable ? 1 : 0;
  c->use_cert = enable == 2 ? 0 : 1; // 2: the shortcut without its certificates
  c->certify_min = enable == 3 ? 0 : CERTIFY_MIN; // 3: certificates checked by their own kernel at any size
  c->local_holdoff = 0; // (and the policy starts afresh)
  c->bad_streak = 0;
  c->hopeless_streak = 0;
  c->direct = false;
  c->noisy = false;
  c->holdoff_len = 8;
  c->probation = false;
  return ICPX_OK;
}

extern "C" int icpx_set_refill(icpx_ctx* c, int lanes)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (lanes < 0 || lanes > 64)
    return fail(ICPX_E_INVALID, "refill threshold must be 0 (by regime) or 1..64 lanes (got %d)", lanes);
  c->refill_override = lanes;
  return ICPX_OK;
}

extern "C" int icpx_set_defer(icpx_ctx* c, int edges)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (edges < 0)
    return fail(ICPX_E_INVALID, "edge lengths must be 0 (never) or positive, got %d", edges);
  c->defer_edges = edges;
  return ICPX_OK;
}

extern "C" int icpx_set_step_budget(icpx_ctx* c, int steps)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (steps < 0)
    return fail(ICPX_E_INVALID, "step budget must be 0 (none) or positive, got %d", steps);
  c->step_budget = steps;
  return ICPX_OK;
}

extern "C" int icpx_set_bail(icpx_ctx* c, int steps /* microseconds */)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (steps < 0)
    return fail(ICPX_E_INVALID, "microseconds must be 0 (never), 1 (by the cloud's size) or a count (got %d)", steps);
  c->bail_mode = steps;
  return ICPX_OK;
}

extern "C" int icpx_set_finalize(icpx_ctx* c, int build)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (build < 0 || build > 2)
    return fail(ICPX_E_INVALID, "finalize build must be 0 (by size), 1 (registers) or 2 (matrix core), got %d", build);
  c->finalize_build = build;
  return ICPX_OK;
}

extern "C" int icpx_set_anchor(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->use_anchor = enable ? 1 : 0;
  return ICPX_OK;
}

extern "C" int icpx_set_lean(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (enable < 0 || enable > 3)
    return fail(ICPX_E_INVALID, "lean passes: 0 (off), 1 (on), 2 (tried at every chance) or 3 (as 2, "
                                "every third one made to fail), got %d", enable);
  c->use_lean = enable;
  return ICPX_OK;
}

extern "C" int icpx_set_shard_order(icpx_ctx* c, int order)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (order < 0 || order > 2)
    return fail(ICPX_E_INVALID, "shard order must be 0 (blocks of kept indices), 1 (ranges of "
                                "the curve order) or 2 (chunks of it dealt out in turn), got %d", order);
  c->shard_order = order;
  return ICPX_OK;
}

extern "C" int icpx_set_wave_blocks(icpx_ctx* c, int blocks_per_cu)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (blocks_per_cu < 1 || blocks_per_cu > 32)
    return fail(ICPX_E_INVALID, "blocks per CU must be 1..32, got %d", blocks_per_cu);
  c->wave_blocks = blocks_per_cu;
  return ICPX_OK;
}

extern "C" int icpx_set_query_curve(icpx_ctx* c, int hilbert)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->query_curve = hilbert ? 1 : 0;
  c->prep_step = -1; // (the queries are ordered anew)
  return ICPX_OK;
}

extern "C" int icpx_set_seed(icpx_ctx* c, int mode)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (mode < 0 || mode > 2)
    return fail(ICPX_E_INVALID, "seed mode must be 0 (curve), 1 (grid and curve) or 2 (grid), got %d", mode);
  c->seed_mode = mode;
  return ICPX_OK;
}

extern "C" int icpx_set_weigh(icpx_ctx* c, int mode)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (mode < 0 || mode > 3)
    return fail(ICPX_E_INVALID, "weigh mode must be 0..3, got %d", mode);
  c->weigh_in_finalize = mode;
  return ICPX_OK;
}

extern "C" int icpx_set_noisy_policy(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->noisy_policy = enable ? 1 : 0;
  return ICPX_OK;
}

extern "C" int icpx_set_certify_min(icpx_ctx* c, int64_t queries)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->certify_min = queries < 0 ? CERTIFY_MIN : queries;
  return ICPX_OK;
}

extern "C" int icpx_set_persistent(icpx_ctx* c, int mode)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (mode != 1 && mode != 3 && mode != 4)
    return fail(ICPX_E_INVALID, "tree kernel must be 1 (pooled lanes), 3 (one walk per "
                                "wavefront) or 4 (by regime); got %d", mode);
  c->persistent = mode;
  return ICPX_OK;
}

extern "C" int icpx_set_exhaustive(icpx_ctx* c, int enable)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->exhaustive = enable ? 1 : 0;
  return ICPX_OK;
}

extern "C" int icpx_reset_hints(icpx_ctx* c)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->hint_valid = false;
  c->stat_valid = false;
  c->stat_pending = false;
  c->local_holdoff = 0;
  c->bad_streak = 0;
  c->hopeless_streak = 0;
  c->direct = false;
  c->noisy = false;
  c->holdoff_len = 8;
  c->probation = false;
  return ICPX_OK;
}

extern "C" double icpx_last_build_seconds(icpx_ctx* c)
{
  return c ? c->build_seconds : 0.0;
}

extern "C" int icpx_get_origin(icpx_ctx* c, double origin[3])
{
  if (!c || !origin)
    return fail(ICPX_E_INVALID, "null argument");
  if (c->nt == 0)
    return fail(ICPX_E_INVALID, "icpx_get_origin: no target set");
  for (int i = 0; i < 3; i++)
    origin[i] = c->origin[i];
  return ICPX_OK;
}

extern "C" int icpx_set_allreduce(icpx_ctx* c, icpx_allreduce_fn fn, void* user)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  c->allreduce = fn;
  c->allreduce_user = user;
  return ICPX_OK;
}

extern "C" void icpx_default_params(icpx_params* p)
{
  if (!p)
    return;
  p->metric = ICPX_METRIC_POINT_TO_PLANE;
  p->max_iterations = 50;
  p->max_landmarks = 200;
  p->max_distance = 10.0;
  p->check_mean_distance = 0;
  p->max_mean_distance = 0.01;
  p->min_normal_dot = 0.0;
  p->rotate_source_normals = 0;
}

// ------------------------------------------------------------------ target ----
extern "C" int icpx_set_target(icpx_ctx* c, const float* xyz, int64_t nv, const int32_t* tri,
  int64_t nt, const float* cell_normals)
{
  if (!c || !xyz || !tri)
    return fail(ICPX_E_INVALID, "icpx_set_target: null argument");
  if (nv <= 0 || nt <= 0)
    return fail(ICPX_E_INVALID, "icpx_set_target: empty target (%lld points, %lld triangles)",
      (long long)nv, (long long)nt);
  if (nv >= (1ll << 31) || nt >= (1ll << 28) * 4)
    return fail(ICPX_E_INVALID, "icpx_set_target: target too large");
  for (int64_t i = 0; i < 3 * nt; i++)
    if (tri[i] < 0 || tri[i] >= nv)
      return fail(ICPX_E_INVALID, "icpx_set_target: triangle %lld refers to point %d of %lld",
        (long long)(i / 3), tri[i], (long long)nv);
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (int64_t i = 0; i < nv; i++)
    for (int a = 0; a < 3; a++)
    {
      float v = xyz[3 * i + a];
      if (!(std::fabs(v) <= 1e12f)) // (squared distances stay finite in single precision)
        return fail(ICPX_E_INVALID, "icpx_set_target: point %lld is not a number, or beyond 1e12 in magnitude",
          (long long)i);
      lo[a] = v < lo[a] ? v : lo[a];
      hi[a] = v > hi[a] ? v : hi[a];
    }

  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  auto t0 = std::chrono::steady_clock::now();
  free_target(c);
  c->hint_valid = false;
  c->prep_step = -1; // query order depends on the target's bounds
  hipStream_t st = c->stream;
  for (int a = 0; a < 3; a++)
  {
    c->box.lo[a] = lo[a];
    float ext = hi[a] - lo[a];
    c->box.inv[a] = ext > 0.f ? 1.f / ext : 0.f;
    c->origin[a] = 0.5 * ((double)lo[a] + (double)hi[a]);
  }
  c->scene_max = 0.f;
  for (int a = 0; a < 3; a++)
    c->scene_max = std::fmax(c->scene_max, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
  {
    double l2 = 0.0;
    for (int a = 0; a < 3; a++)
      l2 += ((double)hi[a] - (double)lo[a]) * ((double)hi[a] - (double)lo[a]);
    c->bounds_length = std::sqrt(l2);
  }
  {
    // typical edge length, from an even sample of the triangles (sizes the neighbour lists),
    // and how widely the triangles' sizes range: the median of the sampled triangles' longest
    // edges over the shortest edge of the sample's smallest thousandth (an evenly meshed
    // surface: 2 to 4; a decimated or graded one: dozens).  65 536 triangles are sampled and
    // a thousandth is looked at because a patch of a hundredth of the triangles is enough:
    // three of a million triangles cut into 64 x 64 each keep one batch of the cold pass
    // walking for 7.6 ms where the pass takes 0.9 without them (tools/shape_probe.py patch3).
    const int64_t stride = std::max<int64_t>(1, nt / 65536);
    double sum = 0.0;
    int64_t cnt = 0;
    std::vector<float> shortest, longest;
    for (int64_t t = 0; t < nt; t += stride, cnt++)
    {
      double emin = INFINITY, emax = 0.0;
      for (int k = 0; k < 3; k++)
      {
        const float* p = xyz + 3 * (int64_t)tri[3 * t + k];
        const float* q = xyz + 3 * (int64_t)tri[3 * t + (k + 1) % 3];
        const double dx = (double)p[0] - q[0], dy = (double)p[1] - q[1], dz = (double)p[2] - q[2];
        const double e = std::sqrt(dx * dx + dy * dy + dz * dz);
        sum += e;
        emin = std::fmin(emin, e);
        emax = std::fmax(emax, e);
      }
      if (emin > 0.0 && std::isfinite(emax))
      {
        shortest.push_back((float)emin);
        longest.push_back((float)emax);
      }
    }
    c->mesh_edge = (float)(sum / (3.0 * (double)cnt));
    c->mesh_grading = 1.f;
    if (shortest.size() >= 64)
    {
      std::nth_element(shortest.begin(), shortest.begin() + shortest.size() / 1000, shortest.end());
      std::nth_element(longest.begin(), longest.begin() + longest.size() / 2, longest.end());
      c->mesh_grading = longest[longest.size() / 2] / shortest[shortest.size() / 1000];
    }
  }

  ALLOC(c->d_xyz, 3 * nv);
  ALLOC(c->d_tri, 3 * nt);
  HIPCHK(hipMemcpyAsync(c->d_xyz, xyz, sizeof(float) * 3 * nv, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->d_tri, tri, sizeof(int32_t) * 3 * nt, hipMemcpyHostToDevice, st));

  // K1/K2: Morton order of the triangles
  uint64_t *k0 = nullptr, *k1 = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr, *scr = nullptr;
  int rc = dev_alloc(&k0, nt);
  if (!rc)
    rc = dev_alloc(&k1, nt);
  if (!rc)
    rc = dev_alloc(&v0, nt);
  if (!rc)
    rc = dev_alloc(&v1, nt);
  if (!rc)
    rc = dev_alloc(&scr, rs_scratch_words(nt));
  if (!rc)
    rc = dev_alloc(&c->d_tverts, 9 * nt);
  if (!rc)
    rc = dev_alloc(&c->d_tid, nt);
  if (!rc)
    rc = dev_alloc(&c->d_tpre, 2 * (nt + TPRE_PAD));
  auto drop_sort = [&]() {
    dev_free(&k0);
    dev_free(&k1);
    dev_free(&v0);
    dev_free(&v1);
    dev_free(&scr);
  };
  if (rc)
  {
    drop_sort();
    return rc;
  }
  k_morton_tris<<<nblk(nt, 256), 256, 0, st>>>(c->d_xyz, c->d_tri, nt, c->box, k0, v0);
  int which = radix_sort_pairs(k0, v0, k1, v1, nt, 63, scr, st);
  k_gather_tris<<<nblk(nt, 256), 256, 0, st>>>(
    c->d_xyz, c->d_tri, which ? v1 : v0, nt, c->d_tverts, c->d_tid);
  k_flag_degenerate<<<nblk(nt, 256), 256, 0, st>>>(c->d_tverts, nt, c->d_tid);
  k_tri_bounds<<<nblk(nt + TPRE_PAD, 256), 256, 0, st>>>(
    c->d_tverts, c->d_tid, nt, nt + TPRE_PAD, c->d_tpre);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    rc = dev_alloc(&c->d_tcodes, nt);
  if (e == hipSuccess && !rc)
    e = hipMemcpyAsync(
      c->d_tcodes, which ? k1 : k0, sizeof(uint64_t) * nt, hipMemcpyDeviceToDevice, st);
  // (about two cells per triangle, 2^24 at most: 64 MB)
  int cbits = 3;
  while (cbits < 24 && ((int64_t)1 << cbits) < 2 * nt)
    cbits += 3; // (whole levels of the octree the codes spell out)
  c->tcell_bits = cbits;
  if (e == hipSuccess && !rc)
    rc = dev_alloc(&c->d_tcell, ((int64_t)1 << cbits) + 1);
  if (e == hipSuccess && !rc)
  {
    k_code_cells<<<nblk(((int64_t)1 << cbits) + 1, 256), 256, 0, st>>>(c->d_tcodes, nt, cbits, c->d_tcell);
    e = hipGetLastError();
  }
  // the seed grid: four cells per triangle at least, 16^3 to 512^3 (half a gigabyte, for ten
  // million triangles: cold pass 22.9 ms with 256^3, 21.3 with 512^3)
  int sdim = 16;
  {
    static const char* cap = std::getenv("ICPX_SEED_DIM"); // (development: the grid's largest size)
    const int dmax = cap ? std::atoi(cap) : 512;
    while (sdim < dmax && (int64_t)sdim * sdim * sdim < 4 * nt)
      sdim *= 2;
  }
  int32_t* seed_tmp = nullptr;
  const int64_t scells = (int64_t)sdim * sdim * sdim;
  if (e == hipSuccess && !rc)
    rc = dev_alloc(&c->d_seed, scells);
  if (e == hipSuccess && !rc)
    rc = dev_alloc(&seed_tmp, scells);
  if (e == hipSuccess && !rc)
  {
    c->seed_dim = sdim;
    e = hipMemsetAsync(c->d_seed, 0xff, sizeof(int32_t) * scells, st);
    k_seed_scatter<<<nblk(nt, 256), 256, 0, st>>>(c->d_tpre, nt, c->box, sdim,
      reinterpret_cast<uint32_t*>(c->d_seed));
    int32_t *src = c->d_seed, *dst = seed_tmp;
    for (int stride = sdim / 2; stride >= 1; stride /= 2)
    {
      k_seed_flood<<<nblk(scells, 256), 256, 0, st>>>(src, dst, c->d_tpre, c->box, sdim, stride);
      std::swap(src, dst);
    }
    if (src != c->d_seed) // (the last pass wrote the other buffer: that one is kept)
      std::swap(c->d_seed, seed_tmp);
    if (e == hipSuccess)
      e = hipGetLastError();
  }
  const hipError_t es = hipStreamSynchronize(st); // (the sort's buffers are in use until here)
  dev_free(&seed_tmp);
  if (e == hipSuccess)
    e = es;
  drop_sort();
  if (rc)
    return rc;
  if (e != hipSuccess)
    return fail(ICPX_E_HIP, "target ordering failed: %s", hipGetErrorString(e));

  // K5: links and pseudo-normal tables
  double* fn64 = nullptr;
  uint32_t *voff = nullptr, *cursor = nullptr, *sscr = nullptr;
  int32_t* vcell = nullptr;
  float* d_cn = nullptr;
  auto drop_tab = [&]() {
    dev_free(&fn64);
    dev_free(&voff);
    dev_free(&cursor);
    dev_free(&sscr);
    dev_free(&vcell);
    dev_free(&d_cn);
  };
  rc = dev_alloc(&fn64, 3 * nt);
  if (!rc)
    rc = dev_alloc(&voff, nv + 1);
  if (!rc)
    rc = dev_alloc(&cursor, nv + 1);
  if (!rc)
    rc = dev_alloc(&sscr, scan_scratch_words(nv + 1));
  if (!rc)
    rc = dev_alloc(&vcell, 3 * nt);
  if (!rc && cell_normals)
    rc = dev_alloc(&d_cn, 3 * nt);
  if (!rc)
    rc = dev_alloc(&c->d_fn, 3 * nt);
  if (!rc)
    rc = dev_alloc(&c->d_fnpos, nt);
  if (!rc)
    rc = dev_alloc(&c->d_en, 9 * nt);
  if (!rc)
    rc = dev_alloc(&c->d_vn, 3 * nv);
  if (rc)
  {
    drop_tab();
    return rc;
  }
  if (cell_normals)
    (void)hipMemcpyAsync(d_cn, cell_normals, sizeof(float) * 3 * nt, hipMemcpyHostToDevice, st);
  k_face_normals<<<nblk(nt, 256), 256, 0, st>>>(c->d_xyz, c->d_tri, d_cn, nt, fn64, c->d_fn);
  k_face_by_position<<<nblk(nt, 256), 256, 0, st>>>(c->d_fn, c->d_tid, nt, c->d_fnpos);
  (void)hipMemsetAsync(voff, 0, sizeof(uint32_t) * (nv + 1), st);
  (void)hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (nv + 1), st);
  k_valence<<<nblk(3 * nt, 256), 256, 0, st>>>(c->d_tri, 3 * nt, voff);
  exclusive_scan(voff, voff, nv + 1, sscr, st);
  k_fill_links<<<nblk(3 * nt, 256), 256, 0, st>>>(c->d_tri, 3 * nt, voff, cursor, vcell);
  k_sort_links<<<nblk(nv, 256), 256, 0, st>>>(voff, nv, vcell);
  k_vertex_normals<<<nblk(nv, 256), 256, 0, st>>>(
    c->d_xyz, c->d_tri, voff, vcell, fn64, nv, c->d_vn);
  k_edge_normals<<<nblk(3 * nt, 256), 256, 0, st>>>(c->d_tri, voff, vcell, fn64, nt, c->d_en);
  e = hipGetLastError();
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  if (e == hipSuccess)
  {
    // the fp64 face normals and the links stay: double-precision rows are made from them
    c->d_fn64 = fn64;
    c->d_voff = voff;
    c->d_vcell = vcell;
    fn64 = nullptr;
    voff = nullptr;
    vcell = nullptr;
  }
  drop_tab();
  if (e != hipSuccess)
    return fail(ICPX_E_HIP, "normal tables failed: %s", hipGetErrorString(e));

  c->nv = nv;
  c->nt = nt;
  c->build_seconds =
    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return ICPX_OK;
}

extern "C" int icpx_point_normals(icpx_ctx* c, const float* xyz, int64_t nv, const int32_t* tri,
  int64_t nt, float* normals_out)
{
  if (!c || !xyz || !tri || !normals_out)
    return fail(ICPX_E_INVALID, "icpx_point_normals: null argument");
  if (nv <= 0 || nt <= 0)
    return fail(ICPX_E_INVALID, "icpx_point_normals: empty mesh (%lld points, %lld triangles)",
      (long long)nv, (long long)nt);
  if (nv >= (1ll << 31) || nt >= (1ll << 28) * 4)
    return fail(ICPX_E_INVALID, "icpx_point_normals: mesh too large");
  for (int64_t i = 0; i < 3 * nt; i++)
    if (tri[i] < 0 || tri[i] >= nv)
      return fail(ICPX_E_INVALID, "icpx_point_normals: triangle %lld refers to point %d of %lld",
        (long long)(i / 3), tri[i], (long long)nv);
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  float *d_xyz = nullptr, *d_pn = nullptr, *fn32 = nullptr;
  int32_t *d_tri = nullptr, *vcell = nullptr;
  double* fn64 = nullptr;
  uint32_t *voff = nullptr, *cursor = nullptr, *sscr = nullptr;
  auto drop = [&]() {
    dev_free(&d_xyz);
    dev_free(&d_pn);
    dev_free(&fn32);
    dev_free(&d_tri);
    dev_free(&vcell);
    dev_free(&fn64);
    dev_free(&voff);
    dev_free(&cursor);
    dev_free(&sscr);
  };
  int rc = dev_alloc(&d_xyz, 3 * nv);
  if (!rc)
    rc = dev_alloc(&d_pn, 3 * nv);
  if (!rc)
    rc = dev_alloc(&fn32, 3 * nt);
  if (!rc)
    rc = dev_alloc(&d_tri, 3 * nt);
  if (!rc)
    rc = dev_alloc(&vcell, 3 * nt);
  if (!rc)
    rc = dev_alloc(&fn64, 3 * nt);
  if (!rc)
    rc = dev_alloc(&voff, nv + 1);
  if (!rc)
    rc = dev_alloc(&cursor, std::max<int64_t>(nv + 1, 4)); // (also three flag words later on)
  if (!rc)
    rc = dev_alloc(&sscr, scan_scratch_words(nv + 1));
  if (rc)
  {
    drop();
    return rc;
  }
  hipError_t e = hipMemcpyAsync(d_xyz, xyz, sizeof(float) * 3 * nv, hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_tri, tri, sizeof(int32_t) * 3 * nt, hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    e = hipMemsetAsync(voff, 0, sizeof(uint32_t) * (nv + 1), st);
  if (e == hipSuccess)
    e = hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (nv + 1), st);
  if (e == hipSuccess)
  {
    // the same links as the target's tables are made from
    k_valence<<<nblk(3 * nt, 256), 256, 0, st>>>(d_tri, 3 * nt, voff);
    exclusive_scan(voff, voff, nv + 1, sscr, st);
    k_fill_links<<<nblk(3 * nt, 256), 256, 0, st>>>(d_tri, 3 * nt, voff, cursor, vcell);
    k_sort_links<<<nblk(nv, 256), 256, 0, st>>>(voff, nv, vcell);
    e = hipGetLastError();
  }
  // ConsistencyOn: cells wound against their piece's lowest cell are reversed first
  c->last_flips = 0;
  c->last_unorientable = false;
  if (e == hipSuccess)
  {
    // (fn64 is not needed yet: its memory holds the neighbour table and the labels meanwhile)
    int32_t* nbr = reinterpret_cast<int32_t*>(fn64);          // 3 nt int32
    uint32_t* lab = reinterpret_cast<uint32_t*>(fn64) + 3 * nt; // nt uint32 (fn64 has 6 nt words)
    uint32_t* flags = cursor; // [0] inconsistent edges, [1] changed, [2] flipped, [3] a piece that cannot be wound consistently
    uint32_t h_flags[4] = { 0, 0, 0, 0 };
    e = hipMemsetAsync(flags, 0, 4 * sizeof(uint32_t), st);
    if (e == hipSuccess)
    {
      k_edge_neighbours<<<nblk(3 * nt, 256), 256, 0, st>>>(d_tri, voff, vcell, nt, nbr, flags);
      e = hipMemcpyAsync(h_flags, flags, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    if (e == hipSuccess && h_flags[0] != 0)
    {
      k_orient_init<<<nblk(nt, 256), 256, 0, st>>>(nt, lab);
      // the smallest cell id of a piece reaches every cell of it after a number of sweeps that
      // grows with the logarithm of the piece's length (the sweeps also follow the labels'
      // own references); look every few dozen sweeps whether anything still moves
      for (int64_t round = 0; e == hipSuccess && round <= nt / 32 + 1; round++)
      {
        e = hipMemsetAsync(flags + 1, 0, sizeof(uint32_t), st);
        for (int k = 0; k < 32 && e == hipSuccess; k++)
        {
          k_orient_sweep<<<nblk(nt, 256), 256, 0, st>>>(nbr, nt, lab, flags + 1, flags + 3);
          e = hipGetLastError();
        }
        if (e == hipSuccess)
          e = hipMemcpyAsync(h_flags + 1, flags + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
          e = hipStreamSynchronize(st);
        if (h_flags[1] == 0)
          break;
      }
      if (e == hipSuccess)
      {
        k_orient_apply<<<nblk(nt, 256), 256, 0, st>>>(lab, nt, d_tri, flags + 2);
        e = hipMemcpyAsync(h_flags + 2, flags + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
      }
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      c->last_flips = h_flags[2];
      c->last_unorientable = h_flags[3] != 0;
    }
  }
  if (e == hipSuccess)
  {
    k_face_normals<<<nblk(nt, 256), 256, 0, st>>>(d_xyz, d_tri, nullptr, nt, fn64, fn32);
    k_point_normals<<<nblk(nv, 256), 256, 0, st>>>(voff, vcell, fn64, nv, d_pn);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(normals_out, d_pn, sizeof(float) * 3 * nv, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  drop();
  if (e != hipSuccess)
    return fail(ICPX_E_HIP, "icpx_point_normals failed: %s", hipGetErrorString(e));
  return ICPX_OK;
}

// K3: the box hierarchy for one fan-out, built on first use.
static int ensure_tree(icpx_ctx* c, int G)
{
  int slot = -1;
  for (int i = 0; i < 5; i++)
    if (kWidths[i] == G)
      slot = i;
  if (slot < 0)
    return fail(ICPX_E_INVALID, "bad group width %d", G);
  Tree& t = c->trees[slot];
  if (t.built)
    return ICPX_OK;
  t.lv = make_levels(c->nt, G);
  int maxlev = (G == 4) ? SearchCfg<4>::LEVELS
                        : (G == 8) ? SearchCfg<8>::LEVELS
                                   : (G == 16) ? SearchCfg<16>::LEVELS
                                               : (G == 32) ? SearchCfg<32>::LEVELS
                                                           : SearchCfg<64>::LEVELS;
  if (t.lv.cnt[t.lv.nlev] != 1 || t.lv.nlev > maxlev || t.lv.cnt[0] >= (1 << 28))
    return fail(ICPX_E_INVALID, "target too large for fan-out %d", G);
  ALLOC(t.nodes, t.lv.total_floats);
  hipStream_t st = c->stream;
  k_build_level1<<<nblk((int64_t)t.lv.cnt[1] * G, 256), 256, 0, st>>>(
    c->d_tverts, c->nt, t.lv, t.nodes);
  for (int L = 2; L <= t.lv.nlev; L++)
    k_build_level<<<nblk((int64_t)t.lv.cnt[L] * G, 256), 256, 0, st>>>(L, t.lv, t.nodes);
  LAUNCHCHK();
  t.built = true;
  return ICPX_OK;
}

static int ensure_heap(icpx_ctx* c, int G)
{
  int slot = -1;
  for (int i = 0; i < 5; i++)
    if (kWidths[i] == G)
      slot = i;
  if (slot < 0)
    return fail(ICPX_E_INVALID, "bad group width %d", G);
  Heap& h = c->heaps[slot];
  if (h.built)
    return ICPX_OK;
  h.t = make_heap_tree(c->nt, G);
  int maxlev = (G == 4) ? Search2Cfg<4>::LEVELS
                        : (G == 8) ? Search2Cfg<8>::LEVELS
                                   : (G == 16) ? Search2Cfg<16>::LEVELS
                                               : (G == 32) ? Search2Cfg<32>::LEVELS
                                                           : Search2Cfg<64>::LEVELS;
  if (h.t.depth > maxlev || h.t.first_leaf + h.t.leaf_cap >= (1ll << 32))
    return fail(ICPX_E_INVALID, "target too large for fan-out %d", G);
  ALLOC(h.nodes, h.t.first_leaf * 6 * G);
  hipStream_t st = c->stream;
  k_heap_bottom<<<nblk(h.t.leaf_cap, 256), 256, 0, st>>>(
    c->d_tverts, c->d_tid, c->nt, h.t, h.nodes);
  for (int k = h.t.depth - 2; k >= 0; k--)
    k_heap_level<<<nblk(ipow(G, k + 1), 256), 256, 0, st>>>(k, h.t, h.nodes);
  LAUNCHCHK();
  h.built = true;
  return ICPX_OK;
}

// ------------------------------------------------------------------ source ----
extern "C" int icpx_set_source(icpx_ctx* c, const float* xyz, int64_t np, const float* normals)
{
  if (!c || !xyz)
    return fail(ICPX_E_INVALID, "icpx_set_source: null argument");
  if (np <= 0 || np >= (1ll << 31))
    return fail(ICPX_E_INVALID, "icpx_set_source: bad point count %lld", (long long)np);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  ALLOC(c->d_src, 3 * np);
  HIPCHK(hipMemcpyAsync(c->d_src, xyz, sizeof(float) * 3 * np, hipMemcpyHostToDevice, c->stream));
  if (normals)
  {
    ALLOC(c->d_srcn, 3 * np);
    HIPCHK(hipMemcpyAsync(
      c->d_srcn, normals, sizeof(float) * 3 * np, hipMemcpyHostToDevice, c->stream));
  }
  else
    dev_free(&c->d_srcn);
  // (largest |coordinate|, for the reach of anchored passes: non-negative floats order as
  // their bits do)
  uint32_t* d_max = nullptr;
  ALLOC(d_max, 1);
  HIPCHK(hipMemsetAsync(d_max, 0, sizeof(uint32_t), c->stream));
  k_abs_max<<<std::min<unsigned>(nblk(3 * np, 256), 4096u), 256, 0, c->stream>>>(c->d_src, 3 * np, d_max);
  uint32_t bits = 0;
  HIPCHK(hipMemcpyAsync(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  dev_free(&d_max);
  std::memcpy(&c->src_max, &bits, sizeof(bits));
  c->src_version++;
  // (the bits of a NaN order above those of every number, so the maximum shows one)
  if (!(c->src_max <= 1e12f))
  {
    c->np = 0; // (what was uploaded is not searched)
    c->src_max = 0.f;
    return fail(ICPX_E_INVALID, "icpx_set_source: a coordinate is not a number, or beyond 1e12 in magnitude");
  }
  c->np = np;
  return ICPX_OK;
}

extern "C" int icpx_set_shard(icpx_ctx* c, int rank, int world)
{
  if (!c)
    return fail(ICPX_E_INVALID, "null context");
  if (world < 1 || rank < 0 || rank >= world)
    return fail(ICPX_E_INVALID, "icpx_set_shard: bad shard %d of %d", rank, world);
  c->rank = rank;
  c->world = world;
  return ICPX_OK;
}

// Which kept points are this context's shard (icpx_ctx::shard_order).  Returns whether the
// shard is cut from the curve order of all nk kept points (then *map says which positions of
// that order are its queries, *nq how many); otherwise it is kept indices [*k0, *k0 + *nq).
static bool shard_extent(const icpx_ctx* c, int64_t nk, int64_t* k0, int64_t* nq, ShardMap* map)
{
  const int r = c->rank, w = c->world;
  *k0 = nk * r / w;
  *nq = nk * (r + 1) / w - *k0;
  map->rank = 0;
  map->world = 1;
  map->chunk = std::max<int64_t>(nk, 1);
  if (w == 1 || c->shard_order == 0)
    return false;
  if (c->shard_order == 1)
  {
    // one range per rank: positions [k0, k0 + nq), i.e. one chunk that starts at k0
    // (pos(j) = j with the order's pointer moved up by k0: prepare_queries does that)
    return true;
  }
  int64_t chunk = 8192;
  while (chunk > 1 && nk < 8 * w * chunk)
    chunk /= 2; // (a small cloud: at least eight chunks per rank)
  const int64_t nchunks = (nk + chunk - 1) / chunk;
  const int64_t mine = r < nchunks ? (nchunks - r + w - 1) / w : 0;
  int64_t n = mine * chunk;
  if (mine > 0 && (nchunks - 1) % w == r && nk % chunk != 0)
    n -= chunk - nk % chunk; // (the order's last chunk is a short one)
  *k0 = 0;
  *nq = n;
  map->rank = r;
  map->world = w;
  map->chunk = chunk;
  return true;
}

// Kept points of this shard, gathered in the order of a space-filling curve through the
// target's bounds (Hilbert's by default, icpx_ctx::query_curve).
// One rank, or shards by blocks of kept indices: the shard's own points are ordered.  Shards
// by the curve (the default for several ranks): ALL kept points are ordered (every rank does
// that for itself, from the same source and the same bounds, with a stable sort: the same
// order everywhere) and the shard is the positions of that order that shard_extent names.
static int prepare_queries(icpx_ctx* c, int64_t step)
{
  if (c->nt == 0)
    return fail(ICPX_E_INVALID, "no target set");
  if (c->np == 0)
    return fail(ICPX_E_INVALID, "no source set");
  if (step < 1)
    return fail(ICPX_E_INVALID, "step must be >= 1");
  if (c->prep_step == step && c->prep_rank == c->rank && c->prep_world == c->world &&
    c->prep_order == c->shard_order && c->prep_version == c->src_version)
    return ICPX_OK;
  const int64_t nk = (c->np + step - 1) / step;
  int64_t k0 = 0, nq = 0;
  ShardMap map;
  const bool curve = shard_extent(c, nk, &k0, &nq, &map);
  const int64_t nsort = curve ? nk : nq; // points that are ordered here
  hipStream_t st = c->stream;
  if (nq > c->q_cap)
  {
    ALLOC(c->d_qpos, 3 * nq);
    ALLOC(c->d_qsn, 3 * nq);
    ALLOC(c->d_qk, nq);
    ALLOC(c->d_hint, nq);
    ALLOC(c->d_kglobal, nq);
    c->q_cap = nq;
  }
  if (nq > 0)
  {
    uint64_t *ka = nullptr, *kb = nullptr;
    uint32_t *va = nullptr, *vb = nullptr, *scr = nullptr;
    int rc = dev_alloc(&ka, nsort);
    if (!rc)
      rc = dev_alloc(&kb, nsort);
    if (!rc)
      rc = dev_alloc(&va, nsort);
    if (!rc)
      rc = dev_alloc(&vb, nsort);
    if (!rc)
      rc = dev_alloc(&scr, rs_scratch_words(nsort));
    hipError_t e = hipSuccess;
    if (!rc && !curve)
    {
      k_morton_points<<<nblk(nq, 256), 256, 0, st>>>(c->d_src, k0, step, nq, c->box, ka, va, c->query_curve);
      int which = radix_sort_pairs(ka, va, kb, vb, nq, 63, scr, st);
      k_gather_queries<<<nblk(nq, 256), 256, 0, st>>>(c->d_src, c->d_srcn, which ? vb : va, k0,
        step, nq, c->d_qpos, c->d_qsn, c->d_qk, c->d_hint);
      e = hipGetLastError();
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    }
    else if (!rc)
    {
      k_morton_points<<<nblk(nk, 256), 256, 0, st>>>(c->d_src, 0, step, nk, c->box, ka, va, c->query_curve);
      const int which = radix_sort_pairs(ka, va, kb, vb, nk, 63, scr, st);
      // (the order stays in one pair of buffers; the other pair and the scratch, which are
      // sized for nk >= nq, serve the second sort: the shard's kept indices, ascending)
      const uint32_t* order = (which ? vb : va) + k0; // (k0: a range's start; 0 for chunks)
      uint64_t* k2a = which ? ka : kb;
      uint32_t* v2a = which ? va : vb;
      // (two more buffers of nq: the sort ping-pongs)
      uint64_t* k2b = nullptr;
      uint32_t* v2b = nullptr;
      rc = dev_alloc(&k2b, nq);
      if (!rc)
        rc = dev_alloc(&v2b, nq);
      if (!rc)
      {
        k_shard_keys<<<nblk(nq, 256), 256, 0, st>>>(order, map, nq, k2a, v2a);
        const int w2 = radix_sort_pairs(k2a, v2a, k2b, v2b, nq, 32, scr, st);
        k_shard_slots<<<nblk(nq, 256), 256, 0, st>>>(w2 ? k2b : k2a, w2 ? v2b : v2a, nq,
          c->d_kglobal, c->d_qk);
        k_gather_queries_curve<<<nblk(nq, 256), 256, 0, st>>>(c->d_src, c->d_srcn, order, map,
          step, nq, c->d_qpos, c->d_qsn, c->d_hint);
        e = hipGetLastError();
        if (e == hipSuccess)
          e = hipStreamSynchronize(st);
      }
      else
        (void)hipStreamSynchronize(st);
      dev_free(&k2b);
      dev_free(&v2b);
    }
    dev_free(&ka);
    dev_free(&kb);
    dev_free(&va);
    dev_free(&vb);
    dev_free(&scr);
    if (rc)
      return rc;
    if (e != hipSuccess)
      return fail(ICPX_E_HIP, "query ordering failed: %s", hipGetErrorString(e));
  }
  c->nq = nq;
  c->nk = nk;
  c->k0 = k0;
  c->curve_shard = curve;
  c->prep_step = step;
  c->prep_rank = c->rank;
  c->prep_world = c->world;
  c->prep_order = c->shard_order;
  c->prep_version = c->src_version;
  c->hint_valid = false;
  return ICPX_OK;
}

static XfRef xf_ref(icpx_ctx* c)
{
  XfRef r;
  r.T = c->d_state->T;
  r.stop = &c->d_state->done;
  return r;
}

// Puts a host transform into the device state the kernels read (stream-ordered).
static int set_transform(icpx_ctx* c, const double T[16], int reset)
{
  Mat16 m;
  std::memcpy(m.m, T, sizeof(m.m));
  k_set_state<<<1, 64, 0, c->stream>>>(c->d_state, m, reset);
  LAUNCHCHK();
  return ICPX_OK;
}

static int pick_width(const icpx_ctx* c)
{
  if (c->group_width)
    return (c->exhaustive && c->group_width == 1) ? 8 : c->group_width;
  // few queries: spend a whole wavefront on each; many: one lane each
  if (c->nq <= 4096)
    return 64;
  if (c->nq <= 32768 || c->exhaustive)
    return 8;
  return 1;
}

static int ensure_bin(icpx_ctx* c, int L)
{
  const int slot = (L == 1) ? 0 : (L == 2) ? 1 : 2;
  Bin& b = c->bins[slot];
  if (b.built)
    return ICPX_OK;
  b.t = make_bin_tree(c->nt, L);
  if (b.t.depth > 30)
    return fail(ICPX_E_INVALID, "target too large");
  ALLOC(b.nodes, (1ll << b.t.depth) * NODE_FLOATS);
  hipStream_t st = c->stream;
  k_bin_bottom<<<nblk(1ll << b.t.depth, 256), 256, 0, st>>>(
    c->d_tverts, c->d_tid, c->nt, b.t, b.nodes);
  for (int k = b.t.depth - 2; k >= 0; k--)
    k_bin_level<<<nblk(2ll << k, 256), 256, 0, st>>>(k, b.nodes);
  ALLOC(b.lpre, 2 * b.t.nleaf);
  k_leaf_bounds<<<nblk(b.t.nleaf, 256), 256, 0, st>>>(c->d_tverts, c->d_tid, c->nt, b.t, b.lpre);
  LAUNCHCHK();
  b.built = true;
  return ICPX_OK;
}

static bool heavy_cloud(const icpx_ctx* c);
static bool leads_kept(const icpx_ctx* c);

template <int L>
static void launch_search3(icpx_ctx* c, const Search3Args& a, unsigned grid, unsigned pgrid,
  unsigned wgrid, int mode, const Search2Args* wide, hipStream_t st)
{
  // The queries a walk gave up on (the batches of the wavefront walk past their deadline, the
  // queries of the pooled lanes past their step budget), one wavefront each; batches' queries,
  // should they be many, on the pooled lanes (most of them are their batch's bystanders).
  auto given_up = [&](const Search3Args& from) {
    Search3Args ab = from;
    ab.qlist = from.bail_list;
    ab.count_slot = from.bail_slot;
    ab.list_lo = 0;
    ab.list_hi = INT64_MAX;
    ab.use_hint = 1;
    ab.setmask = nullptr;
    ab.bail_ticks = 0u;
    ab.step_budget = 0u;
    if (from.bail_ticks)
      k_reset_pools<<<1, 64, 0, st>>>(from.counters); // (the walk has drawn from them)
    else
      ab.small_limit = 0xffffffffu; // (given up on one by one: every one of them is a long walk)
    launch_ev(c, false, k_fallback<L>, pgrid, 256, st, ab, *wide);
  };
  if (pgrid && mode == 1 && a.qlist && a.cert && heavy_cloud(c))
  {
    // The leftovers of a large cloud: the host does not know how many there are.  Most of
    // the cloud (the first passes with the shortcut, when hardly any query holds a
    // certificate yet) is served faster by the wavefront walk, a scattered few percent by
    // the pooled lanes: both are enqueued, and the one whose side of nq/2 the list's length
    // is on does the work (the other's blocks return after one read).
    c->split_passes++;
    c->last_split = true;
    Search3Args lo = a, hi = a;
    lo.list_hi = a.nq / 2;
    hi.list_lo = a.nq / 2;
    if (wide)
      launch_ev(c, false, k_fallback<L>, pgrid, 256, st, lo, *wide);
    else
      launch_ev(c, false, k_search3p<L>, pgrid, 256, st, lo);
    // (the walk over a long list gives up on batches as a whole pass does, onto the list of
    // its own that the launch behind it serves one wavefront per query)
    if (a.bail_ticks && wide)
      launch_ev(c, false, k_search3w<L, true, true>, wgrid, 64, st, hi);
    else
      launch_ev(c, false, k_search3w<L, true>, wgrid, 64, st, hi);
    if ((a.bail_ticks || a.step_budget || a.defer_far > 0.f) && wide)
      given_up(a);
  }
  else if (wide && pgrid && mode == 1 && a.qlist)
  {
    launch_ev(c, false, k_fallback<L>, pgrid, 256, st, a, *wide);
    if (a.step_budget)
      given_up(a);
  }
  else if (mode == 3)
  {
    const bool bail = a.bail_ticks && wide;
    if (a.cert && bail)
      launch_ev(c, false, k_search3w<L, true, true>, wgrid, 64, st, a);
    else if (a.cert)
      launch_ev(c, false, k_search3w<L, true>, wgrid, 64, st, a);
    else if (bail)
      launch_ev(c, false, k_search3w<L, false, true>, wgrid, 64, st, a);
    else
      launch_ev(c, false, k_search3w<L, false>, wgrid, 64, st, a);
    if (bail || (a.defer_far > 0.f && wide))
      given_up(a);
  }
  else
  {
    launch_ev(c, false, k_search3p<L>, pgrid, 256, st, a);
    if (a.step_budget && wide)
      given_up(a);
  }
  unsigned rgrid = grid / 8 + 1; // (a block of four wavefronts weighs the sets of 2048 queries)
  // (at most twice the wavefronts the chip holds: beyond four million queries a wavefront
  // takes several chunks of 512 and weighs their marked queries 64 at a time, whichever chunk
  // they came from; C4's settled pass 0.283 -> 0.252 ms against a chunk per wavefront)
  if (rgrid > 2048)
    rgrid = 2048;
  // (overflows are rare: a bounded number of blocks takes them in turns)
  Search3Args ac = a;
  c->pass_marked = a.setmask != nullptr;
  if (c->weigh_later)
  {
    ac.setmask = nullptr; // (the marked queries' sets are weighed by the finalize that follows)
    rgrid = std::min(rgrid, 1024u);
  }
  launch_ev(c, true, k_contested<L>, rgrid + std::min(grid, 256u), 256, st, ac, rgrid);
}

// Neighbour lists of every leaf of the one-triangle-per-leaf tree for radius eps (see
// local.hpp).  Leaves c->nbr_eps == 0 when the lists would be too long to pay off.
static int build_nbr(icpx_ctx* c, const Bin& b, float eps)
{
  hipStream_t st = c->stream;
  dev_free(&c->d_noff);
  dev_free(&c->d_nbr);
  c->nbr_eps = 0.f;
  const int64_t nl = b.t.nleaf;
  uint32_t* scr = nullptr;
  ALLOC(c->d_noff, nl + 1);
  int rc = dev_alloc(&scr, scan_scratch_words(nl + 1));
  if (rc)
    return rc;
  const float eps2 = nextafterf(eps * eps, INFINITY);
  (void)hipMemsetAsync(c->d_noff + nl, 0, sizeof(uint32_t), st);
  k_nbr<false><<<nblk(nl, 256), 256, 0, st>>>(b.nodes, b.t, eps2, c->d_noff, nullptr, nullptr);
  exclusive_scan(c->d_noff, c->d_noff, nl + 1, scr, st);
  uint32_t total = 0;
  hipError_t e = hipMemcpyAsync(&total, c->d_noff + nl, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  dev_free(&scr);
  if (e != hipSuccess)
    return fail(ICPX_E_HIP, "neighbour count failed: %s", hipGetErrorString(e));
  c->nbr_builds++;
  c->nbr_total = total;
  if ((double)total > 32.0 * (double)nl)
  {
    dev_free(&c->d_noff); // too many neighbours at this radius: the tree walk is better
    return ICPX_OK;
  }
  ALLOC(c->d_nbr, (int64_t)total + 4);
  k_nbr<true><<<nblk(nl, 256), 256, 0, st>>>(b.nodes, b.t, eps2, nullptr, c->d_noff, c->d_nbr);
  ALLOC(c->d_qbox, nl);
  float ext = 0.f;
  for (int a = 0; a < 3; a++)
  {
    ext = std::fmax(ext, c->box.inv[a] > 0.f ? 1.f / c->box.inv[a] : 0.f);
    c->qgrid.base[a] = c->box.lo[a];
  }
  c->qgrid.step = std::fmax(ext * 1.000001f, 1e-30f) / 65535.f;
  c->qgrid.inv_step = 1.f / c->qgrid.step;
  k_quant_boxes<<<nblk(nl, 256), 256, 0, st>>>(b.nodes, b.t, c->qgrid, c->d_qbox);
  ALLOC(c->d_edges, 3 * c->nt);
  k_tri_edges<<<nblk(c->nt, 256), 256, 0, st>>>(c->d_tverts, c->nt, c->d_edges);
  LAUNCHCHK();
  c->nbr_eps = eps;
  return ICPX_OK;
}

// Neighbour lists for the shortcut, built once per target: the radius is a quarter of the
// target's typical edge length, halved up to three times if that makes the lists long.
static int ensure_lists(icpx_ctx* c)
{
  if (c->nbr_state != 0)
    return ICPX_OK;
  const double floor_eps = 256.0 * 2.3841858e-07 * (double)c->scene_max;
  double eps = std::fmax(0.25 * (double)c->mesh_edge, floor_eps);
  int rc = ensure_bin(c, 1);
  if (rc)
    return rc;
  for (int attempt = 0; attempt < 4; attempt++)
  {
    rc = build_nbr(c, c->bins[0], (float)eps);
    if (rc)
      return rc;
    if (c->nbr_eps > 0.f || eps <= floor_eps)
      break;
    eps = std::fmax(0.5 * eps, floor_eps);
  }
  c->nbr_state = c->nbr_eps > 0.f ? 1 : -1;
  if (c->nbr_state == 1)
    (void)ensure_heap(c, 64); // for short lists of leftovers (k_fallback); optional
  return ICPX_OK;
}

// Anchored passes (cert.hpp): the radius about the origin that holds every source point, and
// how far a point may be from where the certificates' transform has it for a pass to leave
// the certificates as they are: one rounding allowance of the scene (a certificate made
// meanwhile loses twice the query's own distance from there, which a settled registration
// keeps a hundred times smaller).
static float anchor_rmax(const icpx_ctx* c)
{
  return 1.7320509f * c->src_max * 1.000001f;
}
static float anchor_reach(const icpx_ctx* c)
{
  return (c->use_anchor && c->src_max >= 0.f) ? std::fmax(c->scene_max, c->src_max) * 2.3841858e-07f : 0.f;
}

// A cloud large enough for its certificates to be checked by k_certify; its passes are long
// enough, too, for the host to stay closer behind the device (run_catch_up).
// The host's policy goes by sizes every rank of a job agrees on (in a loop of several ranks
// all of them must enqueue the same sequence of passes): a shard's size is taken as the
// job's kept points over its ranks, whichever rank got the odd one.
static int64_t policy_nq(const icpx_ctx* c)
{
  return c->world > 1 ? c->nk / c->world : c->nq;
}
// How many queries there are to the mesh's triangles where this rank searches: shards cut
// from the curve cover one stretch of the surface each at the whole cloud's density, blocks
// of kept indices are spread over all of it at a fraction.
// Triangles per point up to which a whole pass goes one walk per wavefront: measured on cold
// passes, the walk against the pooled lanes, 40 000 / 200 000 points on a million triangles
// 1.40 / 0.70 against 2.56 / 2.73 ms, a million / 100 000 on ten million 3.9 / 5.1 against
// 12.5 / 6.9 ms, 40 000 on ten million 6.3 against 7.4 ms (the pooled lanes' cold pass over ten
// million triangles takes 7 ms however few the queries: one lane's longest walk); sparser
// still, the 64 queries of a batch share nothing and their walks run one after the other in
// one wavefront.  (The rule was 4 until it was measured.)
constexpr int64_t WALK_DENSITY = 256;
static int64_t density_nq(const icpx_ctx* c)
{
  return c->curve_shard ? c->nk : policy_nq(c);
}
static bool large_cloud(const icpx_ctx* c)
{
  return c->certify_min >= 0 && policy_nq(c) >= c->certify_min;
}
// Whether the leads of sets' winners are kept (cert.hpp): by the cloud's size itself, whatever
// kernel checks its certificates.
static bool leads_kept(const icpx_ctx* c)
{
  return policy_nq(c) < CERTIFY_MIN;
}
// A cloud whose passes take long enough for the host to stay one pass behind the device in
// the first passes of a run, so that it knows how the first solve went before it picks the
// third pass's kernels.
constexpr int64_t CLOSE_FOLLOW_MIN = 262144;
// From which spread of its triangles' sizes (mesh_grading) a target's whole passes of the
// wavefront walk may give up on batches.  The benchmark's tori come to 2.7 and 4; a torus
// whose grid lines are drawn together on one side (edge lengths over a range of 65) to 40.
constexpr float GRADED_MESH = 8.f;
// A cloud that gets the large cloud's policy (certificates checked by k_certify, its list
// straight to the tree kernels, pair certificates, no certificates from the third pass): a
// large one, and one of some size whose points stay beyond the neighbour lists' reach (a noisy
// scan: the shortcut settles none of them, whatever the cloud's size; an eighth of sixteen
// million noisy points took 35.8 ms per registration with the small cloud's policy and 23.3
// with this one).  `noisy` follows the records, so a clean cloud that merely starts far from
// its place changes over when it arrives.
// (... and a cloud of that size whose target admits no neighbour lists, nbr_state < 0: one
// whose coordinates are a hundred times its extent has rounding allowances the size of the
// lists' radius; without the shortcut its settled passes were whole walks, 0.22 ms for a
// million queries that all stand on their certificates, against 0.04 this way.  The same on
// every rank of a job: the lists are a matter of the target alone.)
static bool heavy_cloud(const icpx_ctx* c)
{
  return large_cloud(c) ||
    (c->noisy_policy && (c->noisy || c->nbr_state < 0) && c->certify_min >= 0 &&
      policy_nq(c) >= CLOSE_FOLLOW_MIN);
}

// A finished pass's figures, for the shortcut's policy (see prepare_local).
static void note_pass(icpx_ctx* c, double n, double mean, double fallback, double listed,
  bool was_local, bool was_listed, double nq_ref)
{
  // The shortcut settles a query within half the lists' radius of its triangle: with the
  // mean residual at twice the radius it settles next to none (a noisy scan, or a cloud
  // still far from its place), whichever kernels the pass ran.
  if (n > 0.0)
    c->noisy = mean > 2.0 * (c->nbr_eps > 0.f ? (double)c->nbr_eps : 0.25 * (double)c->mesh_edge);
  if (!was_local)
    return; // a pass without the shortcut says nothing else about it
  // The shortcut run over k_certify's list: when it settles less than half of a list of
  // some length twice in a row (a noisy cloud: its points are farther from the surface than
  // the neighbour lists reach, whatever the pose), the list goes straight to the tree
  // kernels from then on, until the next cold start.
  if (was_listed && listed >= 1024.0)
  {
    c->hopeless_streak = fallback >= 0.5 * listed ? c->hopeless_streak + 1 : 0;
    if (c->hopeless_streak >= 2)
      c->direct = true;
  }
  const bool bad = n > 0.0 && fallback > 0.5 * nq_ref && 3.0 * mean > (double)c->nbr_eps;
  c->bad_streak = bad ? c->bad_streak + 1 : 0;
  if (!bad && c->local_holdoff == 0)
    c->probation = false;
  // (with certificates the first passes of the shortcut leave most queries to the tree and
  // later ones only those without a certificate: it gets four passes to show that)
  const int patience = c->use_cert ? 4 : 2;
  if (c->bad_streak >= (c->probation ? 1 : patience) && c->local_holdoff == 0)
  {
    c->local_holdoff = c->holdoff_len;
    c->holdoff_len = c->holdoff_len < 64 ? 2 * c->holdoff_len : 64;
    c->bad_streak = 0;
  }
}

// Whether this pass tries the shortcut first: always once the lists exist (a query it
// cannot settle costs a short walk before it joins the tree walk), except for a few
// passes after two in a row that left most queries to the tree walk while the mean
// residual was beyond the lists' reach (noisy scans never get close enough).  Both
// figures reach the host a pass or two late.
static int prepare_local(icpx_ctx* c, bool* use)
{
  *use = false;
  if (!c->use_local || (c->nbr_state < 0 && !heavy_cloud(c)))
    return ICPX_OK;
  if (!c->hint_valid)
  {
    c->after_cold = true;
    return ICPX_OK; // a cold pass: a walk from the Morton seed settles too few queries to pay
  }
  if (c->after_cold)
  {
    // the pass after a cold one: the first step of a registration is its largest, last
    // pass's winners are far from this pass's, and a whole pass of the tree kernel (which
    // re-seeds such queries) is cheaper than a long walk plus the tree for what is left
    c->after_cold = false;
    if (c->persistent >= 3 && density_nq(c) * WALK_DENSITY >= c->nt)
      return ICPX_OK;
  }
  if (c->stat_pending && hipEventQuery(c->ev_stat) == hipSuccess)
  {
    c->stat_pending = false;
    note_pass(c, c->h_stat[0], c->h_stat[0] > 0.0 ? c->h_stat[1] / c->h_stat[0] : 0.0,
      c->h_stat[2], c->h_stat[3], c->stat_local, c->stat_listed, (double)c->nq);
  }
  if (c->local_holdoff > 0)
  {
    c->probation = true;
    c->local_holdoff--;
    return ICPX_OK;
  }
  // A large cloud that stays beyond the lists' reach does without the shortcut: a pass with
  // certificates to check hands k_certify's list straight to the tree kernels, one without
  // is a whole tree pass (which leaves certificates).  The lists are not even built for it.
  if ((c->noisy || c->nbr_state < 0) && heavy_cloud(c))
  {
    if (!(c->use_cert && c->certs_live))
      return ICPX_OK;
    (void)ensure_heap(c, 64); // for short lists of leftovers (k_fallback); optional
    c->direct = true;
    *use = true;
    return ICPX_OK;
  }
  int rc = ensure_lists(c);
  if (rc)
    return rc;
  *use = c->nbr_state == 1;
  return ICPX_OK;
}

static bool finalize_on_matrix_core(const icpx_ctx* c, int64_t nq);
static void fill_final(icpx_ctx* c, FinalArgs& f, double max_dist, double min_dot, int use_sn,
  bool want_sn);
static void fill_lean(icpx_ctx* c, FinalArgs& f);

// Whether this run's settled passes may be lean ones (see icpx_ctx::use_lean): the fused loop
// of one rank, a cloud whose certificates the shortcut itself checks, the default kernels.
static bool lean_capable(const icpx_ctx* c)
{
  if (!(c->use_lean && c->run_active && (c->run_fused || c->run_shared) && !large_cloud(c) &&
        c->use_local == 1 && c->use_cert && c->persistent == 4 && !c->exhaustive))
    return false;
  // (several ranks: all must answer alike, so the size class goes by policy_nq, and the few
  // queries more or fewer of some rank must not change it)
  // (a shard's size is within a chunk of the curve order, 8192 points, of policy_nq)
  if (c->run_shared)
    return c->group_width <= 1 && policy_nq(c) > 32768 + 8192 &&
      finalize_on_matrix_core(c, policy_nq(c) + 8192);
  return pick_width(c) == 1 && finalize_on_matrix_core(c, c->nq);
}
// Whether the pass about to be enqueued is tried as a lean one: the last pass left every
// certificate up to date, and the last record looked at was of a pass that left next to
// nothing to the tree kernels (use_lean >= 2, for tests: whatever the record said, so that
// lean passes fail and are repeated while the cloud still moves).
static bool lean_wanted(const icpx_ctx* c)
{
  return lean_capable(c) && c->lean_base && (c->clean_noted || c->use_lean >= 2) &&
    c->lean_holdoff == 0 && c->hint_valid;
}

static int run_search3(icpx_ctx* c)
{
  const int L = c->leaf_size;
  int rc = ensure_bin(c, L);
  if (rc)
    return rc;
  if (c->nq > c->q3_cap)
  {
    ALLOC(c->d_cand, c->nq * RING3);
    ALLOC(c->d_tie_list, c->nq);
    ALLOC(c->d_counters, PASS_COUNTERS);
    c->counters_clean = false;
    ALLOC(c->d_work, c->nq + CERTIFY_SEGS * 256 * CERTIFY_ROUNDS);
    ALLOC(c->d_ovf_list, c->nq);
    ALLOC(c->d_ovf_best, c->nq);
    ALLOC(c->d_fallback, c->nq);
    ALLOC(c->d_bail, c->nq);
    ALLOC(c->d_cert, c->nq);
    ALLOC(c->d_lead, c->nq);
    ALLOC(c->d_setmask, (c->nq + 63) / 64 + 1);
    c->q3_cap = c->nq;
  }
  const Bin& b = c->bins[(L == 1) ? 0 : (L == 2) ? 1 : 2];
  Search3Args a;
  a.qpos = c->d_qpos;
  a.nq = c->nq;
  a.T = xf_ref(c);
  a.nodes = b.nodes;
  a.tree = b.t;
  a.tverts = c->d_tverts;
  a.tid = c->d_tid;
  a.tpre = c->d_tpre;
  a.lpre = b.lpre;
  a.nt = c->nt;
  a.hint = c->d_hint;
  a.use_hint = c->hint_valid ? 1 : 0;
  a.scene_max = c->scene_max;
  a.tcodes = c->d_tcodes;
  a.tcell = c->d_tcell;
  a.tcell_bits = c->tcell_bits;
  a.box = c->box;
  a.seed_grid = c->seed_mode ? c->d_seed : nullptr;
  a.seed_dim = c->seed_dim;
  a.seed_morton = c->seed_mode == 2 ? 0 : 1;
  a.cand = c->d_cand;
  a.tie_list = c->d_tie_list;
  a.counters = c->d_counters;
  a.ovf_list = c->d_ovf_list;
  a.ovf_best = c->d_ovf_best;
  a.qlist = nullptr;
  a.bail_list = nullptr;
  a.bail_ticks = 0u;
  a.bail_mean = c->bail_mode == 1 ? 1 : 0;
  a.count_slot = 3;
  a.bail_slot = 2;
  a.small_limit = FALLBACK_SMALL;
  a.step_budget = 0u;
  a.defer_far = 0.f;
  a.list_lo = 0;
  a.list_hi = INT64_MAX;
  a.cap2 = c->cap2;
  a.cert = nullptr; // (set below, once the pass's kernels are known)
  a.cert_T = c->d_state->cert_T;
  a.cert_ok = &c->d_state->cert_ok;
  a.cert_cap2 = &c->d_state->cert_cap2;
  a.widen_max = 0.25f * c->mesh_edge;
  // (pairs cost k_contested a few microseconds per pass where they are a thousand of a million:
  // small clouds keep them only where lean passes, which have no k_contested, are to follow)
  // (2: a pair only where a single winner's certificate would be next to nothing; a lean pass
  // lets a single winner through on its certificate alone, a pair it has to weigh)
  a.pairs = heavy_cloud(c) ? 1 : lean_capable(c) ? 2 : 0;
  a.setmask = nullptr; // (set below, once the pass's kernels are known)
  a.refill = c->refill_override > 0 ? c->refill_override : (c->hint_valid ? 40 : REFILL3);
  hipStream_t st = c->stream;
  const unsigned grid = nblk(c->nq, 256);
  bool use_local = false;
  rc = prepare_local(c, &use_local);
  if (rc)
    return rc;
  c->last_local = use_local;
  c->last_listed = false;
  // by regime: a whole pass over a cloud about as dense as the mesh goes one walk per
  // wavefront (neighbouring queries share their walk); what the shortcut left over, or a
  // sparse cloud, is scattered and goes one pooled lane per query
  int mode = c->persistent;
  if (mode == 4)
    mode = (!use_local && density_nq(c) * WALK_DENSITY >= c->nt) ? 3 : 1;
  // the wavefront walk, the pooled lanes and the shortcut keep certificates; the other
  // variants do not, and a pass of theirs leaves the certificates behind
  // (nor does the wavefront walk on a cold pass and the one after it: those moves outrun
  // any certificate; on a large cloud, where making certificates costs a pass milliseconds,
  // nor on the third: measured on 16 M noisy points, the move after it still broke seven
  // eighths of them)
  c->since_cold = c->hint_valid ? c->since_cold + 1 : 0;
  c->pass_certs = c->use_cert && (mode == 1 || (mode == 3 && c->since_cold >= (heavy_cloud(c) ? 3 : 2)));
  a.cert = c->pass_certs ? c->d_cert : nullptr;
  // (not for a cloud of four million points or more: of sixteen million noisy points 6 % stand on sets, and 94 % of those are
  // points nearest to an edge, whose two triangles tie for good; reading the leads cost its
  // passes more than the few clear winners saved: registration 92.6 -> 94.2 ms)
  a.lead = (c->pass_certs && (c->weigh_in_finalize & 2) && leads_kept(c)) ? c->d_lead : nullptr;
  // a kernel that checks every query's certificate runs in this pass (the shortcut or
  // k_certify, or the wavefront walk over the whole cloud): it marks the queries that stand
  // on set certificates for k_contested
  a.setmask = (a.cert && (use_local || mode == 3)) ? c->d_setmask : nullptr;
  ScopedTimer tm(c, 0, true);
  tm.kind = use_local ? 1 : 0;
  tm.hinted = c->hint_valid ? 1 : 0;
  if (!c->counters_clean)
    HIPCHK(hipMemsetAsync(c->d_counters, 0, PASS_COUNTERS * sizeof(uint32_t), st));
  c->counters_clean = false;
  if (use_local)
  {
    LocalArgs la;
    la.s = a;
    la.s.nodes = c->bins[0].nodes; // the lists index the one-triangle-per-leaf tree
    la.s.tree = c->bins[0].t;
    la.noff = c->d_noff;
    la.nbr = c->d_nbr;
    la.qbox = c->d_qbox;
    la.edges = c->d_edges;
    la.grid = c->qgrid;
    la.eps = c->nbr_eps;
    la.fallback = c->d_fallback;
    la.cert_T = c->d_state->cert_T;
    la.cert_ok = &c->d_state->cert_ok;
    la.cert_cap2 = &c->d_state->cert_cap2;
    la.work = nullptr;
    la.work_count = nullptr;
    la.segcap = 0;
    // (lean passes follow the shortcut's own passes only: k_certify_rest reads the neighbour
    // lists, which a cloud that goes by the large cloud's policy may never have built)
    la.pairs_to_tree = (lean_capable(c) && !heavy_cloud(c) && a.cert && c->heaps[4].built &&
                         c->nbr_state == 1 && c->d_noff && c->d_nbr) ? 1 : 0;
    c->lean_pairs_pass = la.pairs_to_tree != 0;
    unsigned lgrid = grid;
    bool listed = false;
    // a large cloud with certificates from the last pass: those are checked by a light
    // kernel of their own, and the shortcut serves the ones that failed
    if (a.cert && c->certs_live && a.use_hint && heavy_cloud(c))
    {
      CertifyArgs ca;
      ca.qpos = c->d_qpos;
      ca.nq = c->nq;
      ca.cert = c->d_cert;
      ca.lead = a.lead;
      ca.T = a.T.T;
      ca.stop = a.T.stop;
      ca.cert_T = la.cert_T;
      ca.cert_ok = la.cert_ok;
      ca.cert_cap2 = la.cert_cap2;
      ca.cap2 = c->cap2;
      ca.work = c->d_work;
      ca.work_count = c->d_counters + 16;
      // (a few thousand blocks, each appending once to its segment)
      ca.rounds = (int)std::min<int64_t>(std::max<int64_t>(c->nq / (256 * 2048), 1), CERTIFY_ROUNDS);
      const unsigned cgrid = nblk(c->nq, 256 * ca.rounds);
      ca.segcap = (int64_t)((cgrid + CERTIFY_SEGS - 1) / CERTIFY_SEGS) * 256 * ca.rounds;
      ca.setmask = c->d_setmask;
      listed = !c->direct;
      ca.direct = listed ? nullptr : c->d_fallback;
      ca.direct_count = c->d_counters + 3;
      launch_ev(c, false, k_certify, cgrid, 256, st, ca);
      c->certify_passes++;
      la.work = c->d_work;
      la.work_count = c->d_counters + 16;
      la.segcap = ca.segcap;
      // (a multiple of the number of segments; four resident rounds of blocks at most)
      lgrid = (lgrid + CERTIFY_SEGS - 1) / CERTIFY_SEGS * CERTIFY_SEGS;
      const unsigned cap = (unsigned)c->num_cus * 16u / CERTIFY_SEGS * CERTIFY_SEGS;
      if (lgrid > cap)
        lgrid = cap;
    }
    if (la.work && !listed)
      c->direct_passes++; // (the list is the tree kernels' as it is, winners of the last pass as hints)
    else if (la.work)
      launch_ev(c, false, k_search_local<true>, lgrid, 256, st, la);
    else
      launch_ev(c, false, k_search_local<false>, lgrid, 256, st, la);
    a.qlist = c->d_fallback; // the tree kernel serves what is left,
    a.use_hint = 1;          // starting from the triangles the shortcut reached
    c->last_listed = listed;
    c->local_passes++;
  }
  else
    c->tree_passes++;
  // persistent lanes: as many wavefronts as stay resident, each drawing queries from a pool
  unsigned pgrid = (unsigned)c->num_cus * 5u;
  if (pgrid > grid)
    pgrid = grid;
  // one walk per wavefront: single-wavefront blocks, a resident set of them
  unsigned wgrid = nblk(c->nq, 64);
  if (wgrid > (unsigned)c->num_cus * (unsigned)c->wave_blocks)
    wgrid = (unsigned)c->num_cus * (unsigned)c->wave_blocks;
  // the shortcut's leftovers: a short list goes one wavefront per query over the 64-ary tree
  Search2Args wide;
  const Search2Args* widep = nullptr;
  // ... and so do the queries of the batches a whole pass of the wavefront walk gives up on:
  // a batch may walk for half of what a wavefront's share of the pass comes to at a quarter
  // of a millisecond per batch (a longer one is what the pass ends up waiting for), 0.8 ms
  // at least (the longest batch of an evenly meshed million triangles takes about 1 ms)
  // (only for a target whose triangles' sizes range widely, or when asked for: the kernel
  // that can give up is a build of its own, 3 % slower on every walk)
  const bool graded = c->bail_mode > 1 || (c->bail_mode == 1 && c->mesh_grading > GRADED_MESH);
  // (... and by a walk over a long list of a heavy cloud's leftovers, the other kernel that
  // runs one walk per wavefront)
  const bool whole_walk = graded && ((!use_local && mode == 3) || (use_local && mode == 1 && heavy_cloud(c)));
  if (graded)
    a.small_limit = 2u * FALLBACK_SMALL;
  if (whole_walk && ensure_heap(c, 64) == ICPX_OK)
  {
    const uint64_t per_wave = ((uint64_t)c->nq / 64u) / std::max(wgrid, 1u);
    a.bail_list = c->d_bail;
    const uint64_t us = c->bail_mode > 1 ? (uint64_t)c->bail_mode
                                         : std::min<uint64_t>(std::max<uint64_t>(125u * per_wave, 800u), 1000000u);
    a.bail_ticks = (uint32_t)(us * 2400u / 64u); // (the shader clock at its 2.4 GHz: a lower clock makes the budget longer)
    c->bail_passes++;
  }
  // the pooled lanes give up on a query whose walk goes on and on (Search3Args::step_budget),
  // given the tree that the kernel behind them serves such queries over
  if (mode == 1 && c->step_budget > 0 && ensure_heap(c, 64) == ICPX_OK)
  {
    a.step_budget = (uint32_t)c->step_budget;
    a.bail_list = c->d_bail;
  }
  // ... and the wavefront walk sets aside the queries that are far out where their batch is
  // not (Search3Args::defer_far: beyond that many of the mesh's edge lengths)
  if (c->defer_edges > 0 && ensure_heap(c, 64) == ICPX_OK)
  {
    const float reach = (float)c->defer_edges * c->mesh_edge;
    a.defer_far = reach * reach;
    a.bail_list = c->d_bail;
  }
  if ((mode == 1 || a.bail_ticks || a.defer_far > 0.f) && c->heaps[4].built)
  {
    const Heap& h = c->heaps[4];
    wide.qpos = c->d_qpos;
    wide.nq = c->nq;
    wide.T = a.T;
    wide.nodes = h.nodes;
    wide.tree = h.t;
    wide.tverts = c->d_tverts;
    wide.tid = c->d_tid;
    wide.nt = c->nt;
    wide.hint = c->d_hint;
    wide.use_hint = 1;
    wide.scene_max = c->scene_max;
    wide.cap2 = c->cap2;
    wide.cert = a.cert;
    wide.cand = c->d_cand;
    wide.widen = a.cert ? a.widen_max : 0.f;
    wide.pairs = a.pairs;
    wide.cert_ok = a.cert ? &c->d_state->cert_ok : nullptr;
    wide.cert_T = c->d_state->cert_T;
    widep = &wide;
  }
  if (L == 1)
    launch_search3<1>(c, a, grid, pgrid, wgrid, mode, widep, st);
  else if (L == 2)
    launch_search3<2>(c, a, grid, pgrid, wgrid, mode, widep, st);
  else
    launch_search3<4>(c, a, grid, pgrid, wgrid, mode, widep, st);
  if (use_local && c->lean_pairs_pass)
  {
    // certificates for what the search kernels left without one (lean.hpp)
    FinalArgs f;
    fill_final(c, f, 0.0, 0.0, 0, false);
    fill_lean(c, f);
    k_certify_rest<<<grid, 256, 0, st>>>(f);
  }
  LAUNCHCHK();
  c->hint_valid = true;
  // (what a lean pass can follow: the shortcut checked every certificate and whatever was
  // searched again, here or by the tree kernels, got a new one)
  // (a heavy cloud's pass, k_certify with its list straight to the tree kernels, as well: the
  // lean kernel reads certificates, sets and winners, never the neighbour lists, which such a
  // cloud may not have; what it cannot have is k_certify_rest's help, so a query that no search
  // kernel can certify fails every lean pass there, and the back-off keeps those few)
  c->lean_base = use_local && !c->last_listed && c->pass_certs && mode == 1 && widep != nullptr &&
    (heavy_cloud(c) ? c->direct : (c->nbr_state == 1 && c->d_noff && c->d_nbr));
  return ICPX_OK;
}


template <int G>
static void launch_search2(const Search2Args& a, unsigned grid, hipStream_t st)
{
  k_search2<G><<<grid, 256, 0, st>>>(a);
}

// The distance beyond which a pass's searches need not look: the filter's threshold, a hair
// wider (a correspondence at the threshold itself is dropped anyway).
static void set_cap(icpx_ctx* c, double max_dist)
{
  c->cap2 = (max_dist > 0.0 && max_dist < 1e100) ? max_dist * max_dist * (1.0 + 1e-12) : DBL_MAX;
}

// The fused finalize: which build (see query.hpp) and how many blocks.  Up to four million
// rows every lane of a resident grid has at most a few to do and the matrix-core build is the
// faster one; beyond, the register build.
constexpr int64_t FINALIZE_MX_MAX = 4000000;
static bool finalize_on_matrix_core(const icpx_ctx* c, int64_t nq)
{
  return c->finalize_build == 0 ? nq <= FINALIZE_MX_MAX : c->finalize_build == 2;
}
static int finalize_grid(const icpx_ctx* c, int64_t nq)
{
  const int per_cu = finalize_on_matrix_core(c, nq) ? 4 : 3; // wavefronts per SIMD = resident blocks per CU
  int grid = (int)nblk(nq, FIN_THREADS);
  if (grid > c->num_cus * per_cu)
    grid = c->num_cus * per_cu;
  if (grid > c->max_blocks)
    grid = c->max_blocks;
  return grid;
}
static void launch_finalize_sums(const icpx_ctx* c, int metric, const FinalArgs& f, int64_t nq,
  int grid, hipStream_t st)
{
  if (finalize_on_matrix_core(c, nq))
  {
    if (metric == 0)
      k_finalize_mx<0, false><<<grid, FIN_THREADS, 0, st>>>(f);
    else if (metric == 1)
      k_finalize_mx<1, false><<<grid, FIN_THREADS, 0, st>>>(f);
    else
      k_finalize_mx<2, false><<<grid, FIN_THREADS, 0, st>>>(f);
  }
  else if (metric == 0)
    k_finalize<0, false><<<grid, FIN_THREADS, 0, st>>>(f);
  else if (metric == 1)
    k_finalize<1, false><<<grid, FIN_THREADS, 0, st>>>(f);
  else
    k_finalize<2, false><<<grid, FIN_THREADS, 0, st>>>(f);
}

static int run_search(icpx_ctx* c)
{
  c->last_local = false;
  c->last_listed = false;
  c->last_split = false;
  c->lean_base = false;
  c->lean_pairs_pass = false;
  c->pass_marked = false;
  c->certs_live = c->pass_certs && c->hint_valid; // (what the last pass left)
  c->pass_certs = false;
  const int G = pick_width(c);
  if (G == 1)
    return run_search3(c);
  int slot = 0;
  for (int i = 0; i < 5; i++)
    if (kWidths[i] == G)
      slot = i;
  const unsigned grid = nblk(c->nq, 256 / G);
  hipStream_t st = c->stream;
  if (!c->exhaustive)
  {
    int rc = ensure_heap(c, G);
    if (rc)
      return rc;
    const Heap& h = c->heaps[slot];
    Search2Args a;
    a.qpos = c->d_qpos;
    a.nq = c->nq;
    a.T = xf_ref(c);
    a.nodes = h.nodes;
    a.tree = h.t;
    a.tverts = c->d_tverts;
    a.tid = c->d_tid;
    a.nt = c->nt;
    a.hint = c->d_hint;
    a.use_hint = c->hint_valid ? 1 : 0;
    a.scene_max = c->scene_max;
    a.cap2 = c->cap2;
    a.cert = nullptr;
    a.cand = nullptr;
    a.widen = 0.f;
    a.pairs = 0;
    a.cert_ok = nullptr;
    a.cert_T = nullptr;
    ScopedTimer tm(c, 0);
    switch (G)
    {
      case 4:
        launch_search2<4>(a, grid, st);
        break;
      case 8:
        launch_search2<8>(a, grid, st);
        break;
      case 16:
        launch_search2<16>(a, grid, st);
        break;
      case 32:
        launch_search2<32>(a, grid, st);
        break;
      default:
        launch_search2<64>(a, grid, st);
        break;
    }
    LAUNCHCHK();
    c->hint_valid = true;
    return ICPX_OK;
  }
  int rc = ensure_tree(c, G);
  if (rc)
    return rc;
  const Tree& t = c->trees[slot];
  SearchArgs a;
  a.qpos = c->d_qpos;
  a.nq = c->nq;
  a.T = xf_ref(c);
  a.nodes = t.nodes;
  a.lv = t.lv;
  a.tverts = c->d_tverts;
  a.tid = c->d_tid;
  a.nt = c->nt;
  a.hint = c->d_hint;
  a.use_hint = c->hint_valid ? 1 : 0;
  ScopedTimer tm(c, 0);
  switch (G)
  {
    case 4:
      k_search<4><<<grid, 256, 0, st>>>(a);
      break;
    case 8:
      k_search<8><<<grid, 256, 0, st>>>(a);
      break;
    case 16:
      k_search<16><<<grid, 256, 0, st>>>(a);
      break;
    case 32:
      k_search<32><<<grid, 256, 0, st>>>(a);
      break;
    default:
      k_search<64><<<grid, 256, 0, st>>>(a);
      break;
  }
  LAUNCHCHK();
  c->hint_valid = true;
  return ICPX_OK;
}

static void fill_final(icpx_ctx* c, FinalArgs& f, double max_dist,
  double min_dot, int use_sn, bool want_sn)
{
  std::memset(&f, 0, sizeof(f));
  f.qpos = c->d_qpos;
  f.qsn = want_sn ? c->d_qsn : nullptr;
  f.qk = c->d_qk;
  f.nq = c->nq;
  f.T = xf_ref(c);
  for (int i = 0; i < 3; i++)
    f.origin[i] = c->origin[i];
  f.tverts = c->d_tverts;
  f.tid = c->d_tid;
  f.tri = c->d_tri;
  f.fnorm = c->d_fn;
  f.fnpos = c->d_fnpos;
  f.enorm = c->d_en;
  f.vnorm = c->d_vn;
  f.hint = c->d_hint;
  f.max_dist = max_dist;
  f.min_dot = min_dot;
  f.use_sn = use_sn;
  f.rotate_sn = use_sn == 2 ? 1 : 0;
  f.partials = c->d_partials;
}

static int check_call(icpx_ctx* c, const double* T, int metric, int use_sn)
{
  if (!c || !T)
    return fail(ICPX_E_INVALID, "null argument");
  if (metric < 0 || metric > 2)
    return fail(ICPX_E_INVALID, "unknown metric %d", metric);
  if (use_sn < 0 || use_sn > 2)
    return fail(ICPX_E_INVALID, "use_source_normals must be 0, 1 or 2 (got %d)", use_sn);
  if ((metric == ICPX_METRIC_PLANE_TO_PLANE || use_sn) && !c->d_srcn)
    return fail(ICPX_E_INVALID, "Plane to plane metric requires source normals");
  for (int i = 0; i < 16; i++)
    if (!std::isfinite(T[i]))
      return fail(ICPX_E_INVALID, "transform is not finite");
  if (!icpx_core::transform_usable(T))
    return fail(ICPX_E_INVALID, "transform beyond 1e6 in its 3x3 or 1e18 in its translation");
  return ICPX_OK;
}

static int run_catch_up(icpx_ctx* c, int lag_override = -1);

// What a lean pass and k_certify_rest need beyond the finalize's arguments.
static void fill_lean(icpx_ctx* c, FinalArgs& f)
{
  f.cert = c->d_cert;
  f.lead = ((c->weigh_in_finalize & 2) && leads_kept(c)) ? c->d_lead : nullptr;
  f.cert_T = c->d_state->cert_T;
  f.cert_ok = &c->d_state->cert_ok;
  f.cert_cap2 = &c->d_state->cert_cap2;
  f.cand = c->d_cand;
  f.cand_rw = c->d_cand;
  f.hint_rw = c->d_hint;
  f.cap2 = c->cap2;
  f.lean_fail = &c->d_state->lean_fail;
  f.noff = c->d_noff;
  f.nbr = c->d_nbr;
  f.nbr_cap = NBR_CAP;
  f.eps = c->nbr_eps;
  f.scene_max = c->scene_max;
}

// A lean pass: the certificate check and the fused sums in one kernel, then the reduction,
// with the solve on the same launch (one rank's fused loop) or left to icpx_run_update behind
// the caller's all-reduce (several ranks: the reduction puts the pass's flag into the sums'
// spare slot, so that a pass that failed on one rank is dropped by all).  Nothing on the host
// changes but the run's position: the shortcut's policy, the pass counters and the
// certificates' standing are those of the last full pass.
static int enqueue_lean(icpx_ctx* c, int metric, double max_dist, double min_dot, int use_sn,
  double* dst, bool fused)
{
  hipStream_t st = c->stream;
  c->last_lean = true;
  if (!fused && !(c->nq > 0 && c->d_cand && c->d_cert && c->d_counters && pick_width(c) == 1 &&
                  finalize_on_matrix_core(c, c->nq)))
  {
    // (every rank of a job decides alike from the same records; a rank that finds itself
    // without what a lean pass reads says so, and the pass is repeated in full by all)
    k_lean_unready<<<1, 64, 0, st>>>(dst, &c->d_state->done, c->nq);
    LAUNCHCHK();
    return ICPX_OK;
  }
  FinalArgs f;
  fill_final(c, f, max_dist, min_dot, use_sn, use_sn || metric == ICPX_METRIC_PLANE_TO_PLANE);
  fill_lean(c, f);
  // (tests: every third lean pass is made to fail, on the even ranks of a job only, so that
  // the odd ones learn of it through the all-reduce)
  f.lean_inject = (c->use_lean == 3 && c->lean_passes % 3 == 1 && c->rank % 2 == 0) ? 1 : 0;
  const int grid = finalize_grid(c, c->nq);
  {
    // (timed as a search pass of kind 2: the one kernel is the pass's first and last)
    ScopedTimer tm(c, 0, true);
    tm.kind = 2;
    tm.hinted = 1;
    if (metric == 0)
      launch_ev(c, true, k_finalize_mx<0, false, true>, (unsigned)grid, FIN_THREADS, st, f);
    else if (metric == 1)
      launch_ev(c, true, k_finalize_mx<1, false, true>, (unsigned)grid, FIN_THREADS, st, f);
    else
      launch_ev(c, true, k_finalize_mx<2, false, true>, (unsigned)grid, FIN_THREADS, st, f);
  }
  if (!fused)
  {
    // (the certificates and the transform they refer to stay as they are: no cert_ok)
    k_reduce_partials<<<ICPX_NSUMS, 64, 0, st>>>(c->d_partials, grid, dst, &c->d_state->done,
      c->d_counters, nullptr, c->d_state->T, c->d_state->cert_T, nullptr, 1, c->cap2,
      &c->d_state->cert_cap2, 1, c->nq, &c->d_state->lean_fail);
    LAUNCHCHK();
    return ICPX_OK;
  }
  const icpx_params& p = c->run_p;
  const int slot = (int)(c->run_it & 3);
  const uint32_t seq = ++c->rec_seq;
#define ICPX_REDUCE_SOLVE(M)                                                                     \
  k_reduce_solve<M><<<ICPX_NSUMS, 64, 0, st>>>(c->d_partials, grid, dst, c->d_counters,          \
    c->d_state, 1, c->origin[0], c->origin[1], c->origin[2], p.check_mean_distance,              \
    p.max_mean_distance, c->d_rec + slot, seq, c->d_ticket, c->cap2, 1, anchor_rmax(c),         \
    anchor_reach(c))
  if (p.metric == 0)
    ICPX_REDUCE_SOLVE(0);
  else if (p.metric == 1)
    ICPX_REDUCE_SOLVE(1);
  else
    ICPX_REDUCE_SOLVE(2);
#undef ICPX_REDUCE_SOLVE
  LAUNCHCHK();
  c->slot_seq[slot] = c->rec_seq;
  c->pass_local[slot] = false;
  c->pass_listed[slot] = false;
  c->pass_split[slot] = false;
  c->pass_lean[slot] = true;
  c->run_it++;
  c->lean_unnoted++;
  c->lean_passes++;
  return ICPX_OK;
}

// Search + fused sums of one pass, enqueued on the stream; the transform is the one in
// the device state.  `stat` asks for the pass's {n, sum |s-t|} to be copied to the host
// behind an event (for callers that do not run the device-resident loop).
// loop: 0 a pass by itself (icpx_accumulate, icpx_run_enqueue); 1 a pass of one rank's fused
// loop (the solve rides on the reduction's launch); 2 a pass of a rank's share of a
// multi-rank loop (icpx_run_advance: the reduction puts the rank's policy figures into the
// sums' spare slots, the solve comes behind the caller's all-reduce).
static int enqueue_pass(icpx_ctx* c, int metric, int64_t step, double max_dist, double min_dot,
  int use_sn, double* dst, bool stat, int loop = 0)
{
  int rc = prepare_queries(c, step);
  if (rc)
    return rc;
  hipStream_t st = c->stream;
  const bool fuse_solve = loop == 1;
  set_cap(c, max_dist);
  c->last_lean = false;
  if (loop)
  {
    // A lean pass when the run has settled (see icpx_ctx::use_lean); a full pass must not be
    // enqueued behind lean ones whose records are still out (if one of them failed, this
    // pass's bookkeeping on the host would be of a pass that never ran).
    bool lean = lean_wanted(c);
    if (!lean && c->lean_unnoted > 0)
    {
      const int64_t it0 = c->run_it;
      rc = run_catch_up(c, 0);
      if (rc)
        return rc;
      if (c->run_done_known)
        return ICPX_OK;
      if (c->run_it == it0)
        lean = lean_wanted(c);
    }
    if (lean)
      return enqueue_lean(c, metric, max_dist, min_dot, use_sn, dst, fuse_solve);
    if (c->lean_holdoff > 0)
      c->lean_holdoff--;
  }
  if (c->nq == 0)
  {
    HIPCHK(hipMemsetAsync(dst, 0, sizeof(double) * ICPX_NSUMS, st));
    return ICPX_OK;
  }
  c->weigh_later = (c->weigh_in_finalize & 1) != 0 && finalize_on_matrix_core(c, c->nq);
  rc = run_search(c);
  c->weigh_later = false;
  if (rc)
    return rc;
  FinalArgs f;
  fill_final(c, f, max_dist, min_dot, use_sn, use_sn || metric == ICPX_METRIC_PLANE_TO_PLANE);
  if ((c->weigh_in_finalize & 1) && c->pass_marked && finalize_on_matrix_core(c, c->nq))
  {
    f.setmask = reinterpret_cast<const unsigned long long*>(c->d_setmask);
    f.cand = c->d_cand;
    f.hint_rw = c->d_hint;
    f.scene_max = c->scene_max;
    f.cert = c->d_cert;
    f.lead = ((c->weigh_in_finalize & 2) && leads_kept(c)) ? c->d_lead : nullptr;
    f.cert_ok = &c->d_state->cert_ok;
    f.cert_T = c->d_state->cert_T;
  }
  const bool want_stat = stat && !c->stat_pending; // (what the shortcut's policy looks at)
  // one resident round of the build that suits the cloud's size
  int grid = finalize_grid(c, c->nq);
  {
    ScopedTimer tm(c, 1);
    launch_finalize_sums(c, metric, f, c->nq, grid, st);
  }
  const int cert_mode = c->pass_certs ? 1 : 0;
  if (fuse_solve)
  {
    // (a run's pass on one rank: the solve rides on the reduction's launch)
    const icpx_params& p = c->run_p;
    const int slot = (int)(c->run_it & 3);
    const uint32_t seq = ++c->rec_seq;
#define ICPX_REDUCE_SOLVE(M)                                                                     \
  k_reduce_solve<M><<<ICPX_NSUMS, 64, 0, st>>>(c->d_partials, grid, dst, c->d_counters,          \
    c->d_state, cert_mode, c->origin[0], c->origin[1], c->origin[2], p.check_mean_distance,      \
    p.max_mean_distance, c->d_rec + slot, seq, c->d_ticket, c->cap2, 0, anchor_rmax(c),         \
    anchor_reach(c))
    if (p.metric == 0)
      ICPX_REDUCE_SOLVE(0);
    else if (p.metric == 1)
      ICPX_REDUCE_SOLVE(1);
    else
      ICPX_REDUCE_SOLVE(2);
#undef ICPX_REDUCE_SOLVE
    c->slot_seq[slot] = c->rec_seq;
    c->pass_local[slot] = c->last_local;
    c->pass_listed[slot] = c->last_listed;
    c->pass_split[slot] = c->last_split;
    c->pass_lean[slot] = false;
    c->run_it++;
  }
  else
    k_reduce_partials<<<ICPX_NSUMS, 64, 0, st>>>(c->d_partials, grid, dst, &c->d_state->done,
      c->d_counters, want_stat ? c->d_stat : nullptr, c->d_state->T, c->d_state->cert_T,
      &c->d_state->cert_ok, cert_mode, c->cap2, &c->d_state->cert_cap2, loop == 2 ? 1 : 0, c->nq,
      nullptr);
  LAUNCHCHK();
  if (c->d_counters)
    c->counters_clean = true;
  if (want_stat)
  {
    HIPCHK(hipEventRecord(c->ev_stat, st));
    c->stat_pending = true;
    c->stat_local = c->last_local;
    c->stat_listed = c->last_listed;
  }
  return ICPX_OK;
}

extern "C" int icpx_accumulate(icpx_ctx* c, int metric, const double T[16], int64_t step,
  double max_dist, double min_dot, int use_sn, double* sums_host, double* sums_device)
{
  int rc = check_call(c, T, metric, use_sn);
  if (rc)
    return rc;
  HIPCHK(hipSetDevice(c->device));
  c->run_active = false;
  rc = set_transform(c, T, 0);
  if (rc)
    return rc;
  hipStream_t st = c->stream;
  double* dst = sums_device ? sums_device : c->d_sums;
  rc = enqueue_pass(c, metric, step, max_dist, min_dot, use_sn, dst, true);
  if (rc)
    return rc;
  if (sums_host)
  {
    HIPCHK(hipMemcpyAsync(
      c->h_sums, dst, sizeof(double) * ICPX_NSUMS, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::memcpy(sums_host, c->h_sums, sizeof(double) * ICPX_NSUMS);
  }
  return ICPX_OK;
}

// icpx_accumulate for callers that cannot allocate.  Nothing here calls hipMalloc, hipFree or
// a routine that does: the kept points are ordered, searched and summed in the static array
// above, the meshes are the context's, the sums go to the caller's device buffer.
extern "C" int icpx_accumulate_static(icpx_ctx* c, int metric, const double T[16], int64_t step,
  double max_dist, double min_dot, int use_sn, double* sums_device)
{
  int rc = check_call(c, T, metric, use_sn);
  if (rc)
    return rc;
  if (!sums_device)
    return fail(ICPX_E_INVALID, "icpx_accumulate_static: sums_device is null");
  if (c->nt == 0)
    return fail(ICPX_E_INVALID, "no target set");
  if (c->np == 0)
    return fail(ICPX_E_INVALID, "no source set");
  if (step < 1)
    return fail(ICPX_E_INVALID, "step must be >= 1");
  const int64_t nk = (c->np + step - 1) / step;
  int64_t k0 = 0, nq = 0;
  ShardMap map;
  const bool curve = shard_extent(c, nk, &k0, &nq, &map);
  if (nq > STATIC_MAX_QUERIES)
    return fail(ICPX_E_INVALID,
      "icpx_accumulate_static: %lld kept source points, the static scratch holds %lld at most "
      "(use icpx_accumulate, or a larger step)",
      (long long)nq, (long long)STATIC_MAX_QUERIES);
  // (shards cut from the curve order: all kept points of the job are ordered here, in the
  // same scratch)
  if (curve && nk > STATIC_MAX_QUERIES)
    return fail(ICPX_E_INVALID,
      "icpx_accumulate_static: %lld kept source points in the job, the static scratch orders "
      "%lld at most (use icpx_accumulate, or a larger step)",
      (long long)nk, (long long)STATIC_MAX_QUERIES);
  if (rs_scratch_words(curve ? nk : nq) > (int64_t)ST_SORT_WORDS)
    return fail(ICPX_E_INVALID, "icpx_accumulate_static: sort scratch too small");
  unsigned char* base = nullptr;
  if (c->device < ST_MAX_DEVICES)
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    base = g_scratch_ptr[c->device];
  }
  if (!base)
    return fail(ICPX_E_INVALID, "icpx_accumulate_static: no scratch address for device %d",
      c->device);
  // The search tree of this size class is built (allocated) by the first ordinary pass: the
  // one icpx_accumulate picks for this many queries, which goes by the development switches
  // (icpx_set_group_width, icpx_set_exhaustive) when they are set.  This entry point runs the
  // wavefront-per-query or the eight-lanes-per-query search over its heap tree, nothing else.
  int G = nq <= 4096 ? 64 : 8;
  if (c->group_width == 8 || c->group_width == 64)
    G = c->group_width;
  const int slot = G == 64 ? 4 : 1;
  const Heap& h = c->heaps[slot];
  if (nq > 0 && !h.built)
    return fail(ICPX_E_INVALID,
      "icpx_accumulate_static: the %d-ary search tree is not built yet; one icpx_accumulate call "
      "with the same step and the default kernels (icpx_set_group_width 0 or %d, "
      "icpx_set_exhaustive 0) builds it (that call allocates)", G, G);
  HIPCHK(hipSetDevice(c->device));
  c->run_active = false;
  hipStream_t st = c->stream;
  if (nq == 0)
  {
    HIPCHK(hipMemsetAsync(sums_device, 0, sizeof(double) * ICPX_NSUMS, st));
    return ICPX_OK;
  }
  uint64_t* ka = reinterpret_cast<uint64_t*>(base + ST_OFF_KA);
  uint64_t* kb = reinterpret_cast<uint64_t*>(base + ST_OFF_KB);
  uint32_t* va = reinterpret_cast<uint32_t*>(base + ST_OFF_VA);
  uint32_t* vb = reinterpret_cast<uint32_t*>(base + ST_OFF_VB);
  uint32_t* sortscr = reinterpret_cast<uint32_t*>(base + ST_OFF_SORT);
  float* qpos = reinterpret_cast<float*>(base + ST_OFF_QPOS);
  float* qsn = reinterpret_cast<float*>(base + ST_OFF_QSN);
  int32_t* qk = reinterpret_cast<int32_t*>(base + ST_OFF_QK);
  int32_t* hint = reinterpret_cast<int32_t*>(base + ST_OFF_HINT);
  double* partials = reinterpret_cast<double*>(base + ST_OFF_PART);

  rc = set_transform(c, T, 0);
  if (rc)
    return rc;
  // the array keeps the last call's contents: everything read below is written first (the
  // sort clears its own histograms, k_gather_queries writes every query's slot, hints start
  // at -1, every finalize block writes its partial)
  if (curve)
  {
    k_morton_points<<<nblk(nk, 256), 256, 0, st>>>(c->d_src, 0, step, nk, c->box, ka, va, c->query_curve);
    const int which = radix_sort_pairs(ka, va, kb, vb, nk, 63, sortscr, st);
    HIPCHK(hipMemsetAsync(qk, 0, sizeof(int32_t) * nq, st)); // (slots matter to rows only)
    k_gather_queries_curve<<<nblk(nq, 256), 256, 0, st>>>(
      c->d_src, c->d_srcn, (which ? vb : va) + k0, map, step, nq, qpos, qsn, hint);
  }
  else
  {
    k_morton_points<<<nblk(nq, 256), 256, 0, st>>>(c->d_src, k0, step, nq, c->box, ka, va, c->query_curve);
    const int which = radix_sort_pairs(ka, va, kb, vb, nq, 63, sortscr, st);
    k_gather_queries<<<nblk(nq, 256), 256, 0, st>>>(
      c->d_src, c->d_srcn, which ? vb : va, k0, step, nq, qpos, qsn, qk, hint);
  }
  set_cap(c, max_dist);
  Search2Args a;
  a.qpos = qpos;
  a.nq = nq;
  a.T = xf_ref(c);
  a.nodes = h.nodes;
  a.tree = h.t;
  a.tverts = c->d_tverts;
  a.tid = c->d_tid;
  a.nt = c->nt;
  a.hint = hint;
  a.use_hint = 0;
  a.scene_max = c->scene_max;
  a.cap2 = c->cap2;
  a.cert = nullptr;
  a.cand = nullptr;
  a.widen = 0.f;
  a.pairs = 0;
  a.cert_ok = nullptr;
  a.cert_T = nullptr;
  if (G == 64)
    launch_search2<64>(a, nblk(nq, 256 / 64), st);
  else
    launch_search2<8>(a, nblk(nq, 256 / 8), st);
  FinalArgs f;
  fill_final(c, f, max_dist, min_dot, use_sn, use_sn || metric == ICPX_METRIC_PLANE_TO_PLANE);
  f.qpos = qpos;
  f.qsn = f.qsn ? qsn : nullptr;
  f.qk = qk;
  f.nq = nq;
  f.hint = hint;
  f.partials = partials;
  int grid = finalize_grid(c, nq); // (the same shape and build as icpx_accumulate's: same sums)
  if (grid > STATIC_MAX_BLOCKS)
    grid = STATIC_MAX_BLOCKS;
  launch_finalize_sums(c, metric, f, nq, grid, st);
  // (no counters, no policy figures, and the context's certificates are left as they are)
  k_reduce_partials<<<ICPX_NSUMS, 64, 0, st>>>(partials, grid, sums_device, &c->d_state->done,
    nullptr, nullptr, c->d_state->T, c->d_state->cert_T, nullptr, 0, c->cap2, nullptr);
  LAUNCHCHK();
  return ICPX_OK;
}

extern "C" int icpx_correspond(icpx_ctx* c, const double T[16], int64_t step, double max_dist,
  double min_dot, int use_sn, float* origin, float* origin_normal, float* closest, float* normal,
  int32_t* cell, double* dist, int64_t* src_index, int64_t* n_out)
{
  int rc = check_call(c, T, 0, use_sn);
  if (rc)
    return rc;
  if (!n_out)
    return fail(ICPX_E_INVALID, "icpx_correspond: n_out is null");
  *n_out = 0;
  HIPCHK(hipSetDevice(c->device));
  rc = prepare_queries(c, step);
  if (rc)
    return rc;
  const int64_t nq = c->nq;
  if (nq == 0)
    return ICPX_OK;
  hipStream_t st = c->stream;
  if (nq > c->rows_cap)
  {
    free_rows(c);
    ALLOC(c->r_origin, 3 * nq);
    ALLOC(c->r_onormal, 3 * nq);
    ALLOC(c->r_closest, 3 * nq);
    ALLOC(c->r_normal, 3 * nq);
    ALLOC(c->r_cell, nq);
    ALLOC(c->r_dist, nq);
    ALLOC(c->r_flag, nq + 1);
    ALLOC(c->r_pos, nq + 1);
    ALLOC(c->r_scan, scan_scratch_words(nq + 1));
    ALLOC(c->c_origin, 3 * nq);
    ALLOC(c->c_onormal, 3 * nq);
    ALLOC(c->c_closest, 3 * nq);
    ALLOC(c->c_normal, 3 * nq);
    ALLOC(c->c_cell, nq);
    ALLOC(c->c_dist, nq);
    ALLOC(c->c_index, nq);
    c->rows_cap = nq;
  }
  c->run_active = false;
  set_cap(c, max_dist);
  rc = set_transform(c, T, 0);
  if (!rc)
    rc = run_search(c);
  if (rc)
    return rc;
  k_cert_mark<<<1, 64, 0, c->stream>>>(c->d_state, c->pass_certs ? 1 : 0, c->cap2);
  FinalArgs f;
  fill_final(c, f, max_dist, min_dot, use_sn, use_sn != 0);
  f.o_origin = c->r_origin;
  f.o_onormal = use_sn ? c->r_onormal : nullptr;
  f.o_closest = c->r_closest;
  f.o_normal = c->r_normal;
  f.o_cell = c->r_cell;
  f.o_dist = c->r_dist;
  f.o_flag = c->r_flag;
  int grid = (int)nblk(nq, FIN_THREADS);
  if (grid > c->max_blocks)
    grid = c->max_blocks;
  HIPCHK(hipMemsetAsync(c->r_flag + nq, 0, sizeof(uint32_t), st));
  k_finalize<0, true><<<grid, FIN_THREADS, 0, st>>>(f);
  exclusive_scan(c->r_flag, c->r_pos, nq + 1, c->r_scan, st);
  CompactArgs ca;
  ca.nq = nq;
  ca.k0 = c->k0;
  ca.kglobal = c->curve_shard ? c->d_kglobal : nullptr;
  ca.step = step;
  ca.flag = c->r_flag;
  ca.pos = c->r_pos;
  ca.origin = c->r_origin;
  ca.onormal = c->r_onormal;
  ca.closest = c->r_closest;
  ca.normal = c->r_normal;
  ca.cell = c->r_cell;
  ca.dist = c->r_dist;
  ca.d_origin = c->c_origin;
  ca.d_onormal = use_sn ? c->c_onormal : nullptr;
  ca.d_closest = c->c_closest;
  ca.d_normal = c->c_normal;
  ca.d_cell = c->c_cell;
  ca.d_dist = c->c_dist;
  ca.d_index = c->c_index;
  k_compact_rows<<<nblk(nq, 256), 256, 0, st>>>(ca);
  LAUNCHCHK();
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, c->r_pos + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (!dst || !bytes)
      return hipSuccess;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  };
  HIPCHK(back(origin, c->c_origin, sizeof(float) * 3 * n));
  if (use_sn)
    HIPCHK(back(origin_normal, c->c_onormal, sizeof(float) * 3 * n));
  HIPCHK(back(closest, c->c_closest, sizeof(float) * 3 * n));
  HIPCHK(back(normal, c->c_normal, sizeof(float) * 3 * n));
  HIPCHK(back(cell, c->c_cell, sizeof(int32_t) * n));
  HIPCHK(back(dist, c->c_dist, sizeof(double) * n));
  HIPCHK(back(src_index, c->c_index, sizeof(int64_t) * n));
  HIPCHK(hipStreamSynchronize(st));
  *n_out = n;
  return ICPX_OK;
}

// The rows in double precision: vtkPolyDataCorrespondenceFilter with OutputPrecision =
// DOUBLE_PRECISION (ICP/vtkPolyDataCorrespondenceFilter.cxx:519-531).
extern "C" int icpx_correspond_f64(icpx_ctx* c, const double T[16], int64_t step, double max_dist,
  double* origin, double* closest, double* normal, int32_t* cell, double* dist,
  int64_t* src_index, int64_t* n_out)
{
  int rc = check_call(c, T, 0, 0);
  if (rc)
    return rc;
  if (!n_out)
    return fail(ICPX_E_INVALID, "icpx_correspond_f64: n_out is null");
  *n_out = 0;
  if (!c->d_fn64 || !c->d_voff || !c->d_vcell)
    return fail(ICPX_E_INVALID, "Can't execute with nullptr or empty target");
  HIPCHK(hipSetDevice(c->device));
  rc = prepare_queries(c, step);
  if (rc)
    return rc;
  const int64_t nq = c->nq;
  if (nq == 0)
    return ICPX_OK;
  hipStream_t st = c->stream;
  if (nq > c->rows_cap)
  {
    // (the flag / position / cell / index arrays are shared with the float rows)
    free_rows(c);
    ALLOC(c->r_origin, 3 * nq);
    ALLOC(c->r_onormal, 3 * nq);
    ALLOC(c->r_closest, 3 * nq);
    ALLOC(c->r_normal, 3 * nq);
    ALLOC(c->r_cell, nq);
    ALLOC(c->r_dist, nq);
    ALLOC(c->r_flag, nq + 1);
    ALLOC(c->r_pos, nq + 1);
    ALLOC(c->r_scan, scan_scratch_words(nq + 1));
    ALLOC(c->c_origin, 3 * nq);
    ALLOC(c->c_onormal, 3 * nq);
    ALLOC(c->c_closest, 3 * nq);
    ALLOC(c->c_normal, 3 * nq);
    ALLOC(c->c_cell, nq);
    ALLOC(c->c_dist, nq);
    ALLOC(c->c_index, nq);
    c->rows_cap = nq;
  }
  if (nq > c->rows64_cap)
  {
    ALLOC(c->r64_origin, 3 * nq);
    ALLOC(c->r64_closest, 3 * nq);
    ALLOC(c->r64_normal, 3 * nq);
    ALLOC(c->r64_dist, nq);
    ALLOC(c->c64_origin, 3 * nq);
    ALLOC(c->c64_closest, 3 * nq);
    ALLOC(c->c64_normal, 3 * nq);
    ALLOC(c->c64_dist, nq);
    c->rows64_cap = nq;
  }
  c->run_active = false;
  set_cap(c, max_dist);
  rc = set_transform(c, T, 0);
  if (!rc)
    rc = run_search(c);
  if (rc)
    return rc;
  k_cert_mark<<<1, 64, 0, st>>>(c->d_state, c->pass_certs ? 1 : 0, c->cap2);
  Rows64Args a;
  a.qpos = c->d_qpos;
  a.qk = c->d_qk;
  a.nq = nq;
  a.T = xf_ref(c);
  a.tverts = c->d_tverts;
  a.tid = c->d_tid;
  a.tri = c->d_tri;
  a.xyz = c->d_xyz;
  a.fn64 = c->d_fn64;
  a.voff = c->d_voff;
  a.vcell = c->d_vcell;
  a.hint = c->d_hint;
  a.max_dist = max_dist;
  a.o_origin = c->r64_origin;
  a.o_closest = c->r64_closest;
  a.o_normal = c->r64_normal;
  a.o_dist = c->r64_dist;
  a.o_cell = c->r_cell;
  a.o_flag = c->r_flag;
  HIPCHK(hipMemsetAsync(c->r_flag + nq, 0, sizeof(uint32_t), st));
  k_rows64<<<nblk(nq, 256), 256, 0, st>>>(a);
  exclusive_scan(c->r_flag, c->r_pos, nq + 1, c->r_scan, st);
  Compact64Args ca;
  ca.nq = nq;
  ca.k0 = c->k0;
  ca.kglobal = c->curve_shard ? c->d_kglobal : nullptr;
  ca.step = step;
  ca.flag = c->r_flag;
  ca.pos = c->r_pos;
  ca.origin = c->r64_origin;
  ca.closest = c->r64_closest;
  ca.normal = c->r64_normal;
  ca.dist = c->r64_dist;
  ca.cell = c->r_cell;
  ca.d_origin = c->c64_origin;
  ca.d_closest = c->c64_closest;
  ca.d_normal = c->c64_normal;
  ca.d_dist = c->c64_dist;
  ca.d_cell = c->c_cell;
  ca.d_index = c->c_index;
  k_compact_rows64<<<nblk(nq, 256), 256, 0, st>>>(ca);
  LAUNCHCHK();
  uint32_t n = 0;
  HIPCHK(hipMemcpyAsync(&n, c->r_pos + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  auto back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    if (!dst || !bytes)
      return hipSuccess;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  };
  HIPCHK(back(origin, c->c64_origin, sizeof(double) * 3 * n));
  HIPCHK(back(closest, c->c64_closest, sizeof(double) * 3 * n));
  HIPCHK(back(normal, c->c64_normal, sizeof(double) * 3 * n));
  HIPCHK(back(cell, c->c_cell, sizeof(int32_t) * n));
  HIPCHK(back(dist, c->c64_dist, sizeof(double) * n));
  HIPCHK(back(src_index, c->c_index, sizeof(int64_t) * n));
  HIPCHK(hipStreamSynchronize(st));
  *n_out = n;
  return ICPX_OK;
}

// -------------------------------------------------------- line intersection ----
// vtkImplicitPolyDataIntersection::IntersectWithLine for n segments at once
// (ICP/vtkImplicitPolyDataIntersection.cxx:110-337), on the target's box tree.
extern "C" int icpx_intersect_lines(icpx_ctx* c, const double* p1, const double* p2, int64_t n,
  int32_t* hit, double* ret, double* closest, double* normal, double* point, int32_t* cell,
  int32_t* kind)
{
  if (!c || !p1 || !p2)
    return fail(ICPX_E_INVALID, "icpx_intersect_lines: null argument");
  if (n < 0 || n >= (1ll << 31))
    return fail(ICPX_E_INVALID, "icpx_intersect_lines: bad segment count %lld", (long long)n);
  if (c->nt == 0 || !c->d_fn64 || !c->d_voff || !c->d_vcell)
    return fail(ICPX_E_INVALID, "No polygons to intersect!");
  for (int64_t i = 0; i < 3 * n; i++)
    if (!std::isfinite(p1[i]) || !std::isfinite(p2[i]))
      return fail(ICPX_E_INVALID, "icpx_intersect_lines: segment %lld is not finite",
        (long long)(i / 3));
  if (n == 0)
    return ICPX_OK;
  HIPCHK(hipSetDevice(c->device));
  const int L = c->leaf_size;
  int rc = ensure_bin(c, L);
  if (rc)
    return rc;
  const Bin& b = c->bins[(L == 1) ? 0 : (L == 2) ? 1 : 2];
  hipStream_t st = c->stream;
  double *d_p = nullptr, *d_out = nullptr; // {p1, p2}; {ret, closest, normal, point}
  int32_t* d_i = nullptr;                  // {hit, cell, kind}
  auto drop = [&]() {
    dev_free(&d_p);
    dev_free(&d_out);
    dev_free(&d_i);
  };
  rc = dev_alloc(&d_p, 6 * n);
  if (!rc)
    rc = dev_alloc(&d_out, 10 * n);
  if (!rc)
    rc = dev_alloc(&d_i, 3 * n);
  if (rc)
  {
    drop();
    return rc;
  }
  hipError_t e = hipMemcpyAsync(d_p, p1, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_p + 3 * n, p2, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
  {
    IntersectArgs a;
    a.p1 = d_p;
    a.p2 = d_p + 3 * n;
    a.n = n;
    a.nodes = b.nodes;
    a.tree = b.t;
    a.tverts = c->d_tverts;
    a.tid = c->d_tid;
    a.nt = c->nt;
    a.tri = c->d_tri;
    a.xyz = c->d_xyz;
    a.fn64 = c->d_fn64;
    a.voff = c->d_voff;
    a.vcell = c->d_vcell;
    a.scene_max = c->scene_max;
    a.novalue = c->bounds_length;
    a.ret = d_out;
    a.closest = d_out + n;
    a.normal = d_out + 4 * n;
    a.x = d_out + 7 * n;
    a.hit = d_i;
    a.cell = d_i + n;
    a.kind = d_i + 2 * n;
    const unsigned grid = nblk(n, ISECT_THREADS);
    if (L == 1)
      k_intersect_lines<1><<<grid, ISECT_THREADS, 0, st>>>(a);
    else if (L == 2)
      k_intersect_lines<2><<<grid, ISECT_THREADS, 0, st>>>(a);
    else
      k_intersect_lines<4><<<grid, ISECT_THREADS, 0, st>>>(a);
    e = hipGetLastError();
  }
  auto back = [&](void* dst, const void* src, size_t bytes) {
    if (e == hipSuccess && dst)
      e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  };
  back(ret, d_out, sizeof(double) * n);
  back(closest, d_out + n, sizeof(double) * 3 * n);
  back(normal, d_out + 4 * n, sizeof(double) * 3 * n);
  back(point, d_out + 7 * n, sizeof(double) * 3 * n);
  back(hit, d_i, sizeof(int32_t) * n);
  back(cell, d_i + n, sizeof(int32_t) * n);
  back(kind, d_i + 2 * n, sizeof(int32_t) * n);
  const hipError_t es = hipStreamSynchronize(st);
  if (e == hipSuccess)
    e = es;
  drop();
  if (e != hipSuccess)
    return fail(ICPX_E_HIP, "icpx_intersect_lines failed: %s", hipGetErrorString(e));
  return ICPX_OK;
}

// ---------------------------------------------------------------- the loop ----
static int check_run(icpx_ctx* c, const icpx_params* p)
{
  if (!c || !p)
    return fail(ICPX_E_INVALID, "icpx_run: null argument");
  if (c->nt == 0)
    return fail(ICPX_E_INVALID, "Can't execute with nullptr or empty target");
  if (c->np == 0)
    return fail(ICPX_E_INVALID, "Can't execute with nullptr or empty input");
  if (p->metric < 0 || p->metric > 2)
    return fail(ICPX_E_INVALID, "unknown metric %d", p->metric);
  if (p->metric == ICPX_METRIC_PLANE_TO_PLANE && !c->d_srcn)
    return fail(ICPX_E_INVALID, "Plane to plane metric requires source normals");
  if (p->max_landmarks < 1)
    return fail(ICPX_E_INVALID, "max_landmarks must be positive");
  return ICPX_OK;
}

static int64_t landmark_step(const icpx_ctx* c, const icpx_params* p)
{
  // vtkPolyDataCorrespondenceFilter.cxx:473-477
  return c->np > p->max_landmarks ? c->np / p->max_landmarks : 1;
}

extern "C" int icpx_run_begin(icpx_ctx* c, const icpx_params* p)
{
  int rc = check_run(c, p);
  if (rc)
    return rc;
  HIPCHK(hipSetDevice(c->device));
  c->run_p = *p;
  c->run_step = landmark_step(c, p);
  c->run_it = 0;
  c->run_noted = 0;
  c->noisy = false;
  c->run_done_known = false;
  c->hint_valid = false; // every Update() starts cold
  c->stat_valid = false;
  c->stat_pending = false;
  c->local_holdoff = 0;
  c->bad_streak = 0;
  c->hopeless_streak = 0;
  c->direct = false;
  c->holdoff_len = 8;
  c->probation = false;
  c->run_fused = false;
  c->run_shared = false;
  c->last_lean = false;
  c->lean_base = false;
  c->clean_noted = false;
  c->lean_unnoted = 0;
  c->lean_holdoff = 0;
  c->lean_backoff = 4;
  for (int s = 0; s < 4; s++)
    c->pass_lean[s] = false;
  double I[16];
  for (int i = 0; i < 16; i++)
    I[i] = (i % 5 == 0) ? 1.0 : 0.0;
  rc = set_transform(c, I, 1);
  if (!rc)
    rc = prepare_queries(c, c->run_step);
  if (rc)
    return rc;
  c->run_active = true;
  return ICPX_OK;
}

// The host stays at most two iterations ahead of the device: before iteration i is
// enqueued, iteration i-2's record (32 bytes, copied behind its solve) is awaited.  That
// keeps the queue fed, bounds the passes enqueued after the run stopped (they return at
// once on the device), and brings the residual the shortcut's policy looks at.
// Which of the two tree kernels enqueued over a large cloud's leftovers did the work in the
// pass behind record `slot` (diagnostic counts; the list's length is only known afterwards).
static void tally_split(icpx_ctx* c, int slot)
{
  if (!c->pass_split[slot])
    return;
  c->pass_split[slot] = false;
  if ((int64_t)c->h_rec[slot].fallback >= c->nq / 2)
    c->split_wave++;
  else
    c->split_pool++;
}

static int await_record(icpx_ctx* c, int slot)
{
  // wait for that solve's serial number to show up in the pinned record (the GPU writes it
  // last); while waiting, look now and then whether the stream has drained without it
  const uint32_t want = c->slot_seq[slot];
  volatile const uint32_t* seen = &c->h_rec[slot].seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0; *seen != want; spins++)
  {
    if ((spins & 0xfffff) == 0xfffff)
    {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess && *seen != want)
        return fail(ICPX_E_HIP, "iteration record %u never arrived", want);
      if (e != hipSuccess && e != hipErrorNotReady)
        return fail(ICPX_E_HIP, "stream failed: %s", hipGetErrorString(e));
      // a kernel that never finishes must not hold the host for ever
      const double waited =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > c->record_wait_limit)
        return fail(ICPX_E_HIP, "iteration record %u did not arrive within %.0f s", want,
          c->record_wait_limit);
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ICPX_OK;
}

static int run_catch_up(icpx_ctx* c, int lag_override)
{
  // Two iterations behind as a rule.  One behind for the first passes of a large cloud: each
  // takes milliseconds, the gap the wait leaves between two of them is some ten microseconds,
  // and what the record says (how far the cloud stays from the surface) decides which kernels
  // the next of these, the most expensive passes of a run after the cold one, are given.
  // (lag_override 0: everything enqueued so far, for a caller that must know how the lean
  // passes still out have ended)
  const int lag = lag_override >= 0 ? lag_override + 1
                                    : (policy_nq(c) >= CLOSE_FOLLOW_MIN && c->since_cold <= 3) ? 1 : 2;
  while (c->run_noted + lag <= c->run_it)
  {
    const int slot = (int)(c->run_noted & 3);
    const int rc = await_record(c, slot);
    if (rc)
      return rc;
    const RunRecord& r = c->h_rec[slot];
    if (c->pass_lean[slot] && c->lean_unnoted > 0)
      c->lean_unnoted--;
    if (r.done == 2)
    {
      // A lean pass met a query whose certificate did not hold: its sums were dropped, the
      // registration stands where it stood before that pass, and that pass and the ones
      // enqueued behind it (lean ones, which returned at once) count for nothing.  The loop
      // goes back to it; the device is told to carry on, and the next passes are full ones.
      c->lean_failures++;
      c->run_it = c->run_noted;
      c->lean_unnoted = 0;
      c->lean_holdoff = c->use_lean >= 2 ? 1 : c->lean_backoff;
      c->lean_backoff = c->lean_backoff < 32 ? 2 * c->lean_backoff : 32;
      c->clean_noted = false;
      k_resume<<<1, 64, 0, c->stream>>>(c->d_state);
      LAUNCHCHK();
      return ICPX_OK;
    }
    if (c->pass_lean[slot])
      c->lean_backoff = 4;
    {
      static const bool trace = std::getenv("ICPX_TRACE") != nullptr; // (development aid)
      if (trace)
        std::fprintf(stderr, "[icpx] pass %lld %s iters %d done %d n %.0f mean %.3e fallback %u "
                             "contested %u listed %u base %d holdoff %d\n",
          (long long)c->run_noted, c->pass_lean[slot] ? "lean" : c->pass_local[slot] ? "local" : "tree",
          r.iters, r.done, r.n, r.mean, r.fallback, r.contested, r.listed, (int)c->lean_base,
          c->lean_holdoff);
    }
    // (several ranks: the record's counts are the job's, and so is the number they are held
    // against)
    const int64_t nq_rec = c->run_shared ? (int64_t)r.nq_total : c->nq;
    // (the few queries that pass left to the tree kernels were given certificates by them)
    // (a heavy cloud's leftovers are certificates that a cloud still on the move uses up, as many
    // again next pass, and a lean pass that meets one fails: there the record must show none)
    c->clean_noted = (c->pass_lean[slot] || (c->pass_local[slot] && !c->pass_listed[slot])) &&
      (int64_t)r.fallback + (int64_t)r.contested <=
        (heavy_cloud(c) ? (int64_t)0 : std::max<int64_t>(16, nq_rec >> 13));
    if (r.done)
      c->run_done_known = true;
    note_pass(c, r.n, r.mean, (double)r.fallback, (double)r.listed, c->pass_local[slot],
      c->pass_listed[slot], (double)nq_rec);
    tally_split(c, slot);
    c->run_noted++;
  }
  return ICPX_OK;
}

static int run_enqueue_impl(icpx_ctx* c, double* sums_device, int* done_out, int loop)
{
  if (!c || !c->run_active)
    return fail(ICPX_E_INVALID, "icpx_run_enqueue: no run in progress (call icpx_run_begin)");
  HIPCHK(hipSetDevice(c->device));
  int rc = run_catch_up(c);
  if (rc)
    return rc;
  if (done_out)
    *done_out = c->run_done_known ? 1 : 0;
  if (c->run_done_known)
    return ICPX_OK;
  const icpx_params& p = c->run_p;
  return enqueue_pass(c, p.metric, c->run_step, p.max_distance, p.min_normal_dot,
    p.metric == ICPX_METRIC_PLANE_TO_PLANE ? (p.rotate_source_normals ? 2 : 1) : 0,
    sums_device ? sums_device : c->d_sums, false, loop);
}

extern "C" int icpx_run_enqueue(icpx_ctx* c, double* sums_device, int* done_out)
{
  if (c && c->run_shared)
    return fail(ICPX_E_INVALID, "icpx_run_enqueue: this run is driven by icpx_run_advance");
  return run_enqueue_impl(c, sums_device, done_out, 0);
}

extern "C" int icpx_run_advance(icpx_ctx* c, double* sums_device, int* more_out)
{
  if (!c || !c->run_active || !more_out)
    return fail(ICPX_E_INVALID, "icpx_run_advance: no run in progress (call icpx_run_begin), or more_out is null");
  if (!sums_device)
    return fail(ICPX_E_INVALID, "icpx_run_advance: sums_device is null");
  if (!c->run_shared && c->run_it != 0)
    return fail(ICPX_E_INVALID, "icpx_run_advance: this run is driven by icpx_run_enqueue");
  *more_out = 0;
  HIPCHK(hipSetDevice(c->device));
  c->run_shared = true;
  const icpx_params& p = c->run_p;
  // (c->run_it is the run's position: a failed lean pass takes it back)
  while (!c->run_done_known && c->run_it >= p.max_iterations)
  {
    if (c->lean_unnoted == 0)
      return ICPX_OK;
    const int rc = run_catch_up(c, 0); // how did the lean passes still out end?
    if (rc)
      return rc;
  }
  if (c->run_done_known)
    return ICPX_OK;
  int done = 0;
  int rc = run_enqueue_impl(c, sums_device, &done, 2);
  if (rc)
    return rc;
  // (nothing was enqueued when the records read on the way said that the run is over; or when
  // they took the position back and left it at the end again: the next call sorts that out)
  if (done || c->run_done_known)
    return ICPX_OK;
  *more_out = 1;
  return ICPX_OK;
}

extern "C" int icpx_run_update(icpx_ctx* c, const double* sums_device)
{
  if (!c || !c->run_active)
    return fail(ICPX_E_INVALID, "icpx_run_update: no run in progress (call icpx_run_begin)");
  if (c->run_done_known)
    return ICPX_OK;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const icpx_params& p = c->run_p;
  const int slot = (int)(c->run_it & 3);
  const uint32_t seq = ++c->rec_seq;
#define ICPX_SOLVE(M)                                                                            \
  k_solve<M><<<1, 64, 0, st>>>(c->d_state, sums_device ? sums_device : c->d_sums, c->origin[0],  \
    c->origin[1], c->origin[2], p.check_mean_distance, p.max_mean_distance, c->d_counters,       \
    c->d_rec + slot, seq, anchor_rmax(c), anchor_reach(c), c->run_shared ? 1 : 0)
  if (p.metric == 0)
    ICPX_SOLVE(0);
  else if (p.metric == 1)
    ICPX_SOLVE(1);
  else
    ICPX_SOLVE(2);
#undef ICPX_SOLVE
  LAUNCHCHK();
  c->slot_seq[slot] = c->rec_seq;
  c->pass_local[slot] = c->last_lean ? false : c->last_local;
  c->pass_listed[slot] = c->last_lean ? false : c->last_listed;
  c->pass_split[slot] = c->last_lean ? false : c->last_split;
  c->pass_lean[slot] = c->last_lean;
  if (c->last_lean)
  {
    c->lean_unnoted++;
    c->lean_passes++;
  }
  c->last_lean = false;
  c->run_it++;
  return ICPX_OK;
}

extern "C" int icpx_run_end(icpx_ctx* c, double T_out[16], int* iterations_out,
  double* mean_distance_out, int* status_out)
{
  if (!c || !c->run_active)
    return fail(ICPX_E_INVALID, "icpx_run_end: no run in progress (call icpx_run_begin)");
  HIPCHK(hipSetDevice(c->device));
  RunState h;
  HIPCHK(hipMemcpyAsync(&h, c->d_state, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->run_active = false;
  for (int s = 0; s < 4; s++) // (the last two iterations' records were not looked at yet)
    if (c->h_rec[s].seq == c->slot_seq[s])
      tally_split(c, s);
  if (T_out)
    std::memcpy(T_out, h.T, sizeof(h.T));
  if (iterations_out)
    *iterations_out = h.iters;
  if (mean_distance_out)
    *mean_distance_out = h.mean;
  if (status_out)
    *status_out = h.status;
  return ICPX_OK;
}

extern "C" int icpx_run(icpx_ctx* c, const icpx_params* p, double T_out[16], int* iterations_out,
  double* mean_distance_out, int* status_out)
{
  if (!T_out)
    return fail(ICPX_E_INVALID, "icpx_run: null argument");
  int rc = check_run(c, p);
  if (rc)
    return rc;
  if (!c->allreduce)
  {
    // one rank: nothing leaves the device until the loop is over
    rc = icpx_run_begin(c, p);
    const bool fuse = !rc && c->nq > 0; // (an empty shard has no reduction to ride on)
    c->run_fused = fuse;
    // (c->run_it is the run's position: a failed lean pass takes it back)
    while (!rc && !c->run_done_known)
    {
      if (c->run_it >= p->max_iterations)
      {
        if (c->lean_unnoted == 0)
          break;
        rc = run_catch_up(c, 0); // how did the lean passes still out end?
        continue;
      }
      int done = 0;
      rc = run_enqueue_impl(c, nullptr, &done, fuse ? 1 : 0);
      if (rc || done)
        break;
      if (!fuse)
        rc = icpx_run_update(c, nullptr);
    }
    if (rc)
    {
      c->run_active = false;
      return rc;
    }
    return icpx_run_end(c, T_out, iterations_out, mean_distance_out, status_out);
  }

  // a host all-reduce hook: sums come to the host every iteration and are solved here
  double T[16];
  for (int i = 0; i < 16; i++)
    T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  c->hint_valid = false; // every Update() starts cold
  const int64_t step = landmark_step(c, p);
  const int use_sn =
    p->metric == ICPX_METRIC_PLANE_TO_PLANE ? (p->rotate_source_normals ? 2 : 1) : 0;
  int iters = 0, status = ICPX_RUN_MAX_ITERATIONS;
  double mean = 0.0;
  double sums[ICPX_NSUMS];
  for (int it = 0; it < p->max_iterations; it++)
  {
    rc = icpx_accumulate(
      c, p->metric, T, step, p->max_distance, p->min_normal_dot, use_sn, sums, nullptr);
    if (rc)
      return rc;
    if (c->allreduce(sums, ICPX_NSUMS, c->allreduce_user))
      return fail(ICPX_E_INVALID, "all-reduce hook reported a failure");
    double M[16];
    int64_t n = 0;
    double md = 0.0;
    rc = icpx_solve(p->metric, sums, c->origin, M, &n, &md);
    mean = md;
    if (p->check_mean_distance)
    {
      if (n > 0 && mean <= p->max_mean_distance)
      {
        status = ICPX_RUN_CONVERGED;
        break;
      }
    }
    if (rc == ICPX_E_FEW)
    {
      status = ICPX_RUN_TOO_FEW;
      break;
    }
    if (rc)
      return rc;
    double R[16];
    for (int r = 0; r < 4; r++)
      for (int k = 0; k < 4; k++)
      {
        double acc = 0.0;
        for (int j = 0; j < 4; j++)
          acc += M[4 * r + j] * T[4 * j + k];
        R[4 * r + k] = acc;
      }
    if (!icpx_core::transform_usable(R))
    {
      status = ICPX_RUN_DIVERGED; // (as the device's loop: the last good transform is kept)
      break;
    }
    iters++;
    std::memcpy(T, R, sizeof(R));
  }
  std::memcpy(T_out, T, sizeof(T));
  if (iterations_out)
    *iterations_out = iters;
  if (mean_distance_out)
    *mean_distance_out = mean;
  if (status_out)
    *status_out = status;
  return ICPX_OK;
}
