// The call sequence a maintainer adds to vtkICP::InternalUpdate (ICP/vtkICP.cxx:130-247),
// compiled against include/icpx.h with plain arrays standing in for vtkPolyData /
// vtkFloatArray / vtkMatrix4x4 (VTK is not needed to check that the documented binding
// type-checks, links and runs).  Reads
//   <dir>/target_xyz.f32  <dir>/target_tri.i32  <dir>/source_xyz.f32  [<dir>/source_normals.f32]
// and prints status, iteration count, mean distance and the 4x4 (%.17g).
//
//   vtkicp_shim <dir> <metric> <max_iterations> <max_landmarks> <max_distance> <check_mean> <max_mean>
#include "icpx.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace
{
template <typename T>
std::vector<T> slurp(const std::string& path, bool optional = false)
{
  std::vector<T> v;
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f)
  {
    if (!optional)
    {
      std::fprintf(stderr, "cannot read %s\n", path.c_str());
      std::exit(64);
    }
    return v;
  }
  std::fseek(f, 0, SEEK_END);
  const long bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  v.resize((size_t)bytes / sizeof(T));
  if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size())
    std::exit(65);
  std::fclose(f);
  return v;
}

// What vtkICP keeps between calls, with the VTK objects reduced to their arrays.
struct ShimICP
{
  // vtkICP members (ICP/vtkICP.h:140-152)
  int MaximumNumberOfIterations = 50, NumberOfIterations = 0, Metric = ICPX_METRIC_POINT_TO_PLANE;
  int64_t MaximumNumberOfLandmarks = 200;
  bool CheckMeanDistance = false;
  double MaximumMeanDistance = 0.01, MaximumDistance = 10.0, MeanDistance = 0.0;
  double Matrix[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
  int Status = 0;
  std::vector<float> TargetPoints, SourcePoints, SourceNormals;
  std::vector<int32_t> TargetTriangles;
  icpx_ctx* Icpx = nullptr;

  ~ShimICP() { icpx_destroy(this->Icpx); }

  // returns false after "vtkErrorMacro" (message on stderr, Matrix left as it was)
  bool InternalUpdate()
  {
    if (this->SourcePoints.empty())
      return this->Error("Can't execute with nullptr or empty input");
    if (this->TargetPoints.empty())
      return this->Error("Can't execute with nullptr or empty target");
    if (!this->Icpx && icpx_create(&this->Icpx, /*device*/ 0) != ICPX_OK)
      return this->Error(icpx_last_error());
    if (icpx_set_target(this->Icpx, this->TargetPoints.data(), (int64_t)this->TargetPoints.size() / 3,
          this->TargetTriangles.data(), (int64_t)this->TargetTriangles.size() / 3,
          /*cell normals*/ nullptr) != ICPX_OK)
      return this->Error(icpx_last_error());
    const float* snrm = nullptr;
    std::vector<float> computed;
    if (this->Metric == ICPX_METRIC_PLANE_TO_PLANE)
    {
      if (this->SourceNormals.empty())
        return this->Error("Plane to plane metric requires source normals");
      snrm = this->SourceNormals.data();
    }
    if (icpx_set_source(this->Icpx, this->SourcePoints.data(), (int64_t)this->SourcePoints.size() / 3,
          snrm) != ICPX_OK)
      return this->Error(icpx_last_error());
    icpx_params p;
    icpx_default_params(&p);
    p.metric = this->Metric;
    p.max_iterations = this->MaximumNumberOfIterations;
    p.max_landmarks = this->MaximumNumberOfLandmarks;
    p.max_distance = this->MaximumDistance;
    p.check_mean_distance = this->CheckMeanDistance ? 1 : 0;
    p.max_mean_distance = this->MaximumMeanDistance;
    double T[16], mean = 0.0;
    int iters = 0, status = 0;
    if (icpx_run(this->Icpx, &p, T, &iters, &mean, &status) != ICPX_OK)
      return this->Error(icpx_last_error());
    this->NumberOfIterations = iters;
    if (this->CheckMeanDistance)
      this->MeanDistance = mean;
    this->Status = status;
    for (int i = 0; i < 16; i++)
      this->Matrix[i] = T[i]; // vtkMatrix4x4::DeepCopy(const double[16]): same element order
    return true;
  }

  bool Error(const char* msg)
  {
    std::fprintf(stderr, "ERROR: vtkICP: %s\n", msg);
    return false;
  }
};
} // namespace

int main(int argc, char** argv)
{
  if (argc != 8)
  {
    std::fprintf(stderr, "usage: %s dir metric iterations landmarks max_distance check_mean max_mean\n",
      argv[0]);
    return 64;
  }
  const std::string dir = argv[1];
  ShimICP icp;
  icp.Metric = std::atoi(argv[2]);
  icp.MaximumNumberOfIterations = std::atoi(argv[3]);
  icp.MaximumNumberOfLandmarks = std::atoll(argv[4]);
  icp.MaximumDistance = std::atof(argv[5]);
  icp.CheckMeanDistance = std::atoi(argv[6]) != 0;
  icp.MaximumMeanDistance = std::atof(argv[7]);
  icp.TargetPoints = slurp<float>(dir + "/target_xyz.f32");
  icp.TargetTriangles = slurp<int32_t>(dir + "/target_tri.i32");
  icp.SourcePoints = slurp<float>(dir + "/source_xyz.f32");
  icp.SourceNormals = slurp<float>(dir + "/source_normals.f32", true);
  if (!icp.InternalUpdate())
    return 2; // the reference returns with its matrix untouched; say so to the caller
  std::printf("status %d\niterations %d\nmean %.17g\n", icp.Status, icp.NumberOfIterations,
    icp.MeanDistance);
  for (int i = 0; i < 16; i++)
    std::printf("%.17g%c", icp.Matrix[i], (i % 4 == 3) ? '\n' : ' ');
  return 0;
}
