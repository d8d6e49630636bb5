// solveRsPnPRansac over the C++ mirror (include/rsba/solve_rs_pnp.hpp): reads a flat problem file, writes the result.
// Used by tests/test_facade.py to check the whole RANSAC flow (host subsets + batched device hypotheses + selection +
// final refinement) against a sequential replay through the oracle.
//   in  (binary, little endian): int32 n, shutter, scan0, scan1, iterations (0: solveRsPnP, -1: the DLT of the GS initialisation only), minInliers, minPoints; float reprojError;
//        uint64 rngState; double cam[9], rvec[3], tvec[3], rvec2[3], tvec2[3]; float opoints[n*3], ipoints[n*2]
//   out: double rvec[3], tvec[3], rvec2[3], tvec2[3]; int32 numInliers, inliers[numInliers]
//   g++ -std=c++17 -O2 -Iinclude examples/pnp_ransac.cpp -Lrsba_amd/_lib -lrsba_amd -Wl,-rpath,... -o pnp_ransac
#include <cstdio>
#include <vector>

#include "rsba/solve_rs_pnp.hpp"

using namespace rsba_amd;

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("problem"); return 2; }
  int32_t hd[7]; float err; uint64_t state; double cam[9], v[12];
  if (!rd(f, hd, 7) || !rd(f, &err, 1) || !rd(f, &state, 1) || !rd(f, cam, 9) || !rd(f, v, 12)) return 2;
  const int n = hd[0];
  std::vector<float> op((size_t)n * 3), ip((size_t)n * 2);
  if (!rd(f, op.data(), op.size()) || !rd(f, ip.data(), ip.size())) return 2;
  std::fclose(f);
  const int scan[2] = {hd[2], hd[3]};
  std::vector<int> inliers;
  try {
    if (hd[4] == -1) {  // iterations -1: the direct linear transform of the global-shutter initialisation alone (host glue; no device)
      const std::vector<double> nrm = pnp_detail::normalised_points(cam, ip.data(), n);
      std::vector<int32_t> all((size_t)n);
      for (int i = 0; i < n; ++i) all[(size_t)i] = i;
      double pose[6];
      if (!pnp_detail::dlt_pose(op.data(), nrm.data(), all.data(), n, pose)) inliers.assign(1, -1);
      else { pnp_detail::from_pose(pose, v, v + 3); pnp_detail::from_pose(pose, v + 6, v + 9); }
    } else if (hd[4] == 0) {   // iterations 0: the single solve (solveRsPnP); -1 inliers = the solution is not usable
      if (!solveRsPnP(op.data(), ip.data(), n, cam, v, v + 3, v + 6, v + 9, (SHUTTER)hd[1], scan)) inliers.assign(1, -1);
    } else
    solveRsPnPRansac(op.data(), ip.data(), n, cam, v, v + 3, v + 6, v + 9, (SHUTTER)hd[1], scan, hd[4], err, hd[5], &inliers, hd[6], state);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  FILE* g = std::fopen(argv[2], "wb");
  if (!g) { std::perror("out"); return 2; }
  std::fwrite(v, sizeof(double), 12, g);
  const int32_t cnt = (int32_t)inliers.size();
  std::fwrite(&cnt, sizeof cnt, 1, g);
  std::vector<int32_t> idx(inliers.begin(), inliers.end());
  std::fwrite(idx.data(), sizeof(int32_t), idx.size(), g);
  std::fclose(g);
  return 0;
}
