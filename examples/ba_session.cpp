// A CeresHandler-shaped host program over the facade: loads a flat scene file, rebuilds the Session
// pointer graph rsba works on, runs BA() (VideoSfMHandler.cc:574-631 mirror) and writes the adjusted
// parameters back.  Used by tests/test_facade.py to check that the C++ host path produces the same
// solve as the C ABI / oracle.
//
//   scene file (binary, little endian): int32 F,P,M,rs,scan0,scan1,calibrated,interp,fixFirstN,fixScale,maxIter, int64 N,
//     double huber, double revalidate (squared px threshold of revalidateReprojections, <= 0 = off), double covFrame (>= 0:
//     calcCovariances, the pp | pe | ee blocks of that frame are appended to the result file), double constFrameVelocity,
//     constFrameAcceleration, interFrameRatio (motion priors, CeresHandler.h:147-185), double cam[9], poses[F*P*6], points[M*3], obs_xy[N*2], int32 obs_frame[N], obs_point[N]
//   g++ -std=c++17 -O2 -Iinclude examples/ba_session.cpp -Lrsba_amd/_lib -lrsba_amd -Wl,-rpath,... -o ba_session
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rsba/ceres_handler.hpp"

namespace ceres = rsba_amd::ceres;
using namespace rsba_amd;

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("scene"); return 2; }
  int32_t hd[11]; int64_t N; double huber, reval, covf, motion[3], cam[9];
  if (!rd(f, hd, 11) || !rd(f, &N, 1) || !rd(f, &huber, 1) || !rd(f, &reval, 1) || !rd(f, &covf, 1) || !rd(f, motion, 3) || !rd(f, cam, 9)) return 2;
  const int F = hd[0], P = hd[1], M = hd[2];
  std::vector<double> poses((size_t)F * P * 6), points((size_t)M * 3), xy((size_t)N * 2);
  std::vector<int32_t> of(N), op(N);
  if (!rd(f, poses.data(), poses.size()) || !rd(f, points.data(), points.size()) || !rd(f, xy.data(), xy.size()) || !rd(f, of.data(), N) || !rd(f, op.data(), N)) return 2;
  std::fclose(f);

  Session sess;
  sess.cam.assign(cam, cam + 9);
  sess.rs = hd[3]; sess.scanlines = {hd[4], hd[5]}; sess.width = 1280; sess.height = 720;
  sess.frames.resize(F); sess.tracks.resize(M);
  for (int i = 0; i < F; ++i) {
    sess.frames[i].__isset.poses = true;
    for (int q = 0; q < P; ++q) sess.frames[i].poses.emplace_back(poses.begin() + ((size_t)i * P + q) * 6, poses.begin() + ((size_t)i * P + q + 1) * 6);
  }
  for (int j = 0; j < M; ++j) { sess.tracks[j].pt.assign(points.begin() + (size_t)j * 3, points.begin() + (size_t)j * 3 + 3); sess.tracks[j].__isset.pt = true; sess.tracks[j].valid = true; }
  for (int64_t i = 0; i < N; ++i) {
    Observation o; o.x = xy[2 * i]; o.y = xy[2 * i + 1]; o.track = op[i]; o.__isset.track = true;
    ObservationRef ref; ref.frame = of[i]; ref.obs = (int32_t)sess.frames[of[i]].obs.size(); ref.valid = true;
    sess.tracks[op[i]].obs.push_back(ref);
    sess.frames[of[i]].obs.push_back(o);
  }
  SfmOptions opt;
  opt.model.rolling_shutter = P == 2; opt.model.calibrated = hd[6] != 0; opt.model.interpolateRotation = hd[7] != 0;
  opt.ceres.fixFirstNCameras = (unsigned)hd[8]; opt.ceres.fixScale = hd[9] != 0; opt.ceres.huberLoss = huber;
  if (reval > 0) { opt.ceres.revalidateReprojections = true; opt.tracks.sqrdThreshold = reval; }
  opt.debug.calcCovariances = covf >= 0;
  opt.ceres.constFrameVelocity = motion[0]; opt.ceres.constFrameAcceleration = motion[1]; opt.ceres.interFrameRatio = motion[2];

  ceres::Solver::Summary summary;
  std::vector<std::vector<double>> covs;
  const bool usable = BA(sess, 0, F - 1, opt, hd[10], &summary, true, &covs);

  FILE* g = std::fopen(argv[2], "wb");
  if (!g) { std::perror("out"); return 2; }
  const double head[6] = {summary.initial_cost, summary.final_cost, (double)summary.iterations.size(), (double)summary.num_residual_blocks_reduced,
                          (double)(int)summary.termination_type, usable ? 1.0 : 0.0};
  std::fwrite(head, sizeof(double), 6, g);
  for (int i = 0; i < F; ++i) for (int q = 0; q < P; ++q) std::fwrite(sess.frames[i].poses[q].data(), sizeof(double), 6, g);
  for (int j = 0; j < M; ++j) std::fwrite(sess.tracks[j].pt.data(), sizeof(double), 3, g);
  if (covf >= 0 && (size_t)covf < covs.size() && covs[(size_t)covf].size() == 108) std::fwrite(covs[(size_t)covf].data(), sizeof(double), 108, g);
  std::fclose(g);
  return usable ? 0 : 1;
}
