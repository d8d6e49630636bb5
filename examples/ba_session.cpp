// A CeresHandler-shaped host program over the facade: loads a flat scene file, rebuilds the Session
// pointer graph rsba works on, runs BA() (VideoSfMHandler.cc:574-631 mirror) and writes the adjusted
// parameters back.  Used by tests/test_facade.py to check that the C++ host path produces the same
// solve as the C ABI / oracle.
//
//   scene file (binary, little endian): int32 F,P,M,rs,scan0,scan1,calibrated,interp,fixFirstN,fixScale,maxIter, int64 N,
//     double huber, double revalidate (squared px threshold of revalidateReprojections, <= 0 = off), double covFrame (>= 0:
//     calcCovariances, the pp | pe | ee blocks of that frame are appended to the result file), double constFrameVelocity,
//     constFrameAcceleration, interFrameRatio (motion priors, CeresHandler.h:147-185), double cam[9], poses[F*P*6], points[M*3], obs_xy[N*2], int32 obs_frame[N], obs_point[N]
//   ba_session --cache session.cache out.bin [fixFirstN=1] [maxIter=20] [huber=0] [calibrated=1] [useOnlyValidMatches=1] [sqrdThreshold=16] [trustRotation=0] [trustPosition=0]
//     replays a Session cache written by the reference (VideoSfMCache, Thrift binary; include/rsba/session_cache.hpp)
//   g++ -std=c++17 -O2 -Iinclude examples/ba_session.cpp -Lrsba_amd/_lib -lrsba_amd -Wl,-rpath,... -o ba_session
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "rsba/ceres_handler.hpp"
#include "rsba/session_cache.hpp"

namespace ceres = rsba_amd::ceres;
using namespace rsba_amd;

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

static int write_result(const char* path, const Session& sess, const ceres::Solver::Summary& summary, bool usable, const std::vector<std::vector<double>>& covs, double covf) {
  FILE* g = std::fopen(path, "wb");
  if (!g) { std::perror("out"); return 2; }
  const double head[6] = {summary.initial_cost, summary.final_cost, (double)summary.iterations.size(), (double)summary.num_residual_blocks_reduced,
                          (double)(int)summary.termination_type, usable ? 1.0 : 0.0};
  std::fwrite(head, sizeof(double), 6, g);
  size_t pmax = 0;
  for (const Frame& fr : sess.frames) pmax = std::max(pmax, fr.poses.size());
  for (const Frame& fr : sess.frames) for (size_t q = 0; q < pmax; ++q) std::fwrite(fr.poses[std::min(q, fr.poses.size() - 1)].data(), sizeof(double), 6, g);   // (a one-pose frame of a two-pose session fills both slots of the [F][P][6] layout)
  for (const Track& t : sess.tracks) std::fwrite(t.pt.data(), sizeof(double), 3, g);
  for (const Frame& fr : sess.frames) if (fr.__isset.priorPoses) for (const auto& pose : fr.priorPoses) std::fwrite(pose.data(), sizeof(double), 6, g);   // solved priorPoses blocks
  if (covf >= 0 && (size_t)covf < covs.size() && covs[(size_t)covf].size() == 108) std::fwrite(covs[(size_t)covf].data(), sizeof(double), 108, g);
  std::fclose(g);
  return usable ? 0 : 1;
}

static int replay_cache(int argc, char** argv) {
  Session sess;
  try { loadCache(argv[2], sess); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 2; }
  if (sess.frames.empty()) { std::fprintf(stderr, "empty session\n"); return 2; }
  SfmOptions opt;
  opt.model.rolling_shutter = sess.frames[0].poses.size() != 1;
  opt.ceres.fixFirstNCameras = argc > 4 ? (unsigned)std::atoi(argv[4]) : 1u;
  const int maxIter = argc > 5 ? std::atoi(argv[5]) : 20;
  opt.ceres.huberLoss = argc > 6 ? std::atof(argv[6]) : 0.0;
  opt.model.calibrated = argc > 7 ? std::atoi(argv[7]) != 0 : true;
  opt.ceres.useOnlyValidMatches = argc > 8 ? std::atoi(argv[8]) != 0 : true;
  if (argc > 9) opt.tracks.sqrdThreshold = std::atof(argv[9]);
  if (argc > 10) opt.ceres.trustPriorCamRotation = std::atof(argv[10]);   // GoodPosePrior on the frames that carry priorPoses (CeresHandler.h:188-204)
  if (argc > 11) opt.ceres.trustPriorCamPosition = std::atof(argv[11]);
  ceres::Solver::Summary summary;
  const bool usable = BA(sess, 0, sess.frames.size() - 1, opt, maxIter, &summary, true, nullptr);
  return write_result(argv[3], sess, summary, usable, {}, -1.0);
}

int main(int argc, char** argv) {
  if (argc >= 4 && !std::strcmp(argv[1], "--cache")) return replay_cache(argc, argv);
  if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin [startFrame] [camEvery] [BA|fullBA|windowedBA] [validTracks] [useOnlyValidMatches] [gsEvery] [manyEvery lines]\n", argv[0]); return 2; }
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
  const int startFrame = argc > 3 ? std::atoi(argv[3]) : 0;            // windowed BA (VideoSfMClient.cc:243): frames [startFrame, F - 1]
  const int camEvery = argc > 4 ? std::atoi(argv[4]) : 0;              // > 0: every camEvery-th frame carries its own Frame.cam (sfm.thrift:48),
  for (int i = 0; camEvery > 0 && i < F; ++i)                          // which CeresHandler::Add uses as that frame's intrinsics block (:260,277)
    if (i % camEvery == camEvery - 1) { sess.frames[i].cam = sess.cam; sess.frames[i].__isset.cam = true; }
  // entry point (default BA): "fullBA" / "windowedBA" run the callers of VideoSfMHandler.cc:153-214 with their option rules;
  // validTracks >= 0 marks only the first validTracks tracks valid (the < 100 valid tracks rule)
  const char* entry = argc > 5 ? argv[5] : "BA";
  const int validTracks = argc > 6 ? std::atoi(argv[6]) : -1;
  for (int j = 0; validTracks >= 0 && j < M; ++j) sess.tracks[j].valid = j < validTracks;
  if (argc > 7) opt.ceres.useOnlyValidMatches = std::atoi(argv[7]) != 0;
  const int gsEvery = argc > 8 ? std::atoi(argv[8]) : 0;               // > 0: every gsEvery-th frame has ONE pose (its first): CeresHandler::Add then gives it
  for (int i = 0; gsEvery > 0 && i < F; ++i)                            // the global-shutter functor inside the rolling-shutter session (CeresHandler.h:245-286)
    if (i % gsEvery == gsEvery - 1 && sess.frames[i].poses.size() == 2) sess.frames[i].poses.resize(1);
  const int manyEvery = argc > 9 ? std::atoi(argv[9]) : 0, lines = argc > 10 ? std::atoi(argv[10]) : 0;   // > 0: every manyEvery-th frame carries `lines` poses,
  for (int i = 0; manyEvery > 0 && lines > 2 && i < F; ++i)             // one per scan line ("fullDoF": samples of its own motion); Add then gives each observation
    if (i % manyEvery == manyEvery - 1 && sess.frames[i].poses.size() == 2) {   // the global-shutter functor on getPose's pick (struct/VideoSfM.cc:83-97, CeresHandler.h:266-285)
      const std::vector<double> a = sess.frames[i].poses[0], b = sess.frames[i].poses[1];
      sess.frames[i].poses.assign((size_t)lines, a);
      for (int l = 0; l < lines; ++l) {
        const double t = (double)l / (lines - 1);   // (numpy.linspace: start + l * step with step = 1 / (lines - 1) — written so that both sides round alike)
        for (int k = 0; k < 6; ++k) sess.frames[i].poses[l][k] = a[k] * (1 - t) + b[k] * t;
      }
    }
  const bool usable = !std::strcmp(entry, "fullBA")       ? fullBA(sess, opt, hd[10], &summary, true, &covs)
                      : !std::strcmp(entry, "windowedBA") ? windowedBA(sess, opt, startFrame, F - 1, hd[10], &summary, true, &covs)
                                                          : BA(sess, startFrame, F - 1, opt, hd[10], &summary, true, &covs);

  return write_result(argv[2], sess, summary, usable, covs, covf);
}
