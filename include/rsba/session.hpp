// Host data model of one SfM session, field-for-field what rsba's Thrift IDL generates
// (/root/reference/src/rsba/sfm.thrift:13-74, gen-cpp/sfm_types.h:51-66,133-146,259-272,332-348) minus
// the RPC / serialisation machinery, plus the option fields the bundle-adjustment path reads
// (/root/reference/src/rsba/SfmOptions.h:23-27,42,46,63-87).  Parameter blocks are the std::vector<double>
// storage inside Frame::poses[i] / Track::pt / Session::cam — block identity is the address, exactly as in
// the reference.
#pragma once
#include <cstdint>
#include <vector>

namespace rsba_amd {

constexpr int NUM_POINT_PARAMS = 3;   // mat/cam.h:19
constexpr int NUM_POSE_PARAMS = 6;    // mat/cam.h:20
constexpr int NUM_CAM_PARAMS = 9;     // mat/cam.h:33
enum SHUTTER { GLOBAL = 0, HORIZONTAL = 1, VERTICAL = 2 };   // mat/cam.h:37-41

struct ObservationRef {   // sfm.thrift:33-42
  int32_t frame = 0;
  int32_t obs = 0;
  bool valid = false;
};

struct Observation {      // sfm.thrift:13-22
  double x = 0, y = 0;
  std::vector<ObservationRef> matches;
  int32_t track = 0;
  struct { bool matches = false, track = false; } __isset;
};

struct Track {            // sfm.thrift:25-30
  std::vector<ObservationRef> obs;
  std::vector<double> pt;
  bool valid = false;
  struct { bool pt = false; } __isset;
};

struct Frame {            // sfm.thrift:45-56
  std::vector<Observation> obs;
  std::vector<std::vector<double>> poses;       // 1 pose (global shutter) or 2 (rolling shutter start / end)
  std::vector<double> cam;
  std::vector<std::vector<double>> priorPoses;
  struct { bool poses = false, cam = false, priorPoses = false; } __isset;
};

struct Session {          // sfm.thrift:62-74; struct/VideoSfM.h:102-126
  std::vector<double> cam;
  std::vector<Frame> frames;
  std::vector<Track> tracks;
  int32_t rs = GLOBAL;                          // unset Session.rs == 0 == GLOBAL (gen-cpp/sfm_types.h:337)
  std::vector<int32_t> scanlines;
  int32_t width = 0, height = 0;
  Session() : cam(NUM_CAM_PARAMS, 0.0), scanlines(2, 0) {}   // struct/VideoSfM.h:104-106
  Track& getTrack(size_t k) { return tracks[k]; }
  const Track& getTrack(size_t k) const { return tracks[k]; }
};

// The fields of SfmOptions the BA path reads (SfmOptions.h), same names and defaults.
struct SfmOptions {
  struct Model {
    bool rolling_shutter = true;
    bool interpolateRotation = true;
    bool use3Dpoints = true;
    bool calibrated = true;
    bool constVelocity = false;
  } model;
  struct Tracks {
    double sqrdThreshold = 16.0;
    unsigned minDistanceToCamera = 0;
  } tracks;
  struct Ceres {
    bool useOnlyValidMatches = true;
    double huberLoss = 0.0;
    bool const3d = false;
    unsigned fixFirstNCameras = 0;
    bool fixScale = false;
    bool fixRotation = false;
    bool fixPosition = false;
    double constFrameVelocity = 0;
    double constFrameAcceleration = 0;
    double interFrameRatio = 1;       // SfmOptions.h:75; 1 = a free, lower-bounded parameter block that Solve optimises (CeresHandler.h:161,172,175)
    double trustPriorCamPosition = 0;
    double trustPriorCamRotation = 0;
    bool revalidateReprojections = false;
  } ceres;
  struct Debug {
    bool calcCovariances = false;   // VideoSfMHandler.cc:599-621
  } debug;
};

}  // namespace rsba_amd
