// Reads / rewrites rsba's Thrift cache files through include/rsba/session_cache.hpp (no libthrift):
//   session_cache_tool dump session|frame <cache>          -> JSON on stdout
//   session_cache_tool copy session|frame <cache> <out>    -> load, then save in the reference's event layout
// Used by tests/test_session_cache.py.   g++ -std=c++17 -O2 -Iinclude examples/session_cache_tool.cpp -o session_cache_tool
#include <cstdio>
#include <cstring>
#include <string>

#include "rsba/session_cache.hpp"

using namespace rsba_amd;

static void put(const std::vector<double>& v) { std::printf("["); for (size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? "," : "", v[i]); std::printf("]"); }
static void put(const std::vector<std::vector<double>>& v) { std::printf("["); for (size_t i = 0; i < v.size(); ++i) { if (i) std::printf(","); put(v[i]); } std::printf("]"); }
static void put(const ObservationRef& r) { std::printf("{\"frame\":%d,\"obs\":%d,\"valid\":%s}", r.frame, r.obs, r.valid ? "true" : "false"); }
static void put(const std::vector<ObservationRef>& v) { std::printf("["); for (size_t i = 0; i < v.size(); ++i) { if (i) std::printf(","); put(v[i]); } std::printf("]"); }
static void put(const Frame& f) {
  std::printf("{\"obs\":[");
  for (size_t i = 0; i < f.obs.size(); ++i) {
    const Observation& o = f.obs[i];
    std::printf("%s{\"x\":%.17g,\"y\":%.17g", i ? "," : "", o.x, o.y);
    if (o.__isset.matches) { std::printf(",\"matches\":"); put(o.matches); }
    if (o.__isset.track) std::printf(",\"track\":%d", o.track);
    std::printf("}");
  }
  std::printf("]");
  if (f.__isset.poses) { std::printf(",\"poses\":"); put(f.poses); }
  if (f.__isset.cam) { std::printf(",\"cam\":"); put(f.cam); }
  if (f.__isset.priorPoses) { std::printf(",\"priorPoses\":"); put(f.priorPoses); }
  std::printf("}");
}
static void put(const Session& s) {
  std::printf("{\"cam\":"); put(s.cam);
  std::printf(",\"frames\":[");
  for (size_t i = 0; i < s.frames.size(); ++i) { if (i) std::printf(","); put(s.frames[i]); }
  std::printf("],\"tracks\":[");
  for (size_t i = 0; i < s.tracks.size(); ++i) {
    const Track& k = s.tracks[i];
    std::printf("%s{\"obs\":", i ? "," : ""); put(k.obs);
    if (k.__isset.pt) { std::printf(",\"pt\":"); put(k.pt); }
    std::printf(",\"valid\":%s}", k.valid ? "true" : "false");
  }
  std::printf("],\"rs\":%d,\"scanlines\":[", s.rs);
  for (size_t i = 0; i < s.scanlines.size(); ++i) std::printf("%s%d", i ? "," : "", s.scanlines[i]);
  std::printf("],\"width\":%d,\"height\":%d}", s.width, s.height);
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s dump|copy session|frame <cache> [out]\n", argv[0]); return 2; }
  const bool copy = !std::strcmp(argv[1], "copy"), session = !std::strcmp(argv[2], "session");
  try {
    if (session) { Session s; loadCache(argv[3], s); if (copy) saveCache(argv[4], s); else put(s); }
    else { Frame f; loadCache(argv[3], f); if (copy) saveCache(argv[4], f); else put(f); }
  } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
  if (!copy) std::printf("\n");
  return 0;
}
