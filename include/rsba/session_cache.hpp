// Loader / writer for rsba's on-disk Session and Frame caches (second half), without libthrift:
//   vision::sfm::VideoSfMCache::save / serialize / unserialize   /root/reference/src/rsba/struct/VideoSfMCache.h:40-69
//   the structs of /root/reference/src/rsba/sfm.thrift:13-74 (Observation, ObservationRef, Track, Frame, Session)
// The reference writes `obj.write(TBinaryProtocol(TFileTransport(path)))`.  Both layers belong to Apache Thrift (an
// un-vendored dependency; rsba's .travis.yml installs 0.9.x), restated here from their published formats — PARITY
// UNPINNED: the reference ships no cache file to check against (tests/test_session_cache.py pins this reader against
// an independent Python encoder of the same specification, and against this writer).
//   * TBinaryProtocol (non-strict struct encoding, no message header): a struct is a sequence of fields
//     [type:u8][id:i16 BE][value], closed by type 0 (STOP); i16/i32/i64 big-endian two's complement; double = its
//     IEEE-754 bits as a big-endian i64; bool = one byte; binary/string = [len:i32 BE][bytes];
//     list = [element type:u8][count:i32 BE][elements].  Type codes: BOOL 2, BYTE 3, DOUBLE 4, I16 6, I32 8, I64 10,
//     STRING 11, STRUCT 12, MAP 13, SET 14, LIST 15.
//   * TFileTransport: the file is a sequence of events [size:u32 host order = little endian][payload]; every
//     transport write() is one event (TBinaryProtocol issues one per primitive), events never straddle a 16 MiB
//     chunk boundary (the tail of a chunk is zero-filled, a zero size means "continue at the next chunk").
//     Reading concatenates the payloads.
// Optional fields the solver side has no use for (descriptor, color) are skipped on read and not written.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "session.hpp"

namespace rsba_amd {
namespace cache_detail {

constexpr size_t kChunk = 16u * 1024u * 1024u;   // TFileTransport DEFAULT_CHUNK_SIZE
enum TType : uint8_t { T_STOP = 0, T_BOOL = 2, T_BYTE = 3, T_DOUBLE = 4, T_I16 = 6, T_I32 = 8, T_I64 = 10, T_STRING = 11, T_STRUCT = 12, T_MAP = 13, T_SET = 14, T_LIST = 15 };

// ---- TFileTransport: events -> one byte stream ----
inline std::vector<uint8_t> read_events(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<uint8_t> raw;
  uint8_t buf[1 << 16]; size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + got);
  std::fclose(f);
  std::vector<uint8_t> out; out.reserve(raw.size());
  size_t pos = 0;
  while (pos + 4 <= raw.size()) {
    const uint32_t size = (uint32_t)raw[pos] | (uint32_t)raw[pos + 1] << 8 | (uint32_t)raw[pos + 2] << 16 | (uint32_t)raw[pos + 3] << 24;
    if (size == 0) { pos = (pos / kChunk + 1) * kChunk; continue; }   // padding: the next event starts at the next chunk
    if (size > kChunk || pos + 4 + size > raw.size()) throw std::runtime_error("corrupt TFileTransport event in " + path);
    out.insert(out.end(), raw.begin() + (long)pos + 4, raw.begin() + (long)(pos + 4 + size));
    pos += 4 + size;
  }
  return out;
}

struct EventWriter {
  FILE* f; size_t offset = 0;
  explicit EventWriter(const std::string& path) : f(std::fopen(path.c_str(), "wb")) { if (!f) throw std::runtime_error("cannot create " + path); }
  ~EventWriter() { if (f) std::fclose(f); }
  void write(const void* p, uint32_t n) {                              // one transport write() = one event
    if (n == 0) return;
    const size_t room = kChunk - offset % kChunk;
    if (4 + (size_t)n > room) { std::vector<uint8_t> z(room, 0); std::fwrite(z.data(), 1, room, f); offset += room; }
    const uint8_t h[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    std::fwrite(h, 1, 4, f); std::fwrite(p, 1, n, f); offset += 4 + (size_t)n;
  }
};

// ---- TBinaryProtocol ----
struct Reader {
  const std::vector<uint8_t>& b; size_t p = 0;
  explicit Reader(const std::vector<uint8_t>& bytes) : b(bytes) {}
  void need(size_t n) const { if (p + n > b.size()) throw std::runtime_error("truncated Thrift stream"); }
  uint8_t u8() { need(1); return b[p++]; }
  int16_t i16() { need(2); const int16_t v = (int16_t)((b[p] << 8) | b[p + 1]); p += 2; return v; }
  int32_t i32() { need(4); const uint32_t v = (uint32_t)b[p] << 24 | (uint32_t)b[p + 1] << 16 | (uint32_t)b[p + 2] << 8 | b[p + 3]; p += 4; return (int32_t)v; }
  int64_t i64() { need(8); uint64_t v = 0; for (int k = 0; k < 8; ++k) v = v << 8 | b[p + k]; p += 8; return (int64_t)v; }
  double f64() { const int64_t v = i64(); double d; std::memcpy(&d, &v, 8); return d; }
  bool boolean() { return u8() != 0; }
  void skip(uint8_t t, int depth = 0) {
    if (depth > 64) throw std::runtime_error("Thrift nesting too deep");
    switch (t) {
      case T_BOOL: case T_BYTE: need(1); p += 1; break;
      case T_I16: need(2); p += 2; break;
      case T_I32: need(4); p += 4; break;
      case T_I64: case T_DOUBLE: need(8); p += 8; break;
      case T_STRING: { const int32_t n = i32(); if (n < 0) throw std::runtime_error("negative length"); need((size_t)n); p += (size_t)n; break; }
      case T_STRUCT: for (;;) { const uint8_t ft = u8(); if (ft == T_STOP) break; (void)i16(); skip(ft, depth + 1); } break;
      case T_LIST: case T_SET: { const uint8_t et = u8(); const int32_t n = i32(); for (int32_t k = 0; k < n; ++k) skip(et, depth + 1); break; }
      case T_MAP: { const uint8_t kt = u8(), vt = u8(); const int32_t n = i32(); for (int32_t k = 0; k < n; ++k) { skip(kt, depth + 1); skip(vt, depth + 1); } break; }
      default: throw std::runtime_error("unknown Thrift type " + std::to_string((int)t));
    }
  }
  // list header: returns the count, checks the element type
  // A count is only believed as far as the bytes that are left can hold that many elements (a corrupt file must not
  // make the caller allocate gigabytes before the first element read throws).
  static size_t min_wire_size(uint8_t t) {
    switch (t) { case T_I16: return 2; case T_I32: case T_STRING: return 4; case T_I64: case T_DOUBLE: return 8; case T_LIST: case T_SET: return 5; case T_MAP: return 6; default: return 1; }
  }
  int32_t list(uint8_t want) {
    const uint8_t et = u8(); const int32_t n = i32();
    if (n < 0 || (n > 0 && et != want)) throw std::runtime_error("unexpected list element type");
    if ((size_t)n > (b.size() - p) / min_wire_size(et)) throw std::runtime_error("truncated Thrift stream (list count exceeds the remaining bytes)");
    return n;
  }
  std::vector<double> doubles() { const int32_t n = list(T_DOUBLE); std::vector<double> v((size_t)n); for (auto& x : v) x = f64(); return v; }
  std::vector<std::vector<double>> double_lists() { const int32_t n = list(T_LIST); std::vector<std::vector<double>> v((size_t)n); for (auto& x : v) x = doubles(); return v; }
};

struct Writer {
  EventWriter& t;
  explicit Writer(EventWriter& tr) : t(tr) {}
  void u8(uint8_t v) { t.write(&v, 1); }
  void i16(int16_t v) { const uint8_t b[2] = {(uint8_t)((uint16_t)v >> 8), (uint8_t)v}; t.write(b, 2); }
  void i32(int32_t v) { const uint32_t u = (uint32_t)v; const uint8_t b[4] = {(uint8_t)(u >> 24), (uint8_t)(u >> 16), (uint8_t)(u >> 8), (uint8_t)u}; t.write(b, 4); }
  void f64(double d) { uint64_t u; std::memcpy(&u, &d, 8); uint8_t b[8]; for (int k = 0; k < 8; ++k) b[k] = (uint8_t)(u >> (56 - 8 * k)); t.write(b, 8); }
  void field(uint8_t type, int16_t id) { u8(type); i16(id); }
  void stop() { u8(T_STOP); }
  void list(uint8_t et, size_t n) { u8(et); i32((int32_t)n); }
  void doubles(const std::vector<double>& v) { list(T_DOUBLE, v.size()); for (double x : v) f64(x); }
  void double_lists(const std::vector<std::vector<double>>& v) { list(T_LIST, v.size()); for (const auto& x : v) doubles(x); }
};

// ---- sfm.thrift structs ----
inline void read(Reader& r, ObservationRef& o) {                        // sfm.thrift:33-42
  o = ObservationRef();
  for (;;) {
    const uint8_t t = r.u8(); if (t == T_STOP) break;
    const int16_t id = r.i16();
    if (id == 1 && t == T_I32) o.frame = r.i32();
    else if (id == 2 && t == T_I32) o.obs = r.i32();
    else if (id == 3 && t == T_BOOL) o.valid = r.boolean();
    else r.skip(t);
  }
}
inline void read(Reader& r, Observation& o) {                           // sfm.thrift:13-22
  o = Observation();
  for (;;) {
    const uint8_t t = r.u8(); if (t == T_STOP) break;
    const int16_t id = r.i16();
    if (id == 1 && t == T_DOUBLE) o.x = r.f64();
    else if (id == 2 && t == T_DOUBLE) o.y = r.f64();
    else if (id == 5 && t == T_LIST) { const int32_t n = r.list(T_STRUCT); o.matches.resize((size_t)n); for (auto& m : o.matches) read(r, m); o.__isset.matches = true; }
    else if (id == 6 && t == T_I32) { o.track = r.i32(); o.__isset.track = true; }
    else r.skip(t);                                                     // descriptor (3), color (4)
  }
}
inline void read(Reader& r, Track& k) {                                 // sfm.thrift:25-30
  k = Track();
  for (;;) {
    const uint8_t t = r.u8(); if (t == T_STOP) break;
    const int16_t id = r.i16();
    if (id == 1 && t == T_LIST) { const int32_t n = r.list(T_STRUCT); k.obs.resize((size_t)n); for (auto& m : k.obs) read(r, m); }
    else if (id == 2 && t == T_LIST) { k.pt = r.doubles(); k.__isset.pt = true; }
    else if (id == 4 && t == T_BOOL) k.valid = r.boolean();
    else r.skip(t);                                                     // color (3)
  }
}
inline void read(Reader& r, Frame& f) {                                 // sfm.thrift:45-56
  f = Frame();
  for (;;) {
    const uint8_t t = r.u8(); if (t == T_STOP) break;
    const int16_t id = r.i16();
    if (id == 1 && t == T_LIST) { const int32_t n = r.list(T_STRUCT); f.obs.resize((size_t)n); for (auto& o : f.obs) read(r, o); }
    else if (id == 2 && t == T_LIST) { f.poses = r.double_lists(); f.__isset.poses = true; }
    else if (id == 3 && t == T_LIST) { f.cam = r.doubles(); f.__isset.cam = true; }
    else if (id == 4 && t == T_LIST) { f.priorPoses = r.double_lists(); f.__isset.priorPoses = true; }
    else r.skip(t);
  }
}
inline void read(Reader& r, Session& s) {                               // sfm.thrift:62-74
  s = Session();
  for (;;) {
    const uint8_t t = r.u8(); if (t == T_STOP) break;
    const int16_t id = r.i16();
    if (id == 1 && t == T_LIST) s.cam = r.doubles();
    else if (id == 2 && t == T_LIST) { const int32_t n = r.list(T_STRUCT); s.frames.resize((size_t)n); for (auto& f : s.frames) read(r, f); }
    else if (id == 3 && t == T_LIST) { const int32_t n = r.list(T_STRUCT); s.tracks.resize((size_t)n); for (auto& k : s.tracks) read(r, k); }
    else if (id == 4 && t == T_I32) s.rs = r.i32();
    else if (id == 5 && t == T_LIST) { const int32_t n = r.list(T_I32); s.scanlines.resize((size_t)n); for (auto& v : s.scanlines) v = r.i32(); }
    else if (id == 6 && t == T_I32) s.width = r.i32();
    else if (id == 7 && t == T_I32) s.height = r.i32();
    else r.skip(t);
  }
}

inline void write(Writer& w, const ObservationRef& o) {
  w.field(T_I32, 1); w.i32(o.frame); w.field(T_I32, 2); w.i32(o.obs); w.field(T_BOOL, 3); w.u8(o.valid ? 1 : 0); w.stop();
}
inline void write(Writer& w, const Observation& o) {
  w.field(T_DOUBLE, 1); w.f64(o.x); w.field(T_DOUBLE, 2); w.f64(o.y);
  if (o.__isset.matches) { w.field(T_LIST, 5); w.list(T_STRUCT, o.matches.size()); for (const auto& m : o.matches) write(w, m); }
  if (o.__isset.track) { w.field(T_I32, 6); w.i32(o.track); }
  w.stop();
}
inline void write(Writer& w, const Track& k) {
  w.field(T_LIST, 1); w.list(T_STRUCT, k.obs.size()); for (const auto& m : k.obs) write(w, m);
  if (k.__isset.pt) { w.field(T_LIST, 2); w.doubles(k.pt); }
  w.field(T_BOOL, 4); w.u8(k.valid ? 1 : 0); w.stop();
}
inline void write(Writer& w, const Frame& f) {
  w.field(T_LIST, 1); w.list(T_STRUCT, f.obs.size()); for (const auto& o : f.obs) write(w, o);
  if (f.__isset.poses) { w.field(T_LIST, 2); w.double_lists(f.poses); }
  if (f.__isset.cam) { w.field(T_LIST, 3); w.doubles(f.cam); }
  if (f.__isset.priorPoses) { w.field(T_LIST, 4); w.double_lists(f.priorPoses); }
  w.stop();
}
inline void write(Writer& w, const Session& s) {
  w.field(T_LIST, 1); w.doubles(s.cam);
  w.field(T_LIST, 2); w.list(T_STRUCT, s.frames.size()); for (const auto& f : s.frames) write(w, f);
  w.field(T_LIST, 3); w.list(T_STRUCT, s.tracks.size()); for (const auto& k : s.tracks) write(w, k);
  if (s.rs != GLOBAL) { w.field(T_I32, 4); w.i32(s.rs); }                 // optional; unset == 0 == GLOBAL
  w.field(T_LIST, 5); w.list(T_I32, s.scanlines.size()); for (int32_t v : s.scanlines) w.i32(v);
  w.field(T_I32, 6); w.i32(s.width); w.field(T_I32, 7); w.i32(s.height);
  w.stop();
}

}  // namespace cache_detail

namespace cache_detail {
// indices a loaded Session carries are used unchecked by CeresHandler::Add (sess.frames[ref.frame].obs[ref.obs],
// sess.getTrack(o.track)): refuse a file whose references point outside what it holds
inline void check_indices(const Frame&) {}
inline void check_indices(const Session& s) {
  auto ref_ok = [&](const ObservationRef& r) { return r.frame >= 0 && (size_t)r.frame < s.frames.size() && r.obs >= 0 && (size_t)r.obs < s.frames[(size_t)r.frame].obs.size(); };
  for (const Frame& f : s.frames) for (const Observation& o : f.obs) {
    if (o.__isset.track && (o.track < 0 || (size_t)o.track >= s.tracks.size())) throw std::runtime_error("Session cache: observation refers to a track that does not exist");
    for (const ObservationRef& r : o.matches) if (!ref_ok(r)) throw std::runtime_error("Session cache: match refers to an observation that does not exist");
  }
  for (const Track& t : s.tracks) for (const ObservationRef& r : t.obs) if (!ref_ok(r)) throw std::runtime_error("Session cache: track refers to an observation that does not exist");
}
}  // namespace cache_detail

// VideoSfMCache::unserialize(T&) for T = Session / Frame.  Throws std::runtime_error on unreadable input.
template <class T>
inline void loadCache(const std::string& path, T& obj) {
  const std::vector<uint8_t> bytes = cache_detail::read_events(path);
  cache_detail::Reader r(bytes);
  cache_detail::read(r, obj);
  cache_detail::check_indices(obj);
}
// VideoSfMCache::save(const T&): the same event / protocol layout the reference's serialize() produces
template <class T>
inline void saveCache(const std::string& path, const T& obj) {
  cache_detail::EventWriter t(path);
  cache_detail::Writer w(t);
  cache_detail::write(w, obj);
}

}  // namespace rsba_amd
