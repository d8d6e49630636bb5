"""An independent Python encoder of rsba's cache files for the tests: the structs of sfm.thrift in Thrift's
TBinaryProtocol, wrapped in TFileTransport events (format notes: include/rsba/session_cache.hpp).  Written from the
published wire formats, not from the C++ reader, so that the two pin each other."""
import struct

import numpy as np

BOOL, DOUBLE, I32, STRING, STRUCT, LIST = 2, 4, 8, 11, 12, 15


def fld(t, i):
    return struct.pack(">bh", t, i)


def lst(t, items):
    return struct.pack(">bi", t, len(items)) + b"".join(items)


def dbl(v):
    return struct.pack(">d", float(v))


def doubles(v):
    return lst(DOUBLE, [dbl(x) for x in v])


def obs_ref(frame, obs, valid):
    return fld(I32, 1) + struct.pack(">i", frame) + fld(I32, 2) + struct.pack(">i", obs) + fld(BOOL, 3) + struct.pack(">b", int(valid)) + b"\x00"


def observation(x, y, track=None, matches=None, descriptor=None, color=None):
    out = fld(DOUBLE, 1) + dbl(x) + fld(DOUBLE, 2) + dbl(y)
    if descriptor is not None:
        out += fld(STRING, 3) + struct.pack(">i", len(descriptor)) + descriptor
    if color is not None:
        out += fld(STRING, 4) + struct.pack(">i", len(color)) + color
    if matches is not None:
        out += fld(LIST, 5) + lst(STRUCT, [obs_ref(*m) for m in matches])
    if track is not None:
        out += fld(I32, 6) + struct.pack(">i", track)
    return out + b"\x00"


def track(refs, pt=None, valid=False, color=None):
    out = fld(LIST, 1) + lst(STRUCT, [obs_ref(*r) for r in refs])
    if pt is not None:
        out += fld(LIST, 2) + doubles(pt)
    if color is not None:
        out += fld(STRING, 3) + struct.pack(">i", len(color)) + color
    return out + fld(BOOL, 4) + struct.pack(">b", int(valid)) + b"\x00"


def frame(obs, poses=None, cam=None, prior_poses=None):
    out = fld(LIST, 1) + lst(STRUCT, obs)
    if poses is not None:
        out += fld(LIST, 2) + lst(LIST, [doubles(p) for p in poses])
    if cam is not None:
        out += fld(LIST, 3) + doubles(cam)
    if prior_poses is not None:
        out += fld(LIST, 4) + lst(LIST, [doubles(p) for p in prior_poses])
    return out + b"\x00"


def session(cam, frames, tracks, rs, scanlines, width, height):
    out = fld(LIST, 1) + doubles(cam) + fld(LIST, 2) + lst(STRUCT, frames) + fld(LIST, 3) + lst(STRUCT, tracks)
    if rs:
        out += fld(I32, 4) + struct.pack(">i", rs)
    out += fld(LIST, 5) + lst(I32, [struct.pack(">i", v) for v in scanlines])
    return out + fld(I32, 6) + struct.pack(">i", width) + fld(I32, 7) + struct.pack(">i", height) + b"\x00"


def file_events(payload: bytes, rng=None, chunk=16 * 1024 * 1024, max_event=64):
    """TFileTransport framing: [u32 LE size][bytes] events of random sizes, zero padding to the chunk boundary when an
    event would straddle it."""
    rng = rng or np.random.default_rng(0)
    out = bytearray()
    pos = 0
    while pos < len(payload):
        n = int(min(len(payload) - pos, rng.integers(1, max_event + 1)))
        room = chunk - len(out) % chunk
        if 4 + n > room:
            out += b"\x00" * room
        out += struct.pack("<I", n) + payload[pos:pos + n]
        pos += n
    return bytes(out)


def session_of_problem(prob, descriptors=False):
    """The Session CeresHandler works on, for a BAProblem (calibrated, shared intrinsics)."""
    F, M = prob.num_frames, prob.num_points
    per_frame = [[] for _ in range(F)]
    refs = [[] for _ in range(M)]
    for i in range(prob.num_observations):
        f, j = int(prob.obs_frame[i]), int(prob.obs_point[i])
        refs[j].append((f, len(per_frame[f]), True))
        per_frame[f].append(observation(prob.obs_xy[i, 0], prob.obs_xy[i, 1], track=j,
                                        descriptor=bytes(range(16)) if descriptors else None, color=b"\x01\x02\x03" if descriptors else None))
    frames = [frame(per_frame[f], poses=prob.poses[f]) for f in range(F)]
    tracks = [track(refs[j], pt=prob.points[j], valid=True, color=b"\x09\x08\x07" if descriptors else None) for j in range(M)]
    return session(prob.intrinsics[0], frames, tracks, int(prob.shutter), list(prob.scanlines), 1280, 720)
