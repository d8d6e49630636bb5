"""The Thrift cache loader (second half; include/rsba/session_cache.hpp) against an independent
Python encoder of the same wire formats (tests/thrift_encode.py), hand-assembled byte vectors, and its own writer."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import thrift_encode as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "examples", "session_cache_tool")


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "session_cache_tool.cpp"), "-o", TOOL], check=True)
    return TOOL


def dump(tool, kind, path):
    r = subprocess.run([tool, "dump", kind, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def test_hand_assembled_frame_bytes(tool, tmp_path):
    """Byte for byte: one observation (x = 1.5, y = -2, track 7), one pose list, in two events."""
    obs = bytes([4, 0, 1]) + bytes.fromhex("3ff8000000000000") + bytes([4, 0, 2]) + bytes.fromhex("c000000000000000") + bytes([8, 0, 6, 0, 0, 0, 7, 0])
    payload = bytes([15, 0, 1, 12, 0, 0, 0, 1]) + obs + bytes([15, 0, 2, 15, 0, 0, 0, 1, 4, 0, 0, 0, 2]) + bytes.fromhex("3ff0000000000000") \
        + bytes.fromhex("4000000000000000") + bytes([0])
    cut = 11
    data = struct.pack("<I", cut) + payload[:cut] + struct.pack("<I", len(payload) - cut) + payload[cut:]
    (tmp_path / "f.cache").write_bytes(data)
    f = dump(tool, "frame", tmp_path / "f.cache")
    assert f == {"obs": [{"x": 1.5, "y": -2.0, "track": 7}], "poses": [[1.0, 2.0]]}


def test_session_round_trip_with_skipped_fields_and_random_event_splits(tool, tmp_path):
    from rsba_amd.scene import make_scene
    prob = make_scene(6, 80, rolling=True, seed=5).problem
    payload = T.session_of_problem(prob, descriptors=True)
    (tmp_path / "s.cache").write_bytes(T.file_events(payload, np.random.default_rng(3)))
    s = dump(tool, "session", tmp_path / "s.cache")
    assert s["cam"] == list(prob.intrinsics[0]) and s["rs"] == prob.shutter and s["scanlines"] == list(prob.scanlines)
    assert (s["width"], s["height"]) == (1280, 720) and len(s["frames"]) == prob.num_frames and len(s["tracks"]) == prob.num_points
    assert np.array_equal(np.array([f["poses"] for f in s["frames"]]), prob.poses)                 # doubles survive bit for bit
    assert np.array_equal(np.array([t["pt"] for t in s["tracks"]]), prob.points) and all(t["valid"] for t in s["tracks"])
    xy = np.array([[o["x"], o["y"]] for f in s["frames"] for o in f["obs"]])
    order = np.argsort(prob.obs_frame, kind="stable")
    assert np.array_equal(xy, prob.obs_xy[order])
    assert [o["track"] for f in s["frames"] for o in f["obs"]] == list(prob.obs_point[order])
    for j, t in enumerate(s["tracks"]):                                  # back references resolve to the observation of that track
        for ref in t["obs"]:
            assert s["frames"][ref["frame"]]["obs"][ref["obs"]]["track"] == j and ref["valid"]
    # the writer emits the reference's layout (one event per primitive); reading it back gives the same session
    r = subprocess.run([tool, "copy", "session", str(tmp_path / "s.cache"), str(tmp_path / "s2.cache")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert dump(tool, "session", tmp_path / "s2.cache") == s
    raw = (tmp_path / "s2.cache").read_bytes()
    assert raw[:4] == struct.pack("<I", 1) and raw[4] == 15           # first event: the 1-byte field type (LIST) of Session.cam


def test_chunk_boundary_padding_and_global_shutter_default(tool, tmp_path):
    fr = T.frame([T.observation(3.0, 4.0, matches=[(0, 0, False)])], cam=[float(k) for k in range(9)], prior_poses=[[0.5] * 6])
    payload = T.session([0.0] * 9, [fr], [T.track([(0, 0, True)])], 0, [0, 0], 640, 480)
    # tiny chunks force zero padding between events
    (tmp_path / "s.cache").write_bytes(T.file_events(payload, np.random.default_rng(1), chunk=16 * 1024 * 1024, max_event=9))
    s = dump(tool, "session", tmp_path / "s.cache")
    assert s["rs"] == 0 and s["frames"][0]["cam"] == [float(k) for k in range(9)] and s["frames"][0]["priorPoses"] == [[0.5] * 6]
    assert s["frames"][0]["obs"][0]["matches"] == [{"frame": 0, "obs": 0, "valid": False}] and "track" not in s["frames"][0]["obs"][0]
    assert "pt" not in s["tracks"][0] and s["tracks"][0]["valid"] is False
    # an event that would straddle the 16 MiB boundary: zero fill, next event at the boundary
    head = payload[:20]
    pad_at = 16 * 1024 * 1024
    data = bytearray(struct.pack("<I", len(head)) + head)
    data += b"\x00" * (pad_at - len(data))
    data += struct.pack("<I", len(payload) - 20) + payload[20:]
    (tmp_path / "big.cache").write_bytes(bytes(data))
    assert dump(tool, "session", tmp_path / "big.cache") == s


def test_corrupt_files_are_refused(tool, tmp_path):
    (tmp_path / "a.cache").write_bytes(struct.pack("<I", 50) + b"\x0f\x00\x01")           # event longer than the file
    assert subprocess.run([tool, "dump", "frame", str(tmp_path / "a.cache")], capture_output=True).returncode == 1
    (tmp_path / "b.cache").write_bytes(struct.pack("<I", 3) + b"\x0f\x00\x01")            # truncated list header
    assert subprocess.run([tool, "dump", "frame", str(tmp_path / "b.cache")], capture_output=True).returncode == 1
    assert subprocess.run([tool, "dump", "frame", str(tmp_path / "missing.cache")], capture_output=True).returncode == 1


def test_huge_list_counts_and_dangling_indices_are_refused(tool, tmp_path):
    """A list count is believed only as far as the remaining bytes can hold it (no multi-GB allocation from 9 bytes), and
    a Session whose track / observation references point outside the file is refused at load time (CeresHandler::Add
    would index with them unchecked)."""
    # Frame: field 3 (cam) = list<double> with a count of 2^30 and no elements behind it
    payload = T.fld(T.LIST, 3) + struct.pack(">bi", T.DOUBLE, 1 << 30)
    (tmp_path / "huge.cache").write_bytes(T.file_events(payload))
    r = subprocess.run([tool, "dump", "frame", str(tmp_path / "huge.cache")], capture_output=True, text=True)
    assert r.returncode == 1 and "list count" in r.stderr
    cam = [800, 800, 0, 0, 0, 0, 0, 640, 360]
    pose = [[0.0] * 6]
    ok = T.session(cam, [T.frame([T.observation(1, 2, track=0)], poses=pose)], [T.track([(0, 0, True)], pt=[0, 0, 5], valid=True)], 0, [0, 0], 1280, 720)
    (tmp_path / "ok.cache").write_bytes(T.file_events(ok))
    assert subprocess.run([tool, "dump", "session", str(tmp_path / "ok.cache")], capture_output=True).returncode == 0
    bad_track = T.session(cam, [T.frame([T.observation(1, 2, track=3)], poses=pose)], [T.track([(0, 0, True)], pt=[0, 0, 5], valid=True)], 0, [0, 0], 1280, 720)
    bad_ref = T.session(cam, [T.frame([T.observation(1, 2, track=0)], poses=pose)], [T.track([(0, 7, True)], pt=[0, 0, 5], valid=True)], 0, [0, 0], 1280, 720)
    bad_match = T.session(cam, [T.frame([T.observation(1, 2, track=0, matches=[(4, 0, True)])], poses=pose)], [T.track([(0, 0, True)], pt=[0, 0, 5], valid=True)], 0, [0, 0], 1280, 720)
    for name, blob in (("t", bad_track), ("r", bad_ref), ("m", bad_match)):
        (tmp_path / f"{name}.cache").write_bytes(T.file_events(blob))
        r = subprocess.run([tool, "dump", "session", str(tmp_path / f"{name}.cache")], capture_output=True, text=True)
        assert r.returncode == 1 and "does not exist" in r.stderr, (name, r.stderr)
