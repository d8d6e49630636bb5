"""The input tables of the reference's only test (src/rsba/test/mat_test.cc:171-214): 11 poses x 18 points x 6 cameras,
including the +-_EPS poses / points and the k1 = +-_EPS cameras.  Data, not code: they are the reference-authored
known-answer inputs for w2c / w2i / distort / validate, replayed against the CPU oracle in
tests/test_oracle_kat.py and against the HIP path in tests/test_gpu_kat.py."""
import math

import numpy as np

EPS = np.finfo(np.float64).eps
PI2 = math.pi / 2

POSE_REF = [0, 0, 0, 20, 20, 0]
POSES = [[0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 1, 1], [0, 0, PI2, 20, 20, 20], [0, PI2, PI2, -2, 20, 20],
         [PI2, PI2, PI2, -2, -2, 20], [-1, -1, -1, -2, -2, -2], [-PI2, -1, -1, -20, -2, -2],
         [0.5, -PI2, -1, -2, -20, -2], [0.5, 0.5, -PI2, 0.2, -2, -20], [EPS] * 6, [-EPS] * 6]
PTS = [[10, 10, 10], [100, 0, 1], [0, 100, 1], [0, 0, 100], [-100, 0, 1], [0, -100, 1], [0, 0, -100], [0, 0, -1],
       [0, 0, 0], [1, 1, 1], [-1, -1, -1], [0.1, 0.1, 0.1], [100, 100, 100], [-100, -100, -100],
       [-0.39, 1.25, 2014], [EPS, EPS, EPS], [EPS, EPS, -EPS], [-EPS, -EPS, -EPS]]
CAMS = [[0.1, 0.1, 0, 0, 0, 0, 0, 0, 0], [100, 100, 0, 0, 0, 0, 0, 0, 0], [500, 500, 0, 0, 0, 0, 0, 640, 480],
        [100, 100, EPS, 0, 0, 0, 0, 0, 0], [500, 500, -EPS, -EPS, 0, 0, 0, 0, 0], [860, 860, 0.001, 0, 0, 0, 0, 100, 200]]


def deep_cases(oracle):
    """The (pose, point, cam) triples that reach mat_test.cc:280-281 — the control flow of mat_test.cc:215-279 replayed with
    the oracle's helpers — with the two observations validated there: img = w2i(cam, pose, pt), img_ref = w2i(cam, POSE_REF, tri)."""
    for pose in POSES:
        for pt in PTS:
            for cam in CAMS:
                ok1, d1 = oracle.direction_world(pose, pt)
                ok2, _ = oracle.c2direction(pose, oracle.w2c(pose, pt))
                if not (ok1 and ok2):
                    continue
                oki, img = oracle.w2i(cam, pose, pt)
                if not oki or not oracle.direction_pixel(cam, pose, img)[0]:
                    continue
                okr, d_ref = oracle.c2direction(POSE_REF, oracle.w2c(POSE_REF, pt))
                if not okr:
                    continue
                okt, tri = oracle.triangulate(pose[3:], d1, POSE_REF[3:], d_ref)
                if not okt:
                    continue
                okw, img_ref = oracle.w2i(cam, POSE_REF, tri)
                if not okw:
                    continue
                yield pose, pt, cam, img, img_ref
