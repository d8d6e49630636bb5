#!/usr/bin/env python3
"""sha256[:16] over the product's device + ABI sources (rsba_amd/csrc/*, include/rsba_amd.h), in path order.

The PMC summaries under profiles/ carry the stamp of the code they were measured on; bench.py computes the same stamp at run
time and marks `roofline.traffic` / `mfma_busy_frac` as stale when the two differ (the GPU box has no .git to ask).
    python tools/source_stamp.py                      -> prints the stamp
    python tools/source_stamp.py --tag FILE.json ...  -> writes "source_sha16" (and "commit" when git knows one) into the summaries
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rsba_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".h")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "rsba_amd.h"))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def commit():
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None
    except Exception:  # noqa: BLE001
        return None


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--tag":
        for path in sys.argv[2:]:
            with open(path) as fh:
                d = json.load(fh)
            d.setdefault("source_sha16", source_sha16())
            c = commit()
            if c:
                d.setdefault("commit", c)
            with open(path, "w") as fh:
                json.dump(d, fh, indent=1)
    else:
        print(source_sha16())
