#!/usr/bin/env python3
"""What a profile was measured on, as JSON: the commit (the GPU box has no .git, so the caller passes it in as
TPL_GIT_HEAD, e.g. `gpurun -- "TPL_GIT_HEAD=$(tools/git_head.sh) tools/profile_step.sh ..."`; here in the build container
git is asked directly), the digest of the library's sources (`_lib._source_digest()`, the same one the stamp beside
lib/libtetris_piclim.so carries) and whether the library on disk was built from exactly those sources."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stamp():
    import tetris_piclim as T
    head = os.environ.get("TPL_GIT_HEAD")
    if not head:
        try:
            head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
            dirty = subprocess.run(["git", "status", "--porcelain", "--untracked-files=no"], cwd=ROOT, capture_output=True,
                                   text=True).stdout.strip()
            head = (head + ("+dirty" if dirty else "")) or None
        except OSError:
            head = None
    digest = T._lib._source_digest()
    built = None
    try:
        built = open(T._lib.LIB_PATH + ".sha256").read().strip()
    except OSError:
        pass
    return {"git_head": head, "source_digest": digest, "library": os.path.relpath(T._lib.LIB_PATH, ROOT),
            "library_built_from_these_sources": built == digest, "taken": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}


if __name__ == "__main__":
    print(json.dumps(stamp()))
