"""Code digest of the kernel sources (csrc/*.hip, csrc/*.h, include/deepq_hip.h): sha256 over the files' CODE -- comments and whitespace stripped, so that
editing a comment moves nothing.  Three users: build.py compiles it INTO the library (dq_build_digest), the tests compare that with the tree's -- the .so that
travels to a GPU box is the one these sources build --, and tools/pmc_traffic.sh stamps a PMC pass with it (bench_loop.pmc_traffic).  No torch import."""
import glob
import hashlib
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def code_only(text):
    """A C / HIP source with comments and all whitespace removed: what the compiler sees, up to token spacing."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def csrc_digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))) + \
            [os.path.join(HERE, "..", "include", "deepq_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()
