"""Build of the HIP library (gfx950 only).  `python -m` free: called by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libltompc.so")
SOURCES = ["ltompc.hip"] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "ltompc.h"), os.path.join("..", "_build.py")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> csrc/libltompc.so (in-tree, so that it travels with the repo snapshot).

    Several processes may call this at once (one rank per GPU under torchrun): the build is serialised with a file lock
    and written to a temporary name first, so that nobody loads a half-written library."""
    if not force and not needs_build():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():  # (somebody else may have built it while we waited)
                hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
                tmp = LIB + f".tmp{os.getpid()}"
                cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=on",
                       os.path.join(CSRC, "ltompc.hip"), "-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB
