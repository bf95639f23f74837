"""In-tree build of the gfx950 engine: gomilp_amd/libgomilp_hip.so (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgomilp_hip.so")
SOURCES = ["simplex_kernels.hip", "fused_kernels.hip", "lu_kernels.hip", "lu_compressed.hip", "tableau_kernels.hip", "bt_kernels.hip", "btg_kernels.hip", "batch_kernels.hip", "general_kernels.hip", "engine.cpp", "engine_batch.cpp", "engine_tableau.cpp", "engine_general.cpp", "c_api.cpp", "comm.cpp"]
HEADERS = ["device_types.h", "kernels_common.h", "engine.hpp", "engine_work.hpp", "engine_batch.hpp", os.path.join(ROOT, "include", "gomilp_lp.h")]
# -ffp-contract=off: the final basis solve must round every multiply and add separately, like the
# reference's SSE2 kernels (DESIGN.md "bit-exact final solve"); the streaming kernels are HBM-bound
# and do not miss the FMAs.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-pragma-unroll-threshold=1000000",
         "-Wno-unused-result", "-Wno-unused-value", "-ldl", "-I" + os.path.join(ROOT, "include")]


def _hipcc() -> str | None:
    return shutil.which("hipcc") or (os.path.exists("/opt/rocm/bin/hipcc") and "/opt/rocm/bin/hipcc") or None


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile to a temporary file next to the target and rename it into place, under a file lock: several ranks that import
    the package on a fresh checkout at once (torchrun) neither compile concurrently into one file nor load a half-written one."""
    if not (force or stale()):
        return LIB
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found and %s is missing or out of date" % LIB)
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():      # another process built it while this one waited for the lock
                return LIB
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
