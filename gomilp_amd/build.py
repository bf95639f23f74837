"""In-tree build of the gfx950 engine: gomilp_amd/libgomilp_hip.so (hipcc cross-compiles without a GPU).

Every source is compiled to an object of its own under gomilp_amd/build/ (in parallel, only what is out of date) and the
objects are linked into the shared library."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# GOMILP_DEBUG_BUILD=1: the diagnostic flavour (-DGOMILP_DEBUG: stderr notes behind GOMILP_DEBUG_* environment variables, the
# fault-injection knob "bt_fault", the panel-shape override GOMILP_LUC_CFG) as a library and object directory of its own; the
# product library contains none of these hooks
DEBUG = os.environ.get("GOMILP_DEBUG_BUILD", "") not in ("", "0")
OBJ = os.path.join(HERE, "build_debug" if DEBUG else "build")
LIB = os.path.join(HERE, "libgomilp_hip_debug.so" if DEBUG else "libgomilp_hip.so")
SOURCES = ["simplex_kernels.hip", "fused_kernels.hip", "lu_kernels.hip", "lu_compressed.hip", "lu_cross.hip", "tableau_kernels.hip", "bt_kernels.hip", "btg_kernels.hip", "btr_kernels.hip",
           "batch_kernels.hip", "res_kernels.hip", "general_kernels.hip", "general_block.hip", "engine.cpp", "engine_batch.cpp", "engine_tableau.cpp", "engine_general.cpp", "gonum_cond.cpp", "c_api.cpp", "comm.cpp"]
HEADERS = ["device_types.h", "kernels_common.h", "bt_loop.h", "batch_dev.h", "engine.hpp", "engine_work.hpp", "engine_batch.hpp", os.path.join(ROOT, "include", "gomilp_lp.h")]
# -ffp-contract=off: the final basis solve must round every multiply and add separately, like the
# reference's SSE2 kernels (DESIGN.md "bit-exact final solve"); the streaming kernels are HBM-bound
# and do not miss the FMAs.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-pragma-unroll-threshold=1000000",
          "-Wno-unused-result", "-Wno-unused-value", "-Wno-return-type-c-linkage", "-I" + os.path.join(ROOT, "include")]
if DEBUG:
    CFLAGS.append("-DGOMILP_DEBUG")
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
STAMP = os.path.join(OBJ, "flags.stamp")


def _hipcc() -> str | None:
    return shutil.which("hipcc") or (os.path.exists("/opt/rocm/bin/hipcc") and "/opt/rocm/bin/hipcc") or None


def _deps() -> list[str]:
    return [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def _obj(src: str) -> str:
    return os.path.join(OBJ, src + ".o")


_stamp_cache: str | None = None


def _stamp() -> str:
    """What the objects depend on besides their sources: compiler version, flags, the source / header lists.  Bit-exactness
    against the reference hangs on -ffp-contract=off, so an object built with other flags must never be reused."""
    global _stamp_cache
    if _stamp_cache is None:
        hipcc = _hipcc()
        ver = ""
        if hipcc:
            try:
                ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
            except Exception:
                ver = "unknown"
        h = hashlib.sha256()
        for part in [ver] + CFLAGS + LDFLAGS + SOURCES + HEADERS:
            h.update(part.encode())
            h.update(b"\0")
        _stamp_cache = h.hexdigest()
    return _stamp_cache


def _stamp_ok() -> bool:
    if _hipcc() is None:   # nothing could be rebuilt anyway (a box without the compiler uses the shipped library)
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() == _stamp()
    except OSError:
        return False


def _obj_stale(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o) or not _stamp_ok():
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + _deps())


def stale() -> bool:
    if not os.path.exists(LIB) or not _stamp_ok():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _deps()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is out of date (everything with force=True), link into a temporary file next to the target and rename it
    into place, under a file lock: several ranks that import the package on a fresh checkout at once (torchrun) neither
    compile concurrently into one file nor load a half-written one."""
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
            os.makedirs(OBJ, exist_ok=True)
            todo = [s for s in SOURCES if force or _obj_stale(s)]
            if os.path.exists(STAMP) and not _stamp_ok():
                os.remove(STAMP)   # (an interrupted rebuild must not leave old objects under a new stamp)

            def compile_one(src: str) -> None:
                tmp = "%s.tmp.%d" % (_obj(src), os.getpid())
                cmd = [hipcc] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.run(cmd, check=True)
                    os.replace(tmp, _obj(src))
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)

            with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1)) or 1) as ex:
                list(ex.map(compile_one, todo))
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [hipcc] + LDFLAGS + [_obj(s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB)
                with open(STAMP, "w") as f:
                    f.write(_stamp() + "\n")
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
