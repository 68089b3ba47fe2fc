"""Builds libmadeleine_amd.so (gfx950) in-tree with hipcc.  No torch headers, no libtorch linkage:
the library is a plain C-ABI shared object (include/madeleine_amd.h)."""
import contextlib
import fcntl
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmadeleine_amd.so")
HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "madeleine_amd.h"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    return [HEADER] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise HipccMissing("hipcc not found: cannot build libmadeleine_amd.so")


class HipccMissing(RuntimeError):
    pass


@contextlib.contextmanager
def _build_lock(objdir):
    """Serialises concurrent builders (every rank of a torchrun launch on a fresh clone calls lib() at once): one
    process compiles, the others wait on the lock and then find everything fresh."""
    with open(os.path.join(objdir, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link the shared library.  Returns its path.  Multi-process safe: a file
    lock serialises builders, objects and the library are written to per-process temp names and renamed into place."""
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    with _build_lock(objdir):
        return _build_locked(force, verbose, objdir)


def _build_locked(force, verbose, objdir):
    srcs, deps = sources(), _deps()
    if not force and os.path.exists(LIB) and not _stale(LIB, srcs + deps):
        return LIB          # another process built it while we waited for the lock
    hipcc = hipcc_path()
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        tmp = "%s.%d.tmp" % (o, os.getpid())
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            with contextlib.suppress(OSError):
                os.remove(tmp)
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        os.replace(tmp, o)
        if verbose:
            print("built", os.path.basename(o), file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        tmp = "%s.%d.tmp" % (LIB, os.getpid())
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            with contextlib.suppress(OSError):
                os.remove(tmp)
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        os.replace(tmp, LIB)
    return LIB


def is_fresh() -> bool:
    if not os.path.exists(LIB):
        return False
    return not _stale(LIB, sources() + _deps())


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
