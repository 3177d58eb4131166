"""Builds libprotgen_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(HERE, "libprotgen_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "rowwise.hip", "misc.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "-I", INCLUDE]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newer(a, b):
    return os.path.getmtime(a) > os.path.getmtime(b)


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hdrs


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = _deps()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        stale = force or not os.path.exists(o) or _newer(s, o) or any(_newer(h, o) for h in hdrs)
        if stale:
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        if verbose:
            print("compiled", os.path.basename(s), file=sys.stderr)

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(LIB_PATH) or any(_newer(o, LIB_PATH) for o in objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
