"""Builds csrc/*.hip + runtime.cpp into csrc/libfdbm_hip.so for gfx950 (in-tree, so the
.so travels with the repository snapshot to the GPU box)."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libfdbm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.cpp")))


STAMP = os.path.join(CSRC, ".build_stamp")


def source_hash():
    """sha256 over every source, header and the flags: what the built library is a function of."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        sorted(glob.glob(os.path.join(os.path.dirname(CSRC), "..", "include", "*.h")))
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def is_stale():
    """The library is current iff it exists and was built from exactly these sources (hash stamp, not mtimes:
    a shipped .so that merely LOOKS newer than the sources must not pass for a build)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_hash()


def build(force=False, verbose=True):
    """-> path of the library.  force (or FDBM_FORCE_BUILD=1): compile even when the stamp matches."""
    force = force or os.environ.get("FDBM_FORCE_BUILD", "") not in ("", "0")
    if not force and not is_stale():
        if verbose:
            print(f"reused {LIB} (source hash {source_hash()[:12]} matches its build stamp)", flush=True)
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd = [HIPCC, "-x", "hip"] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"FAILED: {src}\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    if verbose:
        print(f"compiled {len(objs)} translation units -> {LIB} (source hash {source_hash()[:12]})", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
