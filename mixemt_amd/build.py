"""
In-tree build of libmixemt_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m mixemt_amd.build [--force] [-D NAME=VALUE ...]

The shared object lands in mixemt_amd/lib/ (git-ignored, but it travels with
the tree to the GPU box).  A stamp file holding the hash of the sources and
flags makes repeat builds a no-op.
"""

import hashlib
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SRC = os.path.join(_PKG, "csrc", "mixemt_hip.hip")
HDRS = [os.path.join(_ROOT, "include", "mixemt_hip.h"), os.path.join(_ROOT, "include", "mixemt_hip_tuning.h")]
LIB_DIR = os.path.join(_PKG, "lib")
LIB = os.path.join(LIB_DIR, "libmixemt_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _sources():
    csrc = os.path.dirname(SRC)
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))) + HDRS


def _digest(flags):
    h = hashlib.sha256()
    for path in _sources():
        with open(path, "rb") as fin:
            h.update(fin.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(force=False, defines=(), verbose=False, out=None):
    """Compile csrc/mixemt_hip.hip -> lib/libmixemt_hip.so (or `out`); returns the path."""
    global LIB
    if out is not None:
        LIB = os.path.abspath(out)
    flags = ["--offload-arch=%s" % ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
             "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.dirname(SRC)]
    flags += ["-D%s" % d for d in defines]
    stamp = LIB + ".stamp"
    want = _digest(flags)
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        with open(stamp) as fin:
            if fin.read().strip() == want:
                return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [_hipcc()] + flags + [SRC, "-o", LIB, "-lz"]          # zlib: the BGZF members of csrc/bam_reader.hpp
    if verbose:
        sys.stderr.write(" ".join(cmd) + "\n")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout)
    with open(stamp, "w") as fout:
        fout.write(want + "\n")
    return LIB


if __name__ == "__main__":
    defs = []
    argv = sys.argv[1:]
    i = 0
    while i < len(argv):
        if argv[i] == "-D":
            defs.append(argv[i + 1])
            i += 2
        elif argv[i].startswith("-D"):
            defs.append(argv[i][2:])
            i += 1
        else:
            i += 1
    out_path = argv[argv.index("--out") + 1] if "--out" in argv else None
    print(build(force="--force" in argv, defines=defs, verbose=True, out=out_path))
