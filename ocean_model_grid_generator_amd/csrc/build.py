"""Build libogg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ocean_model_grid_generator_amd.csrc.build [--force] [--verbose]

-ffp-contract=off is part of the arithmetic contract (see ogg_math.h): products and sums are rounded separately
exactly as numpy rounds them; the only fused operations are the explicit fma() calls.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ogg_api.hip", "ogg_axes.hip", "ogg_midas.hip", "ogg_bipolar.hip", "ogg_dpole.hip", "ogg_elementwise.hip", "ogg_latlon_fused.hip", "ogg_pass.hip", "ogg_reduce.hip"]
HEADERS = ["ogg_common.h", "ogg_math.h", "ogg_bipolar_dev.h", "ogg_dpole_dev.h", "ogg_latlon_fused_dev.h", "../../include/ogg_hip.h"]
LIB = os.path.join(HERE, "libogg_hip.so")
HASH_FILE = os.path.join(HERE, "libogg_hip.srchash")   # source hash of the library next to it (a built artefact, git-ignored like the .so)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function"]


def source_hash():
    """First 12 hex digits of the sha256 over the kernel sources and headers, in a fixed order: compiled into the library
    (ogg_version()) and recorded next to every counter file under profiles/, so that bench.py never quotes counters of other kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        path = os.path.join(HERE, f)
        if os.path.exists(path):
            h.update(f.encode() + b"\0" + open(path, "rb").read() + b"\0")
    return h.hexdigest()[:12]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS if os.path.exists(os.path.join(HERE, f))] + [__file__]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # a library built from other sources (a checkout that kept an old .so with fresh mtimes): the hash build() wrote next to it
    try:
        return open(HASH_FILE).read().strip() != source_hash()
    except OSError:
        pass
    try:   # no side file (a library from an older build.py): look for the hash inside the binary, once, and write the side file
        ok = source_hash().encode() in open(LIB, "rb").read()
        if ok:
            open(HASH_FILE, "w").write(source_hash() + "\n")
        return not ok
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    src_hash = source_hash()
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + (['-DOGG_SRC_HASH="%s"' % src_hash] if src == "ogg_api.hip" else []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout)
    open(HASH_FILE, "w").write(src_hash + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
