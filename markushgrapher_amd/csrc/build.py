"""Build the native library.

  build_hip()  hipcc --offload-arch=gfx950 -> markushgrapher_amd/libmgrapher_hip.so   (THE product library)
  build_tools() hipcc -DMG_TOOLS          -> tools/_build/libmgrapher_tools.so        (profiling tools only: trace kernels,
               what-if GEMM variants; never loaded by the product)
  build_emu()  g++ -DMG_EMU               -> tools/simt_emu/_build/libmgrapher_emu.so (test infrastructure:
               same sources on the CPU SIMT emulator, to check index math without a GPU; never loaded by the
               product)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SOURCES = ["k_gemm.hip", "k_gemm_pp.hip", "k_pack.hip", "k_embed.hip", "k_attn.hip", "k_decode.hip", "k_xattn.hip", "k_beam.hip", "k_prep.hip", "c_ops.hip",
           "engine.hip", "dist.hip", "k_ocr.hip", "ocr.hip", "k_swin.hip", "swin.hip"]
HIP_SO = os.path.join(ROOT, "markushgrapher_amd", "libmgrapher_hip.so")
TOOLS_SO = os.path.join(ROOT, "tools", "_build", "libmgrapher_tools.so")
EMU_DIR = os.path.join(ROOT, "tools", "simt_emu")
EMU_SO = os.path.join(EMU_DIR, "_build", "libmgrapher_emu.so")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def _newer(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs += [os.path.join(ROOT, "include", "mgrapher.h"), os.path.join(EMU_DIR, "simt_emu.h")]
    return hs


class _BuildLock:
    """One builder at a time per output directory (pytest-xdist workers all ask for the emulator library at once)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        import fcntl
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def build_tools(force=False, verbose=False):
    """The product sources with -DMG_TOOLS: adds the phase-stamped trace kernels and the what-if GEMM variants (tools/*.py only)."""
    return build_hip(force, verbose, objdir=os.path.join(ROOT, "tools", "_build", "obj_tools"), out=TOOLS_SO, defines=["-DMG_TOOLS"])


def build_hip(force=False, verbose=False, objdir=None, out=None, defines=()):
    objdir = objdir or os.path.join(HERE, "_obj")
    out = out or HIP_SO
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    objs = []
    for s in _sources():
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + _headers()):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *defines,
                         "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=8) as ex:
        for log in ex.map(_run, jobs):
            if verbose and log.strip():
                print(log)
    if jobs or not os.path.exists(out):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_emu(force=False, opt="-O1"):
    objdir = os.path.join(EMU_DIR, "_build")
    os.makedirs(objdir, exist_ok=True)
    with _BuildLock(os.path.join(objdir, ".lock")):
        return _build_emu(force, opt, objdir)


def _build_emu(force, opt, objdir):
    jobs = []
    objs = []
    srcs = [os.path.join(HERE, s) for s in _sources()] + [os.path.join(EMU_DIR, "simt_emu.cpp")]
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + _headers()):
            jobs.append(["g++", "-DMG_EMU", opt, "-g", "-std=c++17", "-fPIC", "-x", "c++", "-I", EMU_DIR, "-I", HERE,
                         "-Wno-unknown-pragmas", "-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(_run, jobs))
    if jobs or not os.path.exists(EMU_SO):
        _run(["g++", "-shared", "-fPIC", "-o", EMU_SO] + objs)
    return EMU_SO


if __name__ == "__main__":
    what = sys.argv[1:] or ["hip"]
    if "emu" in what:
        print(build_emu())
    if "hip" in what:
        print(build_hip(verbose=True))
    if "tools" in what:
        print(build_tools(verbose=True))
