"""Builds the native pieces in-tree (no JIT cache): the HIP library for gfx950,
the host-only synthetic workload generator, and (test infrastructure) the oracle."""
import os
import subprocess
import shutil

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HIP_LIB = os.path.join(PKG, "libvplines_hip.so")
WORKLOAD_LIB = os.path.join(PKG, "workload", "libvplines_workload.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    deps.append(os.path.join(ROOT, "include", "vplines_ba.h"))
    deps.append(os.path.join(ROOT, "include", "vplines_frontend.h"))
    if not force and not _newer(HIP_LIB, deps):
        return HIP_LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-ffp-contract=on",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", HIP_LIB] + srcs
    if os.environ.get("VPL_STAMPS"):
        cmd.insert(1, "-DVPL_STAMPS")
    for d in os.environ.get("VPL_EXTRA_DEFS", "").split():   # A/B experiments: -DNAME[=value] switches of the kernels
        cmd.insert(1, d)
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    out = _run(cmd)
    if verbose:
        print(out)
    return HIP_LIB


def build_workload(force=False):
    src = os.path.join(PKG, "workload", "synth.cpp")
    if force or _newer(WORKLOAD_LIB, [src]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", WORKLOAD_LIB, src])
    return WORKLOAD_LIB


def build_oracle(force=False):
    """TEST INFRASTRUCTURE: compiles oracle/ (building the checker is not using it)."""
    odir = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "clean"], cwd=odir)
    _run(["make", "-j4"], cwd=odir)
    return ORACLE_LIB


def build_all(force=False):
    build_workload(force)
    build_hip(force)
    build_oracle(force)
