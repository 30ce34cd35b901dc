"""Build every native artefact in-tree (no JIT cache, no pip install).

  libmobiclip_hip.so      product: host parser + C ABI + gfx950 kernels          (hipcc)
  libmobiclip_hip_prof.so the same sources with -DMOBI_PROFILING (debug hooks, ablation switches): tools/ and two tests only
  libmobi_streamgen.so    synthetic bitstream generator (input source)           (g++)
  oracle/_build/libmobi_oracle.so   CPU oracle = TEST INFRASTRUCTURE             (gcc)
  tests/tools/libmobi_cmdinterp.so  CPU command-list interpreter = TEST TOOL     (g++)
  tests/tools/libmobi_lsparse_host.so  lock-step parser's lane functions on the CPU = TEST TOOL (g++)
  tests/tools/abi_caller            plain-C caller of the product's C ABI = TEST TOOL (gcc)

hipcc cross-compiles gfx950 without a GPU.  The built .so files are git-ignored but travel to the
GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mobiclipdecoder_amd")
CSRC = os.path.join(PKG, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")

LIB_HIP = os.path.join(PKG, "libmobiclip_hip.so")
LIB_HIP_PROF = os.path.join(PKG, "libmobiclip_hip_prof.so")  # -DMOBI_PROFILING twin: tools/ and the tests that inject planes
LIB_GEN = os.path.join(PKG, "libmobi_streamgen.so")
LIB_ORACLE = os.path.join(ROOT, "oracle", "_build", "libmobi_oracle.so")
LIB_INTERP = os.path.join(ROOT, "tests", "tools", "libmobi_cmdinterp.so")
LIB_LSHOST = os.path.join(ROOT, "tests", "tools", "libmobi_lsparse_host.so")  # the lock-step parser's lane functions on the CPU (test tool)
ABI_CALLER = os.path.join(ROOT, "tests", "tools", "abi_caller")  # plain-C caller of the product library (test tool)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _hdrs(d):
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]


def header_symbols():
    """every function include/*.h declares (the C ABI)"""
    import re
    names = []
    for h in ("mobiclip_hip.h", "mobiclip_demux.h"):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        names += re.findall(r"\b(mobi_\w+)\s*\(", text)
    return sorted(set(names))


def build_hip(force=False, profiling=False):
    """profiling=True (python -m mobiclipdecoder_amd.build --profiling, tools/ only): a SECOND library, libmobiclip_hip_prof.so, with
    -DMOBI_PROFILING: the ablation / occupancy / stage-stop switches (MOBI_INTRA_DBG, MOBI_LDS_PAD, MOBI_INTRA_LDS_PAD, MOBI_STOP_STAGE,
    MOBI_DEBUG=9) and the mobi_debug_* test hooks.  The product library has none of them; tools select the other one with MOBI_LIB."""
    srcs = [os.path.join(CSRC, f) for f in ("mobi_abi.cpp", "mobi_parse.cpp", "mobi_demux.cpp", "mobi_moflex.cpp", "mobi_kernels.hip", "mobi_rgb.hip", "mobi_dparse.hip", "mobi_lsparse.hip", "mobi_gop.hip", "mobi_analysis.hip")]
    deps = srcs + _hdrs(CSRC) + [os.path.join(ROOT, "include", "mobiclip_hip.h"), os.path.join(ROOT, "include", "mobiclip_demux.h"), os.path.abspath(__file__)]
    lib = LIB_HIP_PROF if profiling else LIB_HIP
    if not force and not _newer(lib, deps):
        return lib
    obj = os.path.join(PKG, "_obj_prof" if profiling else "_obj")
    os.makedirs(obj, exist_ok=True)
    prof = ["-DMOBI_PROFILING"] if profiling else []
    # only the entry points include/*.h declare leave the library (MOBI_API); everything else is hidden
    # (-O3: the host parser -- hand-overs, small batches, mobi_decode -- parses a 640x480 P-frame in 0.41 ms instead of 0.47)
    host_flags = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-fwrapv", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include")] + prof
    objs = []
    for s in srcs[:4]:
        o = os.path.join(obj, os.path.basename(s) + ".o")
        _run(["g++"] + host_flags + ["-c", s, "-o", o])
        objs.append(o)
    # the reconstruction kernels are bound by instruction issue: LLVM's ILP-first scheduling is worth 1.5 % on mobi_recon_inter8
    # (same registers, same occupancy; measured A/B on one box); the other kernels keep the default
    # r03: both kernels are bound by vector instruction issue; loops the source does not ask to unroll stay loops (-fno-unroll-loops:
    # 2.51 against 2.57 ms per launch of 8192 clips, tools/exp_kernel_flags.sh)
    extra = {"mobi_kernels.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-unroll-loops"],
             # the lock-step parser is one long dependent chain per wave: 24.8 against 25.6 ms per P-frame step (tools/exp_lsflags.sh)
             "mobi_lsparse.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
    for s in srcs[4:]:
        ko = os.path.join(obj, os.path.basename(s) + ".o")
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + prof + extra.get(os.path.basename(s), []) + ["-c", s, "-o", ko])
        objs.append(ko)
    # What leaves the library is exactly what include/*.h declares (the profiling twin adds its mobi_debug_* hooks): a linker version
    # script written from the headers -- no C++ symbols, no template instantiations, no kernel stubs.
    vs = os.path.join(obj, "exports.map")
    with open(vs, "w") as f:
        f.write("{\n  global:\n" + "".join(f"    {n};\n" for n in header_symbols()) + ("    mobi_debug_*;\n" if profiling else "") + "  local: *;\n};\n")
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs] + objs + ["-o", lib])
    return lib


def build_gen(force=False):
    src = os.path.join(CSRC, "mobi_streamgen.cpp")
    if force or _newer(LIB_GEN, [src] + _hdrs(CSRC)):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", src, "-o", LIB_GEN])
    return LIB_GEN


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    src = os.path.join(odir, "mobi_oracle.c")
    if force or _newer(LIB_ORACLE, [src] + _hdrs(odir)):
        os.makedirs(os.path.dirname(LIB_ORACLE), exist_ok=True)
        _run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-fwrapv", "-ffp-contract=off", "-Wall", src, "-o", LIB_ORACLE])
    return LIB_ORACLE


def build_interp(force=False):
    src = os.path.join(ROOT, "tests", "tools", "mobi_cmd_interp.cpp")
    parse = os.path.join(CSRC, "mobi_parse.cpp")
    if force or _newer(LIB_INTERP, [src, parse] + _hdrs(CSRC)):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fwrapv", src, parse, "-o", LIB_INTERP])
    return LIB_INTERP


def build_lshost(force=False):
    src = os.path.join(ROOT, "tests", "tools", "mobi_lsparse_host.cpp")
    parse = os.path.join(CSRC, "mobi_parse.cpp")
    if force or _newer(LIB_LSHOST, [src, parse] + _hdrs(CSRC)):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fwrapv", "-I" + CSRC, src, parse, "-o", LIB_LSHOST])
    return LIB_LSHOST


def build_caller(force=False):
    src = os.path.join(ROOT, "tests", "tools", "abi_caller.c")
    if force or _newer(ABI_CALLER, [src, os.path.join(ROOT, "include", "mobiclip_hip.h"), LIB_HIP]):
        _run(["gcc", "-O2", "-std=c99", "-Wall", src, "-L" + PKG, "-lmobiclip_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath-link," + os.path.join(ROCM, "lib"), "-o", ABI_CALLER])
    return ABI_CALLER


def build_all(force=False):
    return [build_hip(force), build_hip(force, profiling=True), build_gen(force), build_oracle(force), build_interp(force), build_lshost(force), build_caller(force)]


if __name__ == "__main__":
    if "--profiling" in sys.argv:
        build_hip(force="--force" in sys.argv, profiling=True)
    else:
        build_all(force="--force" in sys.argv)
    print("ok")
