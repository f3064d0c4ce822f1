"""Builds libingvio_hip.so (hipcc, gfx950) and libingvio_host.so (g++) in-tree under ingvio_amd/lib/.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so files
travel to the GPU box with the repository snapshot (they are git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SOURCES = ["kernels_cov.hip", "kernels_msckf.hip", "kernels_ekf.hip", "kernels_factored.hip", "kernels_solve.hip", "kernels_bigwin.hip", "kernels_tri.hip", "kernels_lm.hip", "kernels_qr.hip", "kernels_chol.hip", "kernels_lmbatch.hip", "kernels_lmchol.hip", "kernels_gnss.hip", "kernels_tracks.hip", "capi.hip"]
HIP_LIB = os.path.join(LIB, "libingvio_hip.so")
HOST_LIB = os.path.join(LIB, "libingvio_host.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_deps(dirs):
    out = []
    for d in dirs:
        for root, _, files in os.walk(d):
            out += [os.path.join(root, f) for f in files if f.endswith((".hip", ".h", ".cpp", ".hpp"))]
    return out


# Per-file code-generation flags, each one measured on MI355X: the large-window gate / Gram kernels are long dependent
# MFMA + LDS chains at 1-3 waves per SIMD, where the max-ILP scheduling strategy gains 3-8 %; the small-window kernels lose with it.
EXTRA_FLAGS = {"kernels_bigwin.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def source_build_id(variant_flags=()):
    """Identity of the DEVICE code a libingvio_hip.so is built from: {"tu": {file: sha1 of the translation unit's own text + the
    headers it includes (transitively, csrc/ and include/ingvio_hip.h) + its extra flags}, "kernels": {__global__ name: file}}.  build_hip() embeds it in the
    library (ingvio_build_id()); tools/pmc_summary.py stores it beside the counters it folds, and bench.py refuses to price a
    kernel with counters that were collected on another build of that kernel's translation unit."""
    import hashlib
    import re
    inc_dir = os.path.join(os.path.dirname(HERE), "include")
    texts = {f: open(os.path.join(CSRC, f), "rb").read() for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))}
    texts["ingvio_hip.h"] = open(os.path.join(inc_dir, "ingvio_hip.h"), "rb").read()
    inc_pat = re.compile(rb'#include\s+"([^"]+)"')

    def closure(f, seen):
        for m in inc_pat.finditer(texts[f]):
            h = os.path.basename(m.group(1).decode())
            if h in texts and h not in seen:
                seen.add(h); closure(h, seen)
        return seen
    tu, kernels = {}, {}
    for src in HIP_SOURCES:
        h = hashlib.sha1(texts[src])
        for hd in sorted(closure(src, set())):
            h.update(hd.encode()); h.update(texts[hd])
        h.update(" ".join(EXTRA_FLAGS.get(src, []) + list(variant_flags)).encode())
        tu[src] = h.hexdigest()[:16]
    # kernels defined in a header belong to every translation unit that instantiates them: attributed to the first .hip that
    # includes the header (good enough: a header change flips every hash anyway)
    pat = re.compile(rb"__global__[^;{]{0,300}?\bvoid\s+(k_[A-Za-z0-9_]+)\s*\(", re.S)
    for f, t in texts.items():
        for m in pat.finditer(t):
            name = m.group(1).decode()
            if f.endswith(".hip"):
                kernels[name] = f
            else:
                owner = next((s for s in HIP_SOURCES if ('#include "%s"' % f).encode() in texts[s]), None)
                kernels.setdefault(name, owner or HIP_SOURCES[0])
    out = {"tu": tu, "kernels": kernels}
    if variant_flags:
        out["variant"] = " ".join(variant_flags)      # bench.py prices kernels with the product library's counters only
    return out


def _write_build_id(out_dir=None, variant_flags=()):
    """<out_dir or lib>/build_id.cpp: the JSON of source_build_id() as a string the loaded library returns (ingvio_build_id).  A
    variant build (build_alt: -DINGVIO_ALT_KERNELS) hashes its extra flags into every translation unit's id and carries a `variant`
    field, so counters collected on it are never taken for the product library's."""
    import json
    txt = json.dumps(source_build_id(variant_flags), sort_keys=True)
    src = os.path.join(out_dir or LIB, "build_id.cpp")
    body = 'extern "C" const char* ingvio_build_id(void) { return R"BID(%s)BID"; }\n' % txt
    if not os.path.exists(src) or open(src).read() != body:
        with open(src, "w") as f:
            f.write(body)
    return src


def build_hip(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    deps = _all_deps([CSRC, os.path.join(os.path.dirname(HERE), "include")])
    objs, cmds = [], []
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bid_src = _write_build_id()
    bid_obj = os.path.join(LIB, "build_id.o")
    if force or _newer(bid_obj, [bid_src]):
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", bid_src, "-o", bid_obj])
    for src in HIP_SOURCES:
        obj = os.path.join(LIB, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, deps):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                   ] + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            if os.environ.get("INGVIO_DBG_STAMPS"):
                cmd.insert(1, "-DINGVIO_DBG_STAMPS")
            if verbose:
                print(" ".join(cmd))
            cmds.append(cmd)
    if cmds:                                    # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
            list(ex.map(subprocess.check_call, cmds))
    if force or _newer(HIP_LIB, objs + [bid_obj]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + [bid_obj]
        subprocess.check_call(cmd)
    return HIP_LIB


def build_alt(force=False, verbose=False):
    """build_var/alt/libingvio_hip.so: the same sources with -DINGVIO_ALT_KERNELS - the measured-and-rejected kernel generations and
    the forced code-path switches (INGVIO_GATE=3, INGVIO_INFO_SOLVE=gj, INGVIO_BIG_SOLVE=regs|gj, INGVIO_BIG_APPLY=o|3|6, INGVIO_APPLY_TW=2,
    INGVIO_INFO_GAUGE=off, INGVIO_LM_FRONT=split, INGVIO_LM_SOLVE=sweep, INGVIO_GRAM_CHUNKS, INGVIO_P_PAD, INGVIO_MSCKF_METHOD) that the product
    library no longer carries.  tests/test_gpu_alternatives.py loads it through INGVIO_HIP_LIB."""
    out = os.path.join(os.path.dirname(HERE), "build_var", "alt")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libingvio_hip.so")
    deps = _all_deps([CSRC, os.path.join(os.path.dirname(HERE), "include")])
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, cmds = [], []
    for src in HIP_SOURCES:
        obj = os.path.join(out, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, deps):
            cmds.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DINGVIO_ALT_KERNELS"]
                        + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj])
    if cmds:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
            list(ex.map(subprocess.check_call, cmds))
    bid_src = _write_build_id(out, ("-DINGVIO_ALT_KERNELS",))
    bid_obj = os.path.join(out, "build_id.o")
    if force or _newer(bid_obj, [bid_src]):
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", bid_src, "-o", bid_obj])
    if force or _newer(lib, objs + [bid_obj]):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [bid_obj])
    return lib


def build_host(force=False, verbose=False):
    host_dir = os.path.join(CSRC, "host")
    if not os.path.isdir(host_dir):
        return None
    srcs = sorted(os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".cpp"))
    if not srcs:
        return None
    deps = _all_deps([host_dir, os.path.join(os.path.dirname(HERE), "include")]) + [HIP_LIB]      # an ABI change relinks the shim
    if force or _newer(HOST_LIB, deps):
        cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-I", os.path.join(os.path.dirname(HERE), "include"),
               "-o", HOST_LIB] + srcs + ["-L", LIB, "-lingvio_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_cpp_tests(force=False, verbose=False):
    """tests/cpp/test_host_shim.cpp -> ingvio_amd/lib/test_host_shim (run by pytest -m gpu)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "test_host_shim.cpp")
    exe = os.path.join(LIB, "test_host_shim")
    if not os.path.exists(src) or not os.path.exists(HOST_LIB):
        return None
    deps = _all_deps([os.path.join(CSRC, "host"), os.path.join(root, "include")]) + [src, HOST_LIB, HIP_LIB]
    if force or _newer(exe, deps):
        cmd = ["g++", "-O1", "-std=c++14", "-I", os.path.join(root, "include"), src, "-o", exe, "-L", LIB,
               "-lingvio_host", "-lingvio_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


def build_ros_adapter_test(force=False, verbose=False):
    """tests/cpp/test_ros_adapter.cpp -> ingvio_amd/lib/test_ros_adapter: the ROS1 node's conversion core (ros1/include/RosAdapter.h)
    on mock message structs; ROS itself is not needed (run by pytest -m "not gpu")."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "test_ros_adapter.cpp")
    exe = os.path.join(LIB, "test_ros_adapter")
    if not os.path.exists(src) or not os.path.exists(HOST_LIB):
        return None
    deps = _all_deps([os.path.join(CSRC, "host"), os.path.join(root, "include"), os.path.join(root, "ros1", "include")]) + [src, HOST_LIB, HIP_LIB]
    if force or _newer(exe, deps):
        cmd = ["g++", "-O1", "-std=c++14", "-I", os.path.join(root, "include"), "-I", os.path.join(CSRC, "host"), "-I", os.path.join(root, "ros1", "include"),
               src, "-o", exe, "-L", LIB, "-lingvio_host", "-lingvio_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


def build_tools(force=False, verbose=False):
    """tools/ingvio_replay.cpp -> ingvio_amd/lib/ingvio_replay (the rosbag-free replay driver)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "ingvio_replay.cpp")
    exe = os.path.join(LIB, "ingvio_replay")
    if not os.path.exists(src) or not os.path.exists(HOST_LIB):
        return None
    deps = _all_deps([os.path.join(CSRC, "host"), os.path.join(root, "include")]) + [src, HOST_LIB, HIP_LIB]
    if force or _newer(exe, deps):
        cmd = ["g++", "-O2", "-std=c++14", "-I", os.path.join(root, "include"), src, "-o", exe, "-L", LIB,
               "-lingvio_host", "-lingvio_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


def build_all(force=False, verbose=False):
    a = build_hip(force, verbose)
    build_alt(force, verbose)
    b = build_host(force, verbose)
    build_cpp_tests(force, verbose)
    build_ros_adapter_test(force, verbose)
    build_tools(force, verbose)
    return a, b


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
