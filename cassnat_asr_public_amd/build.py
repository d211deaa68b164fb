"""Builds libcassnat_hip.so - and libcassnat_hip_f16.so, the same sources with half-precision MFMA operands (csrc/common.h:
-DCN_OP16_F16; the engine `--hip_precision fp16`) - in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcassnat_hip.so")
LIB_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcassnat_hip_f16.so")
SOURCES = ["gemm.hip", "fbank.hip", "conv2.hip", "conv1.hip", "rowops.hip", "attention.hip", "ctc_align.hip", "ctc_beam.hip", "fused.hip", "fused_x3.hip", "genmax.hip", "proj_x3.hip", "conformer.hip", "chain.hip", "ast.hip", "model.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# conv1.hip: its matrix-core kernel converts every accumulator right behind the MFMAs - with the results in VGPRs (not AGPRs) the
# 128 v_accvgpr_read per block of 128 cells go (the kernel is bound by its VALU instruction count)
FILE_FLAGS = {"conv1.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir=None):
    """extra_flags / lib / objdir: measurement variants (e.g. -DFX_STAMPS into a library of their own); the product is the default."""
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(CSRC), "..", "include", "cassnat_hip.h"))
    objdir = objdir or os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + headers):
            cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


def build_f16(force=False, verbose=False):
    """The second product library: every source once more with -DCN_OP16_F16 (objects in csrc/build_f16)."""
    return build(force=force, verbose=verbose, extra_flags=["-DCN_OP16_F16"], lib=LIB_F16, objdir=os.path.join(CSRC, "build_f16"))


def build_all(force=False, verbose=False):
    return build(force=force, verbose=verbose), build_f16(force=force, verbose=verbose)


def experiments_lib(extra_flags=(), tag="exp"):
    """The -DCASSNAT_EXPERIMENTS build of the library (csrc/common.h: cn_exp_env) for the measurement tools: reads the CASSNAT_*
    experiment switches from the environment, which the product library does not.  Built into ab/ (git-ignored; travels to the
    GPU box with the snapshot, so build it BEFORE gpurun - the box has hipcc too, this just saves its minutes)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab")
    os.makedirs(root, exist_ok=True)
    return build(extra_flags=["-DCASSNAT_EXPERIMENTS"] + list(extra_flags), lib=os.path.join(root, "libcassnat_hip_%s.so" % tag),
                 objdir=os.path.join(root, "obj_" + tag))


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(experiments_lib())
        sys.exit(0)
    for path in build_all(force="--force" in sys.argv, verbose=True):
        print(path)
