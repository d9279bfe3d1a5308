"""Recipe for ``oracle/_ref``: the reference's OWN four native extensions, compiled for gfx950 from the sources where
they lie under /root/reference.  TEST INFRASTRUCTURE ONLY -- the product never imports, links or executes it.

What is built (one pybind module each, under the reference's own import names):

    _raymarching_face   modules/radnerfs/raymarching/src/{raymarching.cu, bindings.cpp}
    _gridencoder        modules/radnerfs/encoders/gridencoder/src/{gridencoder.cu, bindings.cpp}
    _shencoder          modules/radnerfs/encoders/shencoder/src/{shencoder.cu, bindings.cpp}
    _freqencoder        modules/radnerfs/encoders/freqencoder/src/{freqencoder.cu, bindings.cpp}

How: the .cu files are compiled UNMODIFIED and IN PLACE (``hipcc -x hip <absolute path under /root/reference>``); no
source is copied, hipified or rewritten.  The only glue is ``oracle/ref_shim/`` -- four three-line headers that answer
``#include <cuda.h> / <cuda_runtime.h> / <cuda_fp16.h> / <ATen/cuda/CUDAContext.h>`` with their HIP equivalents.  The
kernels themselves use nothing outside the common CUDA/HIP subset (``atomicAdd``, ``__expf``, ``__sinf``, ``__half2``).
Flags follow the reference's ``backend.py`` files (``-O3``; ``-use_fast_math`` for the frequency encoder, which we
translate to denormal flushing + the ``__sinf`` intrinsic the source already spells out).

Outputs go to ``oracle/_ref/`` only (git-ignored, but NOT gpurun-ignored: the .so files travel to the GPU box like the
product's own library; /root/reference does not exist there).  ``__graft_entry__.build()`` calls ``build()`` here when
/root/reference is present.

Caveat recorded for the parity statement: nvcc and clang both contract ``a*b+c`` to FMA by default in device code, but
nothing guarantees they pick the same sites.  ``_ref`` is therefore "the reference source under clang's contraction";
the tests state, per quantity, whether the comparison is bit-exact or within an ulp-scale tolerance.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "ref_shim")
REF = os.environ.get("GFPP_REFERENCE_ROOT", "/root/reference")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

MODULES = {
    # name: (source dir relative to REF, extra device flags)
    "_raymarching_face": ("modules/radnerfs/raymarching/src", "raymarching.cu", []),
    "_gridencoder": ("modules/radnerfs/encoders/gridencoder/src", "gridencoder.cu", []),
    "_shencoder": ("modules/radnerfs/encoders/shencoder/src", "shencoder.cu", []),
    "_freqencoder": ("modules/radnerfs/encoders/freqencoder/src", "freqencoder.cu", ["-fgpu-flush-denormals-to-zero"]),
}


def available():
    """True when the reference tree is mounted (this container); False on the GPU box."""
    return all(os.path.exists(os.path.join(REF, d, cu)) for d, cu, _ in MODULES.values())


def module_path(name):
    return os.path.join(OUT, name + ".so")


def built():
    return all(os.path.exists(module_path(n)) for n in MODULES)


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for p in ce.include_paths():
        if os.path.isdir(p):
            inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
    link = [f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", f"-Wl,-rpath,{tlib}"]
    return inc, defs, link


def _newer(target, *sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build(verbose=False):
    """Compile the four reference extensions into oracle/_ref/.  Returns the list of .so paths."""
    if not available():
        raise RuntimeError(f"reference sources not found under {REF}; oracle/_ref can only be built where they are mounted")
    os.makedirs(OUT, exist_ok=True)
    inc, defs, link = _torch_flags()
    outs = []
    procs = []
    for name, (d, cu, extra) in MODULES.items():
        src_cu = os.path.join(REF, d, cu)
        src_cpp = os.path.join(REF, d, "bindings.cpp")
        so = module_path(name)
        outs.append(so)
        if _newer(so, src_cu, src_cpp, os.path.abspath(__file__)):
            continue
        # the .cu as HIP for gfx950, the pybind glue as host C++ (it only includes torch/extension.h), then one link
        obj_cu = os.path.join(OUT, name + ".kernels.o")
        obj_cpp = os.path.join(OUT, name + ".bindings.o")
        common = ["-O3", "-std=c++17", "-fPIC", "-w", f"-DTORCH_EXTENSION_NAME={name}", "-I", SHIM] + defs + inc
        script = " && ".join(" ".join(c) for c in (
            [HIPCC, "-x", "hip", f"--offload-arch={ARCH}"] + common + extra + ["-c", src_cu, "-o", obj_cu],
            [HIPCC, "-x", "c++"] + common + ["-c", src_cpp, "-o", obj_cpp],
            [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", obj_cu, obj_cpp, "-o", so] + link,
            ["rm", "-f", obj_cu, obj_cpp]))
        if verbose:
            print(script)
        procs.append((name, subprocess.Popen(script, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"building oracle/_ref/{name}.so failed:\n{out[-4000:]}")
    return outs


def load(name):
    """Import one built module (needs torch; on the GPU box the .so was shipped with the snapshot)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = module_path(name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python oracle/build_ref.py` where /root/reference is mounted")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    for p in build(verbose="-v" in sys.argv):
        print("built", p)
