"""ctypes binding of libgfpp_radnerf.so (the C ABI declared in include/gfpp_radnerf.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the product raises.  The CPU
oracle under oracle/ is test infrastructure and is never imported from here.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
def _library_path():
    """The shipped library -- or, for same-box A/B measurements, an experiment build of tools/build_variant.sh: GFPP_LIB_PATH is honoured ONLY for
    build/variants/lib_<name>.so of this checkout (anything else raises: the package cannot be pointed at an arbitrary library)."""
    shipped = os.path.join(_HERE, "libgfpp_radnerf.so")
    override = os.environ.get("GFPP_LIB_PATH")
    if not override:
        return shipped
    real, variants = os.path.realpath(override), os.path.realpath(os.path.join(_HERE, "..", "build", "variants"))
    name = os.path.basename(real)
    if real == os.path.realpath(shipped):
        return shipped
    if os.path.dirname(real) != variants or not (name.startswith("lib_") and name.endswith(".so")):
        raise RuntimeError(f"GFPP_LIB_PATH={override}: only experiment builds of tools/build_variant.sh (build/variants/lib_<name>.so) may replace the shipped library")
    return real


LIB_PATH = _library_path()
ABI_VERSION = 8          # include/gfpp_radnerf.h GFPP_ABI_VERSION (7: gfpp_torso_fold_batch / gfpp_torso_group_lp, f16 ambient_net steps inside the bf16 image; 6: corner-block grid copies, frame groups, sticky barrier word; 5: gfpp_head_model.occ_aabb; 4: gfpp_frame_ws.counters [192], .snapshots, the persistent 16-bit launch)
_lib = None

c_u32 = ctypes.c_uint32
c_f = ctypes.c_float
c_p = ctypes.c_void_p
c_i = ctypes.c_int

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "gfpp_abi_version": [],
    "gfpp_last_error": [],
    "gfpp_struct_size": [ctypes.c_char_p],
    "gfpp_set_tuning": [c_p],
    "gfpp_get_tuning": [c_p],
    "gfpp_near_far_from_aabb": [c_p, c_p, c_p, c_u32, c_f, c_p, c_p, c_p],
    "gfpp_morton3D": [c_p, c_u32, c_p, c_p],
    "gfpp_morton3D_invert": [c_p, c_u32, c_p, c_p],
    "gfpp_packbits": [c_p, c_u32, c_f, c_p, c_p],
    "gfpp_march_rays": [c_u32, c_u32, c_p, c_p, c_p, c_p, c_f, c_f, c_u32, c_u32, c_u32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gfpp_composite_rays": [c_u32, c_u32, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gfpp_grid_encode_forward": [c_p, c_p, c_p, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_p, c_u32, c_i, c_u32, c_i, c_p],
    "gfpp_sh_encode_forward": [c_p, c_p, c_u32, c_u32, c_u32, c_p, c_p],
    "gfpp_freq_encode_forward": [c_p, c_u32, c_u32, c_u32, c_u32, c_p, c_p],
    "gfpp_sh_encode_backward": [c_p, c_p, c_u32, c_u32, c_u32, c_p, c_p, c_p],
    "gfpp_freq_encode_backward": [c_p, c_p, c_u32, c_u32, c_u32, c_u32, c_p, c_p],
    "gfpp_march_rays_train": [c_p, c_p, c_p, c_f, c_f, c_u32, c_u32, c_u32, c_u32, c_u32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "gfpp_march_rays_train_backward": [c_p, c_p, c_p, c_p, c_u32, c_u32, c_p, c_p, c_p],
    "gfpp_composite_rays_train_forward": [c_p, c_p, c_p, c_p, c_p, c_u32, c_u32, c_f, c_p, c_p, c_p, c_p, c_p],
    "gfpp_composite_rays_train_backward": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_u32, c_u32, c_f, c_p, c_p, c_p, c_p],
    "gfpp_morton3D_dilation": [c_p, c_u32, c_u32, c_p, c_p],
    "gfpp_sph_from_ray": [c_p, c_p, c_f, c_u32, c_p, c_p],
    "gfpp_grid_encode_dydx": [c_p, c_p, c_p, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_u32, c_i, c_u32, c_p],
    "gfpp_grid_encode_backward": [c_p, c_p, c_p, c_p, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_p, c_p, c_u32, c_i, c_u32, c_p],
    "gfpp_grid_encode_backward_xcd": [c_p, c_p, c_p, c_p, c_u32, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_p, c_p, c_u32, c_i, c_u32, c_p, c_p, ctypes.c_uint64],
    "gfpp_grid_backward_bins_bytes": [c_u32, c_u32, c_u32, c_u32],
    "gfpp_grid_encode_input_backward": [c_p, c_i, c_p, c_p, c_p, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_u32, c_i, c_u32, c_p],
    "gfpp_grid_encode_backward_f16": [c_p, c_p, c_p, c_p, c_u32, c_p, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_p, c_p, c_u32, c_i, c_u32, c_p, c_p, ctypes.c_uint64],
    "gfpp_grad_total_variation": [c_p, c_p, c_p, c_p, c_f, c_u32, c_u32, c_u32, c_u32, c_f, c_u32, c_u32, c_i, c_p],
    "gfpp_get_rays": [c_p, c_f, c_f, c_f, c_f, c_u32, c_u32, c_p, c_p, c_p],
    "gfpp_get_rays_at": [c_p, c_f, c_f, c_f, c_f, c_u32, c_u32, c_p, c_u32, c_p, c_p, c_p],
    "gfpp_rgb_to_u8": [c_p, ctypes.c_uint64, c_p, c_p],
}
_RESTYPES = {"gfpp_last_error": ctypes.c_char_p, "gfpp_struct_size": ctypes.c_uint, "gfpp_grid_backward_bins_bytes": ctypes.c_uint64}


class GfppError(RuntimeError):
    """A C-ABI call returned non-zero (the reference surfaces these as RuntimeError from C++ exceptions)."""


def build(verbose=False):
    """Compile every HIP source for gfx950 into libgfpp_radnerf.so (in-tree; hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def declared_symbols():
    return sorted(_SIGNATURES)


def register(name, argtypes, restype=ctypes.c_int):
    """Used by the fused-pipeline bindings to add their entry points to the table."""
    _SIGNATURES[name] = argtypes
    if restype is not ctypes.c_int:
        _RESTYPES[name] = restype
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = restype


def lib():
    """Load the library (once).  Raises if it has not been built -- never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GfppError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"(or `make -C genefaceplusplus_amd/csrc`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        got = handle.gfpp_abi_version()
        if got != ABI_VERSION:
            raise GfppError(f"libgfpp_radnerf.so ABI version {got}, expected {ABI_VERSION} (rebuild: make -C genefaceplusplus_amd/csrc)")
        _lib = handle
        push_tuning()
    return _lib


def push_tuning():
    """Hand tuning.LIB to the loaded library (gfpp_set_tuning); a library that is not loaded yet gets it when it loads."""
    if _lib is None:
        return
    from . import tuning
    check_struct("tuning", tuning.GfppTuning)
    rec = tuning.record()
    rc = _lib.gfpp_set_tuning(ctypes.byref(rec))
    if rc != 0:
        msg = _lib.gfpp_last_error()
        raise GfppError(f"gfpp_set_tuning failed (code {rc}): {msg.decode() if msg else ''}")


def check_struct(name, mirror):
    """A ctypes mirror of a header struct must have the size the library was compiled with (a silent mismatch would corrupt memory)."""
    want = lib().gfpp_struct_size(name.encode())
    got = ctypes.sizeof(mirror)
    if want != got:
        raise GfppError(f"ctypes mirror of gfpp_{name} is {got} bytes, the library's struct is {want}: binding and libgfpp_radnerf.so are out of step")


def call(name, *args):
    """Invoke one entry point and raise GfppError on a non-zero return."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        msg = handle.gfpp_last_error()
        raise GfppError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")
