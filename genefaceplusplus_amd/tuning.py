"""Every measurement switch of the package in ONE place, read from the environment ONCE (at import) -- the library's launch paths never call getenv().

Two kinds of switches:

* `LIB` -- the fields of the C library's `gfpp_tuning` record (include/gfpp_radnerf.h).  `_lib.lib()` hands the record to `gfpp_set_tuning` when it loads the
  library; `set_tuning(**fields)` / `with tuned(**fields):` change it at run time (tests and A/B tools; a captured graph keeps what it was captured with).
* `HOST` -- defaults of class attributes of the Python host side (FramePipeline.lp_kernel, ClipRenderer.precompute_cond, ...): plain attributes, settable per
  object; the environment only supplies the process-wide default.

Defaults are the shipped, measured optimum.  Switches of experiments that were measured and dropped (GFPP_PERSIST_GRID, GFPP_PERSIST_STAGGER, GFPP_EVAL_GRID,
GFPP_EVAL_WAVES: docs/LAB_NOTEBOOK.md) no longer exist.
"""
import contextlib
import ctypes
import os


class GfppTuning(ctypes.Structure):
    """ctypes mirror of gfpp_tuning (size-checked against the library by _lib.lib())."""
    _fields_ = [("size", ctypes.c_uint32), ("trip_pool", ctypes.c_int32), ("lp_separate_trips", ctypes.c_int32), ("occ_clip", ctypes.c_int32),
                ("barrier_spins", ctypes.c_uint32), ("persist_caps", ctypes.c_uint32), ("persist_xcd", ctypes.c_int32), ("torso_group_wgs", ctypes.c_int32),
                ("sr_fuse_first", ctypes.c_int32), ("sr_final_resident", ctypes.c_int32), ("sr_up_poly", ctypes.c_int32), ("grid_bwd_scatter", ctypes.c_int32), ("wgrad_tr", ctypes.c_int32), ("grid_bwd_bins", ctypes.c_int32), ("march_fixed_step", ctypes.c_int32)]


def _caps(text):
    """'2,2,2,4,8' -> eight 4-bit caps (the last value repeated)."""
    v = [max(1, min(8, int(x))) for x in text.split(",") if x.strip()][:8]
    if not v:
        return 0
    v += [v[-1]] * (8 - len(v))
    return sum(c << (4 * k) for k, c in enumerate(v))


def _env(name, default, conv=str):
    raw = os.environ.get(name)
    return default if raw is None or raw == "" else conv(raw)


#: field -> value of the library record; the environment names are the ones rounds 1-5 used
LIB = {
    "trip_pool": _env("GFPP_TRIP_POOL", 1, int),
    "lp_separate_trips": _env("GFPP_LP_SEPARATE_TRIPS", -1, int),
    "occ_clip": _env("GFPP_OCC_CLIP", 1, int),
    "barrier_spins": _env("GFPP_BARRIER_SPINS", 0, int),
    "persist_caps": _env("GFPP_PERSIST_CAPS", 0, _caps),
    "persist_xcd": _env("GFPP_PERSIST_XCD", 0, int),
    "torso_group_wgs": _env("GFPP_TORSO_GROUP_WGS", 0, int),
    "sr_fuse_first": _env("GFPP_SR_FUSE_FIRST", 1, int),
    "sr_final_resident": _env("GFPP_SR_FINAL_RESIDENT", 1, int),
    "sr_up_poly": _env("GFPP_SR_UP_POLY", 0, int),          # (off: the launch disturbs kernels that share a CU with it, see include/gfpp_radnerf.h)
    "grid_bwd_scatter": 1 if _env("GFPP_GRID_BWD", "").startswith("s") else 0,
    "wgrad_tr": _env("GFPP_WGRAD_TR", 1, int),
    "grid_bwd_bins": _env("GFPP_GRID_BWD_BINS", 1, int),
    "march_fixed_step": _env("GFPP_MARCH_FIXED_STEP", 1, int),
}

#: process-wide defaults of host-side attributes
HOST = {
    "lp_kernel": _env("GFPP_LP_KERNEL", "persist"),                 # FramePipeline.lp_kernel: 'persist' (one launch per head pass) | 'trips'
    "fp32_torso": _env("GFPP_FP32_TORSO", "mfma"),                  # FramePipeline.fp32_torso: 'mfma' | 'valu'
    "fuse_begin": _env("GFPP_FUSE_BEGIN", "1") != "0",              # FramePipeline.fuse_begin
    "fuse_tail": _env("GFPP_FUSE_TAIL", "0"),                       # FramePipeline.fuse_tail: '0' | '1' | 'resolve' | 'store'
    "group_torso": _env("GFPP_GROUP_TORSO", "1") != "0",            # FramePipeline.group_torso
    "lp_block_table": _env("GFPP_LP_BLOCK_TABLE", "1") != "0",      # 16-bit corner-block copies of the grid tables
    "trip_margin": _env("GFPP_TRIP_MARGIN", 1, int),                # FramePipeline.calibrate_trip_launches
    "clip_group": _env("GFPP_CLIP_GROUP", 0, int) or 4,             # ClipRenderer: frames per head launch
    "clip_precond": _env("GFPP_CLIP_PRECOND", "1") != "0",          # ClipRenderer.precompute_cond
    "clip_replay": _env("GFPP_CLIP_REPLAY", "c"),                   # ClipRenderer.replay_mode: 'c' | 'python'
    "clip_max_ahead": _env("GFPP_CLIP_MAX_AHEAD", 0, int),          # ClipRenderer.max_ahead
    "sr_store_u8": _env("GFPP_SR_STORE_U8", "1") != "0",            # ClipRenderer: the SR stage's last layer stores the clip frame as uint8 itself (0: a store launch per frame)
    "train_fused_mlp": _env("GFPP_TRAIN_FUSED_MLP", "1") != "0",    # cond_nets.FUSED_MLP
    "train_cond": _env("GFPP_TRAIN_COND", "fused"),                 # head.COND_TRAIN: 'fused' | 'eager'
}


def record():
    t = GfppTuning(**LIB)
    t.size = ctypes.sizeof(GfppTuning)
    return t


def set_tuning(**fields):
    """Change fields of the library record (and push it to the loaded library)."""
    from . import _lib
    unknown = set(fields) - set(LIB)
    if unknown:
        raise KeyError(f"unknown gfpp_tuning field(s) {sorted(unknown)}; known: {sorted(LIB)}")
    for k, v in fields.items():
        LIB[k] = _caps(v) if k == "persist_caps" and isinstance(v, str) else int(v)
    _lib.push_tuning()


@contextlib.contextmanager
def tuned(**fields):
    """`with tuned(sr_fuse_first=0): ...` -- the record changed inside the block, restored after it."""
    keep = {k: LIB[k] for k in fields}
    set_tuning(**fields)
    try:
        yield
    finally:
        set_tuning(**keep)
