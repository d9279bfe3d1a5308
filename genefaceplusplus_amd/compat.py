"""Import-path shim: make ``modules.radnerfs.*`` resolve to this package, so the reference's callers run unmodified.

    import genefaceplusplus_amd.compat as compat
    compat.install()                      # before `from modules.radnerfs.radnerf import RADNeRF` is executed
    # inference/genefacepp_infer.py:39-43 then imports RADNeRF, RADNeRFTorso, RADNeRFTorsowithSR, RADNeRFwithSR and
    # modules.radnerfs.utils.{get_rays, get_bg_coords, convert_poses, nerf_matrix_to_ngp, get_audio_features} from here.

Only the render path's modules are replaced; every other ``modules.*`` package of the reference keeps importing from the
reference tree (the shim extends, not replaces, ``sys.modules['modules']`` when the reference is on sys.path).
"""
import importlib
import sys
import types


def _alias(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__dict__["__gfpp_shim__"] = True
    sys.modules[name] = m
    return m


def install(hparams=None):
    """Register the shim modules.  ``hparams`` (optional) seeds the stand-in for the reference's global hparams dict."""
    from . import radnerfs
    from .radnerfs import camera, cond_nets, encoders, head, torso, raymarching, superres

    if hparams is not None:
        radnerfs.set_runtime_hparams(hparams)

    for pkg in ("modules", "modules.radnerfs", "modules.radnerfs.encoders"):
        if pkg not in sys.modules:
            try:
                importlib.import_module(pkg)           # the reference's own package, if it is on sys.path
            except Exception:
                p = types.ModuleType(pkg)
                p.__path__ = []
                sys.modules[pkg] = p

    _alias("modules.radnerfs.renderer", NeRFRenderer=head.NeRFRenderer)
    _alias("modules.radnerfs.radnerf", RADNeRF=head.RADNeRF)
    _alias("modules.radnerfs.radnerf_torso", RADNeRFTorso=torso.RADNeRFTorso)
    _alias("modules.radnerfs.radnerf_torso_sr", RADNeRFTorsowithSR=torso.RADNeRFTorsowithSR)
    _alias("modules.radnerfs.radnerf_sr", RADNeRFwithSR=torso.RADNeRFwithSR, Superresolution=superres.Superresolution)
    _alias("modules.radnerfs.cond_encoder", AudioNet=cond_nets.AudioNet, AudioAttNet=cond_nets.AudioAttNet, MLP=cond_nets.MLP)
    _alias("modules.radnerfs.utils", get_rays=camera.get_rays, get_bg_coords=camera.get_bg_coords, convert_poses=camera.convert_poses,
           nerf_matrix_to_ngp=camera.nerf_matrix_to_ngp, get_audio_features=camera.get_audio_features, trunc_exp=camera.trunc_exp)
    _alias("modules.radnerfs.raymarching", **{k: getattr(raymarching, k) for k in
                                               ("near_far_from_aabb", "morton3D", "morton3D_invert", "packbits", "march_rays", "composite_rays",
                                                "morton3D_dilation", "sph_from_ray", "march_rays_train", "composite_rays_train")})
    _alias("modules.radnerfs.encoders.encoding", get_encoder=encoders.get_encoder)
    _alias("modules.radnerfs.encoders.gridencoder", GridEncoder=encoders.GridEncoder)
    _alias("modules.radnerfs.encoders.shencoder", SHEncoder=encoders.SHEncoder)
    _alias("modules.radnerfs.encoders.freqencoder", FreqEncoder=encoders.FreqEncoder)
    return sorted(k for k, v in list(sys.modules.items()) if isinstance(v, types.ModuleType) and v.__dict__.get("__gfpp_shim__", False))
