"""Host-side mirror of the reference's ``modules/radnerfs`` package (see compat.py for the import-path shim)."""
from .head import NeRFRenderer, RADNeRF
from .torso import RADNeRFTorso, RADNeRFTorsowithSR, RADNeRFwithSR
from .cond_nets import AudioNet, AudioAttNet, MLP
from .encoders import GridEncoder, SHEncoder, FreqEncoder, get_encoder
from ..configs import CLASSES

_RUNTIME_HPARAMS = {}


def set_runtime_hparams(hp):
    """Stand-in for the reference's global ``utils.commons.hparams.hparams`` dict, for helpers that read it implicitly
    (get_audio_features reads ``smo_win_size``, utils.py:71-104)."""
    _RUNTIME_HPARAMS.clear()
    _RUNTIME_HPARAMS.update(hp)


def runtime_hparams():
    if not _RUNTIME_HPARAMS:
        try:  # running inside the reference tree: use its global dict
            from utils.commons.hparams import hparams as ref_hparams  # type: ignore
            return ref_hparams
        except Exception as exc:  # pragma: no cover
            raise RuntimeError("call genefaceplusplus_amd.radnerfs.set_runtime_hparams(hparams) first") from exc
    return _RUNTIME_HPARAMS
