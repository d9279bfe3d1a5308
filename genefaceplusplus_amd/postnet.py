"""Per-identity conditioning between the (shared) audio2motion result and the renderer -- SURVEY 8f-4.

In a multi-identity job the audio -> motion model runs once and its landmarks are broadcast (frames.share_driving_signals); what remains
per identity is cheap tensor work that the reference does inside its inference call (inference/genefacepp_infer.py:335-424) and that
this module batches on the identity's GPU:

    LLE projection of the predicted landmarks onto the person's training-set manifold   modules/postnet/lle.py:8-93
    blend with lle_percent, normalise with the person's mean / std                      genefacepp_infer.py:364-368, 391-396
    clamp to the 3 % / 97 % quantiles of the person's normalised training landmarks     genefacepp_infer.py:343-344, 407
    smoothing windows around every frame (att_mode 2)                                   genefacepp_infer.py:421-423, radnerfs/utils.py:71-104

The three LLE functions keep the reference's names, argument meaning and return values.  Everything is torch on whatever device the inputs
live on (the database matmul and the batched K-1 x K-1 solves go to rocBLAS on the GPU); there is no hand-written kernel here: the work is
a few MFLOP per clip.  What stays with the reference: the audio2motion VAE / postnet models and the 3DMM (`Face3DHelper`) that turns their
(id, exp) output into the `idexp_lm3d` landmarks this module starts from, and the blink injection, which edits landmarks through the 3DMM's
mean shape."""
import torch

from .radnerfs.camera import get_audio_features


def _as_tensor(x, like=None):
    if torch.is_tensor(x):
        return x
    t = torch.as_tensor(x)
    return t.to(like.device) if like is not None else t


def find_k_nearest_neighbors(feats, feat_database, K=10):
    """Indices [N, K] of the K nearest database rows of every row of `feats` (lle.py:8-29: squared distances through
    |x|^2 + |y|^2 - 2 x.y, `topk(largest=False)`)."""
    feats = _as_tensor(feats)
    feat_database = _as_tensor(feat_database, feats)
    base_norm = (feat_database ** 2).sum(-1)
    feats_norm = (feats ** 2).sum(-1)
    distance = feats_norm.view(-1, 1) + base_norm.view(1, -1) - 2 * feats @ feat_database.t()
    return distance.topk(K, dim=1, largest=False).indices


def solve_LLE_projection_batch(feat, feat_base):
    """min || feat - sum_k w_k feat_base_k ||  s.t.  sum_k w_k = 1, for a batch (lle.py:31-80: eliminate w_0, normal equations with an explicit
    inverse).  feat [N, C], feat_base [N, K, C] -> (feat_fuse [N, C], errors [N] | None, weights [N, K])."""
    feat = _as_tensor(feat)
    feat_base = _as_tensor(feat_base, feat)
    N, K, C = feat_base.shape
    if K == 1:
        return feat_base[:, 0], None, torch.ones(N, 1, device=feat.device, dtype=feat.dtype)
    B = feat - feat_base[:, 0, :]                                             # [N, C]
    A = (feat_base[:, 1:, :] - feat_base[:, 0:1, :]).transpose(1, 2)           # [N, C, K-1]
    AT = A.transpose(1, 2)
    X = torch.bmm(torch.bmm(torch.inverse(torch.bmm(AT, A)), AT), B.unsqueeze(2)).squeeze(2)   # [N, K-1]
    weights = torch.zeros(N, K, device=feat.device, dtype=X.dtype)
    weights[:, 1:] = X
    weights[:, 0] = torch.ones_like(weights[:, 0]) - X.sum(dim=1)
    feat_fuse = torch.bmm(weights.unsqueeze(1), feat_base).squeeze(1)
    errors = (torch.bmm(A, X.unsqueeze(-1)).squeeze(-1) - B).abs().mean(dim=-1)
    return feat_fuse, errors, weights


def compute_LLE_projection(feats, feat_database, K=10):
    """Every row of `feats` as the best affine combination of its K nearest database rows (lle.py:82-98)."""
    feats = _as_tensor(feats)
    feat_database = _as_tensor(feat_database, feats)
    index = find_k_nearest_neighbors(feats, feat_database, K)
    return solve_LLE_projection_batch(feats, feat_database[index])


class IdentityConditioner:
    """One person's statistics (from the landmarks of the person's training set) and the per-clip step that turns shared predicted landmarks
    into the renderer's conditioning windows.  `idexp_lm3d_ds`: [M, 68, 3] (the lm68 subset of the dataset's idexp_lm3d, genefacepp_infer.py:336, 389)."""

    def __init__(self, idexp_lm3d_ds, normalize_cond=True, device=None):
        ds = _as_tensor(idexp_lm3d_ds).float()
        if device is not None:
            ds = ds.to(device)
        if ds.dim() != 3 or ds.shape[1:] != (68, 3):
            raise ValueError(f"idexp_lm3d_ds must be [M, 68, 3], got {tuple(ds.shape)}")
        self.mean = ds.mean(dim=0, keepdim=True)                   # genefacepp_infer.py:337-338
        self.std = ds.std(dim=0, keepdim=True)
        self.normalize_cond = bool(normalize_cond)
        normalized = (ds - self.mean) / self.std if self.normalize_cond else ds
        self.lower = torch.quantile(normalized, q=0.03, dim=0)     # :343-344
        self.upper = torch.quantile(normalized, q=0.97, dim=0)
        self.database = ds.reshape(-1, 68 * 3)                     # :389

    def normalized_landmarks(self, idexp_lm3d, lle_percent=0.2, K=10, clamp=True):
        """[T, 68, 3] predicted landmarks -> the normalised, manifold-projected, clamped landmarks (:390-396, 407)."""
        x = _as_tensor(idexp_lm3d, self.database).float().reshape(-1, 68 * 3).clone()
        if lle_percent:
            fuse, _, _ = compute_LLE_projection(x, self.database, K=K)
            x = lle_percent * fuse + (1 - lle_percent) * x
        x = x.reshape(-1, 68, 3)
        # (the reference normalises with mean / std here even when normalize_cond is off, :395)
        x = (x - self.mean) / self.std
        return torch.clamp(x, min=self.lower, max=self.upper) if clamp else x

    def cond_wins(self, idexp_lm3d, smo_win_size, lle_percent=0.2, K=10):
        """-> [T, smo_win_size, 1, 204]: the `cond_wins` entry of the batch the renderer consumes (:421-423)."""
        x = self.normalized_landmarks(idexp_lm3d, lle_percent, K)
        T = x.shape[0]
        if T == 0:
            return x.new_zeros(0, int(smo_win_size), 1, 68 * 3)
        win = x.reshape(T, 1, -1)
        return torch.stack([get_audio_features(win, att_mode=2, index=i, smo_win_size=smo_win_size) for i in range(T)])
