"""Per-identity conditioning (genefaceplusplus_amd/postnet.py) against the reference's own LLE functions and conditioning
preparation, run on the CPU by tests/golden/make_golden_postnet.py (modules/postnet/lle.py:8-93, inference/genefacepp_infer.py:335-423)."""
import os

import numpy as np
import pytest
import torch

from genefaceplusplus_amd import postnet

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "postnet_golden.npz"))


def _run(device):
    ds, pred = torch.from_numpy(G["ds"]).to(device), torch.from_numpy(G["pred"]).to(device)
    feats, base = pred.reshape(-1, 204), ds.reshape(-1, 204)
    # same op sequence as the reference: bit-equal on the CPU; the GPU's GEMM / inverse round differently (near-singular normal equations amplify it)
    tol = dict(rtol=0, atol=0) if device == "cpu" else dict(rtol=2e-3, atol=2e-4)
    for K in (10, 4, 1):
        ind = postnet.find_k_nearest_neighbors(feats, base, K)
        np.testing.assert_array_equal(ind.cpu().numpy(), G[f"knn_K{K}"])
        fuse, err, w = postnet.compute_LLE_projection(feats, base, K)
        np.testing.assert_allclose(fuse.cpu().numpy(), G[f"fuse_K{K}"], **tol)
        np.testing.assert_allclose(w.cpu().numpy(), G[f"weights_K{K}"], **(tol if device == "cpu" else dict(rtol=0, atol=5e-3)))
        np.testing.assert_allclose(w.sum(1).cpu().numpy(), 1.0, atol=1e-4)
        if K > 1:
            np.testing.assert_allclose(err.cpu().numpy(), G[f"errors_K{K}"], **tol)
        else:
            assert err is None
    fuse, err, w = postnet.solve_LLE_projection_batch(feats[:5], base[:50].reshape(5, 10, 204))
    np.testing.assert_allclose(fuse.cpu().numpy(), G["solve_fuse"], **tol)
    np.testing.assert_allclose(err.cpu().numpy(), G["solve_errors"], **tol)

    cond = postnet.IdentityConditioner(ds, device=device)
    for name in ("mean", "std", "lower", "upper"):
        np.testing.assert_allclose(getattr(cond, name).cpu().numpy(), G[name], rtol=0, atol=0 if device == "cpu" else 1e-6)
    smo = int(G["smo_win_size"])
    for lle_percent, tag in ((0.2, "0p2"), (1.0, "1p0"), (0.0, "0p0")):
        norm = cond.normalized_landmarks(pred, lle_percent)
        np.testing.assert_allclose(norm.cpu().numpy(), G[f"normalized_{tag}"], **(tol if device == "cpu" else dict(rtol=0, atol=2e-3)))
        wins = cond.cond_wins(pred, smo, lle_percent)
        assert tuple(wins.shape) == (30, smo, 1, 204)
        np.testing.assert_allclose(wins.cpu().numpy(), G[f"cond_wins_{tag}"], **(tol if device == "cpu" else dict(rtol=0, atol=2e-3)))


def test_postnet_matches_the_reference_on_cpu():
    _run("cpu")


def test_projection_is_a_projection():
    """Size-independent properties: database rows project onto themselves (K = 1 and K = 10: the row itself is among its neighbours), weights sum
    to one, and a point that IS an affine combination of its neighbours is reproduced."""
    g = torch.Generator().manual_seed(3)
    base = torch.randn(2000, 204, generator=g)
    fuse, _, w = postnet.compute_LLE_projection(base[:64], base, K=1)
    np.testing.assert_array_equal(fuse.numpy(), base[:64].numpy())
    fuse, err, w = postnet.compute_LLE_projection(base[100:164], base, K=10)
    np.testing.assert_allclose(fuse.numpy(), base[100:164].numpy(), atol=1e-4)
    np.testing.assert_allclose(w.sum(1).numpy(), 1.0, atol=1e-5)
    wts = torch.softmax(torch.randn(8, 10, generator=g), dim=1)
    nb = base[:80].reshape(8, 10, 204)
    target = torch.bmm(wts.unsqueeze(1), nb).squeeze(1)
    fuse, err, w = postnet.solve_LLE_projection_batch(target, nb)
    np.testing.assert_allclose(fuse.numpy(), target.numpy(), atol=1e-4)
    np.testing.assert_allclose(w.numpy(), wts.numpy(), atol=1e-3)
    assert float(err.max()) < 1e-5


def test_conditioner_rejects_wrong_shapes_and_handles_empty_clips():
    with pytest.raises(ValueError):
        postnet.IdentityConditioner(torch.zeros(10, 204))
    cond = postnet.IdentityConditioner(torch.from_numpy(G["ds"]))
    assert tuple(cond.normalized_landmarks(torch.zeros(0, 68, 3)).shape) == (0, 68, 3)
    one = cond.cond_wins(torch.from_numpy(G["pred"][:1]), 3)
    assert tuple(one.shape) == (1, 3, 1, 204) and float(one[0, 0].abs().sum()) == 0.0 and float(one[0, 2].abs().sum()) == 0.0   # zero padding either side


@pytest.mark.gpu
def test_postnet_matches_the_reference_on_gpu():
    assert torch.cuda.is_available()
    _run("cuda:0")
