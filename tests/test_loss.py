"""Loss block: utils/loss_utils.py l1_loss + ssim as used by train.py:116-117.

This row is PINNED by the reference itself: tests/golden/ref_loss.npz was produced by importing the reference's
loss_utils.py (values + autograd gradients).  CPU tests check the oracle against it; GPU tests check the HIP kernels
against the fixture and against the oracle on larger images.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_matches_reference_loss_utils():
    fx = np.load(os.path.join(G, "ref_loss.npz"))
    for k in range(int(fx["n"])):
        loss, l1, ss, g = oracle.l1_ssim_loss(fx[f"img{k}"], fx[f"gt{k}"], float(fx[f"res{k}"][3]))
        want = fx[f"res{k}"]
        assert abs(loss - want[0]) < 2e-7 and abs(l1 - want[1]) < 2e-7 and abs(ss - want[2]) < 5e-7
        assert np.abs(g - fx[f"grad{k}"]).max() <= 2e-5 * np.abs(fx[f"grad{k}"]).max()


def test_oracle_loss_edge_cases():
    a = np.random.default_rng(0).uniform(0, 1, (3, 12, 9)).astype(np.float32)
    loss, l1, ss, g = oracle.l1_ssim_loss(a, a, 0.2)  # identical images: L1 = 0, SSIM = 1
    assert l1 == 0.0 and abs(ss - 1.0) < 1e-6 and abs(loss) < 1e-6
    loss, l1, ss, g = oracle.l1_ssim_loss(a[:, :3, :4], a[:, :3, :4] * 0.5, 0.2)  # image smaller than the window
    assert np.isfinite(loss) and np.isfinite(g).all()


@pytest.mark.gpu
def test_hip_loss_matches_reference_fixture():
    from based_on_point_cloud_partitions_amd import loss as L

    fx = np.load(os.path.join(G, "ref_loss.npz"))
    for k in range(int(fx["n"])):
        img = torch.tensor(fx[f"img{k}"], device="cuda", requires_grad=True)
        gt = torch.tensor(fx[f"gt{k}"], device="cuda")
        lam = float(fx[f"res{k}"][3])
        loss = L.l1_ssim_loss(img, gt, lam)
        loss.backward()
        want = fx[f"res{k}"]
        assert abs(loss.item() - want[0]) < 5e-7
        assert np.abs(img.grad.cpu().numpy() - fx[f"grad{k}"]).max() <= 2e-5 * np.abs(fx[f"grad{k}"]).max()
        assert abs(L.l1_loss(img, gt).item() - want[1]) < 5e-7 and abs(L.ssim(img, gt).item() - want[2]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape,lam", [((3, 270, 480), 0.2), ((3, 131, 77), 0.5), ((1, 16, 16), 1.0), ((3, 5, 7), 0.2),
                                       ((3, 1080, 1920), 0.2)])
def test_hip_loss_matches_oracle(shape, lam):
    from based_on_point_cloud_partitions_amd import loss as L

    rng = np.random.default_rng(sum(shape))
    gt = rng.uniform(0, 1, shape).astype(np.float32)
    img = np.clip(gt + rng.normal(0, 0.1, shape), 0, 1.3).astype(np.float32)
    want_loss, want_l1, want_ss, want_g = oracle.l1_ssim_loss(img, gt, lam)
    x = torch.tensor(img, device="cuda", requires_grad=True)
    y = torch.tensor(gt, device="cuda")
    loss = L.l1_ssim_loss(x, y, lam)
    (3.0 * loss).backward()  # upstream gradient != 1
    assert abs(loss.item() - want_loss) < 1e-6
    got = x.grad.cpu().numpy() / 3.0
    assert np.abs(got - want_g).max() <= 3e-5 * np.abs(want_g).max()
    # the separate entry points share the kernels
    s = L.ssim(x, y)
    x.grad = None
    s.backward()
    _, _, _, g1 = oracle.l1_ssim_loss(img, gt, 1.0)
    assert abs(s.item() - want_ss) < 2e-6 and np.abs(x.grad.cpu().numpy() + g1).max() <= 3e-5 * np.abs(g1).max()


@pytest.mark.gpu
def test_hip_loss_is_deterministic_and_rejects_cpu():
    from based_on_point_cloud_partitions_amd import GsrError
    from based_on_point_cloud_partitions_amd import loss as L

    x = torch.rand(3, 200, 300, device="cuda")
    y = torch.rand(3, 200, 300, device="cuda")
    assert L.l1_ssim_loss(x, y).item() == L.l1_ssim_loss(x, y).item()
    with pytest.raises(GsrError):
        L.l1_ssim_loss(x.cpu(), y.cpu())
    with pytest.raises(NotImplementedError):
        L.ssim(x, y, window_size=7)


def test_oracle_matches_reference_decoupled_loss():
    """train.py:112-117 of the reference: L1 on the appearance-decoupled image, SSIM on the rendered one (fixture from its own
    loss_utils.py).  The oracle evaluates the two terms separately; their combination must give the fixture's values."""
    fx = np.load(os.path.join(G, "ref_loss_decoupled.npz"))
    for k in range(int(fx["n"])):
        lam = float(fx[f"res{k}"][3])
        _, l1, _, g_l1 = oracle.l1_ssim_loss(fx[f"dec{k}"], fx[f"gt{k}"], 0.0)
        _, _, ss, g_ss = oracle.l1_ssim_loss(fx[f"img{k}"], fx[f"gt{k}"], 1.0)
        want = fx[f"res{k}"]
        assert abs((1 - lam) * l1 + lam * (1 - ss) - want[0]) < 3e-7 and abs(l1 - want[1]) < 2e-7 and abs(ss - want[2]) < 5e-7
        assert np.abs((1 - lam) * g_l1 - fx[f"grad_dec{k}"]).max() <= 2e-5 * np.abs(fx[f"grad_dec{k}"]).max()
        assert np.abs(lam * g_ss - fx[f"grad_img{k}"]).max() <= 2e-5 * np.abs(fx[f"grad_img{k}"]).max()


@pytest.mark.gpu
def test_hip_decoupled_loss_matches_reference_fixture():
    from based_on_point_cloud_partitions_amd import loss as L

    fx = np.load(os.path.join(G, "ref_loss_decoupled.npz"))
    for k in range(int(fx["n"])):
        img = torch.tensor(fx[f"img{k}"], device="cuda", requires_grad=True)
        dec = torch.tensor(fx[f"dec{k}"], device="cuda", requires_grad=True)
        gt = torch.tensor(fx[f"gt{k}"], device="cuda")
        lam = float(fx[f"res{k}"][3])
        loss = L.l1_ssim_loss(img, gt, lam, l1_image=dec)
        (3.0 * loss).backward()  # an upstream factor must scale both gradients
        want = fx[f"res{k}"]
        assert abs(loss.item() - want[0]) < 5e-7
        assert np.abs(img.grad.cpu().numpy() / 3.0 - fx[f"grad_img{k}"]).max() <= 2e-5 * np.abs(fx[f"grad_img{k}"]).max()
        assert np.abs(dec.grad.cpu().numpy() / 3.0 - fx[f"grad_dec{k}"]).max() <= 2e-5 * np.abs(fx[f"grad_dec{k}"]).max()


@pytest.mark.gpu
def test_hip_decoupled_loss_through_an_appearance_map():
    """decouple_image = mapping_image * image (scene/ptgs/appearance_network.py:15): the image receives gradient through both
    terms; against the oracle's two terms at 1080p."""
    from based_on_point_cloud_partitions_amd import loss as L

    rng = np.random.default_rng(3)
    shape = (3, 1080, 1920)
    gt = rng.uniform(0, 1, shape).astype(np.float32)
    img_np = np.clip(gt + rng.normal(0, 0.1, shape), 0, 1.3).astype(np.float32)
    map_np = rng.uniform(0.8, 1.2, shape).astype(np.float32)
    img = torch.tensor(img_np, device="cuda", requires_grad=True)
    mapping = torch.tensor(map_np, device="cuda")
    loss = L.l1_ssim_loss(img, torch.tensor(gt, device="cuda"), 0.2, l1_image=mapping * img)
    loss.backward()
    _, l1, _, g_l1 = oracle.l1_ssim_loss(map_np * img_np, gt, 0.0)
    _, _, ss, g_ss = oracle.l1_ssim_loss(img_np, gt, 1.0)
    assert abs(loss.item() - (0.8 * l1 + 0.2 * (1 - ss))) < 1e-6
    want = 0.8 * g_l1 * map_np + 0.2 * g_ss
    assert np.abs(img.grad.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
