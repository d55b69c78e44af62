"""Appearance-decoupling tail: scene/ptgs/appearance_network.py:54-60 + :15.

PINNED by the reference itself: tests/golden/ref_appearance.npz came from running its own AppearanceNetwork /
decouple_appearance (values + autograd gradients, make_golden.py ref_appearance).  CPU tests check the numpy oracle against it;
GPU tests check the HIP kernels against the fixture, against the oracle at awkward and full sizes, and the whole
`decouple_appearance` against the same layers run as stock torch ops.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRADS = (("feat", "g_feat{k}"), ("w2", "g_w2_{k}"), ("b2", "g_b2_{k}"), ("w3", "g_w3_{k}"), ("b3", "g_b3_{k}"), ("image", "g_image_direct{k}"))


def test_oracle_matches_reference_appearance_network():
    fx = np.load(os.path.join(G, "ref_appearance.npz"))
    for k in range(int(fx["n"])):
        m, t, saved = oracle.appearance_tail_forward(fx[f"feat{k}"], fx[f"w2_{k}"], fx[f"b2_{k}"], fx[f"w3_{k}"], fx[f"b3_{k}"], fx[f"image{k}"])
        assert np.abs(m - fx[f"mapping{k}"]).max() < 5e-7 and np.abs(t - fx[f"transformed{k}"]).max() < 5e-7
        g = oracle.appearance_tail_backward(saved, fx[f"w2_{k}"], fx[f"w3_{k}"], fx[f"image{k}"], fx[f"r1_{k}"], fx[f"r2_{k}"])
        for name, key in GRADS:
            want = fx[key.format(k=k)]
            assert np.abs(g[name] - want).max() <= 5e-6 * np.abs(want).max(), name


def test_oracle_up_block_matches_reference_module():
    fx = np.load(os.path.join(G, "ref_appearance.npz"))
    y, ps = oracle.up_block_forward(fx["up_x"], fx["up_w"], fx["up_b"])
    assert np.abs(y - fx["up_y"]).max() < 3e-6
    gx, gw, gb = oracle.up_block_backward(ps, fx["up_w"], y, fx["up_g"])
    for got, key in ((gx, "up_gx"), (gw, "up_gw"), (gb, "up_gb")):
        assert np.abs(got - fx[key]).max() <= 5e-6 * np.abs(fx[key]).max(), key


def test_oracle_bilinear_is_torchs():
    import torch.nn.functional as F

    x = torch.randn(3, 5, 7, generator=torch.Generator().manual_seed(0))
    for (H, W) in [(12, 17), (5, 7), (1, 1), (33, 8)]:
        want = F.interpolate(x[None], size=(H, W), mode="bilinear", align_corners=True)[0].numpy()
        assert np.abs(oracle.bilinear_upsample(x.numpy(), H, W) - want).max() < 1e-6


def test_synthetic_appearance_model_has_the_reference_modules_parameters():
    from based_on_point_cloud_partitions_amd.synthetic import make_appearance_model

    fx = np.load(os.path.join(G, "ref_appearance.npz"))
    mine = [f"{name}:{tuple(v.shape)}" for name, v in make_appearance_model(4).appearance_network.state_dict().items()]
    assert mine == list(fx["state_keys"])


def _run_hip(feat, w2, b2, w3, b3, image, r1, r2):
    from based_on_point_cloud_partitions_amd.appearance import appearance_tail

    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    ins = [t(a) for a in (feat, w2, b2, w3, b3, image)]
    transformed, mapping = appearance_tail(*ins)
    loss = (transformed * torch.tensor(r1, device="cuda")).sum()
    if r2 is not None:
        loss = loss + (mapping * torch.tensor(r2, device="cuda")).sum()
    H, W = image.shape[1:]
    saved_a = transformed.grad_fn.saved_tensors[5].view(torch.float32)[:16 * H * W].view(16, H, W).cpu().numpy()  # relu(conv2) as the kernels kept it
    loss.backward()
    names = ("feat", "w2", "b2", "w3", "b3", "image")
    grads = {n: v.grad.cpu().numpy() for n, v in zip(names, ins)}
    grads["saved_a"] = saved_a
    return mapping.detach().cpu().numpy(), transformed.detach().cpu().numpy(), grads


@pytest.mark.gpu
def test_hip_tail_matches_reference_fixture():
    fx = np.load(os.path.join(G, "ref_appearance.npz"))
    for k in range(int(fx["n"])):
        m, t, g = _run_hip(fx[f"feat{k}"], fx[f"w2_{k}"], fx[f"b2_{k}"], fx[f"w3_{k}"], fx[f"b3_{k}"], fx[f"image{k}"], fx[f"r1_{k}"], fx[f"r2_{k}"])
        assert np.abs(m - fx[f"mapping{k}"]).max() < 1e-6 and np.abs(t - fx[f"transformed{k}"]).max() < 1e-6
        for name, key in GRADS:
            want = fx[key.format(k=k)]
            assert np.abs(g[name] - want).max() <= 1e-5 * np.abs(want).max(), (k, name)


def _random_case(H, W, fh, fw, seed):
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32)
    feat = f32(np.maximum(rng.normal(0.2, 1.0, (16, fh, fw)), 0))  # up4 ends in a ReLU
    w2, b2 = f32(rng.normal(0, 0.25, (16, 16, 3, 3))), f32(rng.normal(0, 0.3, 16))
    w3, b3 = f32(rng.normal(0, 0.3, (3, 16, 3, 3))), f32(rng.normal(0, 0.3, 3))
    return feat, w2, b2, w3, b3, f32(rng.uniform(0, 1, (3, H, W))), f32(rng.normal(0, 1, (3, H, W))), f32(rng.normal(0, 1, (3, H, W)))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,fh,fw,with_r2", [(33, 65, 16, 32, True), (16, 64, 16, 16, True), (8, 64, 16, 32, False), (17, 130, 1, 64, True),
                                                (9, 7, 3, 1, True), (1, 1, 1, 1, False), (131, 257, 64, 128, True), (1080, 1920, 528, 960, False)])
def test_hip_tail_matches_oracle(H, W, fh, fw, with_r2):
    feat, w2, b2, w3, b3, image, r1, r2 = _random_case(H, W, fh, fw, H * 1000 + W)
    if not with_r2:
        r2 = None
    m, t, g = _run_hip(feat, w2, b2, w3, b3, image, r1, r2)
    want_m, want_t, saved = oracle.appearance_tail_forward(feat, w2, b2, w3, b3, image)
    assert np.abs(m - want_m).max() < 2e-6 and np.abs(t - want_t).max() < 2e-6
    # the ReLU mask is a discontinuity: a pre-activation within rounding of 0 may fall on either side (at 1080p there are 33 M of
    # them), so the oracle's backward runs on the activations the kernels saved, after checking those against its own
    assert np.abs(g["saved_a"] - saved["a"]).max() <= 1e-5 * max(1.0, np.abs(saved["a"]).max())
    saved["a"] = g["saved_a"]
    want = oracle.appearance_tail_backward(saved, w2, w3, image, r1, r2)
    for name in ("feat", "w2", "b2", "w3", "b3", "image"):
        scale = max(np.abs(want[name]).max(), 1e-6)
        assert np.abs(g[name] - want[name]).max() <= 2e-5 * scale, (name, np.abs(g[name] - want[name]).max(), scale)


def _run_hip_up_tail(x, w4, b4, w2, b2, w3, b3, image, r1, r2):
    from based_on_point_cloud_partitions_amd.appearance import appearance_up_tail

    t = lambda a: torch.tensor(a, device="cuda", requires_grad=True)
    ins = [t(a) for a in (x, w4, b4, w2, b2, w3, b3, image)]
    transformed, mapping = appearance_up_tail(*ins)
    H, W = image.shape[1:]
    st = transformed.grad_fn.saved_tensors  # x, w4, feat, w2, w3, image, mapping, saved
    feat = st[2].cpu().numpy()
    saved_a = st[7].view(torch.float32)[:16 * H * W].view(16, H, W).cpu().numpy()
    loss = (transformed * torch.tensor(r1, device="cuda")).sum() + (mapping * torch.tensor(r2, device="cuda")).sum()
    loss.backward()
    names = ("x", "w4", "b4", "w2", "b2", "w3", "b3", "image")
    return mapping.detach().cpu().numpy(), transformed.detach().cpu().numpy(), feat, saved_a, {n: v.grad.cpu().numpy() for n, v in zip(names, ins)}


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,h,w", [(70, 101, 16, 24), (33, 65, 8, 16), (9, 7, 1, 1), (200, 333, 48, 80), (1080, 1920, 264, 480)])
def test_hip_up_block_and_tail_match_oracle(H, W, h, w):
    """`up4` + tail as one node (x = up3's output [32,h,w] -> feat [16,2h,2w] -> image resolution) against the oracle's two stages."""
    rng = np.random.default_rng(H * 7 + W)
    f32 = lambda a: a.astype(np.float32)
    x = f32(np.maximum(rng.normal(0.1, 1.0, (32, h, w)), 0))
    w4, b4 = f32(rng.normal(0, 0.35, (16, 8, 3, 3))), f32(rng.normal(0, 0.3, 16))
    _, w2, b2, w3, b3, image, r1, r2 = _random_case(H, W, 1, 1, H + W)
    m, t, feat, saved_a, g = _run_hip_up_tail(x, w4, b4, w2, b2, w3, b3, image, r1, r2)
    want_feat, ps = oracle.up_block_forward(x, w4, b4)
    assert np.abs(feat - want_feat).max() <= 1e-5 * max(1.0, np.abs(want_feat).max())
    # both ReLU masks are discontinuities: the oracle continues from the activations the kernels produced (checked above / below)
    want_m, want_t, saved = oracle.appearance_tail_forward(feat, w2, b2, w3, b3, image)
    assert np.abs(m - want_m).max() < 5e-6 and np.abs(t - want_t).max() < 5e-6  # (pre-activations reach +-30 here)
    assert np.abs(saved_a - saved["a"]).max() <= 1e-5 * max(1.0, np.abs(saved["a"]).max())
    saved["a"] = saved_a
    want = oracle.appearance_tail_backward(saved, w2, w3, image, r1, r2)
    gx, gw4, gb4 = oracle.up_block_backward(ps, w4, feat, want["feat"])
    want.update(x=gx, w4=gw4, b4=gb4)
    for name in ("x", "w4", "b4", "w2", "b2", "w3", "b3", "image"):
        scale = max(np.abs(want[name]).max(), 1e-6)
        assert np.abs(g[name] - want[name]).max() <= 3e-5 * scale, (name, np.abs(g[name] - want[name]).max(), scale)


def _sweep_cases():
    rng = np.random.default_rng(2026)
    cases = []
    for i in range(12):
        H, W = int(rng.integers(1, 700)), int(rng.integers(1, 900))
        if i % 3 == 0:  # the network's own geometry: feat = 16 * (size // 32), at least 2
            fh, fw = max(2, 16 * (H // 32)), max(2, 16 * (W // 32))
        else:  # arbitrary ratios, up- and down-sampling: the generic (unstaged / untiled) paths
            fh, fw = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        cases.append((H, W, fh, fw))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,fh,fw", _sweep_cases())
def test_hip_tail_random_sizes(H, W, fh, fw):
    """Ragged tiles, several tiles per persistent workgroup (more than 512 tiles), staged and generic interpolation paths."""
    feat, w2, b2, w3, b3, image, r1, r2 = _random_case(H, W, fh, fw, 31 * H + W)
    m, t, g = _run_hip(feat, w2, b2, w3, b3, image, r1, r2)
    want_m, want_t, saved = oracle.appearance_tail_forward(feat, w2, b2, w3, b3, image)
    assert np.abs(m - want_m).max() < 6e-6 and np.abs(t - want_t).max() < 6e-6  # (contract: 1e-4)
    assert np.abs(g["saved_a"] - saved["a"]).max() <= 1e-5 * max(1.0, np.abs(saved["a"]).max())
    saved["a"] = g["saved_a"]
    want = oracle.appearance_tail_backward(saved, w2, w3, image, r1, r2)
    for name in ("feat", "w2", "b2", "w3", "b3", "image"):
        scale = max(np.abs(want[name]).max(), 1e-6)
        assert np.abs(g[name] - want[name]).max() <= 3e-5 * scale, (name, np.abs(g[name] - want[name]).max(), scale)


def _model(seed):
    from based_on_point_cloud_partitions_amd.synthetic import make_appearance_model

    m = make_appearance_model(n_cameras=5, seed=seed, device="cuda")
    with torch.no_grad():  # the default init gives a nearly constant sigmoid: widen the last layers so every term matters
        m.appearance_network.conv2.weight.mul_(3.0)
        m.appearance_network.conv3.weight.mul_(6.0)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0])
def test_decouple_appearance_matches_the_stock_torch_dataflow(overlap):
    """The whole function (head as the module's own layers + HIP tail) against appearance_network.py:7-17 with every layer a torch
    op, on the GPU: outputs, and the gradients of the image (both paths), the embedding and every network parameter.  overlap = 1: the
    head is one autograd node whose weight gradients run on a side stream and reach `.grad` through the end-of-backward callback
    (appearance._Head); a second backward must ACCUMULATE into the same `.grad` like autograd does."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd.appearance import decouple_appearance

    _lib.set_variant("APPEARANCE_OVERLAP", overlap)

    from based_on_point_cloud_partitions_amd.synthetic import decouple_appearance_torch

    H, W = 200, 333
    a, b = _model(3), _model(3)
    img = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    r = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    ia, ib = img.clone().requires_grad_(True), img.clone().requires_grad_(True)
    ta, ma = decouple_appearance(ia, a, 3)
    tb, mb = decouple_appearance_torch(ib, b, 3)
    assert ta.shape == tb.shape == (3, H, W) and ma.shape == mb.shape
    assert (ta - tb).abs().max().item() < 2e-6 and (ma - mb).abs().max().item() < 2e-6
    (ta * r).sum().backward()
    (tb * r).sum().backward()
    close = lambda u, v: (u - v).abs().max().item() <= 2e-4 * max(v.abs().max().item(), 1e-6)
    assert close(ia.grad, ib.grad) and close(a._appearance_embeddings.grad, b._appearance_embeddings.grad)
    for (n, pa), (_, pb) in zip(a.appearance_network.named_parameters(), b.appearance_network.named_parameters()):
        assert close(pa.grad, pb.grad), n
    first = {n: p.grad.clone() for n, p in a.appearance_network.named_parameters()}
    ta2, _ = decouple_appearance(img.clone().requires_grad_(True), a, 3)
    (ta2 * r).sum().backward()
    torch.cuda.synchronize()
    for n, p in a.appearance_network.named_parameters():
        assert (p.grad - 2 * first[n]).abs().max().item() <= 1e-5 * max(first[n].abs().max().item(), 1e-6), n
    _lib.set_variant("APPEARANCE_OVERLAP", 1)


@pytest.mark.gpu
def test_head_weight_gradients_survive_reuse_of_the_saved_activations():
    """The head's weight-gradient kernels run on a side stream and read activations allocated on the main stream.  Right after
    `backward()` autograd drops those tensors; the caching allocator must not hand their blocks to main-stream work that runs beside
    the side stream (in training: the rasterizer backward's scratch).  Here the main stream allocates and overwrites same-sized
    blocks immediately, many times over, on a net whose weight gradients are large enough to keep the side stream busy."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd.appearance import decouple_appearance
    from based_on_point_cloud_partitions_amd.synthetic import decouple_appearance_torch

    _lib.set_variant("APPEARANCE_OVERLAP", 1)
    H, W = 544, 960
    a, b = _model(5), _model(5)
    img = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    r = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    tb, _ = decouple_appearance_torch(img.clone().requires_grad_(True), b, 1)
    (tb * r).sum().backward()
    want = {n: p.grad.clone() for n, p in b.appearance_network.named_parameters()}
    for trial in range(3):
        for p_ in a.appearance_network.parameters():
            p_.grad = None
        ta, _ = decouple_appearance(img.clone().requires_grad_(True), a, 1)
        loss = (ta * r).sum()
        shapes = [(256, H // 32, W // 32), (64, H // 16, W // 16), (32, H // 8, W // 8), (32, H // 4, W // 4), (16, H // 2, W // 2)]
        del ta
        loss.backward()
        del loss
        for _ in range(4):  # the freed activations' size classes, poisoned on the main stream at once
            junk = [torch.full((1,) + s_, float("nan"), device="cuda") for s_ in shapes for _ in range(3)]
            del junk
        torch.cuda.synchronize()
        for n, p_ in a.appearance_network.named_parameters():
            assert torch.isfinite(p_.grad).all(), (trial, n)
            assert (p_.grad - want[n]).abs().max().item() <= 2e-4 * max(want[n].abs().max().item(), 1e-6), (trial, n)


@pytest.mark.gpu
def test_head_backward_on_another_stream_and_hooked_weights():
    """`backward()` under a stream that is not the forward's: the `.grad` accumulation is ordered after the side stream on THAT stream.
    A weight with a tensor hook takes the plain autograd path (the hook sees the gradient)."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd.appearance import decouple_appearance
    from based_on_point_cloud_partitions_amd.synthetic import decouple_appearance_torch

    _lib.set_variant("APPEARANCE_OVERLAP", 1)
    H, W = 200, 333
    a, b, c = _model(7), _model(7), _model(7)
    img = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    r = torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)).cuda()
    tb, _ = decouple_appearance_torch(img.clone().requires_grad_(True), b, 2)
    (tb * r).sum().backward()
    ta, _ = decouple_appearance(img.clone().requires_grad_(True), a, 2)
    loss = (ta * r).sum()
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        loss.backward()
        grads = {n: p.grad.clone() for n, p in a.appearance_network.named_parameters()}  # read on the backward's stream, no device sync
    other.synchronize()
    close = lambda u, v: (u - v).abs().max().item() <= 2e-4 * max(v.abs().max().item(), 1e-6)
    for n, p in b.appearance_network.named_parameters():
        assert close(grads[n], p.grad), n
    seen = []
    c.appearance_network.conv1.weight.register_hook(lambda g: seen.append(g.clone()))
    tc, _ = decouple_appearance(img.clone().requires_grad_(True), c, 2)
    (tc * r).sum().backward()
    torch.cuda.synchronize()
    assert len(seen) == 1 and close(seen[0], b.appearance_network.conv1.weight.grad)
    for (n, pc), (_, pb) in zip(c.appearance_network.named_parameters(), b.appearance_network.named_parameters()):
        assert close(pc.grad, pb.grad), n


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W", [(256, 33, 60), (32, 264, 480), (5, 1, 1), (64, 7, 130)])
def test_fused_bias_relu_matches_torch(C, H, W):
    from based_on_point_cloud_partitions_amd.appearance import _BiasReLU

    gen = torch.Generator().manual_seed(C + H)
    x0 = torch.randn(1, C, H, W, generator=gen).cuda()
    b0 = torch.randn(C, generator=gen).cuda()
    g = torch.randn(1, C, H, W, generator=gen).cuda()
    xa, ba = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    xb, bb = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    ya = _BiasReLU.apply(xa * 1.0, ba)  # (* 1.0: a non-leaf, as after a convolution -- the kernel writes in place)
    yb = torch.relu(xb + bb[None, :, None, None])
    assert torch.equal(ya, yb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(xa.grad, xb.grad)
    assert (ba.grad - bb.grad).abs().max().item() <= 1e-5 * max(1.0, bb.grad.abs().max().item())


@pytest.mark.gpu
def test_appearance_tail_rejects_cpu_and_wrong_shapes():
    from based_on_point_cloud_partitions_amd import GsrError
    from based_on_point_cloud_partitions_amd.appearance import appearance_tail

    feat, w2, b2, w3, b3, image, _, _ = (torch.tensor(a) for a in _random_case(16, 64, 8, 32, 0))
    with pytest.raises(GsrError):
        appearance_tail(feat, w2, b2, w3, b3, image)
    c = lambda t: t.cuda()
    with pytest.raises(GsrError):
        appearance_tail(c(feat)[:8], c(w2), c(b2), c(w3), c(b3), c(image))
    with pytest.raises(GsrError):
        appearance_tail(c(feat), c(w2)[:, :8].contiguous(), c(b2), c(w3), c(b3), c(image))
