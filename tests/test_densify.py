"""Per-iteration densification statistics and point pruning: train.py:140-142,
GaussianModel.add_densification_stats / prune_points / _prune_optimizer (scene/gaussian_model.py:297-331,433-435).

The oracle for the statistics is a numpy restatement pinned against the reference's own torch lines on CPU; the oracle
for the compaction is torch's boolean indexing itself (`t[mask]`).
"""
import numpy as np
import pytest
import torch

from oracle import oracle


def _stats_inputs(P, seed):
    gen = torch.Generator().manual_seed(seed)
    radii = torch.randint(-1, 40, (P,), generator=gen, dtype=torch.int32).clamp_min(0)
    grad = torch.randn(P, 3, generator=gen) * 1e-3
    max_r = torch.rand(P, generator=gen) * 30
    acc = torch.rand(P, 1, generator=gen)
    den = torch.randint(0, 50, (P, 1), generator=gen).float()
    return radii, grad, max_r, acc, den


def test_oracle_stats_match_the_reference_lines_on_cpu():
    P = 30011
    radii, grad, max_r, acc, den = _stats_inputs(P, 0)
    m, a, d = max_r.numpy().copy(), acc.numpy().copy(), den.numpy().copy()
    oracle.densification_stats(radii.numpy(), grad.numpy(), m, a, d)
    # the reference's lines, verbatim semantics
    vis = radii > 0
    max_r[vis] = torch.max(max_r[vis], radii[vis])
    acc[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
    den[vis] += 1
    assert np.array_equal(m, max_r.numpy()) and np.array_equal(d, den.numpy())
    np.testing.assert_allclose(a, acc.numpy(), rtol=2e-7, atol=0)  # torch's vectorised norm differs in the last bit


class _Model:
    def __init__(self, P, device, seed=0, with_state=True):
        from based_on_point_cloud_partitions_amd.optim import Adam
        gen = torch.Generator().manual_seed(seed)
        mk = lambda *s: nn_param(torch.randn(*s, generator=gen).to(device))
        self._xyz, self._features_dc, self._features_rest = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3)
        self._opacity, self._scaling, self._rotation = mk(P, 1), mk(P, 3), mk(P, 4)
        self.embed = nn_param(torch.randn(7, 4, generator=gen).to(device))
        groups = [{"params": [self._xyz], "lr": 1e-3, "name": "xyz"}, {"params": [self._features_dc], "lr": 1e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 1e-3, "name": "f_rest"}, {"params": [self._opacity], "lr": 1e-3, "name": "opacity"},
                  {"params": [self._scaling], "lr": 1e-3, "name": "scaling"}, {"params": [self._rotation], "lr": 1e-3, "name": "rotation"},
                  {"params": [self.embed], "lr": 1e-3, "name": "appearance_embeddings"}]
        self.optimizer = Adam(groups, lr=0.0, eps=1e-15)
        if with_state:
            for g in groups:
                g["params"][0].grad = torch.randn(g["params"][0].shape, generator=gen).to(device)
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
        self.xyz_gradient_accum = torch.rand(P, 1, generator=gen).to(device)
        self.denom = torch.rand(P, 1, generator=gen).to(device)
        self.max_radii2D = torch.rand(P, generator=gen).to(device)


def nn_param(t):
    return torch.nn.Parameter(t.requires_grad_(True))


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 255, 256, 30011])
def test_hip_stats_bit_exact_vs_oracle(P):
    from based_on_point_cloud_partitions_amd.densify import add_densification_stats

    radii, grad, max_r, acc, den = _stats_inputs(P, P)
    m, a, d = max_r.numpy().copy(), acc.numpy().copy(), den.numpy().copy()
    oracle.densification_stats(radii.numpy(), grad.numpy(), m, a, d)

    class G:
        pass
    g = G()
    g.max_radii2D, g.xyz_gradient_accum, g.denom = max_r.cuda(), acc.cuda(), den.cuda()
    vsp = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vsp.grad = grad.cuda()
    add_densification_stats(g, vsp, radii.cuda())
    assert np.array_equal(g.max_radii2D.cpu().numpy(), m)
    assert np.array_equal(g.xyz_gradient_accum.cpu().numpy(), a)
    assert np.array_equal(g.denom.cpu().numpy(), d)


@pytest.mark.gpu
@pytest.mark.parametrize("P,frac", [(1, 1.0), (1, 0.0), (257, 0.5), (20011, 0.9), (20011, 0.0), (20011, 1.0), (100_000, 0.3)])
def test_compact_rows_equals_boolean_indexing(P, frac):
    from based_on_point_cloud_partitions_amd.densify import compact_rows

    gen = torch.Generator().manual_seed(P)
    keep = (torch.rand(P, generator=gen) < frac).cuda()
    tensors = [torch.randn(P, 3, generator=gen).cuda(), torch.randn(P, 15, 3, generator=gen).cuda(), torch.randn(P, generator=gen).cuda(),
               torch.randn(P, 1, generator=gen).cuda(), torch.randn(P, 4, generator=gen).cuda()]
    outs = compact_rows(tensors, keep)
    for t, o in zip(tensors, outs):
        assert o.shape == t[keep].shape and torch.equal(o, t[keep])


@pytest.mark.gpu
def test_prune_points_matches_the_reference_surgery():
    from based_on_point_cloud_partitions_amd.densify import prune_points

    P = 5003
    a, b = _Model(P, "cuda", seed=2), _Model(P, "cuda", seed=2)
    mask = (torch.rand(P, generator=torch.Generator().manual_seed(7)) < 0.25).cuda()
    prune_points(a, mask)
    # reference code path on the twin (gaussian_model.py:297-331), with torch indexing
    valid = ~mask
    for group in b.optimizer.param_groups:
        if group["name"] in ("appearance_embeddings", "appearance_network"):
            continue
        st = b.optimizer.state.get(group["params"][0], None)
        st["exp_avg"] = st["exp_avg"][valid]
        st["exp_avg_sq"] = st["exp_avg_sq"][valid]
        del b.optimizer.state[group["params"][0]]
        group["params"][0] = torch.nn.Parameter(group["params"][0][valid].requires_grad_(True))
        b.optimizer.state[group["params"][0]] = st
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert pa.shape == pb.shape and torch.equal(pa, pb) and pa.requires_grad and pa.is_leaf
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert int(sa["step"]) == int(sb["step"]) == 1
    assert a._xyz is a.optimizer.param_groups[0]["params"][0] and a._rotation is a.optimizer.param_groups[5]["params"][0]
    assert a.optimizer.param_groups[6]["params"][0].shape == (7, 4)  # appearance group untouched
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum[valid]) and torch.equal(a.max_radii2D, b.max_radii2D[valid])
    assert a.denom.shape == (int(valid.sum()), 1)
    # the pruned model keeps training
    for g in a.optimizer.param_groups:
        g["params"][0].grad = torch.ones_like(g["params"][0])
    a.optimizer.step()
    assert int(a.optimizer.state[a._xyz]["step"]) == 2


def _reference_densify_and_prune(m, max_grad, min_opacity, extent, max_screen_size, split_samples):
    """scene/gaussian_model.py:297-431 restated with torch indexing only (the reference's own data flow), on a twin model."""
    import torch.nn as nn
    opt = m.optimizer
    skip = ("appearance_embeddings", "appearance_network")
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}

    def cat(d):
        for g in opt.param_groups:
            if g["name"] in skip:
                continue
            ext = d[g["name"]]
            st = opt.state.get(g["params"][0], None)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del opt.state[g["params"][0]]
            g["params"][0] = nn.Parameter(torch.cat((g["params"][0], ext), dim=0).requires_grad_(True))
            opt.state[g["params"][0]] = st
            setattr(m, names[g["name"]], g["params"][0])
        P = m._xyz.shape[0]
        m.xyz_gradient_accum = torch.zeros((P, 1), device="cuda"); m.denom = torch.zeros((P, 1), device="cuda")
        m.max_radii2D = torch.zeros((P,), device="cuda")

    def prune(mask):
        valid = ~mask
        for g in opt.param_groups:
            if g["name"] in skip:
                continue
            st = opt.state.get(g["params"][0], None)
            st["exp_avg"] = st["exp_avg"][valid]; st["exp_avg_sq"] = st["exp_avg_sq"][valid]
            del opt.state[g["params"][0]]
            g["params"][0] = nn.Parameter(g["params"][0][valid].requires_grad_(True))
            opt.state[g["params"][0]] = st
            setattr(m, names[g["name"]], g["params"][0])
        m.xyz_gradient_accum = m.xyz_gradient_accum[valid]; m.denom = m.denom[valid]; m.max_radii2D = m.max_radii2D[valid]

    grads = m.xyz_gradient_accum / m.denom
    grads[grads.isnan()] = 0.0
    sel = torch.logical_and(torch.norm(grads, dim=-1) >= max_grad, torch.max(torch.exp(m._scaling), dim=1).values <= m.percent_dense * extent)
    cat({k: getattr(m, a)[sel] for k, a in names.items()})
    n_init = m._xyz.shape[0]
    padded = torch.zeros((n_init,), device="cuda"); padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.logical_and(padded >= max_grad, torch.max(torch.exp(m._scaling), dim=1).values > m.percent_dense * extent)
    N = 2
    from based_on_point_cloud_partitions_amd.densify import _build_rotation
    rots = _build_rotation(m._rotation[sel]).repeat(N, 1, 1)
    new_xyz = torch.bmm(rots, split_samples.unsqueeze(-1)).squeeze(-1) + m._xyz[sel].repeat(N, 1)
    new_scaling = torch.log(torch.exp(m._scaling[sel]).repeat(N, 1) / (0.8 * N))
    cat({"xyz": new_xyz, "f_dc": m._features_dc[sel].repeat(N, 1, 1), "f_rest": m._features_rest[sel].repeat(N, 1, 1),
         "opacity": m._opacity[sel].repeat(N, 1), "scaling": new_scaling, "rotation": m._rotation[sel].repeat(N, 1)})
    prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), device="cuda", dtype=bool))))
    pm = (torch.sigmoid(m._opacity) < min_opacity).squeeze()
    if max_screen_size:
        pm = torch.logical_or(torch.logical_or(pm, m.max_radii2D > max_screen_size), torch.exp(m._scaling).max(dim=1).values > 0.1 * extent)
    prune(pm)


@pytest.mark.gpu
def test_densify_and_prune_matches_the_reference_code_path():
    from based_on_point_cloud_partitions_amd import densify

    P = 6007
    a, b = _Model(P, "cuda", seed=4), _Model(P, "cuda", seed=4)
    gen = torch.Generator().manual_seed(12)
    for m in (a, b):
        m.percent_dense = 0.01
        with torch.no_grad():
            m._scaling.copy_(torch.log(0.002 + 0.05 * torch.rand(P, 3, generator=torch.Generator().manual_seed(1))).cuda())
        m.xyz_gradient_accum = (torch.rand(P, 1, generator=torch.Generator().manual_seed(2)) * 4e-4).cuda()
        m.denom = torch.randint(0, 3, (P, 1), generator=torch.Generator().manual_seed(3)).float().cuda()  # zeros -> NaN grads -> 0
        m.max_radii2D = (torch.rand(P, generator=torch.Generator().manual_seed(5)) * 30).cuda()
    extent, max_grad = 5.0, 0.0002
    # the split's random draw: fix it for both paths (count of selected points is deterministic)
    grads = a.xyz_gradient_accum / a.denom
    grads[grads.isnan()] = 0.0
    clone_sel = torch.logical_and(torch.norm(grads, dim=-1) >= max_grad, torch.max(torch.exp(a._scaling), dim=1).values <= 0.01 * extent)
    split_sel = torch.logical_and(grads.squeeze() >= max_grad, torch.max(torch.exp(a._scaling), dim=1).values > 0.01 * extent)
    assert int(clone_sel.sum()) > 50 and int(split_sel.sum()) > 50
    samples = (torch.randn(2 * int(split_sel.sum()), 3, generator=gen) * 0.01).cuda()
    densify.densify_and_prune(a, max_grad, 0.005, extent, 20, split_samples=samples)
    _reference_densify_and_prune(b, max_grad, 0.005, extent, 20, samples)
    assert a._xyz.shape == b._xyz.shape and a._xyz.shape[0] != P
    for attr in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a, attr), getattr(b, attr)), attr
    for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
        if ga["name"] == "appearance_embeddings":
            continue
        sa, sb = a.optimizer.state[ga["params"][0]], b.optimizer.state[gb["params"][0]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # opacity reset (gaussian_model.py:231-234)
    densify.reset_opacity(a)
    assert float(torch.sigmoid(a._opacity.detach()).max()) <= 0.01 + 1e-6
    st = a.optimizer.state[a._opacity]
    assert float(st["exp_avg"].abs().max()) == 0.0 and a.optimizer.param_groups[3]["params"][0] is a._opacity


@pytest.mark.gpu
def test_densify_and_prune_with_nothing_selected_and_with_everything_pruned():
    from based_on_point_cloud_partitions_amd import densify

    m = _Model(1000, "cuda", seed=1)
    m.percent_dense = 0.01
    m.xyz_gradient_accum = torch.zeros(1000, 1, device="cuda")
    m.denom = torch.ones(1000, 1, device="cuda")
    m.max_radii2D = torch.zeros(1000, device="cuda")
    before = m._xyz.detach().clone()
    densify.densify_and_prune(m, 1e9, 0.0, 5.0, None)  # nothing cloned, split or pruned: empty gathers everywhere
    assert torch.equal(m._xyz.detach(), before) and m.optimizer.state[m._xyz]["exp_avg"].shape == (1000, 3)
    m.xyz_gradient_accum = torch.ones(1000, 1, device="cuda")
    densify.densify_and_prune(m, 0.0, 2.0, 5.0, 1)  # every point selected, then every point pruned (opacity < 2)
    assert m._xyz.shape == (0, 3) and m.max_radii2D.shape == (0,) and m.optimizer.state[m._xyz]["exp_avg_sq"].shape == (0, 3)
    densify.reset_opacity(m)  # on an empty model
    assert m._opacity.shape == (0, 1)
