"""networks.ResnetGenerator / NLayerDiscriminator / PatchDiscriminator / Downsample / Upsample
against fixtures generated from the reference's own network/networks.py (tests/golden/networks_zoo.npz)."""
import numpy as np
import pytest
import torch

from conftest import rel_err, l2_rel

pytestmark = pytest.mark.gpu


def _load(module, g, prefix, seed):
    import zlib
    keys = [str(k) for k in g[prefix + "_keys"]]
    sd = module.state_dict()
    assert list(sd.keys()) == keys, (list(sd.keys())[:6], keys[:6])
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g[prefix + "_shapes"]]
    new = {}
    for k, v in sd.items():
        if k.endswith("filt"):
            new[k] = v.clone()
            continue
        rs = np.random.RandomState((zlib.crc32(k.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        a = rs.standard_normal(tuple(v.shape))
        a = a / np.sqrt(np.prod(v.shape[1:])) if v.dim() >= 2 else 0.1 * a
        new[k] = torch.from_numpy(np.ascontiguousarray(a)).float()
    module.load_state_dict(new)
    return module.cuda()


def _check_grads(module, g, prefix):
    grads = dict(module.named_parameters())
    floor = 1e-3 * float(np.max(g[prefix + "_grad_l2"]))
    for n, ref in zip([str(n) for n in g[prefix + "_grad_names"]], g[prefix + "_grad_l2"]):
        got = float(grads[n].grad.double().norm())
        if ref < floor:
            # a conv bias in front of a non-affine InstanceNorm has an exactly-zero gradient in real arithmetic:
            # what both sides hold there is rounding noise -- only require that it stays noise
            assert got < 10 * floor, (n, got, ref)
            continue
        assert abs(got - ref) <= 5e-3 * ref + 1e-7, (n, got, ref)
    for k in g.files:
        if k.startswith(prefix + "_grad::") and float(np.linalg.norm(g[k])) >= floor:
            assert l2_rel(grads[k[len(prefix) + 7:]].grad.cpu().numpy(), g[k]) < 5e-3, k


def test_blur_pool_pair_and_padding(golden):
    import smsut_amd
    from smsut_amd import ops
    from smsut_amd.network import networks
    g = golden("networks_zoo")
    t = torch.from_numpy(g["t"]).cuda()
    assert rel_err(networks.Downsample(3).cuda()(t).cpu().numpy(), g["down"]) < 1e-6
    assert rel_err(networks.Upsample(3).cuda()(t).cpu().numpy(), g["up"]) < 1e-6          # == x2 bilinear
    for mode, ref_mod in (("reflect", torch.nn.ReflectionPad2d), ("replicate", torch.nn.ReplicationPad2d),
                          ("zero", torch.nn.ZeroPad2d)):
        x = torch.from_numpy(g["t"]).requires_grad_(True)
        y = ref_mod((3, 2, 1, 3))(x)
        gy = torch.from_numpy(np.random.RandomState(1).standard_normal(tuple(y.shape))).float()
        y.backward(gy)
        xd = torch.from_numpy(g["t"]).cuda().requires_grad_(True)
        yd = ops.pad2d(xd, (3, 2, 1, 3), mode)
        yd.backward(gy.cuda())
        assert rel_err(yd.detach().cpu().numpy(), y.detach().numpy()) < 1e-7, mode
        assert rel_err(xd.grad.cpu().numpy(), x.grad.numpy()) < 1e-6, mode
    x = torch.from_numpy(g["t"]).requires_grad_(True)
    torch.nn.functional.pad(x, (-1, -2, -1, -1)).sum().backward()
    xd = torch.from_numpy(g["t"]).cuda().requires_grad_(True)
    yc = ops.pad2d(xd, (-1, -2, -1, -1))
    yc.sum().backward()
    assert yc.shape == (2, 3, 8, 9) and rel_err(xd.grad.cpu().numpy(), x.grad.numpy()) < 1e-7
    # blur-pool adjoint: <down(x), g> == <x, down^T(g)>
    xb = torch.from_numpy(g["t"]).cuda().requires_grad_(True)
    yb = ops.blur_down2(xb)
    gb = torch.randn_like(yb)
    yb.backward(gb)
    x2 = torch.randn_like(xb)
    assert abs((ops.blur_down2(x2) * gb).sum().item() - (x2 * xb.grad).sum().item()) < 1e-3


def test_resnet_generator(golden):
    import smsut_amd
    from smsut_amd.network import networks
    g = golden("networks_zoo")
    G = _load(networks.ResnetGenerator(1, 1, ngf=8, norm_layer=networks.get_norm_layer("instance"), n_blocks=2), g, "g",
              int(g["seed"]))
    x = torch.from_numpy(g["g_x"]).cuda().requires_grad_(True)
    y = G(x)
    assert rel_err(y.detach().cpu().numpy(), g["g_y"]) < 1e-3
    y.backward(torch.from_numpy(g["g_gy"]).cuda())
    assert l2_rel(x.grad.cpu().numpy(), g["g_gx"]) < 5e-3
    _check_grads(G, g, "g")
    feats = G(x.detach(), layers=[0, 4, 8], encode_only=True)
    assert len(feats) == 3 and rel_err(feats[2].detach().cpu().numpy(), g["g_feat8"]) < 1e-3
    # torch's own norm-layer factory is accepted too (constructor drop-in)
    import functools
    G2 = networks.ResnetGenerator(1, 1, ngf=8, n_blocks=1,
                                  norm_layer=functools.partial(torch.nn.InstanceNorm2d, affine=False, track_running_stats=False))
    assert any(k.endswith("model.1.bias") for k in G2.state_dict())


def test_nlayer_and_patch_discriminator(golden):
    import smsut_amd
    from smsut_amd.network import networks
    g = golden("networks_zoo")
    norm = networks.get_norm_layer("instance")
    D = _load(networks.NLayerDiscriminator(1, ndf=8, n_layers=3, norm_layer=norm), g, "d", int(g["seed"]) + 5)
    x = torch.from_numpy(g["d_x"]).cuda().requires_grad_(True)
    y = D(x)
    assert y.shape == tuple(g["d_y"].shape) and rel_err(y.detach().cpu().numpy(), g["d_y"]) < 1e-3
    y.backward(torch.from_numpy(g["d_gy"]).cuda())
    assert l2_rel(x.grad.cpu().numpy(), g["d_gx"]) < 5e-3
    _check_grads(D, g, "d")
    P = networks.PatchDiscriminator(1, ndf=8, norm_layer=norm)
    import zlib
    sd = {}
    for k, v in P.state_dict().items():
        if k.endswith("filt"):
            sd[k] = v.clone(); continue
        rs = np.random.RandomState((zlib.crc32(k.encode()) ^ ((int(g["seed"]) + 9) * 2654435761)) & 0x7FFFFFFF)
        a = rs.standard_normal(tuple(v.shape))
        sd[k] = torch.from_numpy(np.ascontiguousarray(a / np.sqrt(np.prod(v.shape[1:])) if v.dim() >= 2 else 0.1 * a)).float()
    P.load_state_dict(sd)
    from oracle import recipe
    yp = P.cuda()(recipe.synth_images((1, 1, 32, 32), int(g["seed"]) + 10).cuda())
    assert rel_err(yp.detach().cpu().numpy(), g["p_y"]) < 1e-3
